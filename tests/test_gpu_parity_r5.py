"""Round-5 device parity (VERDICT r4 / ADVICE r4).  Everything goes through the C-ABI (``lora_amd/_C.py``)."""
import numpy as np
import pytest
import torch

import lora_amd as L
from lora_amd import _C, ops
from lora_amd import trainer as T
from oracle import lora_numpy as O
from tests.test_gpu_kernels import close, n, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# ----------------------------------------------------------------------------- stale group weights
def test_group_forward_after_an_optimizer_step_runs_on_this_steps_weights():
    """An eager forward outside ``trainer.forward_backward`` after ``state.step()`` (validation, sampling): the FIRST LoRA call
    of a UNet forward is the attn1 q / k / v GROUP; it must re-merge before its GEMM reads the scratch weight (round 4: only the
    next non-group site triggered the refresh, so the group ran on the previous step's W_eff).  Checked against the oracle's
    forward on the factors AFTER the step."""
    M, K, N, r, s = 512, 320, 320, 4, 1.0
    torch.manual_seed(3)
    mods = []
    for _ in range(3):
        m = L.LoraInjectedLinear(K, N, False, r=r, dropout_p=0.0, scale=s).to(DEV).to(torch.bfloat16)
        m.linear.requires_grad_(False)
        T.promote_lora_to_fp32(m)
        m.lora_up.weight.data.normal_(0, 0.05)
        mods.append(m)
    holder = torch.nn.ModuleList(mods)
    state = T.FlatLoraState([{"params": T.lora_params(holder), "lr": 5e-2, "weight_decay": 0.0}], max_grad_norm=0.0, device=DEV)
    state.attach_direct_grads(holder)
    mw = state.enable_merged_weights(holder)
    x = rnd((M, K), "bf16", seed=7).requires_grad_(True)
    gs = [rnd((M, N), "bf16", seed=8 + i) for i in range(3)]
    mw.refresh()
    outs = L.lora_linear_group(mods, x)
    assert outs is not None and mw.groups, "the group path must be the one under test"
    torch.autograd.backward(outs, gs)
    before = [n(m.lora_up.weight).copy() for m in mods]
    state.step(state.all_reduce())  # lr 5e-2: every factor moves by ~5e-2 — far above bf16 resolution of the outputs
    assert all(np.abs(n(m.lora_up.weight) - b).max() > 1e-2 for m, b in zip(mods, before))
    refreshes = mw.refreshes
    with torch.no_grad():
        outs2 = L.lora_linear_group(mods, x)  # NO explicit refresh: the group lookup has to notice the step
    assert mw.refreshes == refreshes + 1
    X = n(x)
    for m, y, y_old in zip(mods, outs2, outs):
        W, A, U = n(m.linear.weight), n(m.lora_down.weight), n(m.lora_up.weight)
        yo, _ = O.lora_linear_forward(X, W, None, A, U, s)
        absy = np.abs(X) @ (np.abs(W) + s * np.abs(U) @ np.abs(A)).T
        assert np.all(np.abs(n(y) - yo) <= 2.0 ** -8 * absy + 2.0 ** -8 * np.abs(yo) + 1e-3)
        assert np.abs(n(y) - n(y_old)).max() > 0.05  # and it is NOT the previous step's output
