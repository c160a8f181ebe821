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


# ----------------------------------------------------------------------------- whole step vs the reference's OWN precision
def _oracle_step_on_device(ref, ref_params, lat, noise, ts, ehs, autocast):
    from tests import helpers as H

    return H.oracle_step_on_device(ref, ref_params, lat, noise, ts, ehs, autocast)


def test_batch4_step_error_is_bracketed_by_the_bf16_autocast_reference(monkeypatch):
    """the whole-step tests compared with the f32 oracle at fixed loose bounds.  Here the BASELINE
    configs[1] step at BATCH 4 (bench configuration: bf16, channels_last, head-padded + grouped projections, merged weights,
    hipGraph) is placed against TWO runs of the oracle's op sequence on the same values: f32, and under torch.autocast(bf16) —
    the reference's own arithmetic.  Required: the device path's distance to the f32 result is at most 1.5 x the bf16
    reference's distance to it, on the UNet output, on the loss, and on the LoRA gradients in aggregate and for the median
    tensor; no single tensor both 4 x the reference's error and 3 % off."""
    from lora_amd.standin import DDPMScheduler, fused
    from tests import helpers as H
    from tests.test_gpu_parity_r3 import _sd15_twins

    ref, ref_params, unet = _sd15_twins()
    ref.to(DEV)
    g = torch.Generator().manual_seed(77)
    B = 4
    lat = (torch.randn(B, 4, 64, 64, generator=g) * 0.18215).to(torch.bfloat16).float().to(DEV)
    ehs = torch.randn(B, 77, 768, generator=g).to(torch.bfloat16).float().to(DEV)
    noise = torch.randn(B, 4, 64, 64, generator=g).to(torch.bfloat16).float().to(DEV)
    ts = torch.randint(0, 1000, (B,), generator=g).to(DEV)
    with H.oracle_on_device():   # library kernels only: the oracle must not run through csrc/hostops.hip
        p32, l32, g32 = _oracle_step_on_device(ref, ref_params, lat, noise, ts, ehs, autocast=False)
        pbf, lbf, gbf = _oracle_step_on_device(ref, ref_params, lat, noise, ts, ehs, autocast=True)
    del ref
    torch.cuda.empty_cache()

    monkeypatch.setenv("LORA_AMD_HEAD_PAD", "1")
    monkeypatch.setenv("LORA_AMD_GROUP_QKV", "1")
    monkeypatch.setattr(fused, "_ENABLED", True)
    unet.to(memory_format=torch.channels_last)
    st = T.FlatLoraState([{"params": T.lora_params(unet), "lr": 1e-4, "weight_decay": 1e-2}], max_grad_norm=1.0, device=DEV)
    st.attach_direct_grads(unet)
    merged = st.enable_merged_weights(unet)
    sched = DDPMScheduler()
    fmt = torch.channels_last
    latd = lat.to(torch.bfloat16).contiguous(memory_format=fmt)
    ehsd = ehs.to(torch.bfloat16)
    noised = noise.to(torch.bfloat16).contiguous(memory_format=fmt)

    def fwd_bwd(l_, c_):
        return T.forward_backward(unet, sched, l_, c_, T.StepConfig(), noise=noised, timesteps=ts, merged=merged)

    try:
        for _ in range(2):
            fwd_bwd(latd, ehsd)
            st.zero_grad()
        runner = T.GraphedForwardBackward(fwd_bwd, latd, ehsd, st)
        st.zero_grad()
        loss_dev = float(runner(latd, ehsd))
        gdev = st.flat_g.clone()
        with torch.no_grad():
            merged.refresh()
            pdev = unet(sched.add_noise(latd.float(), noised.float(), ts).to(latd.dtype), ts, ehsd).sample.float()
    finally:
        for m in unet.modules():
            m.__dict__.pop("_grad_sink", None)
            m.__dict__.pop("_merged", None)

    e_dev, e_bf = float((pdev - p32).norm()), float((pbf - p32).norm())
    print(f"UNet output: |dev - f32| = {e_dev:.4e}, |bf16 ref - f32| = {e_bf:.4e}, ratio {e_dev / e_bf:.3f}; "
          f"loss f32 {l32:.6f} bf16-ref {lbf:.6f} dev {loss_dev:.6f}")
    assert e_dev <= 1.5 * e_bf, (e_dev, e_bf)
    # (the bf16 reference's own loss error moves between 2e-5 and 2e-4 with the library's attention kernel picks: a floor of
    # 0.05 % of the loss keeps the bracket meaningful when it happens to be tiny)
    assert abs(loss_dev - l32) <= 1.5 * abs(lbf - l32) + 5e-4 * abs(l32), (loss_dev, lbf, l32)
    # aggregate 1.27-1.41 and median 1.20-1.30 over five boxes: asserted at + 10 % (round 6; round 5: 1.6 / 1.5).  Single
    # tensors: the device step rounds differently from autocast in places that are the HOST model's policy, not the adapters'
    # (bf16-resident weights and residual stream against f32 weights + per-op casts — measured in round 6 by running the SAME
    # adapters under the reference's policy, test_gpu_parity_r6.py::test_bracket_under_the_reference_precision_policy): a tensor
    # may land at a few times the bf16 reference's error where that error happens to be small (worst ratio 2.5-4.2 from box
    # to box, always a `down` gradient at 0.6-2 % relative error); none may be BOTH > 4 x the reference's and > 3 % off
    H.assert_bracket(H.bracket(g32, gbf, gdev, "bench configuration, batch 4"))


# ----------------------------------------------------------------------------- the dithered rounding of the in-step merge
@pytest.mark.parametrize("frac", [0.5, 0.3])
def test_merge_step_dither_is_unbiased_down_the_columns_and_uncorrelated_between_neighbours(frac):
    """``ROUND_DITHER`` (csrc/merge_step.hip) takes the dithers of the 8 elements of a 16-byte chunk
    from ONE 64-bit hash — element i reads bytes (i, i + 1 mod 8), so neighbours share a byte — and round 4 only looked at row
    sums.  Here every element of W is 2^-4 and the delta is `frac` of its bf16 ulp, so "rounded up" is a Bernoulli(frac)
    indicator per element: (a) its mean down every COLUMN (what the GEMM on W_eff^T sums) and along every row is `frac` within
    4.5 sigma of a fair coin's; (b) the eight positions of a chunk have the same rate; (c) the indicators of neighbouring
    elements — along k inside a chunk, across the chunk boundary, and along n — are uncorrelated (|rho| < 0.01 with 1.3 M
    samples: sigma 0.0009); (d) W_eff^T carries the same bits."""
    N, K, r = 2048, 640, 4
    w = torch.full((N, K), 2.0 ** -4, dtype=torch.bfloat16, device=DEV)
    ulp = 2.0 ** -11
    up = torch.zeros(N, r, device=DEV)
    up[:, 0] = 1.0
    down = torch.zeros(r, K, device=DEV)
    down[0] = frac * ulp
    out, out_t = torch.empty_like(w), torch.empty(K, N, dtype=w.dtype, device=DEV)
    _C.MergeStepPlan([dict(w=w, up=up, down=down, out=out, out_t=out_t, row_heads=None, col_heads=None, key=5)]).launch(
        1.0, _C.ROUND_DITHER)
    assert torch.equal(out_t, out.t())
    b = (out.float() != 2.0 ** -4)
    assert bool(((out.float() == 2.0 ** -4) | (out.float() == 2.0 ** -4 + ulp)).all())
    x = b.double()
    sig = (frac * (1 - frac)) ** 0.5
    assert abs(float(x.mean()) - frac) < 4.5 * sig / (N * K) ** 0.5
    col, row = x.mean(0), x.mean(1)
    assert float((col - frac).abs().max()) < 4.5 * sig / N ** 0.5, float((col - frac).abs().max())
    assert float((row - frac).abs().max()) < 4.5 * sig / K ** 0.5, float((row - frac).abs().max())
    pos = x.view(N, K // 8, 8).mean((0, 1))
    assert float((pos - frac).abs().max()) < 4.5 * sig / (N * K / 8) ** 0.5, pos.tolist()

    def rho(a, c):
        a, c = a - a.mean(), c - c.mean()
        return float((a * c).mean() / (a.std(unbiased=False) * c.std(unbiased=False)))
    ch = x.view(N, K // 8, 8)
    inside = max(abs(rho(ch[:, :, i], ch[:, :, i + 1])) for i in range(7))
    wrap = abs(rho(ch[:, :, 7], ch[:, :, 0]))                      # bytes (7, 0): the pair that shares byte 0
    across = abs(rho(ch[:, :-1, 7], ch[:, 1:, 0]))                 # chunk boundary: different hashes
    rows = abs(rho(x[:-1], x[1:]))
    assert max(inside, wrap, across, rows) < 0.01, (inside, wrap, across, rows)
