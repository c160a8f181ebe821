"""Host logic of the training step on CPU tensors: flat buffers, clip+AdamW maths, LR schedules, step parity
with the oracle's restatement of the reference step."""
import copy
import os

import numpy as np
import pytest
import torch

import lora_amd as L
from lora_amd import trainer as T
from lora_amd.standin import DDPMScheduler, tiny_unet
from oracle import torch_ref as TR
from tests import helpers as H


def _tiny_pair(r=2, seed=0):
    torch.manual_seed(seed)
    a = tiny_unet()
    b = copy.deepcopy(a)
    a.requires_grad_(False), b.requires_grad_(False)
    torch.manual_seed(5)
    L.inject_trainable_lora(a, r=r)
    torch.manual_seed(5)
    ref_params = TR.inject(b, L.UNET_DEFAULT_TARGET_REPLACE, r=r)
    for (up, down), site in zip(L.extract_lora_ups_down(a), TR.sites_of(b)):
        site.down.data.copy_(down.weight.data)
        up.weight.data.normal_(0, 0.05)
        site.up.data.copy_(up.weight.data)
    return a, b, ref_params


def test_flat_state_aliases_params_and_matches_torch_adamw():
    d = dict(np.load(os.path.join(H.GOLDEN, "optimizer_case.npz")))
    p1, p2 = torch.nn.Parameter(torch.from_numpy(d["p0"][:700].copy())), torch.nn.Parameter(torch.from_numpy(d["p0"][700:].copy()))
    st = T.FlatLoraState([{"params": [p1], "lr": 1e-2}, {"params": [p2], "lr": 5e-3}], max_grad_norm=1.0)
    assert st.n == 1000 and st.payload_bytes == 4000 and p1.data.data_ptr() == st.flat_p.data_ptr()
    assert p2.grad.data_ptr() == st.flat_g[700:].data_ptr()
    for step in (1, 2, 3):
        st.flat_g.copy_(torch.from_numpy(d[f"g{step}"]))
        assert abs(st.grad_norm().item() - float(d[f"norm{step}"])) < 1e-4 * max(1, float(d[f"norm{step}"]))
        st.step()
        np.testing.assert_allclose(st.flat_p.numpy(), d[f"p{step}"], rtol=3e-6, atol=3e-7)
        assert st.flat_g.abs().sum() == 0
    np.testing.assert_allclose(p1.detach().numpy(), d["p3"][:700], rtol=3e-6, atol=3e-7)  # views stay live


def test_training_steps_match_oracle_restatement_of_reference_step():
    a, b, ref_params = _tiny_pair()
    sched = DDPMScheduler()
    st = T.FlatLoraState([{"params": T.lora_params(a), "lr": 1e-3, "weight_decay": 1e-2}], max_grad_norm=1.0)
    opt = torch.optim.AdamW(ref_params, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    a.train(), b.train()
    for cfg in (T.StepConfig(), T.StepConfig(with_prior_preservation=True, prior_loss_weight=0.5)):
        for it in range(3):
            g = torch.Generator().manual_seed(it)
            lat, ehs = torch.randn(4, 4, 16, 16, generator=g), torch.randn(4, 7, 32, generator=g)
            noise, t = torch.randn(4, 4, 16, 16, generator=g), torch.randint(0, 1000, (4,), generator=g)
            la = T.forward_backward(a, sched, lat, ehs, cfg, noise=noise, timesteps=t)
            st.step(st.all_reduce())
            lb = TR.dreambooth_step(lambda x, tt, c: b(x, tt, c).sample, ref_params, opt, lat, noise, t, ehs,
                                    sched.alphas_cumprod, 1.0, cfg.with_prior_preservation, cfg.prior_loss_weight)
            assert abs(la.item() - lb.item()) < 1e-5 * max(1.0, abs(lb.item()))
    for (up, down), site in zip(L.extract_lora_ups_down(a), TR.sites_of(b)):
        np.testing.assert_allclose(H.t2n(up.weight), H.t2n(site.up), rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(H.t2n(down.weight), H.t2n(site.down), rtol=2e-4, atol=2e-6)


def test_lora_params_order_is_up_then_down():
    a, _, _ = _tiny_pair()
    ps = T.lora_params(a)
    pairs = L.extract_lora_ups_down(a)
    assert len(ps) == 2 * len(pairs) and ps[0] is pairs[0][0].weight and ps[1] is pairs[0][1].weight


@pytest.mark.parametrize("name", ["constant", "constant_with_warmup", "linear", "cosine", "cosine_with_restarts", "polynomial"])
def test_lr_schedules(name):
    f = T.get_lr_lambda(name, 10, 100)
    if name != "constant":
        assert f(0) == 0.0 and abs(f(5) - 0.5) < 1e-9
    assert abs(f(10) - 1.0) < 1e-6
    if name in ("linear", "cosine"):
        assert f(100) < 1e-6 and 0 < f(55) < 1
    with pytest.raises(ValueError):
        T.get_lr_lambda("nope", 0, 1)


def test_promote_and_v_prediction():
    u = tiny_unet().to(torch.bfloat16)
    L.inject_trainable_lora(u, r=2)
    assert all(p.dtype == torch.bfloat16 for p in T.lora_params(u))  # inject casts to the frozen dtype (ref :295)
    T.promote_lora_to_fp32(u)
    assert all(p.dtype == torch.float32 for p in T.lora_params(u))
    s = DDPMScheduler(prediction_type="v_prediction")
    x, n, t = torch.randn(2, 4, 8, 8), torch.randn(2, 4, 8, 8), torch.tensor([10, 700])
    a = s.alphas_cumprod[t].view(-1, 1, 1, 1)
    assert torch.allclose(s.get_velocity(x, n, t), a.sqrt() * n - (1 - a).sqrt() * x, atol=1e-6)
