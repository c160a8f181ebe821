"""The stand-in host models reproduce the index layout of the reference's shipped .safetensors fixtures."""
import json
import os

import pytest
import torch
import torch.nn as nn

import lora_amd as L
from lora_amd.standin import DDPMScheduler, clip_text_model, sd15_lora_site_shapes, sd15_unet, tiny_unet
from oracle import lora_numpy as O
from tests import helpers as H

MAN = json.load(open(os.path.join(H.GOLDEN, "example_loras_manifest.json")))


def test_sd15_parameter_count_and_site_count():
    with torch.device("meta"):
        u = sd15_unet()
    assert sum(p.numel() for p in u.parameters()) == 859_520_964  # the well-known SD1.5 UNet size
    assert len(sd15_lora_site_shapes()) == 144 and len(sd15_lora_site_shapes(extended=True)) == 224


@pytest.mark.parametrize("fn,r", [("analog_svd_rank4.safetensors", 4), ("lora_disney.safetensors", 1)])
def test_unet_index_layout_matches_shipped_fixture(fn, r):
    tensors = MAN[fn]["tensors"]
    shapes = sd15_lora_site_shapes()
    assert sum(1 for k in tensors if k.startswith("unet:") and k.endswith(":up")) == len(shapes) == 144
    for i, (N, K) in enumerate(shapes):
        assert tensors[f"unet:{i}:up"]["shape"] == [N, r], i
        assert tensors[f"unet:{i}:down"]["shape"] == [r, K], i
    assert set(json.loads(MAN[fn]["metadata"]["unet"])) == L.UNET_DEFAULT_TARGET_REPLACE
    assert all(MAN[fn]["metadata"][f"unet:{i}:rank"] == str(r) for i in range(144))


def test_clip_index_layout_matches_shipped_fixture():
    with torch.device("meta"):
        clip = clip_text_model()
    sites = [m for _, _, m in L._find_modules(clip, {"CLIPAttention"}, search_class=[nn.Linear])]
    tensors = MAN["analog_svd_rank4.safetensors"]["tensors"]
    assert len(sites) == 48 == sum(1 for k in tensors if k.startswith("text_encoder:") and k.endswith(":up"))
    for i, m in enumerate(sites):
        assert tensors[f"text_encoder:{i}:up"]["shape"] == [m.out_features, 4]
        assert tensors[f"text_encoder:{i}:down"]["shape"] == [4, m.in_features]


@pytest.mark.skipif(not os.path.exists("/root/reference/example_loras/analog_svd_rank4.safetensors"),
                    reason="shipped fixture not mounted")
def test_patch_shipped_fixture_into_standin_and_resave(tmp_path):
    from safetensors import safe_open

    path = "/root/reference/example_loras/analog_svd_rank4.safetensors"
    with torch.device("meta"):
        unet, clip = sd15_unet(), clip_text_model()

    class Pipe:
        pass

    pipe = Pipe()
    pipe.unet, pipe.text_encoder = unet, clip
    parsed = L.load_safeloras(path)
    assert parsed["unet"][1] == [4] * 144 and parsed["text_encoder"][1] == [4] * 48
    # keep real (CPU) factors: patch with the frozen weights on meta is fine for structure checks
    L.monkeypatch_or_replace_safeloras(pipe, safe_open(path, framework="pt", device="cpu"))
    ads = [m for m in unet.modules() if isinstance(m, L.LoraInjectedLinear)]
    assert len(ads) == 144 and all(a.r == 4 for a in ads)
    assert len([m for m in clip.modules() if isinstance(m, L.LoraInjectedLinear)]) == 48
    # re-save from the parsed tensors: byte-identical tensor payloads and metadata (targets compared as sets)
    src = safe_open(path, framework="pt", device="cpu")
    weights = {k: src.get_tensor(k) for k in src.keys()}
    L.safe_save(weights, str(tmp_path / "resaved.safetensors"), src.metadata())
    dst = safe_open(str(tmp_path / "resaved.safetensors"), framework="pt", device="cpu")
    assert list(dst.keys()) == list(src.keys()) and dst.metadata() == src.metadata()
    for k in list(src.keys())[::17]:
        assert torch.equal(dst.get_tensor(k), src.get_tensor(k))


def test_scheduler_matches_oracle():
    s = DDPMScheduler()
    a = O.ddpm_alphas_cumprod()
    assert torch.allclose(s.alphas_cumprod, torch.from_numpy(a), rtol=1e-6)
    x, n = torch.randn(3, 4, 8, 8), torch.randn(3, 4, 8, 8)
    t = torch.tensor([0, 500, 999])
    got = s.add_noise(x, n, t)
    assert torch.allclose(got, torch.from_numpy(O.add_noise(x.numpy(), n.numpy(), t.numpy(), a)), rtol=1e-5, atol=1e-6)
    assert s.config.num_train_timesteps == 1000 and s.config.prediction_type == "epsilon"


def test_tiny_unet_runs_and_checkpoints():
    torch.manual_seed(0)
    u = tiny_unet()
    x, ctx, t = torch.randn(2, 4, 16, 16), torch.randn(2, 7, 32), torch.tensor([1, 900])
    y = u(x, t, ctx).sample
    assert y.shape == x.shape
    u.enable_gradient_checkpointing()
    u.train()
    L.inject_trainable_lora(u, r=2)
    y2 = u(x, t, ctx).sample
    assert torch.allclose(y, y2, atol=1e-5)  # up = 0 at init: adapters are the identity (ref:51)
    y2.sum().backward()


def test_transformer2d_token_major_projection_equals_the_1x1_convolutions():
    """proj_in / proj_out run as token-major GEMMs on NCHW activations; channels_last keeps the convolutions: same map."""
    import torch.nn.functional as F
    from lora_amd.standin.unet import Transformer2DModel

    torch.manual_seed(0)
    m = Transformer2DModel(32, 2, 16, groups=8)
    x = torch.randn(2, 32, 4, 6, requires_grad=True)
    ctx = torch.randn(2, 5, 16)
    y1 = m(x, ctx)
    (g1,) = torch.autograd.grad(y1.square().sum(), x)
    yc = m(x.contiguous(memory_format=torch.channels_last), ctx)  # token view is free on NHWC activations
    assert yc.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(y1, yc.contiguous(), rtol=1e-5, atol=1e-5)

    class Conv1x1(nn.Conv2d):  # any subclass keeps the module call (what an injected adapter looks like)
        pass

    for name in ("proj_in", "proj_out"):
        old = getattr(m, name)
        new = Conv1x1(32, 32, 1)
        new.load_state_dict(old.state_dict())
        setattr(m, name, new)
    y0 = m(x, ctx)
    (g0,) = torch.autograd.grad(y0.square().sum(), x)
    torch.testing.assert_close(y1, y0, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g1, g0, rtol=1e-4, atol=1e-5)
    # an adapter injected into the projection (custom target class) keeps the module call
    torch.manual_seed(0)
    m2 = Transformer2DModel(32, 2, 16, groups=8)
    L.inject_trainable_lora_extended(m2, target_replace_module={"Transformer2DModel"}, r=2)
    assert type(m2.proj_in).__name__ == "LoraInjectedConv2d"
    y2 = m2(x, ctx)  # same seed, lora_up zero-initialised: same output through the module path
    torch.testing.assert_close(y2, y1, rtol=1e-5, atol=1e-5)
    del F


def test_fused_host_passes_fall_back_to_aten_on_cpu():
    import torch.nn.functional as F
    from lora_amd.standin import fused

    x = torch.randn(2, 16, 4, 4)
    gn = nn.GroupNorm(4, 16).requires_grad_(False)
    torch.testing.assert_close(fused.group_norm_act(x, gn, True), F.silu(gn(x)))
    torch.testing.assert_close(fused.group_norm_act(x, gn, False), gn(x))
    ln = nn.LayerNorm(16).requires_grad_(False)
    t = torch.randn(3, 5, 16)
    torch.testing.assert_close(fused.layer_norm(t, ln), ln(t))
    y = torch.randn(3, 5, 32)
    h, g = y.chunk(2, dim=-1)
    torch.testing.assert_close(fused.geglu(y), h * F.gelu(g))


def test_resnet_block_folds_conv_bias_and_time_embedding_into_one_addend():
    import torch.nn.functional as F
    from lora_amd.standin.unet import ResnetBlock2D

    torch.manual_seed(0)
    blk = ResnetBlock2D(16, 32, temb_channels=24, groups=4)
    x, temb = torch.randn(2, 16, 6, 6, requires_grad=True), torch.randn(2, 24)
    y = blk(x, temb)
    h = blk.conv1(F.silu(blk.norm1(x)))  # the reference order of operations (diffusers ResnetBlock2D)
    h = h + blk.time_emb_proj(F.silu(temb))[:, :, None, None]
    h = blk.conv2(F.silu(blk.norm2(h)))
    want = blk.conv_shortcut(x) + h
    torch.testing.assert_close(y, want, rtol=1e-5, atol=1e-5)
    (g1,) = torch.autograd.grad(y.square().sum(), x, retain_graph=True)
    (g0,) = torch.autograd.grad(want.square().sum(), x)
    torch.testing.assert_close(g1, g0, rtol=1e-4, atol=1e-5)
    # channels_last input: same numbers (ATen fallbacks on CPU)
    yc = blk(x.contiguous(memory_format=torch.channels_last), temb)
    torch.testing.assert_close(yc.contiguous(), want, rtol=1e-5, atol=1e-5)


def test_attention_padded_head_layout_plumbing_equals_regular_path():
    """The padded-head-layout route through CrossAttention (projections via ``forward_heads``, core on padded tensors)
    gives the regular route's output and gradients; on CPU the layout conversions are dense copies."""
    from lora_amd import ops
    from lora_amd.standin import attention
    from lora_amd.standin.unet import CrossAttention

    torch.manual_seed(0)
    y = torch.randn(2, 5, 8 * 40)
    p = ops.pack_heads(y, (8, 40, 64))
    assert p.shape == (2, 5, 512) and torch.equal(ops.unpack_heads(p, (8, 40, 64)), y)
    assert float(p.view(2, 5, 8, 64)[..., 40:].abs().max()) == 0
    for ctx_dim, inject in ((None, True), (48, True), (None, False)):
        att = CrossAttention(64, ctx_dim, heads=4, dim_head=24)
        if inject:
            L.inject_trainable_lora(att, target_replace_module={"CrossAttention"}, r=4)
            for m in att.modules():
                if type(m).__name__ == "LoraInjectedLinear":
                    nn.init.normal_(m.lora_up.weight, std=0.1)
        x = torch.randn(2, 9, 64, requires_grad=True)
        c = None if ctx_dim is None else torch.randn(2, 7, 48)
        leaves = [x] + [q for q in att.parameters() if q.requires_grad]
        try:
            attention.FORCE_PAD = None
            y0 = att(x, c)
            g0 = torch.autograd.grad(y0.square().sum(), leaves)
            attention.FORCE_PAD = 32
            y1 = att(x, c)
            g1 = torch.autograd.grad(y1.square().sum(), leaves)
        finally:
            attention.FORCE_PAD = None
        torch.testing.assert_close(y1, y0, rtol=1e-5, atol=1e-6)
        for a, b in zip(g1, g0):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


def test_unet_channels_last_equals_nchw():
    """The whole stand-in UNet on channels_last activations (token views instead of copies in Transformer2DModel,
    GroupNorm addend, add+LayerNorm) computes what it computes on NCHW; on CPU every fused pass is its ATen fallback."""
    torch.manual_seed(0)
    unet = tiny_unet()
    L.inject_trainable_lora(unet, r=2)
    for m in unet.modules():
        if type(m).__name__ == "LoraInjectedLinear":
            nn.init.normal_(m.lora_up.weight, std=0.05)
    lat, t, ctx = torch.randn(2, 4, 16, 16), torch.tensor([3, 700]), torch.randn(2, 7, 32)
    leaves = [p for p in unet.parameters() if p.requires_grad]
    y0 = unet(lat, t, ctx).sample
    g0 = torch.autograd.grad(y0.square().mean(), leaves)
    unet.to(memory_format=torch.channels_last)
    y1 = unet(lat.contiguous(memory_format=torch.channels_last), t, ctx).sample
    g1 = torch.autograd.grad(y1.square().mean(), leaves)
    torch.testing.assert_close(y1.contiguous(), y0, rtol=1e-4, atol=1e-5)
    for a, b in zip(g1, g0):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-6)


def test_frozen_twins_sit_on_the_adapter_sites_and_change_nothing_on_cpu():
    """standin/frozen.py (bench.py's `frozen_only` leg): a FrozenSite on exactly the modules inject_trainable_lora adapts
    (144 on the SD1.5 shape, 48 on CLIP), same parameter objects, and on CPU the model computes what it computed before."""
    from lora_amd.standin.frozen import FrozenSite, install_frozen_twins

    with torch.device("meta"):
        big = sd15_unet()
    assert install_frozen_twins(big) == 144
    assert install_frozen_twins(big) == 0   # idempotent: children of a twin are not sites
    torch.manual_seed(0)
    unet = tiny_unet()
    x, t, c = torch.randn(2, 4, 16, 16), torch.tensor([3, 500]), torch.randn(2, 7, 32)
    want = unet(x, t, c).sample
    w0 = unet.down_blocks[0].attentions[0].transformer_blocks[0].attn1.to_q.weight
    n_sites = install_frozen_twins(unet)
    twin = unet.down_blocks[0].attentions[0].transformer_blocks[0].attn1.to_q
    assert isinstance(twin, FrozenSite) and twin.linear.weight is w0
    ref = tiny_unet()
    assert n_sites == len(L.inject_trainable_lora(ref)[1]) // 2 or n_sites == sum(
        isinstance(m, L.LoraInjectedLinear) for m in ref.modules())
    torch.testing.assert_close(unet(x, t, c).sample, want)
