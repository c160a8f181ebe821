"""Round-2 host-side additions: the torch-only safetensors reader, fp16 loss scaling, lr floor of the polynomial
schedule, cache invalidation, console entry points."""
import json
import os

import numpy as np
import pytest
import torch

import lora_amd as L
from lora_amd import _C, trainer as T
from lora_amd.safe_open import safe_open as py_safe_open
from tests import helpers as H

G = H.GOLDEN


def test_pure_python_safe_open_equals_safetensors_on_reference_file():
    """lora_amd/safe_open.py (role of lora_diffusion/safe_open.py:1-68) against the real library on a file the
    reference wrote, and through parse_safeloras (ref lora.py:538-596)."""
    from safetensors import safe_open

    path = os.path.join(G, "mini_ref.safetensors")
    a, b = py_safe_open(path, framework="pt", device="cpu"), safe_open(path, framework="pt", device="cpu")
    assert sorted(a.keys()) == sorted(b.keys()) and dict(a.metadata()) == dict(b.metadata())
    for k in b.keys():
        ta, tb = a.get_tensor(k), b.get_tensor(k)
        assert ta.dtype == tb.dtype and ta.shape == tb.shape and torch.equal(ta, tb), k
    pa, pb = L.parse_safeloras(a), L.parse_safeloras(b)
    assert set(pa) == set(pb)
    for name in pa:
        assert pa[name][1] == pb[name][1] and sorted(pa[name][2]) == sorted(pb[name][2])
        assert all(torch.equal(x, y) for x, y in zip(pa[name][0], pb[name][0]))
    assert set(L.parse_safeloras_embeds(a)) == set(L.parse_safeloras_embeds(b))
    with pytest.raises(ValueError):
        py_safe_open(path, framework="np")


def test_pure_python_safe_open_rejects_truncated_files(tmp_path):
    p = tmp_path / "bad.safetensors"
    hdr = json.dumps({"x": {"dtype": "F32", "shape": [4], "data_offsets": [0, 16]}}).encode()
    p.write_bytes(len(hdr).to_bytes(8, "little") + hdr + b"\0" * 8)  # payload shorter than the header says
    with pytest.raises(ValueError):
        py_safe_open(str(p))
    p.write_bytes(b"\1\2")
    with pytest.raises(ValueError):
        py_safe_open(str(p))


def test_polynomial_schedule_floor_is_absolute_lr_end():
    """diffusers get_polynomial_decay_schedule_with_warmup: lr decays to lr_end = 1e-7 ABSOLUTE (ref :737-742 calls
    get_scheduler, un-vendored), i.e. the multiplier floor is 1e-7 / lr_init."""
    f = T.get_lr_lambda("polynomial", 2, 10, lr_init=1e-4)
    assert f(0) == 0.0 and f(1) == 0.5
    assert abs(f(2) - 1.0) < 1e-12 and abs(f(6) - ((1 - 1e-3) * 0.5 + 1e-3)) < 1e-12
    assert abs(f(10) - 1e-3) < 1e-12 and abs(f(11) - 1e-3) < 1e-12
    assert abs(1e-4 * f(11) - 1e-7) < 1e-15


def test_loss_scaling_cpu_path_skips_nonfinite_steps_and_adapts():
    p = torch.nn.Parameter(torch.ones(16))
    st = T.FlatLoraState([{"params": [p], "lr": 1e-2, "weight_decay": 0.0}], max_grad_norm=0.0)
    sc = st.enable_loss_scaling(init_scale=8.0, growth_interval=2)
    assert float(st.loss_scale) == 8.0
    # a scaled gradient: the update must equal the one of the un-scaled gradient
    ref = torch.nn.Parameter(torch.ones(16))
    st_ref = T.FlatLoraState([{"params": [ref], "lr": 1e-2, "weight_decay": 0.0}], max_grad_norm=0.0)
    g = torch.linspace(-1, 1, 16)
    st.flat_g.copy_(g * 8.0), st_ref.flat_g.copy_(g)
    st.step(), st_ref.step()
    np.testing.assert_allclose(st.flat_p.numpy(), st_ref.flat_p.numpy(), rtol=1e-6)
    assert sc.tolist()[:2] == [8.0, 1.0]
    # an overflowed backward: parameters and moments untouched, scale halves, the step does not count
    before, m_before = st.flat_p.clone(), st.exp_avg.clone()
    st.flat_g.copy_(g * 8.0)
    st.flat_g[3] = float("inf")
    st.step()
    assert torch.equal(st.flat_p, before) and torch.equal(st.exp_avg, m_before) and st.step_count == 1
    assert sc.tolist()[0] == 4.0 and sc.tolist()[3] == 0.0 and float(st.flat_g.abs().sum()) == 0.0
    # two finite steps in a row (growth_interval=2): scale doubles
    for _ in range(2):
        st.flat_g.copy_(g * float(st.loss_scale))
        st.step()
    assert sc.tolist()[0] == 8.0 and st.step_count == 3


def test_shadow_cache_cannot_alias_a_replaced_parameter():
    """a compute-dtype shadow keyed on (data_ptr, version) alone can hit a NEW Parameter that the allocator
    placed at the freed address; the entry now also requires the same tensor object."""
    m = L.LoraInjectedLinear(8, 8, r=2)
    w0 = m.linear.weight
    w0.requires_grad_(False)
    s0 = m._shadow(w0, torch.bfloat16, "w")
    assert m._shadow(w0, torch.bfloat16, "w") is s0  # resident
    twin = torch.nn.Parameter(w0.data, requires_grad=False)  # same storage, same version, different object
    twin.data = w0.data
    assert twin.data_ptr() == w0.data_ptr()
    assert m._shadow(twin, torch.bfloat16, "w") is not s0
    m.to(torch.float64)  # Module._apply drops the derived layouts
    assert "_shadow_cache" not in m.__dict__
    L.invalidate_caches(m)  # public hook for in-place edits through .data


def test_console_entry_points_match_reference_setup():
    """ref setup.py:14-21: lora_add / lora_pti / lora_distill (lora_ppim = preprocessing, out of scope)."""
    import importlib

    src = open(os.path.join(H.REPO, "pyproject.toml")).read()
    block = src.split("[project.scripts]")[1].split("[")[0]
    entries = dict(line.replace('"', "").split(" = ") for line in block.strip().splitlines())
    assert set(entries) == {"lora_add", "lora_pti", "lora_distill"}
    for target in entries.values():
        mod, fn = target.split(":")
        assert mod.split(".")[-1] in ("cli_lora_add", "cli_lora_pti", "cli_svd")
        assert callable(getattr(importlib.import_module(mod), fn))


def test_load_host_models_real_branch_with_fake_diffusers(tmp_path, monkeypatch):
    """standin.io.load_host_models takes its diffusers branch when the package is importable and the path is a
    directory (ref train_lora_dreambooth.py:566-594, 678-680): exercised with a fake ``diffusers`` whose classes hand
    back small modules, incl. an attention block with the processor API."""
    import sys
    import types

    import torch.nn as nn

    from lora_amd.diffusers_glue import LoraAmdAttnProcessor
    from lora_amd.standin import io as SIO

    class Attention(nn.Module):  # the shape of diffusers.models.attention_processor.Attention that matters here
        def __init__(self, dim=16, heads=2, ctx=None):
            super().__init__()
            self.heads, self.scale = heads, (dim // heads) ** -0.5
            self.to_q = nn.Linear(dim, dim, bias=False)
            self.to_k = nn.Linear(ctx or dim, dim, bias=False)
            self.to_v = nn.Linear(ctx or dim, dim, bias=False)
            self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
            self.processor = None

        def set_processor(self, p):
            self.processor = p

        def forward(self, x, encoder_hidden_states=None):
            return self.processor(self, x, encoder_hidden_states)

    class FakeUNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.attn1, self.attn2 = Attention(), Attention(ctx=8)

        @classmethod
        def from_pretrained(cls, path, subfolder=None, revision=None):
            return cls()

    class Loader:
        @classmethod
        def from_pretrained(cls, *a, **k):
            return nn.Identity()

        @classmethod
        def from_config(cls, *a, **k):
            return "sched"

    fake = types.ModuleType("diffusers")
    fake.AutoencoderKL, fake.DDPMScheduler, fake.UNet2DConditionModel = Loader, Loader, FakeUNet
    monkeypatch.setitem(sys.modules, "diffusers", fake)
    import transformers

    monkeypatch.setattr(transformers.CLIPTokenizer, "from_pretrained", classmethod(lambda cls, *a, **k: "tok"))
    monkeypatch.setattr(transformers.CLIPTextModel, "from_pretrained", classmethod(lambda cls, *a, **k: nn.Identity()))
    tok, te, vae, unet, sched, what = SIO.load_host_models(str(tmp_path), None, None, None, "cpu")
    assert tok == "tok" and sched == "sched" and isinstance(unet, FakeUNet) and "2 attention blocks" in what
    assert isinstance(unet.attn1.processor, LoraAmdAttnProcessor)
    # the processor computes plain multi-head attention (self and cross), also with adapters injected (CPU: module by module)
    torch.manual_seed(0)
    x, ctx = torch.randn(2, 5, 16), torch.randn(2, 3, 8)

    def ref(a, x, c):
        q, k, v = a.to_q(x), a.to_k(c), a.to_v(c)
        B, T, _ = x.shape
        q, k, v = (t.view(B, t.shape[1], a.heads, -1).transpose(1, 2) for t in (q, k, v))
        o = torch.softmax(q @ k.transpose(-1, -2) * a.scale, -1) @ v
        return a.to_out[0](o.transpose(1, 2).reshape(B, T, -1))

    assert torch.allclose(unet.attn1(x), ref(unet.attn1, x, x), atol=1e-5)
    assert torch.allclose(unet.attn2(x, ctx), ref(unet.attn2, x, ctx), atol=1e-5)
    type(unet.attn1).__name__ = "Attention"
    L.inject_trainable_lora(unet, r=2)
    assert isinstance(unet.attn1.to_q, L.LoraInjectedLinear)
    assert torch.allclose(unet.attn1(x), ref(unet.attn1, x, x), atol=1e-5)


def test_conv_layout_routing_is_decided_from_strides_only():
    """ops.lora_conv picks the channels-last kernels from the memory format (no device needed to decide): NHWC-only
    tensors at 1x1 / 3x3 stride-1 sites in 16-bit; everything else keeps the NCHW kernels or the library-conv branch."""
    import torch
    from lora_amd import ops

    x = torch.zeros(2, 64, 8, 8, dtype=torch.bfloat16)
    xl = x.contiguous(memory_format=torch.channels_last)
    w3, w1 = torch.zeros(32, 64, 3, 3, dtype=torch.bfloat16), torch.zeros(32, 64, 1, 1, dtype=torch.bfloat16)
    geom3, geom1 = ((1, 1), (1, 1), (1, 1), 1), ((1, 1), (0, 0), (1, 1), 1)
    assert not ops.conv_nhwc_ok(x, w3, 4, *geom3)  # plain NCHW
    assert ops.conv_nhwc_ok(xl, w3, 4, *geom3) and ops.conv_nhwc_ok(xl, w1, 4, *geom1)
    assert not ops.conv_nhwc_ok(xl, w3, 3, *geom3)  # rank 3: not a multiple of 4 -> NCHW kernels
    assert ops.conv_nhwc_ok(xl, w1, 3, *geom1)      # 1x1 = the Linear adapter: any rank
    assert not ops.conv_nhwc_ok(xl, w3, 4, (2, 2), (1, 1), (1, 1), 1)  # strided (Downsample2D)
    assert not ops.conv_nhwc_ok(xl.float(), w3.float(), 4, *geom3)     # f32 activations
    one = torch.zeros(2, 64, 1, 1, dtype=torch.bfloat16)  # H = W = 1: both formats at once -> NCHW path
    assert not ops.conv_nhwc_ok(one, w1, 4, *geom1)
    assert ops.WS_DROPOUT and ops.WS_DROPOUT_WIDE and ops.WS_DROPOUT_WIDE_BWD  # defaults
    # the one A/B switch: LORA_AMD_AB flips module constants by name, and nothing else
    ns = {"CONCAT_GROUPS": True, "WS_DROPOUT": True}
    assert ops.apply_ab_overrides("CONCAT_GROUPS=0, WS_DROPOUT=1", ns) == {"CONCAT_GROUPS": False, "WS_DROPOUT": True}
    assert ns == {"CONCAT_GROUPS": False, "WS_DROPOUT": True}
    with pytest.raises(ValueError):
        ops.apply_ab_overrides("NOT_A_CONSTANT=1", {})


def test_weight_cache_keys_accept_inference_tensors():
    """tensors created under torch.inference_mode() do not track a version counter (reading ``_version``
    raises); the layout caches must key them anyway."""
    from lora_amd import _C

    with torch.inference_mode():
        w = torch.randn(8, 8)
        assert _C._tensor_version(w) == 0
    assert _C._tensor_version(torch.randn(2, 2)) == 0
    v = torch.randn(2, 2)
    v.add_(1)
    assert _C._tensor_version(v) == 1
    import lora_amd as L

    m = L.LoraInjectedLinear(8, 8, r=2)
    with torch.inference_mode():
        sh = m._shadow(torch.randn(8, 8), torch.bfloat16, "w")  # shadow built from an inference tensor
        assert sh.dtype == torch.bfloat16


def test_diffusers_processor_hands_back_variants_it_does_not_model():
    """qk-norm / upcast / fused-projection attention blocks must go to the previous processor, never be approximated."""
    import types

    from lora_amd.diffusers_glue import LoraAmdAttnProcessor

    calls = []
    proc = LoraAmdAttnProcessor(lambda *a, **k: calls.append(1) or "fallback")
    x = torch.randn(1, 3, 8)
    for extra in ({"norm_q": torch.nn.LayerNorm(4)}, {"upcast_softmax": True}, {"fused_projections": True},
                  {"upcast_attention": True}, {"norm_k": torch.nn.LayerNorm(4)}):
        attn = types.SimpleNamespace(heads=2, **extra)
        assert proc(attn, x) == "fallback"
    assert len(calls) == 5


def test_host_options_bind_into_fake_diffusers_blocks_and_keep_the_maths():
    """the stand-in's frozen-host passes (GroupNorm+SiLU with the time-embedding addend, residual
    add + LayerNorm, GEGLU, tuned attention processor) bound into ``diffusers``-shaped blocks by
    ``diffusers_glue.install_host_options``: same outputs and gradients as the blocks' own forward (CPU: the bound passes
    take their ATen sequences), unsupported variants keep their original forward."""
    import torch.nn as nn
    import torch.nn.functional as F

    from lora_amd.diffusers_glue import LoraAmdAttnProcessor, install_host_options

    class Attention(nn.Module):
        def __init__(self, dim=32, heads=2, ctx=None):
            super().__init__()
            self.heads, self.scale = heads, (dim // heads) ** -0.5
            self.to_q = nn.Linear(dim, dim, bias=False)
            self.to_k = nn.Linear(ctx or dim, dim, bias=False)
            self.to_v = nn.Linear(ctx or dim, dim, bias=False)
            self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
            self.processor = default_processor  # diffusers blocks always carry one (AttnProcessor / AttnProcessor2_0)

        def set_processor(self, p):
            self.processor = p

        def forward(self, x, encoder_hidden_states=None, attention_mask=None, **kw):
            return self.processor(self, x, encoder_hidden_states, attention_mask)

    def default_processor(attn, x, encoder_hidden_states=None, attention_mask=None, temb=None, *a, **k):
        c = x if encoder_hidden_states is None else encoder_hidden_states
        B, T, _ = x.shape
        q, k_, v = (t.view(B, t.shape[1], attn.heads, -1).transpose(1, 2) for t in (attn.to_q(x), attn.to_k(c), attn.to_v(c)))
        s_ = q @ k_.transpose(-1, -2) * attn.scale
        if attention_mask is not None:
            s_ = s_ + attention_mask[:, None]
        return attn.to_out[1](attn.to_out[0]((torch.softmax(s_, -1) @ v).transpose(1, 2).reshape(B, T, -1)))

    class GEGLU(nn.Module):  # diffusers.models.activations.GEGLU
        def __init__(self, dim_in, dim_out):
            super().__init__()
            self.proj = nn.Linear(dim_in, dim_out * 2)

        def forward(self, hidden_states, *a, **k):
            hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
            return hidden_states * F.gelu(gate)

    class FeedForward(nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

        def forward(self, x):
            for m in self.net:
                x = m(x)
            return x

    class BasicTransformerBlock(nn.Module):  # diffusers.models.attention.BasicTransformerBlock, SD1.x configuration
        def __init__(self, dim=32, ctx=16, ada=False):
            super().__init__()
            self.norm1, self.attn1 = nn.LayerNorm(dim), Attention(dim)
            self.norm2, self.attn2 = nn.LayerNorm(dim), Attention(dim, ctx=ctx)
            self.norm3, self.ff = nn.LayerNorm(dim), FeedForward(dim)
            self.use_ada_layer_norm, self.only_cross_attention, self.pos_embed, self._chunk_size = ada, False, None, None
            self.norm_type = "layer_norm"

        def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                    timestep=None, cross_attention_kwargs=None, class_labels=None, added_cond_kwargs=None):
            x = hidden_states
            x = self.attn1(self.norm1(x), attention_mask=attention_mask) + x
            x = self.attn2(self.norm2(x), encoder_hidden_states=encoder_hidden_states) + x
            return self.ff(self.norm3(x)) + x

    class ResnetBlock2D(nn.Module):  # diffusers.models.resnet.ResnetBlock2D
        def __init__(self, cin=32, cout=64, temb=24, norm="default", osf=1.0):
            super().__init__()
            self.norm1, self.conv1 = nn.GroupNorm(8, cin), nn.Conv2d(cin, cout, 3, padding=1)
            self.time_emb_proj = nn.Linear(temb, cout)
            self.norm2, self.dropout, self.conv2 = nn.GroupNorm(8, cout), nn.Dropout(0.0), nn.Conv2d(cout, cout, 3, padding=1)
            self.nonlinearity = nn.SiLU()
            self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None
            self.time_embedding_norm, self.up, self.down, self.upsample, self.downsample = norm, False, False, None, None
            self.output_scale_factor = osf

        def forward(self, input_tensor, temb, *a, **k):
            h = self.conv1(self.nonlinearity(self.norm1(input_tensor)))
            h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
            h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
            sc = input_tensor if self.conv_shortcut is None else self.conv_shortcut(input_tensor)
            return (sc + h) / self.output_scale_factor

    class FakeUNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.res = ResnetBlock2D(osf=2.0)
            self.res_odd = ResnetBlock2D(norm="scale_shift")
            self.blk = BasicTransformerBlock()
            self.blk_ada = BasicTransformerBlock(ada=True)

    torch.manual_seed(0)
    unet = FakeUNet()
    x4, temb = torch.randn(2, 32, 6, 6, requires_grad=True), torch.randn(2, 24)
    xt, ctx = torch.randn(2, 9, 32, requires_grad=True), torch.randn(2, 5, 16)

    def run():
        outs = [unet.res(x4, temb), unet.blk(xt, encoder_hidden_states=ctx), unet.blk_ada(xt, encoder_hidden_states=ctx),
                unet.res_odd(x4, temb)]
        grads = torch.autograd.grad(sum(o.square().sum() for o in outs), [x4, xt])
        return [o.detach() for o in outs], grads

    want, gwant = run()
    counts = install_host_options(unet)
    assert counts == {"attention": 4, "transformer_block": 1, "resnet": 1, "geglu": 2}, counts
    assert "_lora_amd_forward" in unet.res.__dict__ and "_lora_amd_forward" not in unet.res_odd.__dict__
    assert "_lora_amd_forward" not in unet.blk_ada.__dict__
    assert isinstance(unet.blk.attn1.processor, LoraAmdAttnProcessor) and unet.blk.attn1.processor.tuned
    got, ggot = run()
    for a, b in zip(want + list(gwant), got + list(ggot)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)
    assert install_host_options(unet) == {"attention": 4, "transformer_block": 0, "resnet": 0, "geglu": 0}  # idempotent
    # a masked call of a bound block goes to the block's own forward
    m = torch.zeros(2, 9, 9)
    assert torch.allclose(unet.blk(xt, attention_mask=m, encoder_hidden_states=ctx).detach(),
                          unet.blk._lora_amd_forward(xt, attention_mask=m, encoder_hidden_states=ctx).detach())
    # adapters injected afterwards keep working through the bound blocks
    type(unet.blk.attn1).__name__ = "Attention"
    L.inject_trainable_lora(unet, r=2)
    assert isinstance(unet.blk.ff.net[0].proj, L.LoraInjectedLinear)
    got2, _ = run()
    for a, b in zip(want, got2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)  # up = 0 at injection: same function
