#!/usr/bin/env python
"""Generate tests/golden/* by RUNNING the real reference (/root/reference/lora_diffusion/lora.py).

Run in the build container (the reference is not available on the GPU box):
    python scripts/make_golden.py
The outputs are small, committed fixtures; tests compare the oracle and lora_amd against them.
Nothing here is copied from the reference: it is imported by path and executed.
"""
from __future__ import annotations

import hashlib
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests import helpers as H  # noqa: E402

OUT = H.GOLDEN
ref = H.load_reference()
torch.set_num_threads(1)  # fixed accumulation order in the reference's CPU GEMMs


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def linear_cases():
    out = {}
    cases = [("a", 6, 16, 24, 4, True, 1.0, False), ("b", 5, 8, 8, 1, False, 0.5, False),
             ("c", 7, 32, 16, 8, True, 2.0, True), ("d", 33, 40, 24, 4, True, 0.3, False)]
    for tag, M, K, N, r, bias, scale, diag in cases:
        torch.manual_seed(hash(tag) % 1000 + 1 if False else ord(tag))
        m = ref.LoraInjectedLinear(K, N, bias, r=r, dropout_p=0.0, scale=scale)
        m.lora_up.weight.data.normal_(0, 0.2)
        if diag:
            m.set_selector_from_diag(torch.linspace(0.5, 1.5, r))
        x = torch.randn(M, K, requires_grad=True)
        gy = torch.randn(M, N)
        y = m(x)
        (y * gy).sum().backward()
        out.update({f"{tag}_x": H.t2n(x), f"{tag}_W": H.t2n(m.linear.weight), f"{tag}_down": H.t2n(m.lora_down.weight),
                    f"{tag}_up": H.t2n(m.lora_up.weight), f"{tag}_gy": H.t2n(gy), f"{tag}_y": H.t2n(y),
                    f"{tag}_dx": H.t2n(x.grad), f"{tag}_ddown": H.t2n(m.lora_down.weight.grad),
                    f"{tag}_dup": H.t2n(m.lora_up.weight.grad), f"{tag}_scale": np.float32(scale)})
        if bias:
            out[f"{tag}_b"] = H.t2n(m.linear.bias)
        if diag:
            out[f"{tag}_sel"] = H.t2n(m.selector.weight)
    np.savez(os.path.join(OUT, "linear_cases.npz"), **out)


def conv_cases():
    out = {}
    cases = [("a", 2, 8, 12, 6, 6, 3, 1, 1, 4, 1.0), ("b", 1, 4, 8, 5, 7, 1, 1, 0, 2, 0.5),
             ("c", 2, 8, 8, 8, 8, 3, 2, 1, 4, 1.5)]
    for tag, B, Ci, Co, Hh, Ww, k, s, p, r, scale in cases:
        torch.manual_seed(100 + ord(tag))
        m = ref.LoraInjectedConv2d(Ci, Co, k, s, p, r=r, dropout_p=0.0, scale=scale)
        m.lora_up.weight.data.normal_(0, 0.2)
        x = torch.randn(B, Ci, Hh, Ww, requires_grad=True)
        y = m(x)
        gy = torch.randn_like(y)
        (y * gy).sum().backward()
        out.update({f"{tag}_x": H.t2n(x), f"{tag}_W": H.t2n(m.conv.weight), f"{tag}_b": H.t2n(m.conv.bias),
                    f"{tag}_down": H.t2n(m.lora_down.weight), f"{tag}_up": H.t2n(m.lora_up.weight),
                    f"{tag}_gy": H.t2n(gy), f"{tag}_y": H.t2n(y), f"{tag}_dx": H.t2n(x.grad),
                    f"{tag}_ddown": H.t2n(m.lora_down.weight.grad), f"{tag}_dup": H.t2n(m.lora_up.weight.grad),
                    f"{tag}_geom": np.array([k, s, p, r], dtype=np.int64), f"{tag}_scale": np.float32(scale)})
    np.savez(os.path.join(OUT, "conv_cases.npz"), **out)


def conv_native_cases():
    """LoraInjectedConv2d on geometries the native HIP path accepts (stride 1, same padding, W % 8 == 0)."""
    out = {}
    cases = [("n1", 2, 16, 24, 8, 16, 3, 4, 1.0), ("n2", 3, 20, 12, 4, 8, 1, 2, 0.7),
             ("n3", 1, 24, 16, 16, 24, 3, 8, 1.3), ("n4", 2, 12, 40, 8, 8, 3, 6, 0.9),
             ("n5", 5, 9, 17, 2, 8, 1, 5, 1.1)]
    for tag, B, Ci, Co, Hh, Ww, k, r, scale in cases:
        torch.manual_seed(300 + int(tag[1]))
        m = ref.LoraInjectedConv2d(Ci, Co, k, 1, (k - 1) // 2, r=r, dropout_p=0.0, scale=scale)
        m.lora_up.weight.data.normal_(0, 0.2)
        x = torch.randn(B, Ci, Hh, Ww, requires_grad=True)
        y = m(x)
        gy = torch.randn_like(y)
        (y * gy).sum().backward()
        out.update({f"{tag}_x": H.t2n(x), f"{tag}_W": H.t2n(m.conv.weight), f"{tag}_b": H.t2n(m.conv.bias),
                    f"{tag}_down": H.t2n(m.lora_down.weight), f"{tag}_up": H.t2n(m.lora_up.weight),
                    f"{tag}_gy": H.t2n(gy), f"{tag}_y": H.t2n(y), f"{tag}_dx": H.t2n(x.grad),
                    f"{tag}_ddown": H.t2n(m.lora_down.weight.grad), f"{tag}_dup": H.t2n(m.lora_up.weight.grad),
                    f"{tag}_geom": np.array([k, 1, (k - 1) // 2, r], dtype=np.int64), f"{tag}_scale": np.float32(scale)})
    np.savez(os.path.join(OUT, "conv_native_cases.npz"), **out)


def collapse_cases():
    """collapse_lora (ref:635-669) on single-site trees, several dtype combinations."""
    out = {}
    cases = [("lin_f32", "linear", "f32", "f32", 1.0), ("lin_f32_a", "linear", "f32", "f32", 0.7),
             ("lin_bf16_f32", "linear", "bf16", "f32", 1.0), ("lin_bf16_bf16", "linear", "bf16", "bf16", 0.6),
             ("lin_f16_f16", "linear", "f16", "f16", 1.3), ("conv_f32", "conv", "f32", "f32", 0.9),
             ("conv_bf16_f32", "conv", "bf16", "f32", 1.0)]
    Holder = H.named_class("CrossAttention")
    RHolder = H.named_class("ResnetBlock2D")
    for i, (tag, kind, wdt, abdt, alpha) in enumerate(cases):
        torch.manual_seed(200 + i)
        if kind == "linear":
            m = ref.LoraInjectedLinear(40, 24, False, r=4, scale=3.0)  # scale must be ignored
            frozen = m.linear
            root = Holder()
        else:
            m = ref.LoraInjectedConv2d(8, 12, 3, 1, 1, r=4, scale=3.0)
            frozen = m.conv
            root = RHolder()
        m.lora_up.weight.data.normal_(0, 0.3)
        frozen.weight.data = frozen.weight.data.to(H.TORCH_DT[wdt])
        m.lora_up.weight.data = m.lora_up.weight.data.to(H.TORCH_DT[abdt])
        m.lora_down.weight.data = m.lora_down.weight.data.to(H.TORCH_DT[abdt])
        root.add_module("site", m)
        w0 = H.t2n(frozen.weight)
        quiet(ref.collapse_lora, root, alpha)
        assert frozen.weight.dtype == H.TORCH_DT[wdt]
        out.update({f"{tag}_W": w0, f"{tag}_up": H.t2n(m.lora_up.weight), f"{tag}_down": H.t2n(m.lora_down.weight),
                    f"{tag}_out": H.t2n(frozen.weight), f"{tag}_alpha": np.float32(alpha)})
    np.savez(os.path.join(OUT, "collapse_cases.npz"), **out)
    with open(os.path.join(OUT, "collapse_cases.json"), "w") as f:
        json.dump([{"tag": c[0], "kind": c[1], "w_dtype": c[2], "ab_dtype": c[3], "alpha": c[4]} for c in cases], f,
                  indent=1)


def traversal_cases():
    """_find_modules_v2 (ref:189-232) yield order on synthetic trees."""
    kind2cls = {"linear": nn.Linear, "conv": nn.Conv2d, "lora_linear": ref.LoraInjectedLinear,
                "lora_conv": ref.LoraInjectedConv2d}
    trees = {"toy_unet": H.toy_unet_spec(), "toy_clip": H.toy_clip_spec(), "nested": H.nested_spec()}
    queries = [
        (sorted(ref.UNET_DEFAULT_TARGET_REPLACE), ["linear"]),
        (sorted(ref.UNET_EXTENDED_TARGET_REPLACE), ["linear", "conv"]),
        (sorted(ref.UNET_EXTENDED_TARGET_REPLACE), ["lora_linear", "lora_conv"]),
        (["CLIPAttention"], ["linear"]),
        (["CrossAttention", "GEGLU", "Attention"], ["linear", "lora_linear"]),
        (None, ["lora_linear", "lora_conv"]),
        (None, ["linear"]),
    ]
    out = []
    for tname, spec in trees.items():
        root = H.build_tree(spec, ref.LoraInjectedLinear, ref.LoraInjectedConv2d)
        paths = H.module_paths(root)
        for anc, kinds in queries:
            got = list(ref._find_modules_v2(root, set(anc) if anc is not None else None,
                                            search_class=[kind2cls[k] for k in kinds]))
            out.append({"tree": tname, "ancestors": anc, "kinds": kinds,
                        "paths": [paths[id(m)] for _, _, m in got],
                        "names": [n for _, n, _ in got]})
    with open(os.path.join(OUT, "traversal_cases.json"), "w") as f:
        json.dump({"trees": trees, "cases": out}, f, indent=1)


def injection_and_file_cases():
    """inject -> train-ish perturbation -> save_safeloras_with_embeds / save_lora_weight -> parse (ref:255-596)."""
    torch.manual_seed(7)
    unet = H.build_tree(H.toy_unet_spec())
    clip = H.build_tree(H.toy_clip_spec())
    params, names = ref.inject_trainable_lora(unet, r=2, scale=0.7)
    tparams, tnames = ref.inject_trainable_lora(clip, target_replace_module={"CLIPAttention"}, r=3)
    state = {}
    for tag, model, tgt in (("unet", unet, ref.DEFAULT_TARGET_REPLACE), ("text_encoder", clip, {"CLIPAttention"})):
        for i, (up, down) in enumerate(ref.extract_lora_ups_down(model, tgt)):
            up.weight.data.normal_(0, 0.1)
            state[f"{tag}_{i}_up"] = H.t2n(up.weight)
            state[f"{tag}_{i}_down"] = H.t2n(down.weight)
    embeds = {"<s1>": torch.randn(8), "<s2>": torch.randn(8)}
    path = os.path.join(OUT, "mini_ref.safetensors")
    quiet(ref.save_safeloras_with_embeds, {"unet": (unet, ref.DEFAULT_TARGET_REPLACE),
                                          "text_encoder": (clip, {"CLIPAttention"})}, embeds, path)
    quiet(ref.save_lora_weight, unet, os.path.join(OUT, "mini_ref.pt"))
    state["embed_s1"], state["embed_s2"] = H.t2n(embeds["<s1>"]), H.t2n(embeds["<s2>"])
    np.savez(os.path.join(OUT, "mini_ref_state.npz"), **state)
    parsed = ref.load_safeloras(path)
    info = {"inject_names_unet": names, "inject_names_text": tnames, "n_param_groups_unet": len(params),
            "parsed": {k: {"ranks": v[1], "targets": sorted(v[2]), "shapes": [list(w.shape) for w in v[0]],
                           "dtypes": [str(w.dtype) for w in v[0]]} for k, v in parsed.items()},
            "embeds": sorted(ref.load_safeloras_embeds(path).keys()),
            "scale_unet": 0.7}
    # extended injection names/kinds on a fresh tree
    unet2 = H.build_tree(H.toy_unet_spec())
    _, names_ext = ref.inject_trainable_lora_extended(unet2, r=2)
    info["inject_names_unet_extended"] = names_ext
    info["extended_kinds"] = [type(m).__name__ for _, _, m in ref._find_modules_v2(
        unet2, ref.UNET_EXTENDED_TARGET_REPLACE, search_class=[ref.LoraInjectedLinear, ref.LoraInjectedConv2d])]
    with open(os.path.join(OUT, "mini_ref_info.json"), "w") as f:
        json.dump(info, f, indent=1)

    # monkeypatch_add_lora blend + tune_lora_scale (ref:850-880)
    torch.manual_seed(8)
    m = H.build_tree(H.attn_spec())
    ref.inject_trainable_lora(m, r=2)
    cur = [H.t2n(p) for up, down in ref.extract_lora_ups_down(m) for p in (up.weight, down.weight)]
    new = [torch.randn_like(p) for up, down in ref.extract_lora_ups_down(m) for p in (up.weight, down.weight)]
    for up, down in ref.extract_lora_ups_down(m):
        up.weight.data.normal_(0, 0.1)
    cur = [H.t2n(p) for up, down in ref.extract_lora_ups_down(m) for p in (up.weight, down.weight)]
    ref.monkeypatch_add_lora(m, [t.clone() for t in new], alpha=0.3, beta=0.9)
    after = [H.t2n(p) for up, down in ref.extract_lora_ups_down(m) for p in (up.weight, down.weight)]
    np.savez(os.path.join(OUT, "add_lora_case.npz"), **{f"cur{i}": a for i, a in enumerate(cur)},
             **{f"new{i}": H.t2n(a) for i, a in enumerate(new)}, **{f"after{i}": a for i, a in enumerate(after)})


def example_lora_manifest():
    """Structure of the reference's shipped fixtures: keys, shapes, dtypes, metadata, data hashes."""
    from safetensors import safe_open

    man = {}
    for fn in ("analog_svd_rank4.safetensors", "lora_disney.safetensors"):
        p = os.path.join("/root/reference/example_loras", fn)
        f = safe_open(p, framework="pt", device="cpu")
        keys = list(f.keys())
        ent = {"size": os.path.getsize(p), "metadata": f.metadata(), "keys_in_file_order": keys, "tensors": {}}
        for k in keys:
            t = f.get_tensor(k)
            ent["tensors"][k] = {"shape": list(t.shape), "dtype": str(t.dtype).replace("torch.", ""),
                                 "sha256": hashlib.sha256(t.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]}
        parsed = ref.parse_safeloras(f)
        ent["parsed"] = {k: {"n": len(v[0]), "ranks": v[1], "targets": sorted(v[2])} for k, v in parsed.items()}
        man[fn] = ent
    with open(os.path.join(OUT, "example_loras_manifest.json"), "w") as f:
        json.dump(man, f)


def optimizer_cases():
    """torch.optim.AdamW + clip_grad_norm_ as driven by ref train_lora_dreambooth.py:651-676, 878-888."""
    torch.manual_seed(11)
    n1, n2 = 700, 300
    p1, p2 = nn.Parameter(torch.randn(n1)), nn.Parameter(torch.randn(n2))
    opt = torch.optim.AdamW([{"params": [p1], "lr": 1e-2}, {"params": [p2], "lr": 5e-3}], lr=1e-2, betas=(0.9, 0.999),
                            weight_decay=1e-2, eps=1e-8)
    out = {"p0": np.concatenate([H.t2n(p1), H.t2n(p2)])}
    for step in range(1, 4):
        g = torch.randn(n1 + n2) * (3.0 if step == 2 else 0.01)
        p1.grad, p2.grad = g[:n1].clone(), g[n1:].clone()
        total = torch.nn.utils.clip_grad_norm_([p1, p2], 1.0)
        opt.step()
        out[f"g{step}"] = H.t2n(g)
        out[f"norm{step}"] = np.float32(total.item())
        out[f"p{step}"] = np.concatenate([H.t2n(p1), H.t2n(p2)])
    np.savez(os.path.join(OUT, "optimizer_case.npz"), **out)


def hostops_cases():
    """The ATen sequences the training step runs inside the UNet blocks between the adapter sites (diffusers'
    ResnetBlock2D / BasicTransformerBlock / GEGLU call exactly these torch functions): GroupNorm -> SiLU, LayerNorm,
    chunk -> gelu -> mul, with their autograd input gradients.  CPU, fp32, one thread."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(21)
    out = {}
    for i, (B, C, hh, ww, G, act) in enumerate([(2, 64, 8, 8, 8, 1), (1, 96, 4, 6, 3, 0), (2, 32, 8, 8, 32, 1)]):
        x = (torch.randn(B, C, hh, ww, generator=g) * 1.5 + torch.randn(1, C, 1, 1, generator=g) * 3.0).requires_grad_(True)
        w, b = torch.randn(C, generator=g) * 0.5 + 1.0, torch.randn(C, generator=g) * 0.3
        go = torch.randn(B, C, hh, ww, generator=g)
        y = F.group_norm(x, G, w, b, 1e-5)
        if act:
            y = F.silu(y)
        y.backward(go)
        out.update({f"gn{i}_x": H.t2n(x), f"gn{i}_w": H.t2n(w), f"gn{i}_b": H.t2n(b), f"gn{i}_go": H.t2n(go),
                    f"gn{i}_y": H.t2n(y), f"gn{i}_dx": H.t2n(x.grad), f"gn{i}_meta": np.array([G, act])})
    for i, shape in enumerate([(3, 7, 64), (2, 5, 320)]):
        x = (torch.randn(*shape, generator=g) * 1.5 + torch.randn(shape[-1], generator=g) * 2.0).requires_grad_(True)
        w, b = torch.randn(shape[-1], generator=g) * 0.5 + 1.0, torch.randn(shape[-1], generator=g) * 0.3
        go = torch.randn(*shape, generator=g)
        y = F.layer_norm(x, (shape[-1],), w, b, 1e-5)
        y.backward(go)
        out.update({f"ln{i}_x": H.t2n(x), f"ln{i}_w": H.t2n(w), f"ln{i}_b": H.t2n(b), f"ln{i}_go": H.t2n(go),
                    f"ln{i}_y": H.t2n(y), f"ln{i}_dx": H.t2n(x.grad)})
    for i, shape in enumerate([(2, 9, 64), (1, 3, 32)]):
        yin = (torch.randn(*shape, generator=g) * 2.0).requires_grad_(True)
        go = torch.randn(*shape[:-1], shape[-1] // 2, generator=g)
        h, gate = yin.chunk(2, dim=-1)
        o = h * F.gelu(gate)
        o.backward(go)
        out.update({f"gg{i}_y": H.t2n(yin), f"gg{i}_go": H.t2n(go), f"gg{i}_out": H.t2n(o), f"gg{i}_dy": H.t2n(yin.grad)})
    np.savez(os.path.join(OUT, "hostops_cases.npz"), **out)


def _reference_function(path: str, name: str, namespace: dict):
    """Compile ONE top-level function of a reference file whose module cannot be imported here (its imports need
    diffusers / fire) and return it, executed in ``namespace`` — the reference's own code object, nothing restated."""
    import ast

    tree = ast.parse(open(path).read(), filename=path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    mod = ast.Module(body=[fn], type_ignores=[])
    exec(compile(mod, path, "exec"), namespace)
    return namespace[name]


def pti_loss_cases():
    """``loss_step`` of the reference's cli_lora_pti.py (:260-370), executed from its own source: the t_mutliplier
    range of the timestep draw, both prediction types, the masked-MSE branch with a temperature, gradients w.r.t. the
    toy UNet's parameters.  The draws are recovered by re-seeding and repeating the two calls loss_step makes."""
    import torch.nn.functional as F
    from lora_amd.standin import DDPMScheduler

    loss_step = _reference_function("/root/reference/lora_diffusion/cli_lora_pti.py", "loss_step", {"torch": torch, "F": F})
    torch.manual_seed(77)
    unet, text = H.PtiToyUNet(), H.PtiToyText()
    out = {f"unet_{k.replace('.', '_')}": H.t2n(v) for k, v in unet.state_dict().items()}
    out["text_emb"] = H.t2n(text.emb.weight)
    cases = [("plain", "epsilon", 1.0, False, 1.0), ("tmul", "epsilon", 0.8, False, 1.0),
             ("mask", "epsilon", 0.8, True, 1.0), ("mask_temp", "epsilon", 0.8, True, 2.5),
             ("vpred_mask", "v_prediction", 0.8, True, 0.5)]
    g = torch.Generator().manual_seed(5)
    for i, (tag, ptype, tmul, masked, temp) in enumerate(cases):
        B, hw = 3, 6
        sched = DDPMScheduler(prediction_type=ptype)
        batch = {"pixel_values": torch.randn(B, 4, hw, hw, generator=g) * 0.18215,
                 "input_ids": torch.randint(0, 20, (B, 5), generator=g)}
        if masked:  # face-segmentation style: [B, H*8, W*8] with a soft blob, values in [0, 1]
            m = torch.rand(B, hw * 8, hw * 8, generator=g)
            m[:, : hw * 4] *= 0.1
            batch["mask"] = m
        unet.zero_grad()
        torch.manual_seed(1000 + i)
        loss = loss_step(batch, unet, None, text, sched, t_mutliplier=tmul, mask_temperature=temp, cached_latents=True)
        loss.backward()
        torch.manual_seed(1000 + i)  # the same two draws, in loss_step's order (:296, :299-305)
        noise = torch.randn_like(batch["pixel_values"])
        ts = torch.randint(0, int(1000 * tmul), (B,)).long()
        assert int(ts.max()) < int(1000 * tmul)
        out.update({f"{tag}_latents": H.t2n(batch["pixel_values"]), f"{tag}_ids": batch["input_ids"].numpy(),
                    f"{tag}_noise": H.t2n(noise), f"{tag}_t": ts.numpy(), f"{tag}_loss": H.t2n(loss),
                    f"{tag}_dconv": H.t2n(unet.conv.weight.grad), f"{tag}_dctx": H.t2n(unet.ctx.weight.grad),
                    f"{tag}_meta": np.array([tmul, temp, float(ptype == "v_prediction"), 1000 + i], dtype=np.float64)})
        if masked:
            out[f"{tag}_mask"] = H.t2n(batch["mask"])
    # non-cached branch (:272-277): a deterministic toy VAE (latent_dist.sample() returns the mean)
    import types

    class ToyVAE:
        def encode(self, px):
            lat = F.avg_pool2d(px, 8)[:, [0, 1, 2, 0]] * torch.tensor([1.0, -0.5, 0.25, 2.0]).view(1, 4, 1, 1)
            return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: lat))

    px = torch.rand(2, 3, 48, 48, generator=g) * 2 - 1
    batch = {"pixel_values": px, "input_ids": torch.randint(0, 20, (2, 5), generator=g)}
    torch.manual_seed(2000)
    loss = loss_step(batch, unet, ToyVAE(), text, DDPMScheduler(), t_mutliplier=1.0, cached_latents=False)
    out.update({"vae_pixels": H.t2n(px), "vae_ids": batch["input_ids"].numpy(), "vae_loss": H.t2n(loss)})
    np.savez(os.path.join(OUT, "pti_loss_cases.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1:  # regenerate only the named fixture sets, e.g. `make_golden.py conv_native_cases`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    linear_cases()
    conv_cases()
    conv_native_cases()
    collapse_cases()
    traversal_cases()
    injection_and_file_cases()
    example_lora_manifest()
    optimizer_cases()
    hostops_cases()
    pti_loss_cases()
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn:40s} {os.path.getsize(os.path.join(OUT, fn)):>9d} B")
