"""CPU restatement (numpy) of the reference's LoRA hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``lora_amd/``) never does.

Each function restates one piece of ``/root/reference`` (cited as ``ref: file:line``).  The
arithmetic of that path lives in PyTorch ATen (un-vendored); it is restated here as plain
numpy float32 maths.  The restatement is pinned against the reference itself: the vectors in
``tests/golden/`` were produced by importing ``/root/reference/lora_diffusion/lora.py`` in
the build container (``scripts/make_golden.py``), and ``tests/test_oracle_pins.py`` checks
this file against them.
"""
from __future__ import annotations

import json
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- dtype emulation
def round_bf16(x: np.ndarray) -> np.ndarray:
    """float32 -> bfloat16 (round-to-nearest-even) -> float32, as torch's ``.to(bfloat16)``."""
    x = np.ascontiguousarray(x, dtype=F32)
    u = x.view(np.uint32)
    bias = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    r = ((u + bias) & np.uint32(0xFFFF0000)).view(F32)
    return np.where(np.isnan(x), x, r).astype(F32)


def round_f16(x: np.ndarray) -> np.ndarray:
    return np.asarray(x, dtype=F32).astype(np.float16).astype(F32)


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    if dtype == "f32":
        return np.asarray(x, dtype=F32)
    if dtype == "bf16":
        return round_bf16(x)
    if dtype == "f16":
        return round_f16(x)
    raise ValueError(dtype)


# --------------------------------------------------------------------------- adapters (L1)
def lora_linear_forward(x, W, b, down, up, scale=1.0, selector=None, mask=None):
    """ref: lora_diffusion/lora.py:53-58.
    ``linear(x) + dropout(lora_up(selector(lora_down(x)))) * scale``;
    x [M,K], W [N,K], b [N]|None, down [r,K], up [N,r], selector [r,r]|None,
    mask [M,N]|None = the dropout multiplier (0 or 1/(1-p))."""
    x, W, down, up = (np.asarray(a, dtype=F32) for a in (x, W, down, up))
    t = x @ down.T
    if selector is not None:
        t = t @ np.asarray(selector, dtype=F32).T  # nn.Linear(r, r): t @ S^T (ref:63-70)
    branch = t @ up.T
    if mask is not None:
        branch = branch * np.asarray(mask, dtype=F32)
    y = x @ W.T
    if b is not None:
        y = y + np.asarray(b, dtype=F32)
    return (y + branch * F32(scale)).astype(F32), t.astype(F32)


def lora_linear_backward(g, x, W, down, up, scale=1.0, selector=None, mask=None):
    """Autograd of lora.py:53-58 (implicit in the reference): returns dx, d_down, d_up, dW, db."""
    g, x, W, down, up = (np.asarray(a, dtype=F32) for a in (g, x, W, down, up))
    gm = g if mask is None else g * np.asarray(mask, dtype=F32)
    t_pre = x @ down.T
    t = t_pre if selector is None else t_pre @ np.asarray(selector, dtype=F32).T
    d_up = F32(scale) * (gm.T @ t)                       # [N, r]
    gt = F32(scale) * (gm @ up)                           # dL/dt  [M, r]
    if selector is not None:
        gt = gt @ np.asarray(selector, dtype=F32)         # back through t_pre @ S^T
    d_down = gt.T @ x                                     # [r, K]
    dx = g @ W + gt @ down                                # [M, K]
    return dx.astype(F32), d_down.astype(F32), d_up.astype(F32), (g.T @ x).astype(F32), g.sum(0).astype(F32)


def _im2col(x, kh, kw, stride, padding, dilation):
    B, C, H, W = x.shape
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    cols = np.zeros((B, C, kh, kw, Ho, Wo), dtype=F32)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, i, j] = xp[:, :, i * dh: i * dh + sh * Ho: sh, j * dw: j * dw + sw * Wo: sw]
    return cols.reshape(B, C * kh * kw, Ho * Wo), Ho, Wo


def conv2d(x, w, b=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
    """Plain NCHW cross-correlation (groups=1), float32 — what nn.Conv2d computes."""
    x, w = np.asarray(x, dtype=F32), np.asarray(w, dtype=F32)
    Co, Ci, kh, kw = w.shape
    cols, Ho, Wo = _im2col(x, kh, kw, stride, padding, dilation)
    y = np.einsum("ok,bkp->bop", w.reshape(Co, -1), cols).reshape(x.shape[0], Co, Ho, Wo)
    if b is not None:
        y = y + np.asarray(b, dtype=F32).reshape(1, -1, 1, 1)
    return y.astype(F32)


def lora_conv2d_forward(x, W, b, down, up, scale=1.0, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mask=None,
                        selector=None):
    """ref: lora_diffusion/lora.py:130-135 (ctor :94-128): ``conv(x) + dropout(up1x1(down_kxk(x))) * scale``;
    down [r,C,kh,kw] shares the frozen conv's geometry, up [Co,r,1,1]."""
    t = conv2d(x, down, None, stride, padding, dilation)
    if selector is not None:
        t = np.einsum("aj,bjhw->bahw", np.asarray(selector, dtype=F32), t).astype(F32)
    branch = conv2d(t, up)
    if mask is not None:
        branch = branch * np.asarray(mask, dtype=F32)
    return (conv2d(x, W, b, stride, padding, dilation) + branch * F32(scale)).astype(F32), t


def _col2im(dcols, x_shape, kh, kw, stride, padding, dilation):
    """Adjoint of ``_im2col``: scatter-add [B, C*kh*kw, Ho*Wo] back onto the (padded) input grid."""
    B, C, H, W = x_shape
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    d6 = np.asarray(dcols, dtype=F32).reshape(B, C, kh, kw, Ho, Wo)
    dxp = np.zeros((B, C, H + 2 * ph, W + 2 * pw), dtype=F32)
    for i in range(kh):
        for j in range(kw):
            dxp[:, :, i * dh: i * dh + sh * Ho: sh, j * dw: j * dw + sw * Wo: sw] += d6[:, :, i, j]
    return dxp[:, :, ph: ph + H, pw: pw + W]


def lora_conv2d_backward(g, x, W, down, up, scale=1.0, stride=(1, 1), padding=(0, 0), dilation=(1, 1), selector=None,
                         mask=None):
    """Autograd of lora.py:130-135 (implicit in the reference): returns dx, d_down, d_up for
    ``conv(x; W) + scale * mask * up1x1(S . down_kxk(x))`` given the output gradient ``g`` [B,Co,Ho,Wo].
    ``selector`` is the [r, r] matrix applied across the rank channels between down and up (lora.py:140-156)."""
    g, x, W, down, up = (np.asarray(a, dtype=F32) for a in (g, x, W, down, up))
    Co, Ci, kh, kw = W.shape
    r = down.shape[0]
    cols, Ho, Wo = _im2col(x, kh, kw, stride, padding, dilation)          # [B, Ci*kh*kw, P]
    B, P = x.shape[0], Ho * Wo
    t = np.einsum("jk,bkp->bjp", down.reshape(r, -1), cols)
    if selector is not None:
        t = np.einsum("aj,bjp->bap", np.asarray(selector, dtype=F32), t)
    gm = g.reshape(B, Co, P)
    if mask is not None:
        gm = gm * np.asarray(mask, dtype=F32).reshape(B, Co, P)
    up2 = up.reshape(Co, r)
    d_up = F32(scale) * np.einsum("bop,bjp->oj", gm, t)
    gt = F32(scale) * np.einsum("oj,bop->bjp", up2, gm)
    if selector is not None:
        gt = np.einsum("aj,bap->bjp", np.asarray(selector, dtype=F32), gt)   # S^T gt
    d_down = np.einsum("bjp,bkp->jk", gt, cols).reshape(down.shape)
    dcols = np.einsum("jk,bjp->bkp", down.reshape(r, -1), gt) + np.einsum("ok,bop->bkp", W.reshape(Co, -1),
                                                                            g.reshape(B, Co, P))
    dx = _col2im(dcols, x.shape, kh, kw, stride, padding, dilation)
    return dx.astype(F32), d_down.astype(F32), d_up.reshape(up.shape).astype(F32)


def realize_as_lora(up, down, scale):
    """ref: lora.py:60-61, 137-138 — scale is folded into ``up`` only."""
    return np.asarray(up, dtype=F32) * F32(scale), np.asarray(down, dtype=F32)


# --------------------------------------------------------------------------- merge (K3)
def collapse(W, up, down, alpha=1.0, w_dtype="f32", ab_dtype="f32"):
    """ref: lora_diffusion/lora.py:646-655 (Linear) / :659-669 (Conv2d, factors flattened from dim 1).

    ``W.data + alpha * (up @ down).type(W.dtype)`` with torch's per-op rounding:
    the matmul result is rounded to the factors' dtype, ``.type`` rounds to W's dtype, the scalar
    multiply rounds to W's dtype, the add rounds to W's dtype.  ``.scale`` is not involved.
    Inputs are float32 arrays holding values representable in their nominal dtypes."""
    W = np.asarray(W, dtype=F32)
    up2 = np.asarray(up, dtype=F32).reshape(up.shape[0], -1)
    down2 = np.asarray(down, dtype=F32).reshape(down.shape[0], -1)
    p = round_to(round_to(up2 @ down2, ab_dtype), w_dtype).reshape(W.shape)
    q = round_to(F32(alpha) * p, w_dtype)
    return round_to(W + q, w_dtype)


def add_lora_blend(cur, new, alpha=1.0, beta=1.0):
    """ref: lora.py:865-872 — ``new * alpha + cur * beta``."""
    return (np.asarray(new, dtype=F32) * F32(alpha) + np.asarray(cur, dtype=F32) * F32(beta)).astype(F32)


def inspect_moved(up, down):
    """ref: lora.py:1031-1036 — mean |up.flatten(1) @ down.flatten(1)|."""
    up2 = np.asarray(up, dtype=F32).reshape(up.shape[0], -1)
    down2 = np.asarray(down, dtype=F32).reshape(down.shape[0], -1)
    return float(np.abs(up2 @ down2).mean())


# --------------------------------------------------------------------------- traversal (L2)
class Node:
    """A module tree stand-in: class name, kind ('linear'|'conv'|'lora_linear'|'lora_conv'|'other'),
    ordered children."""

    def __init__(self, cls: str, kind: str = "other", children: Optional[List[Tuple[str, "Node"]]] = None):
        self.cls, self.kind, self.children = cls, kind, list(children or [])

    @staticmethod
    def from_spec(spec) -> "Node":
        return Node(spec["cls"], spec.get("kind", "other"),
                    [(n, Node.from_spec(c)) for n, c in spec.get("children", [])])

    def walk(self, prefix=""):
        """pre-order (name, node), like nn.Module.named_modules()."""
        yield prefix, self
        for name, child in self.children:
            yield from child.walk(f"{prefix}.{name}" if prefix else name)


def find_modules(root: Node, ancestor_class: Optional[Iterable[str]], search_kinds: Sequence[str],
                 exclude_parent_kinds: Sequence[str] = ("lora_linear", "lora_conv")) -> List[str]:
    """ref: lora_diffusion/lora.py:189-232 (_find_modules_v2), on an immutable tree: the dotted paths
    (relative to ``root``) of every node of a searched kind below every ancestor whose class name is in
    ``ancestor_class`` (every node if None), in yield order, skipping children of adapters."""
    if ancestor_class is not None:
        ancestors = [(p, n) for p, n in root.walk() if n.cls in set(ancestor_class)]
    else:
        ancestors = list(root.walk())
    out = []
    for apath, anc in ancestors:
        parents = {"": anc}
        for path, node in anc.walk():
            parents[path] = node
        for path, node in anc.walk():
            if node.kind not in search_kinds:
                continue
            owner = parents[path.rpartition(".")[0]]
            if owner.kind in exclude_parent_kinds:
                continue
            out.append(f"{apath}.{path}".strip(".") if apath else path)
    return out


# --------------------------------------------------------------------------- file format (L2)
def safeloras_layout(models: Dict[str, Tuple[List[Tuple[np.ndarray, np.ndarray]], Iterable[str]]],
                     embeds: Optional[Dict[str, np.ndarray]] = None):
    """ref: lora_diffusion/lora.py:463-483 — tensor keys and metadata of a saved file.
    ``models[name] = ([(up, down), ...] already realised, targets)``."""
    weights, meta = {}, {}
    for name, (pairs, targets) in models.items():
        meta[name] = json.dumps(list(targets))
        for i, (up, down) in enumerate(pairs):
            meta[f"{name}:{i}:rank"] = str(down.shape[0])
            weights[f"{name}:{i}:up"] = up
            weights[f"{name}:{i}:down"] = down
    for tok, t in (embeds or {}).items():
        meta[tok] = "<embed>"
        weights[tok] = t
    return weights, meta


def parse_order(keys: Sequence[str], metadata: Dict[str, str]):
    """ref: lora_diffusion/lora.py:556-594 — for each model name: positions ``idx*2 + (direction=='down')``
    -> key, the ranks list, the target list; TI embeds skipped."""
    name_of = lambda k: k.split(":")[0]
    out = {}
    for name in dict.fromkeys(name_of(k) for k in sorted(keys, key=name_of)):
        info = metadata.get(name)
        if not info:
            raise ValueError(f"Tensor {name} has no metadata - is this a Lora safetensor?")
        if info == "<embed>":
            continue
        grp = [k for k in keys if name_of(k) == name]
        order = [None] * len(grp)
        ranks = [4] * (len(grp) // 2)
        for k in grp:
            _, idx, direction = k.split(":")
            idx = int(idx)
            ranks[idx] = int(metadata[f"{name}:{idx}:rank"])
            order[idx * 2 + (1 if direction == "down" else 0)] = k
        out[name] = (order, ranks, json.loads(info))
    return out


# --------------------------------------------------------------------------- training step (L3)
def ddpm_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """SD1.5's DDPMScheduler config (scaled_linear betas) — diffusers is un-vendored; call site
    ref: training_scripts/train_lora_dreambooth.py:678-680."""
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas).astype(F32)


def add_noise(latents, noise, timesteps, alphas_cumprod):
    """ref: train_lora_dreambooth.py:837 (DDPMScheduler.add_noise): sqrt(a_t) x + sqrt(1-a_t) eps."""
    a = alphas_cumprod[np.asarray(timesteps)].astype(F32)
    sa, sb = np.sqrt(a).reshape(-1, 1, 1, 1), np.sqrt(1 - a).reshape(-1, 1, 1, 1)
    return (sa * np.asarray(latents, dtype=F32) + sb * np.asarray(noise, dtype=F32)).astype(F32)


def dreambooth_loss(pred, target, with_prior_preservation=False, prior_loss_weight=1.0):
    """ref: train_lora_dreambooth.py:855-875."""
    pred, target = np.asarray(pred, dtype=F32), np.asarray(target, dtype=F32)
    if not with_prior_preservation:
        return float(((pred - target) ** 2).mean())
    h = pred.shape[0] // 2
    inst = ((pred[:h] - target[:h]) ** 2).mean(axis=(1, 2, 3)).mean()
    prior = ((pred[h:] - target[h:]) ** 2).mean()
    return float(inst + F32(prior_loss_weight) * prior)


def clip_grad_norm(grads: List[np.ndarray], max_norm: float):
    """ref: train_lora_dreambooth.py:884 (torch.nn.utils.clip_grad_norm_, L2):
    total = ||(||g_i||)||; coef = min(1, max_norm / (total + 1e-6))."""
    total = F32(np.sqrt(sum(float((np.asarray(g, dtype=np.float64) ** 2).sum()) for g in grads)))
    coef = min(F32(1.0), F32(max_norm) / (total + F32(1e-6)))
    return [np.asarray(g, dtype=F32) * F32(coef) for g in grads], float(total)


def adamw_step(p, g, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2):
    """ref: train_lora_dreambooth.py:651, 670-676, 885 (torch.optim.AdamW, single-tensor maths)."""
    p, g, m, v = (np.asarray(a, dtype=F32).copy() for a in (p, g, m, v))
    p *= F32(1 - lr * weight_decay)
    m += (g - m) * F32(1 - beta1)
    v = v * F32(beta2) + F32(1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / F32(np.sqrt(bc2)) + F32(eps)
    p -= F32(lr / bc1) * (m / denom)
    return p.astype(F32), m.astype(F32), v.astype(F32)


# --------------------------------------------------------------------------- frozen host-model passes
# The UNet blocks the reference trains through (diffusers ResnetBlock2D / BasicTransformerBlock / GEGLU, un-vendored)
# call torch's group_norm, silu, layer_norm and gelu between the adapter sites; restated here in float64 numpy and
# pinned against torch CPU outputs (tests/golden/hostops_cases.npz, scripts/make_golden.py::hostops_cases).
def _erf(x: np.ndarray) -> np.ndarray:
    import math

    return np.vectorize(math.erf, otypes=[np.float64])(x)


def _silu_and_grad(z: np.ndarray):
    s = 1.0 / (1.0 + np.exp(-z))
    return z * s, s * (1.0 + z * (1.0 - s))


def group_norm_act(x, groups: int, weight, bias, eps: float = 1e-5, act: bool = True):
    """y = silu(group_norm(x)) (``act``) or group_norm(x) for NCHW x; returns (y, cache)."""
    x = np.asarray(x, np.float64)
    B, C = x.shape[:2]
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(xg.var(-1, keepdims=True) + eps)
    xh = ((xg - mean) * rstd).reshape(x.shape)
    shape = (1, C) + (1,) * (x.ndim - 2)
    z = xh * np.asarray(weight, np.float64).reshape(shape) + np.asarray(bias, np.float64).reshape(shape)
    y = _silu_and_grad(z)[0] if act else z
    return y, (xh, rstd, z, np.asarray(weight, np.float64).reshape(shape), groups, act)


def group_norm_act_backward(gout, cache):
    """Input gradient of :func:`group_norm_act` (affine parameters frozen)."""
    xh, rstd, z, w, groups, act = cache
    dz = np.asarray(gout, np.float64) * (_silu_and_grad(z)[1] if act else 1.0)
    t = dz * w
    B = xh.shape[0]
    tg, xg = t.reshape(B, groups, -1), xh.reshape(B, groups, -1)
    dx = rstd * (tg - tg.mean(-1, keepdims=True) - xg * (tg * xg).mean(-1, keepdims=True))
    return dx.reshape(xh.shape)


def layer_norm(x, weight, bias, eps: float = 1e-5):
    x = np.asarray(x, np.float64)
    mean = x.mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(x.var(-1, keepdims=True) + eps)
    xh = (x - mean) * rstd
    return xh * np.asarray(weight, np.float64) + np.asarray(bias, np.float64), (xh, rstd, np.asarray(weight, np.float64))


def layer_norm_backward(gout, cache):
    xh, rstd, w = cache
    t = np.asarray(gout, np.float64) * w
    return rstd * (t - t.mean(-1, keepdims=True) - xh * (t * xh).mean(-1, keepdims=True))


def geglu(y):
    """h * gelu(gate) for y = [h | gate] along the last axis (erf form, F.gelu's default)."""
    y = np.asarray(y, np.float64)
    h, g = np.split(y, 2, axis=-1)
    return h * (g * 0.5 * (1.0 + _erf(g / np.sqrt(2.0))))


def geglu_backward(y, gout):
    y, gout = np.asarray(y, np.float64), np.asarray(gout, np.float64)
    h, g = np.split(y, 2, axis=-1)
    cdf = 0.5 * (1.0 + _erf(g / np.sqrt(2.0)))
    pdf = np.exp(-0.5 * g * g) / np.sqrt(2.0 * np.pi)
    return np.concatenate([gout * g * cdf, gout * h * (cdf + g * pdf)], axis=-1)
