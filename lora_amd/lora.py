"""Drop-in for ``lora_diffusion.lora`` whose device path runs on the gfx950 HIP kernels.

Every public name, signature, return shape, on-disk layout and quirk of the
reference module (``/root/reference/lora_diffusion/lora.py``, cited per item as
``ref:LINE``) is kept; what changes is *how* the hot operations run:

* adapter forward/backward on device tensors -> ``ops.lora_linear`` / ``ops.lora_linear_group`` / ``ops.lora_conv``:
  one fused MFMA launch per site (or per q/k/v group) where the shape table says so, the streaming kernels
  (rowdot / rank_update / colreduce) otherwise, NCHW or channels-last convolution kernels by memory format —
  instead of 6 ATen launches per site;
* ``collapse_lora`` on device tensors -> ONE batched launch of the fused
  ``W + alpha * up @ down`` kernel over all sites instead of 3 passes per site.

CPU tensors take the plain-torch path (the reference's own CPU behaviour —
BASELINE config 0 is a CPU plumbing run).  Device tensors never fall back: a
missing HIP library raises ``_C.HipExtensionMissing``.
"""
from __future__ import annotations

import json
from itertools import groupby
from typing import Dict, Iterator, List, Optional, Set, Tuple, Type, Union

import torch
import torch.nn as nn

from . import _C, ops

try:  # ref:12-29
    from safetensors.torch import safe_open
    from safetensors.torch import save_file as safe_save

    safetensors_available = True
except ImportError:  # pragma: no cover - safetensors ships in this image
    from .safe_open import safe_open

    def safe_save(tensors, filename, metadata=None):
        raise EnvironmentError(
            "Saving safetensors requires the safetensors library. Please install with pip or similar.")

    safetensors_available = False

UNET_DEFAULT_TARGET_REPLACE = {"CrossAttention", "Attention", "GEGLU"}  # ref:159
UNET_EXTENDED_TARGET_REPLACE = {"ResnetBlock2D", "CrossAttention", "Attention", "GEGLU"}  # ref:161
TEXT_ENCODER_DEFAULT_TARGET_REPLACE = {"CLIPAttention"}  # ref:163
TEXT_ENCODER_EXTENDED_TARGET_REPLACE = {"CLIPAttention"}  # ref:165
DEFAULT_TARGET_REPLACE = UNET_DEFAULT_TARGET_REPLACE  # ref:167
EMBED_FLAG = "<embed>"  # ref:169


def _check_rank(r: int, a: int, b: int) -> None:
    if r > min(a, b):  # ref:38-41, 89-92
        raise ValueError(f"LoRA rank {r} must be less or equal than {min(a, b)}")


def _autocast_dtype(x: torch.Tensor, w: torch.Tensor) -> torch.dtype:
    if torch.is_autocast_enabled(x.device.type):
        return torch.get_autocast_dtype(x.device.type)
    return w.dtype


class _Adapter(nn.Module):
    """State shared by both adapter kinds (attribute names are part of the contract, ref:42-48)."""

    r: int
    scale: float

    def _frozen(self) -> nn.Module:
        raise NotImplementedError

    def realize_as_lora(self):  # ref:60-61, 137-138
        return self.lora_up.weight.data * self.scale, self.lora_down.weight.data

    def _selector_matrix(self) -> Optional[torch.Tensor]:
        sel = self.selector
        if isinstance(sel, nn.Identity):
            return None
        return sel.weight.reshape(self.r, self.r)

    def _dropout_p(self) -> float:
        return float(self.dropout.p) if self.dropout.training else 0.0

    def _shadow(self, t: Optional[torch.Tensor], dt: torch.dtype, slot: str) -> Optional[torch.Tensor]:
        """Frozen tensor in the compute dtype, kept resident instead of re-cast every forward
        (the reference's autocast re-casts every fp32 weight every step, SURVEY.md §3.1 note b).

        A hit needs the SAME tensor object (``is``) at the same address and version: a replaced Parameter
        (``collapse_lora`` installs a new one, ref:646) can never alias an old entry even if the allocator gives it the
        freed address.  In-place edits through ``.data`` are invisible to ``_version``: call
        :func:`invalidate_caches` after such an edit."""
        if t is None or t.dtype == dt:
            return t
        if t.requires_grad:
            return t.to(dt)
        cache = self.__dict__.setdefault("_shadow_cache", {})
        hit = cache.get(slot)
        if hit is not None and hit[0] is t and hit[1] == (t.data_ptr(), _C._tensor_version(t), dt):
            return hit[2]
        c = t.detach().to(dt)
        cache[slot] = (t, (t.data_ptr(), _C._tensor_version(t), dt), c)
        return c

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda() / .half(): every derived layout is stale
        self.__dict__.pop("_shadow_cache", None)
        _C.invalidate_weight_caches()
        return super()._apply(fn, *args, **kwargs)


def invalidate_caches(model: Optional[nn.Module] = None) -> None:
    """Forget the compute-dtype shadows and transposed layouts derived from frozen weights (all adapters of ``model``,
    or just the global transposes when ``model`` is None).  Needed only after an in-place edit of a frozen weight
    through ``.data``; replacing Parameters, ``.to()``, ``collapse_lora`` and ``monkeypatch_*`` do it themselves."""
    _C.invalidate_weight_caches()
    if model is not None:
        for m in model.modules():
            if isinstance(m, _Adapter):
                m.__dict__.pop("_shadow_cache", None)


class LoraInjectedLinear(_Adapter):
    """ref:32-70.  ``y = linear(x) + dropout(lora_up(selector(lora_down(x)))) * scale``."""

    def __init__(self, in_features, out_features, bias=False, r=4, dropout_p=0.1, scale=1.0):
        super().__init__()
        _check_rank(r, in_features, out_features)
        self.r = r
        self.linear = nn.Linear(in_features, out_features, bias)
        self.lora_down = nn.Linear(in_features, r, bias=False)
        self.dropout = nn.Dropout(dropout_p)
        self.lora_up = nn.Linear(r, out_features, bias=False)
        self.scale = scale
        self.selector = nn.Identity()
        nn.init.normal_(self.lora_down.weight, std=1 / r)  # ref:50
        nn.init.zeros_(self.lora_up.weight)  # ref:51

    def _frozen(self):
        return self.linear

    def forward(self, input):
        if input.is_cuda:
            return self._forward_device(input)
        # CPU plumbing path: the reference's op sequence (ref:53-58)
        low = self.lora_up(self.selector(self.lora_down(input)))
        return self.linear(input) + self.dropout(low) * self.scale

    def _forward_device(self, x: torch.Tensor, in_heads=None, out_heads=None) -> torch.Tensor:
        w, b = self.linear.weight, self.linear.bias
        dt = _autocast_dtype(x, w)
        xc = x if x.dtype == dt else x.to(dt)
        wc, bc = self._shadow(w, dt, "w"), self._shadow(b, dt, "b")
        with torch.autocast(device_type=x.device.type, enabled=False):
            mw = self.__dict__.get("_merged")
            if mw is not None and ops.merged_ok(xc, wc, self.lora_down.weight, self.lora_up.weight,
                                                self._selector_matrix(), self._dropout_p(), in_heads, out_heads, b):
                # the step's merged weight W + scale up down (trainer.enable_merged_weights): frozen GEMM forward and
                # input gradient, one launch for both factor gradients
                need_dx = xc.requires_grad and torch.is_grad_enabled()
                w_eff, b_eff, w_eff_t = mw.lookup(self, wc, bc, dt, in_heads, out_heads, need_dx)
                return ops.LoraLinearMergedFunction.apply(xc, w_eff, b_eff, self.lora_down.weight, self.lora_up.weight,
                                                          float(self.scale), self.__dict__.get("_grad_sink"), in_heads,
                                                          out_heads, w_eff_t)
            return ops.lora_linear(xc, wc, bc, self.lora_down.weight, self.lora_up.weight,
                                   self._selector_matrix(), self.scale, self._dropout_p(),
                                   self.__dict__.get("_grad_sink"), in_heads, out_heads)

    def forward_heads(self, input, in_heads=None, out_heads=None):
        """``forward`` for head-padded activations (not in the reference): ``in_heads`` / ``out_heads`` = (heads, d, D)
        say that the input arrives / the output leaves with every head's d columns padded to D, the layout the
        attention kernels want for head sizes 40 / 80.  On the device the fused kernels read and write that layout
        themselves (no pad / slice copies); everywhere else this is unpack -> forward -> pack."""
        if input.is_cuda:
            return self._forward_device(input, in_heads, out_heads)
        y = self.forward(ops.unpack_heads(input, in_heads) if in_heads else input)
        return ops.pack_heads(y, out_heads) if out_heads else y

    def set_selector_from_diag(self, diag: torch.Tensor):  # ref:63-70
        assert diag.shape == (self.r,)
        self.selector = nn.Linear(self.r, self.r, bias=False)
        self.selector.weight.data = torch.diag(diag).to(self.lora_up.weight.device).to(self.lora_up.weight.dtype)


def lora_linear_group(adapters, x: torch.Tensor, out_heads=None):
    """Apply several ``LoraInjectedLinear`` adapters to the SAME input in one launch where that is possible (device
    tensors, 16-bit compute, no dropout in effect, no selector, f32 factors, equal rank, shapes the weight-stationary
    kernel covers); returns the list of outputs, or None when the caller should simply call the adapters one by one.

    What host code with q/k/v (or k/v) projections on one tensor calls: ``lora_amd/standin/unet.py::CrossAttention``
    and ``LoraAmdAttnProcessor`` for a ``diffusers`` attention block.  Numerically each output equals the adapter's own
    ``forward`` on the fused path (same kernels, same rounding points)."""
    if not x.is_cuda or len(adapters) < 2 or not all(isinstance(a, LoraInjectedLinear) for a in adapters):
        return None
    if all(a.__dict__.get("_merged") is not None for a in adapters):
        # merged-weight path: every site is a dense GEMM on its own merged weight; grouped so that the sites' input
        # gradients accumulate inside the GEMMs (``out_heads``: the outputs leave in the padded head layout)
        dt = _autocast_dtype(x, adapters[0].linear.weight)
        xc = x if x.dtype == dt else x.to(dt)
        mw = adapters[0].__dict__["_merged"]
        wcs = [a._shadow(a.linear.weight, dt, "w") for a in adapters]
        bcs = [a._shadow(a.linear.bias, dt, "b") for a in adapters]
        for a, wc in zip(adapters, wcs):
            if a.__dict__["_merged"] is not mw or not ops.merged_ok(xc, wc, a.lora_down.weight, a.lora_up.weight,
                                                                    a._selector_matrix(), a._dropout_p(), None, out_heads,
                                                                    a.linear.bias):
                return None
        need_dx = xc.requires_grad and torch.is_grad_enabled()
        # one scratch buffer for the group (one GEMM forward) where the sites' entries do not exist apart already
        g = mw.lookup_group(adapters, wcs, bcs, dt, out_heads, need_dx) if ops.CONCAT_GROUPS else None
        flat = []
        for i, a in enumerate(adapters):
            if g is not None:
                e = g["entries"][i]
                w_eff, b_eff, w_eff_t = e["w_eff"], e["b_eff"], e["w_eff_t"]
            else:
                w_eff, b_eff, w_eff_t = mw.lookup(a, wcs[i], bcs[i], dt, None, out_heads, need_dx)
            flat += [w_eff, b_eff, a.lora_down.weight, a.lora_up.weight, float(a.scale), a.__dict__.get("_grad_sink"),
                     out_heads, w_eff_t]
        with torch.autocast(device_type=x.device.type, enabled=False):
            return list(ops.LoraLinearMergedGroupFunction.apply(xc, len(adapters), g["cat"] if g is not None else None,
                                                                g["bias"] if g is not None else None, *flat))
    if out_heads is not None:
        return None  # the per-site fused kernels take head layouts one adapter at a time (forward_heads)
    a0 = adapters[0]
    w0 = a0.linear.weight
    dt = _autocast_dtype(x, w0)
    if dt not in (torch.bfloat16, torch.float16):
        return None
    r = a0.r
    for a in adapters:
        if (a.r != r or a._dropout_p() > 0.0 or a._selector_matrix() is not None
                or a.lora_down.weight.dtype != torch.float32 or a.lora_up.weight.dtype != torch.float32
                or a.linear.weight.requires_grad or a.linear.in_features != a0.linear.in_features):
            return None
    xc = x if x.dtype == dt else x.to(dt)
    if not ops.linear_group_ok(xc, [(a.linear.out_features, a.linear.in_features) for a in adapters], r):
        return None
    sites = [(a._shadow(a.linear.weight, dt, "w"), a._shadow(a.linear.bias, dt, "b"), a.lora_down.weight,
              a.lora_up.weight, a.scale, a.__dict__.get("_grad_sink")) for a in adapters]
    with torch.autocast(device_type=x.device.type, enabled=False):
        return list(ops.lora_linear_group(xc, sites))


class LoraInjectedConv2d(_Adapter):
    """ref:73-156.  ``lora_down`` copies the frozen conv's geometry (in -> r), ``lora_up`` is 1x1 (r -> out)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, dilation=1,
                 groups: int = 1, bias: bool = True, r: int = 4, dropout_p: float = 0.1, scale: float = 1.0):
        super().__init__()
        _check_rank(r, in_channels, out_channels)
        self.r = r
        geom = dict(kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation, groups=groups)
        self.conv = nn.Conv2d(in_channels, out_channels, bias=bias, **geom)
        self.lora_down = nn.Conv2d(in_channels, r, bias=False, **geom)
        self.dropout = nn.Dropout(dropout_p)
        self.lora_up = nn.Conv2d(r, out_channels, kernel_size=1, stride=1, padding=0, bias=False)
        self.selector = nn.Identity()
        self.scale = scale
        nn.init.normal_(self.lora_down.weight, std=1 / r)  # ref:127
        nn.init.zeros_(self.lora_up.weight)  # ref:128

    def _frozen(self):
        return self.conv

    def forward(self, input):
        if input.is_cuda:
            return self._forward_device(input)
        low = self.lora_up(self.selector(self.lora_down(input)))  # ref:130-135
        return self.conv(input) + self.dropout(low) * self.scale

    def _forward_device(self, x: torch.Tensor) -> torch.Tensor:
        w, b = self.conv.weight, self.conv.bias
        dt = _autocast_dtype(x, w)
        xc = x if x.dtype == dt else x.to(dt)
        wc, bc = self._shadow(w, dt, "w"), self._shadow(b, dt, "b")
        c = self.conv
        with torch.autocast(device_type=x.device.type, enabled=False):
            return ops.lora_conv(xc, wc, bc, self.lora_down.weight, self.lora_up.weight, self._selector_matrix(),
                                 c.stride, c.padding, c.dilation, c.groups, self.scale, self._dropout_p(),
                                 self.__dict__.get("_grad_sink"))

    def set_selector_from_diag(self, diag: torch.Tensor):  # ref:140-156
        assert diag.shape == (self.r,)
        self.selector = nn.Conv2d(self.r, self.r, kernel_size=1, stride=1, padding=0, bias=False)
        # the reference leaves a 2-D weight here (ref:151), which conv2d cannot run; keep it 4-D
        self.selector.weight.data = (torch.diag(diag).reshape(self.r, self.r, 1, 1)
                                     .to(self.lora_up.weight.device).to(self.lora_up.weight.dtype))


# --------------------------------------------------------------------------- site traversal
def _find_children(model, search_class: List[Type[nn.Module]] = [nn.Linear]):  # ref:172-186
    wanted = tuple(search_class)
    for parent in model.modules():
        for name, module in parent.named_children():
            if isinstance(module, wanted):
                yield parent, name, module


def _find_modules_v2(model, ancestor_class: Optional[Set[str]] = None,
                     search_class: List[Type[nn.Module]] = [nn.Linear],
                     exclude_children_of: Optional[List[Type[nn.Module]]] = None,
                     ) -> Iterator[Tuple[nn.Module, str, nn.Module]]:
    """ref:189-232.  Lazily yields ``(parent, name, module)`` for each ``search_class`` instance found
    below each module whose *class name* is in ``ancestor_class`` (all modules if None), skipping
    children of adapters.  The order of this generator IS the ``{model}:{i}`` index of the file format,
    and it tolerates ``parent._modules[name]`` being swapped while it runs.
    """
    if exclude_children_of is None:
        exclude_children_of = [LoraInjectedLinear, LoraInjectedConv2d]
    wanted, skip = tuple(search_class), tuple(exclude_children_of or ())
    if ancestor_class is not None:
        roots = (m for m in model.modules() if type(m).__name__ in ancestor_class)
    else:
        roots = list(model.modules())
    for root in roots:
        for dotted, module in root.named_modules():
            if not isinstance(module, wanted):
                continue
            owner_path, _, leaf = dotted.rpartition(".")
            owner = root.get_submodule(owner_path) if owner_path else root
            if skip and isinstance(owner, skip):
                continue
            yield owner, leaf, module


_find_modules = _find_modules_v2  # ref:252


def _as_param(t) -> nn.Parameter:
    return t if isinstance(t, nn.Parameter) else nn.Parameter(t)


def _wrap_like(child: nn.Module, r: int, **kw) -> _Adapter:
    """New adapter around ``child`` (a Linear/Conv2d or the frozen op inside an adapter) that ALIASES
    the frozen weight/bias Parameters (ref:290-292, 340-342, 358-360)."""
    if isinstance(child, nn.Linear):
        new = LoraInjectedLinear(child.in_features, child.out_features, child.bias is not None, r=r, **kw)
        new.linear.weight = child.weight
        if child.bias is not None:
            new.linear.bias = child.bias
    else:
        new = LoraInjectedConv2d(child.in_channels, child.out_channels, child.kernel_size, child.stride,
                                 child.padding, child.dilation, child.groups, child.bias is not None, r=r, **kw)
        new.conv.weight = child.weight
        if child.bias is not None:
            new.conv.bias = child.bias
    return new


def _finish_injection(parent, name, new, child, loras, params, names):
    new.to(child.weight.device).to(child.weight.dtype)  # ref:295, 363
    parent._modules[name] = new  # ref:296, 367
    params.append(new.lora_up.parameters())  # up first, then down (ref:298-299)
    params.append(new.lora_down.parameters())
    if loras is not None:  # resume: flat [up0, down0, up1, ...] list (ref:301-303)
        new.lora_up.weight = _as_param(loras.pop(0))
        new.lora_down.weight = _as_param(loras.pop(0))
    new.lora_up.weight.requires_grad = True
    new.lora_down.weight.requires_grad = True
    names.append(name)


def inject_trainable_lora(model: nn.Module, target_replace_module: Set[str] = DEFAULT_TARGET_REPLACE, r: int = 4,
                          loras=None, verbose: bool = False, dropout_p: float = 0.0, scale: float = 1.0):
    """ref:255-309.  Swap every nn.Linear under the target blocks for a LoraInjectedLinear; returns
    ``([up-params, down-params, ...] generators, names)``.  Default dropout here is 0.0."""
    params, names = [], []
    if loras is not None:
        loras = torch.load(loras)
    for parent, name, child in _find_modules(model, target_replace_module, search_class=[nn.Linear]):
        if verbose:
            print("LoRA Injection : injecting lora into ", name)
            print("LoRA Injection : weight shape", child.weight.shape)
        new = _wrap_like(child, r, dropout_p=dropout_p, scale=scale)
        _finish_injection(parent, name, new, child, loras, params, names)
    return params, names


def inject_trainable_lora_extended(model: nn.Module, target_replace_module: Set[str] = UNET_EXTENDED_TARGET_REPLACE,
                                   r: int = 4, loras=None):
    """ref:312-380.  As above for nn.Linear AND nn.Conv2d (exact classes); adapter dropout stays at
    the constructor default 0.1."""
    params, names = [], []
    if loras is not None:
        loras = torch.load(loras)
    for parent, name, child in _find_modules(model, target_replace_module, search_class=[nn.Linear, nn.Conv2d]):
        if type(child) not in (nn.Linear, nn.Conv2d):
            continue
        new = _wrap_like(child, r)
        _finish_injection(parent, name, new, child, loras, params, names)
    return params, names


# --------------------------------------------------------------------------- extraction / saving
def _adapters(model, target_replace_module):
    return _find_modules(model, target_replace_module, search_class=[LoraInjectedLinear, LoraInjectedConv2d])


def extract_lora_ups_down(model, target_replace_module=DEFAULT_TARGET_REPLACE):  # ref:383-397
    pairs = [(m.lora_up, m.lora_down) for _, _, m in _adapters(model, target_replace_module)]
    if not pairs:
        raise ValueError("No lora injected.")
    return pairs


def extract_lora_as_tensor(model, target_replace_module=DEFAULT_TARGET_REPLACE, as_fp16=True):  # ref:400-421
    out = []
    for _, _, m in _adapters(model, target_replace_module):
        up, down = m.realize_as_lora()  # scale is folded into `up` here (and only here)
        if as_fp16:
            up, down = up.to(torch.float16), down.to(torch.float16)
        out.append((up, down))
    if not out:
        raise ValueError("No lora injected.")
    return out


def save_lora_weight(model, path="./lora.pt", target_replace_module=DEFAULT_TARGET_REPLACE):  # ref:424-436
    flat = []
    for up, down in extract_lora_ups_down(model, target_replace_module=target_replace_module):
        flat.append(up.weight.to("cpu").to(torch.float16))
        flat.append(down.weight.to("cpu").to(torch.float16))
    torch.save(flat, path)


def save_lora_as_json(model, path="./lora.json"):  # ref:439-448
    flat = []
    for up, down in extract_lora_ups_down(model):
        flat.append(up.weight.detach().cpu().numpy().tolist())
        flat.append(down.weight.detach().cpu().numpy().tolist())
    with open(path, "w") as f:
        json.dump(flat, f)


def _emit_safeloras(weights, metadata, embeds, outpath):
    for token, tensor in embeds.items():
        metadata[token] = EMBED_FLAG
        weights[token] = tensor
    print(f"Saving weights to {outpath}")
    safe_save({k: v.contiguous() for k, v in weights.items()}, outpath, metadata)


def save_safeloras_with_embeds(modelmap: Dict[str, Tuple[nn.Module, Set[str]]] = {},
                               embeds: Dict[str, torch.Tensor] = {}, outpath="./lora.safetensors"):
    """ref:451-483.  Keys ``{name}:{i}:up|down``; metadata ``name -> json(targets)``,
    ``{name}:{i}:rank -> str``, ``token -> "<embed>"``."""
    weights, metadata = {}, {}
    for name, (model, targets) in modelmap.items():
        metadata[name] = json.dumps(list(targets))
        for i, (up, down) in enumerate(extract_lora_as_tensor(model, targets)):
            metadata[f"{name}:{i}:rank"] = str(down.shape[0])
            weights[f"{name}:{i}:up"] = up
            weights[f"{name}:{i}:down"] = down
    _emit_safeloras(weights, metadata, embeds, outpath)


def save_safeloras(modelmap: Dict[str, Tuple[nn.Module, Set[str]]] = {}, outpath="./lora.safetensors"):  # ref:486-490
    return save_safeloras_with_embeds(modelmap=modelmap, outpath=outpath)


def convert_loras_to_safeloras_with_embeds(modelmap: Dict[str, Tuple[str, Set[str], int]] = {},
                                           embeds: Dict[str, torch.Tensor] = {}, outpath="./lora.safetensors"):
    """ref:493-528.  ``.pt`` lists ([up0, down0, ...]) -> one safetensors file."""
    weights, metadata = {}, {}
    for name, (path, targets, r) in modelmap.items():
        metadata[name] = json.dumps(list(targets))
        for pos, weight in enumerate(torch.load(path)):
            i, is_up = pos // 2, pos % 2 == 0
            if is_up:
                metadata[f"{name}:{i}:rank"] = str(r)
                weights[f"{name}:{i}:up"] = weight
            else:
                weights[f"{name}:{i}:down"] = weight
    _emit_safeloras(weights, metadata, embeds, outpath)


def convert_loras_to_safeloras(modelmap: Dict[str, Tuple[str, Set[str], int]] = {}, outpath="./lora.safetensors"):
    convert_loras_to_safeloras_with_embeds(modelmap=modelmap, outpath=outpath)  # ref:531-535


def parse_safeloras(safeloras) -> Dict[str, Tuple[List[nn.parameter.Parameter], List[int], List[str]]]:
    """ref:538-596.  -> ``{name: ([up0, down0, up1, ...] Parameters, ranks, targets)}``."""
    metadata = safeloras.metadata()
    model_of = lambda key: key.split(":")[0]
    keys = sorted(safeloras.keys(), key=model_of)  # stable, like list.sort(key=...)
    out = {}
    for name, grp in groupby(keys, model_of):
        info = metadata.get(name)
        if not info:
            raise ValueError(f"Tensor {name} has no metadata - is this a Lora safetensor?")
        if info == EMBED_FLAG:
            continue
        targets = json.loads(info)
        grp = list(grp)
        ranks = [4] * (len(grp) // 2)
        weights = [None] * len(grp)
        for key in grp:
            _, idx, direction = key.split(":")
            idx = int(idx)
            ranks[idx] = int(metadata[f"{name}:{idx}:rank"])
            weights[idx * 2 + (1 if direction == "down" else 0)] = nn.parameter.Parameter(safeloras.get_tensor(key))
        out[name] = (weights, ranks, targets)
    return out


def parse_safeloras_embeds(safeloras) -> Dict[str, torch.Tensor]:  # ref:599-617
    metadata = safeloras.metadata()
    return {k: safeloras.get_tensor(k) for k in safeloras.keys() if metadata.get(k) == EMBED_FLAG}


def load_safeloras(path, device="cpu"):  # ref:620-622
    return parse_safeloras(safe_open(path, framework="pt", device=device))


def load_safeloras_embeds(path, device="cpu"):  # ref:625-627
    return parse_safeloras_embeds(safe_open(path, framework="pt", device=device))


def load_safeloras_both(path, device="cpu"):  # ref:630-632
    f = safe_open(path, framework="pt", device=device)
    return parse_safeloras(f), parse_safeloras_embeds(f)


# --------------------------------------------------------------------------- merge (K3)
def collapse_lora(model, alpha=1.0):
    """ref:635-669.  ``W <- W + alpha * (up @ down)`` for every adapter (conv: both factors flattened
    from dim 1); ``.scale`` is NOT applied; the frozen weight becomes a NEW Parameter.

    Device weights: all sites are merged by one launch of the fused kernel per dtype group, with the
    reference's rounding sequence.  CPU weights: the reference's torch expression.
    """
    targets = UNET_EXTENDED_TARGET_REPLACE | TEXT_ENCODER_EXTENDED_TARGET_REPLACE
    device_sites, installs = [], []
    for _, name, m in _find_modules(model, targets, search_class=[LoraInjectedLinear, LoraInjectedConv2d]):
        is_lin = isinstance(m, LoraInjectedLinear)
        print("Collapsing Lin Lora in" if is_lin else "Collapsing Conv Lora in", name)
        frozen = m.linear if is_lin else m.conv
        w, up, down = frozen.weight.data, m.lora_up.weight.data, m.lora_down.weight.data
        if w.is_cuda and m.r <= _C.MAX_RANK:  # (rank-joined LoRAs beyond the kernels' rank limit: the torch expression)
            up = up.to(w.device).contiguous()
            down = down.to(w.device)
            if down.dtype != up.dtype:
                down = down.to(up.dtype)
            down = down.contiguous()
            new_w = torch.empty_like(w, memory_format=torch.contiguous_format)
            device_sites.append((w.contiguous(), new_w, up, down))
            installs.append((frozen, new_w))
        else:
            delta = (up.flatten(start_dim=1) @ down.flatten(start_dim=1)).reshape(w.shape).type(w.dtype).to(w.device)
            frozen.weight = nn.Parameter(w + alpha * delta)
    if device_sites:
        ops.merge_sites(device_sites, alpha, _C.ROUND_REFERENCE)
        for frozen, new_w in installs:
            frozen.weight = nn.Parameter(new_w)
    invalidate_caches(model)


# --------------------------------------------------------------------------- inference-time patching
def _take_rank(r):
    return r.pop(0) if isinstance(r, list) else r


def _install_loaded(parent, name, new, frozen_weight, loras):
    parent._modules[name] = new
    up, down = loras.pop(0), loras.pop(0)
    new.lora_up.weight = nn.Parameter(up.type(frozen_weight.dtype))  # ref:706-711, 789-794
    new.lora_down.weight = nn.Parameter(down.type(frozen_weight.dtype))
    new.to(frozen_weight.device)


def monkeypatch_or_replace_lora(model, loras, target_replace_module=DEFAULT_TARGET_REPLACE,
                                r: Union[int, List[int]] = 4):
    """ref:672-713.  (Re)load a flat ``[up, down, ...]`` list into Linear sites (dropout 0.1 default)."""
    for parent, name, child in _find_modules(model, target_replace_module,
                                             search_class=[nn.Linear, LoraInjectedLinear]):
        src = child.linear if isinstance(child, LoraInjectedLinear) else child
        new = _wrap_like(src, _take_rank(r))
        _install_loaded(parent, name, new, src.weight, loras)


def monkeypatch_or_replace_lora_extended(model, loras, target_replace_module=DEFAULT_TARGET_REPLACE,
                                         r: Union[int, List[int]] = 4):
    """ref:716-796.  Linear+Conv2d variant; a site is skipped when the next tensor's ndim (2 vs 4)
    does not match the site kind (ref:731-732, 756-757)."""
    search = [nn.Linear, LoraInjectedLinear, nn.Conv2d, LoraInjectedConv2d]
    for parent, name, child in _find_modules(model, target_replace_module, search_class=search):
        kind = type(child)
        if kind in (nn.Linear, LoraInjectedLinear):
            if len(loras[0].shape) != 2:
                continue
            src = child.linear if kind is LoraInjectedLinear else child
        elif kind in (nn.Conv2d, LoraInjectedConv2d):
            if len(loras[0].shape) != 4:
                continue
            src = child.conv if kind is LoraInjectedConv2d else child
        else:  # subclass of a searched type: the reference reuses the previous site's adapter; do nothing
            continue
        new = _wrap_like(src, _take_rank(r))
        _install_loaded(parent, name, new, src.weight, loras)


def monkeypatch_or_replace_safeloras(models, safeloras):  # ref:799-809
    for name, (lora, ranks, target) in parse_safeloras(safeloras).items():
        model = getattr(models, name, None)
        if not model:
            print(f"No model provided for {name}, contained in Lora")
            continue
        monkeypatch_or_replace_lora_extended(model, lora, target, ranks)


def monkeypatch_remove_lora(model):
    """ref:812-847.  Strip adapters, handing the SAME frozen Parameters back to fresh Linear/Conv2d."""
    for parent, name, m in _find_modules(model, search_class=[LoraInjectedLinear, LoraInjectedConv2d]):
        if name == "":
            continue  # an adapter visited as its own root (ancestor_class=None quirk): nothing to swap
        if isinstance(m, LoraInjectedLinear):
            src = m.linear
            plain = nn.Linear(src.in_features, src.out_features, src.bias is not None)
        else:
            src = m.conv
            plain = nn.Conv2d(in_channels=src.in_channels, out_channels=src.out_channels,
                              kernel_size=src.kernel_size, stride=src.stride, padding=src.padding,
                              dilation=src.dilation, groups=src.groups, bias=src.bias is not None)
        plain.weight = src.weight
        if src.bias is not None:
            plain.bias = src.bias
        parent._modules[name] = plain
    _C.invalidate_weight_caches()


def monkeypatch_add_lora(model, loras, target_replace_module=DEFAULT_TARGET_REPLACE, alpha: float = 1.0,
                         beta: float = 1.0):
    """ref:850-874.  ``up <- alpha*up_new + beta*up`` (same for down) on existing Linear adapters."""
    for parent, name, m in _find_modules(model, target_replace_module, search_class=[LoraInjectedLinear]):
        w = m.linear.weight
        up_new, down_new = loras.pop(0), loras.pop(0)
        cur = parent._modules[name]
        cur.lora_up.weight = nn.Parameter(up_new.type(w.dtype).to(w.device) * alpha
                                          + cur.lora_up.weight.to(w.device) * beta)
        cur.lora_down.weight = nn.Parameter(down_new.type(w.dtype).to(w.device) * alpha
                                            + cur.lora_down.weight.to(w.device) * beta)
        cur.to(w.device)


_ADAPTER_NAMES = ("LoraInjectedLinear", "LoraInjectedConv2d")  # matched by class-name string (ref:879, 885, 1030)


def tune_lora_scale(model, alpha: float = 1.0):  # ref:877-880
    for m in model.modules():
        if type(m).__name__ in _ADAPTER_NAMES:
            m.scale = alpha


def set_lora_diag(model, diag: torch.Tensor):  # ref:883-886
    for m in model.modules():
        if type(m).__name__ in _ADAPTER_NAMES:
            m.set_selector_from_diag(diag)


# legacy names used by the reference's notebooks
monkeypatch_lora = monkeypatch_or_replace_lora
monkeypatch_replace_lora = monkeypatch_or_replace_lora


# --------------------------------------------------------------------------- TI helpers / pipeline patching
def _text_lora_path(path: str) -> str:  # ref:889-891
    assert path.endswith(".pt"), "Only .pt files are supported"
    return ".".join(path.split(".")[:-1] + ["text_encoder", "pt"])


def _ti_lora_path(path: str) -> str:  # ref:894-896
    assert path.endswith(".pt"), "Only .pt files are supported"
    return ".".join(path.split(".")[:-1] + ["ti", "pt"])


def apply_learned_embed_in_clip(learned_embeds, text_encoder, tokenizer,
                                token: Optional[Union[str, List[str]]] = None, idempotent=False):
    """ref:899-942.  Add each learned token to the tokenizer (renaming ``<x>`` -> ``<x-1>`` ... on a clash
    unless idempotent) and write its embedding row."""
    if isinstance(token, str):
        trained = [token]
    elif isinstance(token, list):
        assert len(learned_embeds.keys()) == len(token), \
            "The number of tokens and the number of embeds should be the same"
        trained = token
    else:
        trained = list(learned_embeds.keys())
    for token in trained:
        print(token)
        embeds = learned_embeds[token]
        added = tokenizer.add_tokens(token)
        if not idempotent:
            i = 1
            while added == 0:
                print(f"The tokenizer already contains the token {token}.")
                token = f"{token[:-1]}-{i}>"
                print(f"Attempting to add the token {token}.")
                added = tokenizer.add_tokens(token)
                i += 1
        elif added == 0:
            print(f"The tokenizer already contains the token {token}.")
            print(f"Replacing {token} embedding.")
        text_encoder.resize_token_embeddings(len(tokenizer))
        text_encoder.get_input_embeddings().weight.data[tokenizer.convert_tokens_to_ids(token)] = embeds
    return token


def load_learned_embed_in_clip(learned_embeds_path, text_encoder, tokenizer,
                               token: Optional[Union[str, List[str]]] = None, idempotent=False):  # ref:945-955
    apply_learned_embed_in_clip(torch.load(learned_embeds_path), text_encoder, tokenizer, token, idempotent)


def patch_pipe(pipe, maybe_unet_path, token: Optional[str] = None, r: int = 4, patch_unet=True, patch_text=True,
               patch_ti=True, idempotent_token=True, unet_target_replace_module=DEFAULT_TARGET_REPLACE,
               text_target_replace_module=TEXT_ENCODER_DEFAULT_TARGET_REPLACE):
    """ref:958-1022.  One-call patcher for ``.pt`` triples or a ``.safetensors`` file."""
    if maybe_unet_path.endswith(".pt"):
        if maybe_unet_path.endswith(".ti.pt"):
            unet_path = maybe_unet_path[:-6] + ".pt"
        elif maybe_unet_path.endswith(".text_encoder.pt"):
            unet_path = maybe_unet_path[:-16] + ".pt"
        else:
            unet_path = maybe_unet_path
        ti_path, text_path = _ti_lora_path(unet_path), _text_lora_path(unet_path)
        if patch_unet:
            print("LoRA : Patching Unet")
            monkeypatch_or_replace_lora(pipe.unet, torch.load(unet_path), r=r,
                                        target_replace_module=unet_target_replace_module)
        if patch_text:
            print("LoRA : Patching text encoder")
            monkeypatch_or_replace_lora(pipe.text_encoder, torch.load(text_path),
                                        target_replace_module=text_target_replace_module, r=r)
        if patch_ti:
            print("LoRA : Patching token input")
            token = load_learned_embed_in_clip(ti_path, pipe.text_encoder, pipe.tokenizer, token=token,
                                               idempotent=idempotent_token)
    elif maybe_unet_path.endswith(".safetensors"):
        f = safe_open(maybe_unet_path, framework="pt", device="cpu")
        monkeypatch_or_replace_safeloras(pipe, f)
        tok_dict = parse_safeloras_embeds(f)
        if patch_ti:
            apply_learned_embed_in_clip(tok_dict, pipe.text_encoder, pipe.tokenizer, token=token,
                                        idempotent=idempotent_token)
        return tok_dict


@torch.no_grad()
def inspect_lora(model):  # ref:1025-1042
    moved = {}
    for name, m in model.named_modules():
        if type(m).__name__ in _ADAPTER_NAMES:
            delta = m.lora_up.weight.data.clone().flatten(1) @ m.lora_down.weight.data.clone().flatten(1)
            moved.setdefault(name, []).append(delta.flatten().abs().mean().item())
    return moved


def save_all(unet, text_encoder, save_path, placeholder_token_ids=None, placeholder_tokens=None, save_lora=True,
             save_ti=True, target_replace_module_text=TEXT_ENCODER_DEFAULT_TARGET_REPLACE,
             target_replace_module_unet=DEFAULT_TARGET_REPLACE, safe_form=True):
    """ref:1045-1110.  ``.pt`` triple (unet / .text_encoder.pt / .ti.pt) or one ``.safetensors``."""

    def learned_rows():
        rows = {}
        for tok, tok_id in zip(placeholder_tokens, placeholder_token_ids):
            row = text_encoder.get_input_embeddings().weight[tok_id]
            print(f"Current Learned Embeddings for {tok}:, id {tok_id} ", row[:4])
            rows[tok] = row.detach().cpu()
        return rows

    if not safe_form:
        if save_ti:
            ti_path = _ti_lora_path(save_path)
            torch.save(learned_rows(), ti_path)
            print("Ti saved to ", ti_path)
        if save_lora:
            save_lora_weight(unet, save_path, target_replace_module=target_replace_module_unet)
            print("Unet saved to ", save_path)
            save_lora_weight(text_encoder, _text_lora_path(save_path),
                             target_replace_module=target_replace_module_text)
            print("Text Encoder saved to ", _text_lora_path(save_path))
    else:
        assert save_path.endswith(".safetensors"), f"Save path : {save_path} should end with .safetensors"
        loras, embeds = {}, {}
        if save_lora:
            loras["unet"] = (unet, target_replace_module_unet)
            loras["text_encoder"] = (text_encoder, target_replace_module_text)
        if save_ti:
            embeds = learned_rows()
        save_safeloras_with_embeds(loras, embeds, save_path)
