"""Frozen twins of the adapter sites: the SAME launches the merged-weight path issues, minus everything LoRA.

``bench.py --adapters none`` (the ``frozen_only`` leg behind ``lora_overhead_ms``) used to run plain ``nn.Linear`` modules:
no grouped q / k / v GEMM (three GEMMs where the adapter step runs one), dense pack / unpack copies around the head-padded
attention core, ``G @ W`` input gradients where the adapter step multiplies by a transposed scratch weight — the
"adapter-free" step was SLOWER than the adapter step and the difference said nothing about the LoRA kernels (VERDICT r5,
measurement honesty).  A :class:`FrozenSite` stands where ``inject_trainable_lora`` would have put a
``LoraInjectedLinear`` and runs that site through the merged path's own autograd nodes
(``ops.LoraLinearMergedFunction`` / ``ops.LoraLinearMergedGroupFunction``) with ``down = up = None``:

* forward  = the frozen GEMM on the weight in the layout the adapter path's scratch weight has (head-padded rows / columns;
  q / k / v of a self-attention block, k / v of a cross-attention block: row ranges of ONE concatenated buffer);
* backward = the frozen GEMM on the transposed copy (column ranges of one transposed buffer for a group), accumulated by
  ``addmm_`` inside a group — and nothing else: no merge launch, no factor pass, no fold.

The layouts are built ONCE (the weights are frozen).  ``step(adapters) - step(frozen twins)`` is therefore what the
hand-written LoRA launches cost: merge_step (W -> W_eff, W_eff^T), factor pack + pass + fold, and the flat optimiser state.
Host-model code (outside SURVEY.md section 8), used by the measurement only."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _rows_padded(w: torch.Tensor, lay) -> torch.Tensor:
    h, d, D = lay   # [h d, K] -> [h D, K], every head's d rows followed by D - d zero rows
    return F.pad(w.view(h, d, w.shape[1]), (0, 0, 0, D - d)).reshape(h * D, w.shape[1]).contiguous()


def _cols_padded(w: torch.Tensor, lay) -> torch.Tensor:
    h, d, D = lay   # [N, h d] -> [N, h D]
    return F.pad(w.view(w.shape[0], h, d), (0, D - d)).reshape(w.shape[0], h * D).contiguous()


class FrozenSite(nn.Module):
    """A frozen ``nn.Linear`` at an adapter site, run through the merged path's launches (module docstring).  Keeps the
    original module as ``.linear`` (same parameter objects; state-dict keys gain ``.linear`` exactly as injection does)."""

    def __init__(self, linear: nn.Linear):
        super().__init__()
        self.linear = linear
        self._lay = {}

    def layout(self, in_heads, out_heads, need_t: bool):
        key = (in_heads, out_heads)
        e = self._lay.get(key)
        if e is None:
            w, b = self.linear.weight.detach(), self.linear.bias
            if out_heads:
                w = _rows_padded(w, out_heads)
            if in_heads:
                w = _cols_padded(w, in_heads)
            b_eff = None if b is None else (ops.pack_heads(b.detach(), out_heads).contiguous() if out_heads else b.detach())
            e = self._lay[key] = dict(w_eff=w, b_eff=b_eff, w_eff_t=None)
        if need_t and e["w_eff_t"] is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FrozenSite: a transposed layout must exist before hipGraph capture")
            e["w_eff_t"] = e["w_eff"].t().contiguous()
        return e

    def forward_heads(self, x, in_heads=None, out_heads=None):
        if not x.is_cuda:
            y = self.linear(ops.unpack_heads(x, in_heads) if in_heads else x)
            return ops.pack_heads(y, out_heads) if out_heads else y
        need_dx = x.requires_grad and torch.is_grad_enabled()
        e = self.layout(in_heads, out_heads, need_dx)
        xc = x if x.dtype == e["w_eff"].dtype else x.to(e["w_eff"].dtype)
        with torch.autocast(device_type=x.device.type, enabled=False):
            return ops.LoraLinearMergedFunction.apply(xc, e["w_eff"], e["b_eff"], None, None, 1.0, None, in_heads,
                                                      out_heads, e["w_eff_t"] if need_dx else None)

    def forward(self, x):
        return self.forward_heads(x, None, None)


_GROUPS = {}


def frozen_linear_group(sites, x: torch.Tensor, out_heads=None) -> Optional[list]:
    """Several :class:`FrozenSite` on ONE input as one GEMM on the concatenated weight (the frozen twin of
    ``lora.lora_linear_group`` on the merged path); None when the caller should call the modules one by one."""
    if not x.is_cuda or len(sites) < 2 or not all(isinstance(s, FrozenSite) for s in sites):
        return None
    need_dx = x.requires_grad and torch.is_grad_enabled()
    key = tuple(id(s) for s in sites) + (out_heads,)
    g = _GROUPS.get(key)
    if g is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("frozen_linear_group: a new group appeared during hipGraph capture")
        es = [s.layout(None, out_heads, False) for s in sites]
        cat = torch.cat([e["w_eff"] for e in es]).contiguous()
        bias = None
        if any(e["b_eff"] is not None for e in es):
            bias = torch.cat([e["b_eff"] if e["b_eff"] is not None else
                              torch.zeros(e["w_eff"].shape[0], dtype=cat.dtype, device=cat.device) for e in es])
        g = _GROUPS[key] = dict(cat=cat, cat_t=None, bias=bias, splits=[e["w_eff"].shape[0] for e in es], keep=sites)
    if need_dx and g["cat_t"] is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("frozen_linear_group: the transposed buffer must exist before hipGraph capture")
        g["cat_t"] = g["cat"].t().contiguous()
    xc = x if x.dtype == g["cat"].dtype else x.to(g["cat"].dtype)
    flat, pos = [], 0
    for n_o in g["splits"]:
        flat += [g["cat"][pos:pos + n_o], None, None, None, 1.0, None, out_heads,
                 g["cat_t"][:, pos:pos + n_o] if need_dx else None]
        pos += n_o
    with torch.autocast(device_type=x.device.type, enabled=False):
        return list(ops.LoraLinearMergedGroupFunction.apply(xc, len(sites), g["cat"], g["bias"], *flat))


def install_frozen_twins(model: nn.Module, target_replace_module=None) -> int:
    """Put a :class:`FrozenSite` on every ``nn.Linear`` that ``inject_trainable_lora`` would adapt (same finder, same
    target classes); returns the number of sites."""
    from ..lora import DEFAULT_TARGET_REPLACE, _find_modules_v2 as _find_modules

    n = 0
    for parent, name, child in list(_find_modules(model, target_replace_module or DEFAULT_TARGET_REPLACE,
                                                  search_class=[nn.Linear], exclude_children_of=[FrozenSite])):
        parent._modules[name] = FrozenSite(child)
        n += 1
    return n
