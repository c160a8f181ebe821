"""Attention core of the stand-in UNet: ``scaled_dot_product_attention`` with a per-shape choice of library kernel.

Host-model plumbing, not a product kernel: the dense QK^T / PV contractions stay on the library's MFMA flash kernels
(north star: "MFMA only for the frozen attention/ResNet GEMMs").  SD1.5's head sizes are 40 / 80 / 160; on MI355X /
PyTorch-ROCm 2.10 the flash kernel at D=40, S=4096 runs 2.7x slower than the memory-efficient kernel on the same
tensors zero-padded to D=64 (scripts/sdpa_probe.py: 2673 us vs 984 us fwd+bwd), so each (Sq, Sk, D, dtype) shape
picks the fastest of {flash, efficient} x {as is, head dim padded to the next 64/128/256} once, timed on first use
(outside hipGraph capture), and sticks to it.  Zero padding of the head dimension leaves QK^T and the kept columns of
PV unchanged; the softmax scale is passed explicitly.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F

try:
    from torch.nn.attention import SDPBackend, sdpa_kernel
except ImportError:  # pragma: no cover
    SDPBackend = sdpa_kernel = None

from .._C import _TuneCache  # noqa: E402  (shared JSON cache of per-shape choices, LORA_AMD_TUNE_CACHE)

_CHOICE = _TuneCache("sdpa")
_AGREED = set()  # shapes whose kernel choice has been agreed across the ranks of a data-parallel job
_TUNE = os.environ.get("LORA_AMD_SDPA_TUNE", "1") != "0"


def _padded(d: int) -> int:
    return 64 if d <= 64 else 128 if d <= 128 else 256 if d <= 256 else d


def _run(q, k, v, backend: Optional[str], pad_to: int, pad_v: bool = True):
    """``pad_to`` > d zero-pads the head dimension of q and k (and of v when ``pad_v``; with v left as is the output
    comes back unpadded — one pad and one slice copy fewer each way — where the kernel accepts Dv != Dk)."""
    d = q.shape[-1]
    scale = d ** -0.5
    if pad_to > d:
        q, k = F.pad(q, (0, pad_to - d)), F.pad(k, (0, pad_to - d))
        if pad_v:
            v = F.pad(v, (0, pad_to - d))
    if backend is None or sdpa_kernel is None:
        o = F.scaled_dot_product_attention(q, k, v, scale=scale)
    else:
        with sdpa_kernel(getattr(SDPBackend, backend)):
            o = F.scaled_dot_product_attention(q, k, v, scale=scale)
    return o[..., :d] if (pad_to > d and pad_v) else o


def _pad_candidates(d: int):
    """Head sizes worth timing: as is, and the next multiples of 16 / 32 / 64-128-256 (what the library kernels are
    specialised for: 40 -> 48, 64; 80 -> 96, 128)."""
    return sorted({d, -(-d // 16) * 16, -(-d // 32) * 32, _padded(d)})


def _tune(q, k, v) -> Tuple[Optional[str], int]:
    d = q.shape[-1]
    cands = [(None, d), ("EFFICIENT_ATTENTION", d)]
    for pad in _pad_candidates(d):
        if pad > d:
            cands += [("EFFICIENT_ATTENTION", pad), ("FLASH_ATTENTION", pad), ("EFFICIENT_ATTENTION", pad, False)]
    best, best_t = (None, d), float("inf")
    for cand in cands:
        try:
            qq, kk, vv = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
            go = torch.randn_like(q)
            _run(qq, kk, vv, *cand).backward(go)  # warm: lazy kernel loading / JIT is not what is compared
            t = float("inf")
            for _ in range(3):  # minimum over repeats: one stall must not pick the wrong kernel for the whole run
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(2):
                    _run(qq, kk, vv, *cand).backward(go)
                b.record()
                torch.cuda.synchronize()
                t = min(t, a.elapsed_time(b))
        except Exception:  # noqa: BLE001 - a backend that rejects the shape is simply not a candidate
            continue
        if t < best_t:
            best, best_t = tuple(cand), t
    return best


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v for [B, H, S, d] tensors."""
    if not q.is_cuda:
        return F.scaled_dot_product_attention(q, k, v)
    key = repr((q.shape[0], q.shape[1], q.shape[2], k.shape[2], q.shape[3], str(q.dtype),
                q.requires_grad or k.requires_grad))
    choice = _CHOICE.get(key)
    if choice is not None:
        choice = tuple(choice)
    # multi-rank agreement only inside a training forward: every rank reaches the same attention calls in the same order
    # there (same model, same step), which a rank-0-only evaluation / sampling call or an uneven last batch does not
    # guarantee — a collective entered by one rank alone would hang the job
    multi = (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and torch.is_grad_enabled()
             and (q.requires_grad or k.requires_grad or v.requires_grad))
    if (choice is None or (multi and key not in _AGREED)) and not torch.cuda.is_current_stream_capturing():
        if choice is None and _TUNE and q.dtype in (torch.bfloat16, torch.float16):
            with torch.enable_grad():
                choice = _tune(q, k, v)
        if multi:
            # data-parallel replicas run the same kernels: every rank adopts rank 0's pick the first time it meets a
            # shape, instead of each timing the candidates on its own GPU and possibly settling on different ones
            box = [list(choice) if choice is not None else None]
            dist.broadcast_object_list(box, src=0)
            choice = tuple(box[0]) if box[0] is not None else None
            _AGREED.add(key)
        if choice is None:
            choice = (None, q.shape[-1])  # the agreed default is cached too: one collective per shape, not one per call
        _CHOICE[key] = list(choice)
    if choice is None:
        choice = (None, q.shape[-1])
    return _run(q, k, v, *choice)


FORCE_PAD: Optional[int] = None  # tests: take the padded-layout path with this head size, on any device


def padded_choice(B: int, H: int, Sq: int, Sk: int, d: int, dtype, grad: bool):
    """(backend, D) if the kernel chosen for this shape runs on q, k, v all zero-padded to head size D > d, else None.
    Only a choice that has already been made counts (the first call of a shape goes through :func:`sdpa`, which times
    the candidates); callers that can produce / consume the padded layout directly then skip the pad and slice copies."""
    if FORCE_PAD is not None:
        return (None, FORCE_PAD) if FORCE_PAD > d else None
    c = _CHOICE.get(repr((B, H, Sq, Sk, d, str(dtype), grad)))
    if c is None or c[1] <= d or (len(c) > 2 and not c[2]):
        return None
    return c[0], c[1]


def sdpa_padded(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, d: int, backend: Optional[str]) -> torch.Tensor:
    """The attention core on tensors that already carry the padded head size (pad columns zero); ``d`` is the true
    head size (softmax scale)."""
    scale = d ** -0.5
    if backend is None or sdpa_kernel is None or not q.is_cuda:
        return F.scaled_dot_product_attention(q, k, v, scale=scale)
    with sdpa_kernel(getattr(SDPBackend, backend)):
        return F.scaled_dot_product_attention(q, k, v, scale=scale)


def choices():
    return dict(_CHOICE)
