"""ctypes binding of the input-stationary K1/K2 experiment (``scripts/gemm_xs/libgemm_xs.so``; ``make -C scripts/gemm_xs``).

Round 5 built the kernel as the measured answer to "own the GEMM half"; the step never routed to it (it wins the plain
K = 320 product by 1.1-1.2x over hipBLASLt, loses fused and at K = 640; worth 0.2 % of the headline step).  Round 6
 moved it out of the product library and the stable C-ABI: it lives here with its kbench leg
(``scripts/kbench.py --what xs``) and its parity checks (``scripts/gemm_xs/check_gemm_xs.py``, run with
``python -m pytest scripts/gemm_xs/check_gemm_xs.py -m gpu``)."""
import ctypes as C
import os
import subprocess

import torch

from lora_amd import _C
from lora_amd._C import WsSite, _check, _off, _ptr, _stream, dtype_code, ws_pack

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def require():
    """The experiment's library (built on first use; needs hipcc).  Loads the product library first (error plumbing)."""
    global _lib
    if _lib is None:
        _C.require()
        so = os.path.join(HERE, "libgemm_xs.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-C", HERE], check=True)
        lib = C.CDLL(so, mode=C.RTLD_GLOBAL)
        i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
        lib.lora_amd_xs_config.argtypes = [i32, C.POINTER(i32), C.POINTER(i32)]
        lib.lora_amd_linear_xs.argtypes = [vp, i64, i64, i32, i32, C.POINTER(WsSite), vp]
        lib.lora_amd_xs_config.restype = lib.lora_amd_linear_xs.restype = C.c_int
        lib.lora_amd_xs_set_tuning.argtypes = [i32, i32]
        lib.lora_amd_xs_set_tuning.restype = None
        _lib = lib
    return _lib


_XS_K = (320, 640)


def xs_supported(x: torch.Tensor, K: int, N: int, r: int) -> bool:
    """Can the input-stationary kernel (scripts/gemm_xs/gemm_xs.hip) run a site of contraction length K, width N, rank r on rows ``x``?"""
    return (K in _XS_K and x.dtype in (torch.bfloat16, torch.float16) and r <= 16 and N % 4 == 0 and x.dim() == 2
            and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0)


def linear_xs(x: torch.Tensor, s: dict):
    """ONE site through the input-stationary kernel: the site dict of :func:`linear_ws` (``down`` / ``up`` None: the plain
    product on the packed weight).  Returns (y, t)."""
    lib = require()
    M, K = x.shape
    d = WsSite()
    N, fl = int(s["N"]), int(s.get("flayout", 0))
    down, up = s.get("down"), s.get("up")
    r = 0
    if down is not None:
        r = down.shape[1] if fl & 1 else down.shape[0]
        if down.dtype != torch.float32 or up.dtype != torch.float32 or not down.is_contiguous() or not up.is_contiguous():
            raise ValueError("linear_xs: contiguous f32 factors expected")
    y = s.get("y")
    if y is None:
        y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    t = torch.empty((M, r), dtype=torch.float32, device=x.device) if (r and s.get("want_t", True)) else None
    bias = s.get("bias")
    d.wp, d.bias, d.y, d.down, d.up, d.t_out = s["wp"].data_ptr(), _ptr(bias), y.data_ptr(), _ptr(down), _ptr(up), _ptr(t)
    d.ldy, d.N, d.r, d.panel_begin, d.flayout = y.stride(0), N, r, 0, fl
    d.reserved = 1 if s.get("rowmajor") else 0   # ``wp`` = the [N, K] weight itself (contiguous rows) instead of its pack
    d.scale, d.t_scale = float(s.get("scale", 1.0)), float(s.get("t_scale", 1.0))
    off_s, off_p = _off(s.get("off", 0))
    d.dropout_p, d.seed, d.offset, d.offset_dev = float(s.get("p", 0.0)), int(s.get("seed", 0)), off_s, off_p
    _check(lib.lora_amd_linear_xs(x.data_ptr(), x.stride(0), M, K, dtype_code(x.dtype), C.byref(d), _stream()),
           "lora_amd_linear_xs")
    return y, t


def xs_set_tuning(slabs: int = 0, panels_per_group: int = 0) -> None:
    require().lora_amd_xs_set_tuning(int(slabs), int(panels_per_group))


def linear_xs_fwd(x, weight, bias, down, up, scale, p: float = 0.0, seed: int = 0, off=0):
    """(y, t) of one site through the input-stationary kernel (same contract as :func:`linear_ws_fwd`)."""
    return linear_xs(x, dict(wp=ws_pack(weight), N=weight.shape[0], bias=bias, down=down, up=up, scale=scale, p=p, seed=seed,
                             off=off))


def linear_xs_dx(g, weight, down, up, scale, p: float = 0.0, seed: int = 0, off=0):
    """(dX, Gt) of one site through the input-stationary kernel (same contract as :func:`linear_ws_dx`)."""
    return linear_xs(g, dict(wp=ws_pack(weight, True), N=weight.shape[1], down=up, up=down, scale=scale, t_scale=scale,
                             flayout=3, p=p, seed=seed, off=off))
