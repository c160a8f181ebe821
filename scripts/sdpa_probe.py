#!/usr/bin/env python
"""Which scaled_dot_product_attention backend is fastest for SD1.5's head sizes (40/80/160) on this GPU?
fwd+bwd time per call, hipGraph-replayed.  Plumbing probe for lora_amd/standin/unet.py (not a product kernel)."""
import json
import sys
import os

import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.kbench import timeit  # noqa: E402

DEV = "cuda:0"


def run(B, H, Sq, Sk, D, backend, pad_to=None):
    q = torch.randn(B, H, Sq, D, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B, H, Sk, D, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B, H, Sk, D, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    go = torch.randn(B, H, Sq, D, device=DEV, dtype=torch.bfloat16)

    def fn():
        qq, kk, vv = q, k, v
        scale = D ** -0.5
        if pad_to:
            qq, kk, vv = (F.pad(t, (0, pad_to - D)) for t in (q, k, v))
        with sdpa_kernel(backend):
            o = F.scaled_dot_product_attention(qq, kk, vv, scale=scale)
        if pad_to:
            o = o[..., :D]
        o.backward(go)
        q.grad = k.grad = v.grad = None

    try:
        med, _ = timeit(fn, iters=5, inner=5)
        return round(med * 1e6, 1)
    except Exception as e:  # noqa: BLE001
        return f"{type(e).__name__}: {str(e)[:80]}"


if __name__ == "__main__":
    print(torch.__version__, torch.cuda.get_device_name(0))
    for (B, H, Sq, Sk, D) in ((4, 8, 4096, 4096, 40), (4, 8, 1024, 1024, 80), (4, 8, 256, 256, 160),
                              (4, 8, 64, 64, 160), (4, 8, 4096, 77, 40), (4, 8, 1024, 77, 80)):
        res = dict(B=B, H=H, Sq=Sq, Sk=Sk, D=D)
        for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION),
                         ("math", SDPBackend.MATH)):
            res[name] = run(B, H, Sq, Sk, D, be)
        padto = 64 if D <= 64 else 128 if D <= 128 else 256
        res[f"flash_pad{padto}"] = run(B, H, Sq, Sk, D, SDPBackend.FLASH_ATTENTION, padto)
        res[f"efficient_pad{padto}"] = run(B, H, Sq, Sk, D, SDPBackend.EFFICIENT_ATTENTION, padto)
        print(json.dumps(res), flush=True)
