#!/usr/bin/env python
"""Where the time of one batched SVD-distillation group goes (host-timed phases with synchronisation)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lora_amd import _C, cli_svd as S

dev = "cuda:0"
def T(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r

for (B, N, K) in ((30, 320, 320), (48, 1280, 1280), (17, 1280, 11520), (6, 10240, 1280), (5, 1280, 23040), (1, 640, 17280)):
    g = torch.Generator(device=dev).manual_seed(0)
    tuned = torch.randn(B, N, K, device=dev, generator=g); base = torch.randn(B, N, K, device=dev, generator=g)
    l = 16
    t_stack, res = T(lambda: torch.stack([(t.float() - b.float()).flatten(1) for t, b in zip(list(tuned), list(base))]))
    om = torch.randn(B, l, K, device=dev)
    t_rd, y = T(lambda: _C.rowdot_batched(res, om, _C.FACTOR_RK))
    t_orth, q = T(lambda: S._orth(y))
    t_cr, z = T(lambda: _C.colreduce_batched(res, q, _C.FACTOR_KR))
    t_rd2, _ = T(lambda: _C.rowdot_batched(res, z, _C.FACTOR_KR))
    t_orthz, _ = T(lambda: S._orth(z))
    b = _C.colreduce_batched(res, q, _C.FACTOR_RK)
    t_svd, (ub, s, vh) = T(lambda: torch.linalg.svd(b, full_matrices=False))
    def gram_route():
        gm = torch.bmm(b, b.transpose(1, 2))
        w, v = torch.linalg.eigh(gm)
        return w, v
    t_eigh, _ = T(gram_route)
    U = torch.randn(B, N, 8, device=dev); Vh = torch.randn(B, 8, K, device=dev)
    t_clamp, _ = T(lambda: S._clamp_pairs(U, Vh, 0.99))
    t_tr, _ = T(lambda: res.transpose(1, 2).contiguous())
    t_all, _ = T(lambda: S.distill_group(list(tuned), list(base), 8, 0.99, None), 2)
    GB = B * N * K * 4 / 1e9
    print(f"B={B} {N}x{K} ({GB:.2f} GB): stack {t_stack:.2f} ms | rowdot {t_rd:.2f} ({GB/t_rd*1e3:.0f} GB/s) | colreduce {t_cr:.2f} ({GB/t_cr*1e3:.0f} GB/s) | "
          f"rowdot_kr {t_rd2:.2f} | transpose {t_tr:.2f} | orth[N] {t_orth:.2f} orth[K] {t_orthz:.2f} | svd {t_svd:.2f} eigh-route {t_eigh:.2f} | clamp {t_clamp:.2f} | distill_group {t_all:.2f}", flush=True)
