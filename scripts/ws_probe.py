#!/usr/bin/env python
"""A few eager launches of the fused GEMM kernels per site shape, for rocprofv3 (kernel durations without launch gaps,
SQ counters):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ws_trace -o p -- python scripts/ws_probe.py
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA \\
              SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/ws_pmc -o p -- python scripts/ws_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lora_amd import _C  # noqa: E402

DEV = "cuda:0"
r = 4
shapes = [(16384, 320, 320), (16384, 320, 2560), (4096, 640, 640), (1024, 1280, 1280), (308, 768, 320)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (M, K, N) in shapes:
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV).to(torch.bfloat16)
    A = torch.randn(r, K, device=DEV) * 0.25
    B = torch.randn(N, r, device=DEV) * 0.05
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    site = dict(wp=_C.ws_pack(W), N=N, bias=bias, down=A, up=B, scale=1e-3, y=y)
    for _ in range(10):
        _C.linear_ws(x, [site])
        torch.cuda.synchronize()
    for _ in range(10):
        _C.linear_gemm_fwd(x, W, bias, A, B, 1e-3, 22)
        torch.cuda.synchronize()
    for _ in range(10):
        torch.nn.functional.linear(x, W, bias)
        torch.cuda.synchronize()
print("done")
