#!/usr/bin/env python
"""Workload + reducer for HBM-traffic counters of the adapter kernels (K1/K2/K4), calibrated on a same-run copy.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o p -- python scripts/pmc_kernels.py run
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o p -- python scripts/pmc_kernels.py run
    python scripts/pmc_kernels.py reduce gpurun_out/pmc_f gpurun_out/pmc_w > profiles/<tag>_adapter_pmc.json
(separate passes per counter, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; FETCH_SIZE under-reports wide
streams on gfx950, hence the calibration on a 16 B/lane copy of known size in the same run.)
"""
import csv
import glob
import json
import os
import sys

CASES = {  # name -> algorithmic bytes (bf16 activations, f32 factors, r = 4); M, K, N = 16384, 320, 2560
    "linear_fwd_kernel": ("fwd (16384,320,2560): read X, read+write Y", 16384 * 320 * 2 + 2 * 16384 * 2560 * 2),
    "linear_bwd_g_kernel": ("G pass (16384,2560): read G", 16384 * 2560 * 2),
    "linear_bwd_x_kernel": ("X pass (16384,320): read X, read+write dX", 3 * 16384 * 320 * 2),
    "linear_ws_kernel": ("K1 fused, weight-stationary (16384,320,2560): read X, W, write Y",
                         (16384 * 320 + 2560 * 320 + 16384 * 2560) * 2),
    "linear_gemm_fwd_kernel": ("K1 fused, LDS ring tile 24 (16384,320,2560): read X, W, write Y",
                               (16384 * 320 + 2560 * 320 + 16384 * 2560) * 2),
    "linear_bwd_factors_kernel": ("dUp + dDown partials in one launch (16384,320,2560): read G, read X",
                                  16384 * 2560 * 2 + 16384 * 320 * 2),
    "conv_down_fwd_kernel": ("conv down 3x3 (4,320,64x64): read X", 4 * 320 * 4096 * 2),
    "conv_up_fwd_kernel": ("conv up (4,320,64x64): read+write Y", 2 * 4 * 320 * 4096 * 2),
    "conv_bwd_g_kernel": ("conv G pass: read G", 4 * 320 * 4096 * 2),
    "conv_bwd_down_kernel": ("conv dDown pass: read X", 4 * 320 * 4096 * 2),
    "conv_bwd_dx_kernel": ("conv dX pass: read+write dX", 2 * 4 * 320 * 4096 * 2),
    # channels-last form (csrc/conv_nhwc.hip), same site at rank 16: X once / dX read + write / X once; the packed
    # factor (92 KB), T and Gt (1 MB each) are counted in the algorithmic bytes, the dDown partials are not
    "conv3_down_nhwc_kernel": ("NHWC T = conv3x3(X; down) (4,320,64x64), r16: read X, pf; write T",
                               4 * 320 * 4096 * 2 + 9 * 320 * 16 * 2 + 16384 * 16 * 4),
    "conv3_dx_nhwc_kernel": ("NHWC dX += (4,320,64x64), r16: read+write dX, read Gt, pd",
                             2 * 4 * 320 * 4096 * 2 + 16384 * 16 * 4 + 320 * 5 * 32 * 2),
    "conv3_ddown_nhwc_kernel": ("NHWC dDown partials (4,320,64x64), r16: read X, Gt",
                                4 * 320 * 4096 * 2 + 16384 * 16 * 4),
}
COPY_ELEMS = 192_634_880


def run():
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lora_amd import _C, ops

    DEV = "cuda:0"
    a = torch.randn(COPY_ELEMS // 64, device=DEV).to(torch.bfloat16).repeat(64)
    b = torch.empty_like(a)
    for _ in range(3):
        torch.neg(a, out=b)
    del a, b
    M, K, N, r = 16384, 320, 2560, 4
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    y = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    g = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    dx = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    A, B = torch.randn(r, K, device=DEV) * 0.25, torch.randn(N, r, device=DEV) * 0.05
    plan = _C.linear_plan(M, K, N, r)
    gt_part, up_part, down_part = (torch.empty(k, device=DEV) for k in
                                   (plan.gt_part_floats, plan.up_part_floats, plan.down_part_floats))
    flush = torch.empty(400_000_000 // 4, device=DEV)  # > 256 MiB L3: evict between launches
    for _ in range(3):
        flush.fill_(1.0)
        t = _C.linear_fwd_(x, y, A, B, 1e-3, None, 0.0, 0, 0)
        flush.fill_(2.0)
        _C.linear_bwd_g(g, t, B, gt_part, up_part, 1.0, 0.0, 0, 0)
        flush.fill_(3.0)
        _C.linear_bwd_x(x, dx, gt_part, plan.nct_g, A, None, down_part)
    W_ = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV).to(torch.bfloat16)
    yb = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    site = dict(wp=_C.ws_pack(W_), N=N, bias=bias, down=A, up=B, scale=1e-3, y=yb)
    gt1 = torch.randn(M, r, device=DEV)
    for _ in range(3):
        flush.fill_(1.0)
        _C.linear_ws(x, [site])
        flush.fill_(2.0)
        _C.linear_gemm_fwd(x, W_, bias, A, B, 1e-3, 24)
        flush.fill_(3.0)
        _C.linear_bwd_factors(g, t, up_part, x, gt1, down_part, r, 1.0)
    Bc, C, Hh, ks = 4, 320, 64, 3
    cp = _C.conv_plan(Bc, C, C, Hh, Hh, ks, r)
    xc = torch.randn(Bc, C, Hh, Hh, device=DEV).to(torch.bfloat16)
    yc, gc, dxc = (torch.randn_like(xc) for _ in range(3))
    down, up = torch.randn(r, C, ks, ks, device=DEV) * 0.1, torch.randn(C, r, 1, 1, device=DEV) * 0.05
    t_part, gtp, gt, upp, dnp = ops.conv_buffers(cp, Bc, r, Hh * Hh, DEV)
    tc = torch.empty(Bc, r, Hh, Hh, device=DEV)
    for _ in range(3):
        flush.fill_(1.0)
        _C.conv_down_fwd(xc, down, None, t_part, tc, ks)
        flush.fill_(2.0)
        _C.conv_up_fwd_(yc, tc, up, 1e-3, 0.0, 0, 0)
        flush.fill_(3.0)
        _C.conv_bwd_g(gc, tc, up, None, gtp, gt, upp, 1.0, 0.0, 0, 0)
        flush.fill_(4.0)
        _C.conv_bwd_x(xc, dxc, gt, down, dnp, ks)
    r16 = 16
    xl = xc.contiguous(memory_format=torch.channels_last)
    dxl = torch.randn_like(xc).contiguous(memory_format=torch.channels_last)
    down16 = torch.randn(r16, C, 3, 3, device=DEV) * 0.1
    gt16 = torch.randn(Bc * Hh * Hh, r16, device=DEV)
    np_ = _C.conv3_nhwc_plan(Bc, C, Hh, Hh, r16)
    pf, pd = _C.conv3_nhwc_pack(down16, torch.bfloat16, np_)
    part = torch.empty(int(np_.down_part_floats), device=DEV)
    for _ in range(3):
        flush.fill_(1.0)
        _C.conv3_nhwc_down_fwd(xl, pf, r16)
        flush.fill_(2.0)
        _C.conv3_nhwc_bwd_dx_(dxl, gt16, pd)
        flush.fill_(3.0)
        _C.conv3_nhwc_bwd_down(xl, gt16, part)
    torch.cuda.synchronize()


def reduce(fdir, wdir):
    def per_kernel(dirpath, counter):
        path = glob.glob(dirpath + "/**/*counter_collection.csv", recursive=True)[0]
        out = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter:
                    out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
        return out

    def pick(d, key):
        vals = [v for k in d if key in k for v in d[k]]
        return sorted(vals)[len(vals) // 2] if vals else None

    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    cf, cw = pick(fetch, "neg_kernel"), pick(write, "neg_kernel")
    res = {"calibration": {"copy_bytes_each_way": COPY_ELEMS * 2, "fetch_counts": cf, "write_counts": cw},
           "note": "bytes = counts * (known copy bytes / copy counts); L3 flushed between launches", "kernels": {}}
    for key, (what, alg) in CASES.items():
        kf, kw = pick(fetch, key), pick(write, key)
        if kf is None or kw is None:
            continue
        rd, wr = kf * COPY_ELEMS * 2 / cf, kw * COPY_ELEMS * 2 / cw
        res["kernels"][key] = {"what": what, "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr),
                               "algorithmic_bytes": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 3)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        reduce(sys.argv[2], sys.argv[3])
