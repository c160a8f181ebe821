#!/usr/bin/env python
"""Kernel micro-benchmarks (HIP events on the launch stream): algorithmic GB/s of each primitive."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lora_amd import _C  # noqa: E402

if os.environ.get("LORA_AMD_LIB"):   # a measurement build of the same ABI (scripts/fm_trace/); kbench only
    _C.LIB_PATH = os.path.abspath(os.environ["LORA_AMD_LIB"])
from lora_amd.standin import sd15_lora_site_shapes  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20, warm=3, inner=20):
    """Median / best time of ONE call of ``fn``: ``inner`` back-to-back calls are captured into a hipGraph and the
    replay is bracketed by HIP events on the launch stream, so the host's ctypes/launch cost (~5-10 us per call,
    more than many of these kernels take) is not what is measured.  Includes the ~1.5 us dependent-launch gap."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        for _ in range(inner):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        graph.replay()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e-3 / inner for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def merge_plan(qkvo_only=False, r=4, wdt=torch.bfloat16, abdt=torch.float32, inplace=False):
    shapes = sd15_lora_site_shapes()
    if qkvo_only:
        shapes = [s for i, s in enumerate(shapes) if i % 9 != 4]
    sites = []
    for N, K in shapes:
        w = (torch.randn(N, K, device=DEV) * 0.03).to(wdt)
        sites.append((w, w if inplace else torch.empty_like(w), (torch.randn(N, r, device=DEV) * 0.05).to(abdt),
                      (torch.randn(r, K, device=DEV) * 0.25).to(abdt)))
    return _C.MergePlan(sites)


def bench_merge(args):
    out = []
    for qkvo in (False, True):
        for tile, bpc in ((16384, 4), (16384, 104), (32768, 4), (32768, 104), (8192, 104)):
            _C.merge_set_tuning(tile, bpc)
            plan = merge_plan(qkvo)
            med, best = timeit(lambda: plan.launch(0.7), iters=args.iters)
            gbs = plan.bytes_algorithmic / med / 1e9
            out.append(dict(kernel="merge", qkvo=qkvo, tile=tile, blocks_per_cu=bpc, sites=plan.n_sites,
                            tiles=plan.total_tiles, MB=plan.bytes_algorithmic / 1e6, us=med * 1e6, best_us=best * 1e6,
                            GBs=gbs, frac8=gbs / 8000))
            print(json.dumps(out[-1]), flush=True)
    _C.merge_set_tuning(16384, 104)
    # plain device copy of the same bytes as a ceiling reference
    n = int(385e6 / 2)
    a = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    b = torch.empty_like(a)
    med, _ = timeit(lambda: b.copy_(a), iters=args.iters)
    print(json.dumps(dict(kernel="torch_copy", MB=2 * n * 2 / 1e6, us=med * 1e6, GBs=2 * n * 2 / med / 1e9)), flush=True)
    return out


def mstep_tensors(r=4):
    shapes = sd15_lora_site_shapes()
    tens = []
    for N, K in shapes:
        w = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
        tens.append((w, torch.randn(N, r, device=DEV) * 0.05, torch.randn(r, K, device=DEV) * 0.25, torch.empty_like(w),
                     torch.empty(K, N, dtype=w.dtype, device=DEV) if K != 768 else None))
    return tens


def mstep_plan(tens):
    return _C.MergeStepPlan([dict(w=w, up=u, down=d, out=o, out_t=ot, row_heads=None, col_heads=None, key=i)
                             for i, (w, u, d, o, ot) in enumerate(tens)])


def bench_mstep(args):
    """The in-step merge (lora_amd_merge_step) on the 144 Linear sites of the SD1.5 UNet, W_eff^T for the sites whose input
    takes a gradient (all but the cross-attention k / v): tile geometry x dither form x rounding."""
    tens = mstep_tensors(args.rank)
    for tile in (0, 1, 2, 3):
        for dith, rounding in ((2, _C.ROUND_DITHER), (1, _C.ROUND_DITHER), (2, _C.ROUND_ONCE)):
            _C.merge_step_set_tuning(tile, dith)
            plan = mstep_plan(tens)
            med, best = timeit(lambda: plan.launch(0.7, rounding), iters=args.iters)
            gbs = plan.bytes_algorithmic / med / 1e9
            print(json.dumps(dict(kernel="merge_step", tile=("128x64", "64x128", "128x128", "256x64")[tile],
                                  dither={1: "per element", 2: "per chunk"}[dith] if rounding == _C.ROUND_DITHER else "none",
                                  sites=plan.n_sites, tiles=plan.total_tiles, MB=plan.bytes_algorithmic / 1e6,
                                  us=med * 1e6, best_us=best * 1e6, GBs=gbs, frac8=gbs / 8000)), flush=True)
    _C.merge_step_set_tuning(2, 2)


def bench_r16(args):
    """Rank-16 kernels of BASELINE configs[3] (768^2, batch 1, dropout 0.1): matrix-core form (csrc/rank16_mfma.hip) against
    the VALU kernel it replaces, same call, hook on / off.  Bytes: the activation once (rowdot, bwd_g) or twice (rank_update)."""
    r, p, dt = 16, 0.1, torch.bfloat16
    for (M, C) in ((9216, 320), (9216, 2560), (2304, 640), (2304, 5120), (576, 1280), (576, 10240), (144, 1280), (77, 1280)):
        a = torch.randn(M, C, device=DEV).to(dt)
        t = torch.randn(M, r, device=DEV)
        fkr, frk = torch.randn(C, r, device=DEV) * 0.2, torch.randn(r, C, device=DEV) * 0.2
        plan = _C.linear_plan(M, 320, C, r)
        gt_part = torch.empty(plan.gt_part_floats, device=DEV)
        up_part = torch.empty(plan.up_part_floats, device=DEV)
        calls = {"rowdot_masked": (lambda: _C.rowdot(a, fkr, _C.FACTOR_KR, 0.7, None, False, p, 5, 9), M * C * 2),
                 "rowdot": (lambda: _C.rowdot(a, frk, _C.FACTOR_RK, 1.0), M * C * 2),
                 "rank_update": (lambda: _C.rank_update_(a, t, fkr, _C.FACTOR_KR, 1e-3, p, 5, 9), 2 * M * C * 2),
                 "bwd_g": (lambda: _C.linear_bwd_g(a, t, fkr, gt_part, up_part, 0.7, p, 5, 9), M * C * 2)}
        for name, (fn, byts) in calls.items():
            res = dict(kernel=name, M=M, C=C, r=r)
            for tag, on in (("mfma", 1), ("valu", 0)):
                prev = _C.rank16_mfma(on)
                med, _ = timeit(fn, args.iters)
                _C.rank16_mfma(prev)
                res[tag + "_us"], res[tag + "_GBs"] = round(med * 1e6, 2), round(byts / med / 1e9, 1)
            print(json.dumps(res), flush=True)


def bench_linear(args):
    r = 4
    for (M, K, N) in ((16384, 320, 320), (16384, 320, 2560), (4096, 640, 640), (4096, 640, 5120), (1024, 1280, 1280),
                      (1024, 1280, 10240), (308, 768, 320)):
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        y = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        A = torch.randn(r, K, device=DEV) * 0.25
        B = torch.randn(N, r, device=DEV) * 0.05
        t = _C.rowdot(x, A, _C.FACTOR_RK)
        res = dict(M=M, K=K, N=N)
        med, _ = timeit(lambda: _C.rowdot(x, A, _C.FACTOR_RK), args.iters)
        res["rowdot_us"], res["rowdot_GBs"] = med * 1e6, (M * K * 2 + M * r * 4) / med / 1e9
        med, _ = timeit(lambda: _C.rank_update_(y, t, B, _C.FACTOR_KR, 1e-3), args.iters)
        res["rank_update_us"], res["rank_update_GBs"] = med * 1e6, (2 * M * N * 2 + M * r * 4) / med / 1e9
        med, _ = timeit(lambda: _C.colreduce(y, t, _C.FACTOR_KR, 1.0), args.iters)
        res["colreduce_us"], res["colreduce_GBs"] = med * 1e6, (M * N * 2 + M * r * 4) / med / 1e9
        # fused kernels (what the adapter actually launches)
        plan = _C.linear_plan(M, K, N, r)
        g = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        dx = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        gt_part, up_part, down_part = (torch.empty(k, device=DEV) for k in
                                       (plan.gt_part_floats, plan.up_part_floats, plan.down_part_floats))
        med, _ = timeit(lambda: _C.linear_fwd_(x, y, A, B, 1e-3, None, 0.0, 0, 0), args.iters)
        res["fwd_us"], res["fwd_GBs"] = med * 1e6, (M * K * 2 + 2 * M * N * 2) / med / 1e9
        med, _ = timeit(lambda: _C.linear_bwd_g(g, t, B, gt_part, up_part, 1.0, 0.0, 0, 0), args.iters)
        res["bwd_g_us"], res["bwd_g_GBs"] = med * 1e6, (M * N * 2) / med / 1e9
        med, _ = timeit(lambda: _C.linear_bwd_x(x, dx, gt_part, plan.nct_g, A, None, down_part), args.iters)
        res["bwd_x_us"], res["bwd_x_GBs"] = med * 1e6, (3 * M * K * 2) / med / 1e9
        # K1 fully fused (one MFMA kernel: frozen GEMM + LoRA branch) per output-tile choice
        Wf = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV).to(torch.bfloat16)
        if _C.gemm_supported(x, Wf, N, r):
            tiles = (21, 22, 23, 24, 31, 32, 33, 34)
            for tile in tiles:
                med, _ = timeit(lambda: _C.linear_gemm_fwd(x, Wf, bias, A, B, 1e-3, tile), args.iters)
                res[f"gemm_fused_t{tile}_us"] = med * 1e6
            best = min(res[f"gemm_fused_t{t}_us"] for t in tiles)
            res["gemm_fused_best_us"] = best
            res["gemm_fused_TF"] = 2 * M * K * N / best / 1e6
            res["gemm_fused_GBs"] = (M * K + N * K + M * N) * 2 / best / 1e3
        med, _ = timeit(lambda: _C.linear_fwd_(x, torch.nn.functional.linear(x, Wf, bias), A, B, 1e-3, None, 0.0, 0, 0),
                        args.iters)
        res["gemm_plus_lora_us"] = med * 1e6
        # what the reference's op sequence costs on the same device (5 ATen launches)
        W = torch.randn(N, K, device=DEV).to(torch.bfloat16)
        Ab, Bb = A.to(torch.bfloat16), B.to(torch.bfloat16)

        def ref_branch():
            return y + torch.nn.functional.dropout(torch.nn.functional.linear(torch.nn.functional.linear(x, Ab), Bb), 0.0) * 0.5

        med, _ = timeit(ref_branch, args.iters)
        res["aten_branch_us"] = med * 1e6
        med, _ = timeit(lambda: torch.nn.functional.linear(x, W), args.iters)
        res["frozen_gemm_us"] = med * 1e6
        res["frozen_gemm_TF"] = 2 * M * K * N / med / 1e12
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)


def bench_xs(args):
    """K1/K2 input-stationary kernel (scripts/gemm_xs/gemm_xs.hip, an experiment outside the product library) per site shape: forward with the LoRA branch, with dropout, the plain
    product on a packed weight (what a merged-weight site would run), the input gradient — next to the weight-stationary
    kernel, the library GEMM alone and the library GEMM + one branch launch."""
    from scripts.gemm_xs import xs as XS
    r = args.rank
    HBM = 8.0e12
    shapes = ((16384, 320, 320), (16384, 320, 960), (16384, 320, 2560), (4096, 640, 640), (4096, 640, 1920), (4096, 640, 5120),
              (9216, 320, 320), (9216, 320, 2560), (2304, 640, 640), (2304, 640, 5120))
    for (M, K, N) in shapes:
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        W = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV).to(torch.bfloat16)
        A = torch.randn(r, K, device=DEV) * 0.25
        B = torch.randn(N, r, device=DEV) * 0.05
        byts = (M * K + N * K + M * N) * 2 + (N + K) * r * 4 + M * r * 4
        res = dict(M=M, K=K, N=N, r=r, MB=round(byts / 1e6, 2), floor_us_8TBs=round(byts / HBM * 1e6, 2))
        wp = _C.ws_pack(W)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        site = dict(wp=wp, N=N, bias=bias, down=A, up=B, scale=1e-3, y=y)
        res["xs_us"] = timeit(lambda: XS.linear_xs(x, site), args.iters)[0] * 1e6
        res["xs_frac8"] = byts / (res["xs_us"] * 1e-6) / HBM
        sweep = {}
        for sl in ((1, 2, 4) if K == 320 else (1, 2)):
            for pg in (1, 2, 4, 8, 64):
                XS.xs_set_tuning(sl, pg)
                sweep[f"sl{sl}_pg{pg}"] = round(timeit(lambda: XS.linear_xs(x, dict(wp=wp, N=N, bias=bias, y=y)), args.iters)[0] * 1e6, 2)
        XS.xs_set_tuning(0, 0)
        res["xs_plain_sweep"] = sweep
        res["xs_plain_best"] = min(sweep.items(), key=lambda kv: kv[1])
        res["xs_plain_rowmajor_us"] = timeit(lambda: XS.linear_xs(x, dict(wp=W, N=N, bias=bias, y=y, rowmajor=True)), args.iters)[0] * 1e6
        res["xs_drop_us"] = timeit(lambda: XS.linear_xs(x, dict(site, p=0.1, seed=7, off=11)), args.iters)[0] * 1e6
        res["xs_plain_us"] = timeit(lambda: XS.linear_xs(x, dict(wp=wp, N=N, bias=bias, y=y)), args.iters)[0] * 1e6
        res["ws_us"] = timeit(lambda: _C.linear_ws(x, [site]), args.iters)[0] * 1e6
        res["ws_drop_us"] = timeit(lambda: _C.linear_ws(x, [dict(site, p=0.1, seed=7, off=11)]), args.iters)[0] * 1e6
        res["lib_gemm_us"] = timeit(lambda: torch.nn.functional.linear(x, W, bias), args.iters)[0] * 1e6
        if _C.fused_ok(x, N, r):
            res["lib_gemm_plus_lora_us"] = timeit(lambda: _C.linear_fwd_(x, torch.nn.functional.linear(x, W, bias), A, B, 1e-3,
                                                                         None, 0.0, 0, 0), args.iters)[0] * 1e6
        if N in (320, 640):
            g = torch.randn(M, N, device=DEV).to(torch.bfloat16)
            wpt = _C.ws_pack(W, True)
            dx = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
            st = dict(wp=wpt, N=K, down=B, up=A, scale=1.0, t_scale=1.0, flayout=3, y=dx)
            res["xs_dx_us"] = timeit(lambda: XS.linear_xs(g, st), args.iters)[0] * 1e6
            res["ws_dx_us"] = timeit(lambda: _C.linear_ws(g, [st]), args.iters)[0] * 1e6
            res["lib_dx_us"] = timeit(lambda: g @ W, args.iters)[0] * 1e6
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)


def bench_ws(args):
    """K1/K2 weight-stationary fused GEMM (gemm_ws.hip) vs the LDS-ring kernel and the library GEMM + linear_fwd."""
    r = 4
    HBM = 8.0e12
    for (M, K, N) in ((16384, 320, 320), (16384, 320, 2560), (4096, 640, 640), (4096, 640, 5120), (1024, 1280, 1280),
                      (1024, 1280, 10240), (256, 1280, 1280), (308, 768, 320), (308, 768, 1280)):
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        W = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV).to(torch.bfloat16)
        A = torch.randn(r, K, device=DEV) * 0.25
        B = torch.randn(N, r, device=DEV) * 0.05
        byts = (M * K + N * K + M * N) * 2 + (N + K) * r * 4 + M * r * 4
        res = dict(M=M, K=K, N=N, MB=round(byts / 1e6, 2), floor_us_8TBs=round(byts / HBM * 1e6, 2))
        wp = _C.ws_pack(W)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        site = dict(wp=wp, N=N, bias=bias, down=A, up=B, scale=1e-3, y=y)
        for rg in (0, 8, 16, 32, 64, 128, 256):
            med, best = timeit(lambda: _C.linear_ws(x, [site], rg), args.iters)
            res[f"ws_rg{rg}_us"] = med * 1e6
        bestk = min((v, k) for k, v in res.items() if k.startswith("ws_rg"))
        res["ws_best"], res["ws_best_us"] = bestk[1], bestk[0]
        res["ws_frac8"] = byts / (bestk[0] * 1e-6) / HBM
        if _C.gemm_supported(x, W, N, r):
            res["ring_best_us"] = min(timeit(lambda: _C.linear_gemm_fwd(x, W, bias, A, B, 1e-3, t_), args.iters)[0]
                                      for t_ in (22, 23, 24, 32, 34)) * 1e6
        plan_ok = _C.fused_ok(x, N, r)
        if plan_ok:
            med, _ = timeit(lambda: _C.linear_fwd_(x, torch.nn.functional.linear(x, W, bias), A, B, 1e-3, None, 0.0, 0, 0),
                            args.iters)
            res["lib_gemm_plus_lora_us"] = med * 1e6
        med, _ = timeit(lambda: torch.nn.functional.linear(x, W, bias), args.iters)
        res["lib_gemm_us"] = med * 1e6
        # input gradient: contraction over N (only where N is a supported contraction length)
        if N in (320, 640, 768, 1280):
            g = torch.randn(M, N, device=DEV).to(torch.bfloat16)
            wpt = _C.ws_pack(W, True)
            dx = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
            st = dict(wp=wpt, N=K, down=B, up=A, scale=1.0, t_scale=1.0, flayout=3, y=dx)
            med, _ = timeit(lambda: _C.linear_ws(g, [st]), args.iters)
            res["ws_dx_us"] = med * 1e6
            med, _ = timeit(lambda: g @ W, args.iters)
            res["lib_dx_us"] = med * 1e6
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)
    # q | k | v of one attention block in ONE launch vs three
    for (M, K) in ((16384, 320), (4096, 640), (1024, 1280)):
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        sites = []
        for i in range(3):
            W = (torch.randn(K, K, device=DEV) * 0.03).to(torch.bfloat16)
            sites.append(dict(wp=_C.ws_pack(W), N=K, down=torch.randn(r, K, device=DEV) * 0.25,
                              up=torch.randn(K, r, device=DEV) * 0.05, scale=1e-3,
                              y=torch.empty(M, K, dtype=torch.bfloat16, device=DEV)))
        one, _ = timeit(lambda: _C.linear_ws(x, sites), args.iters)
        three, _ = timeit(lambda: [_C.linear_ws(x, [s_]) for s_ in sites], args.iters)
        byts = (M * K + 3 * K * K + 3 * M * K) * 2
        print(json.dumps(dict(qkv=[M, K], one_launch_us=round(one * 1e6, 2), three_launches_us=round(three * 1e6, 2),
                              frac8_one=round(byts / one / HBM, 3))), flush=True)


SELF_SHAPES = [(16384, 320, 320), (4096, 640, 640), (1024, 1280, 1280), (256, 1280, 1280), (308, 768, 320),
               (308, 768, 1280), (16384, 320, 2560), (4096, 640, 5120), (1024, 1280, 10240)]


def bench_self(args):
    """The merged-weight path's pieces per site shape (M, K, N): the library GEMMs on W_eff (forward, input gradient),
    lora_amd_linear_bwd_factors_self (T and Gt recomputed inside) next to lora_amd_linear_bwd_factors (T and Gt given) and
    the two rowdot launches that would produce them."""
    r, dt = args.rank, torch.bfloat16
    for (M, K, N) in SELF_SHAPES:
        x = torch.randn(M, K, device=DEV).to(dt)
        g = torch.randn(M, N, device=DEV).to(dt)
        w = (torch.randn(N, K, device=DEV) * 0.03).to(dt)
        down, up = torch.randn(r, K, device=DEV) * 0.25, torch.randn(N, r, device=DEV) * 0.05
        rec = {"M": M, "K": K, "N": N, "r": r, "GX_MB": round((M * K + M * N) * 2 / 1e6, 2)}
        floor = (M * K + M * N) * 2 / 8e12
        sp = _C.factors_self_plan(M, K, N, r)
        if sp.supported:
            up_part = torch.empty(int(sp.up_part_floats), device=DEV)
            down_part = torch.empty(int(sp.down_part_floats), device=DEV)
            t, _ = timeit(lambda: _C.linear_bwd_factors_self(g, x, down, up, up_part, down_part, 1.0))
            rec["self_us"] = round(t * 1e6, 2)
            rec["self_frac8"] = round(floor / t, 3)
            rec["self_nparts"] = sp.nparts
        lp = _C.linear_plan(M, K, N, r)
        if lp.fused:
            tt, gt = torch.randn(M, r, device=DEV), torch.randn(M, r, device=DEV)
            up_p = torch.empty(int(lp.up_part_floats), device=DEV)
            down_p = torch.empty(int(lp.down_part_floats), device=DEV)
            t, _ = timeit(lambda: _C.linear_bwd_factors(g, tt, up_p, x, gt, down_p, r, 1.0))
            rec["factors_given_T_Gt_us"] = round(t * 1e6, 2)
        t, _ = timeit(lambda: (_C.rowdot(x, down, _C.FACTOR_RK), _C.rowdot(g, up, _C.FACTOR_KR)))
        rec["two_rowdots_us"] = round(t * 1e6, 2)
        t, _ = timeit(lambda: torch.nn.functional.linear(x, w))
        rec["lib_fwd_us"] = round(t * 1e6, 2)
        t, _ = timeit(lambda: g @ w)
        rec["lib_dx_us"] = round(t * 1e6, 2)
        print(json.dumps(rec), flush=True)


def sd15_site_list(batch=4, hw=64, seq=77):
    """(M, K, N) of the 144 Linear adapter sites of the default injection at one step's shapes: attention
    projections per level, cross-attention k / v on the text states, GEGLU projections."""
    out = []
    for lvl, (c, nblk) in enumerate(((320, 5), (640, 5), (1280, 5))):
        tokens = batch * (hw >> lvl) ** 2
        for _ in range(nblk):
            out += [(tokens, c, c)] * 4                     # attn1 q k v out
            out += [(tokens, c, c), (batch * seq, 768, c), (batch * seq, 768, c), (tokens, c, c)]  # attn2 q k v out
            out += [(tokens, c, 8 * c)]                     # GEGLU proj
    tokens = batch * (hw >> 3) ** 2                          # mid block
    out += [(tokens, 1280, 1280)] * 4 + [(tokens, 1280, 1280), (batch * seq, 768, 1280), (batch * seq, 768, 1280),
                                          (tokens, 1280, 1280)] + [(tokens, 1280, 10240)]
    return out


def bench_fm(args):
    """The factor-gradient pass of a whole step (144 sites, batch 4, 512^2) both ways: the VALU pass
    (lora_amd_linear_bwd_factors_self_ragged: 128-row blocks, each row block read twice) and the matrix-core pass
    (lora_amd_factor_pack + lora_amd_linear_bwd_factors_mfma_ragged per LDS class: read once), each followed by the fold
    (lora_amd_reduce_batched); algorithmic bytes = G + X once.  LORA_AMD_FM_ROWS / LORA_AMD_FM_NB select
    variants."""
    r, dt = args.rank, torch.bfloat16
    sites = sd15_site_list()
    byts = sum((M * K + M * N) * 2 for M, K, N in sites)
    gs, xs, downs, ups = {}, {}, [], []
    shapes = list(sites)
    shared = os.environ.get("LORA_AMD_FM_SHARED", "0") == "1"   # rounds 4-5: one (G, X) per SHAPE — L2 / MALL hits flattered the pass
    sites = [(M, K, N) if shared else (M, K, N, i) for i, (M, K, N) in enumerate(sites)]
    # as in the model: q / k / v of a self-attention block read ONE x, k / v of a cross-attention block the text states (9 sites
    # per transformer block, sd15_site_list's order); LORA_AMD_FM_OWN_X=1 gives every site its own x
    own_x = os.environ.get("LORA_AMD_FM_OWN_X", "0") == "1"
    xkey = {}
    for i, key in enumerate(sites):
        j = i % 9
        xkey[key] = key if (shared or own_x or j not in (1, 2, 6)) else sites[i - (j if j < 3 else 1)]
    for i, key in enumerate(sites):  # activations per site (1.97 GB; round 6) or per shape, factors per site
        M, K, N = key[:3]
        if key not in gs:
            gs[key] = torch.randn(M, N, device=DEV).to(dt)
            xs[key] = xs[xkey[key]] if xkey[key] is not key else torch.randn(M, K, device=DEV).to(dt)
        downs.append(torch.randn(r, K, device=DEV) * 0.25)
        ups.append(torch.randn(N, r, device=DEV) * 0.05)
    rec = {"sites": len(sites), "activations": "per shape (shared)" if shared else "per site", "GX_GB": round(byts / 1e9, 4), "floor_us_8TBs": round(byts / 8e12 * 1e6, 1)}
    # ---- VALU pass
    vs, rows_v = [], []
    for key, down, up in zip(sites, downs, ups):
        M, K, N = key[:3]
        pl = _C.factors_self_plan(M, K, N, r, _C.SELF_ROWS_DEFERRED)
        up_part, down_part = torch.empty(int(pl.up_part_floats), device=DEV), torch.empty(int(pl.down_part_floats), device=DEV)
        vs.append((gs[key], xs[key], down, up, up_part, down_part, 1.0, None, None))
        rows_v += [(up_part, torch.empty(N, r, device=DEV), pl.nparts, pl.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
                   (down_part, torch.empty(r, K, device=DEV), pl.nparts, pl.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)]
    arr, grid = _C.factors_self_ragged_table(vs, dt)
    tab_v = _C.table_to_device(arr, DEV)
    red_v = _C.make_reduce_table(rows_v, DEV)
    t, _ = timeit(lambda: _C.linear_bwd_factors_self_ragged(tab_v, len(vs), grid, r, dt), inner=5)
    rec["valu_pass_us"], rec["valu_frac8"] = round(t * 1e6, 1), round(byts / 8e12 / t, 3)
    t, _ = timeit(lambda: _C.reduce_batched(*red_v), inner=5)
    rec["valu_fold_us"] = round(t * 1e6, 1)
    # ---- matrix-core pass
    by_cls, packs, rows_m, part_bytes = {}, [], [], 0
    for key, down, up in zip(sites, downs, ups):
        M, K, N = key[:3]
        pl = _C.factors_mfma_plan(M, K, N, r, dt)
        assert pl.supported, (M, K, N)
        up_part, down_part = torch.empty(int(pl.up_part_floats), device=DEV), torch.empty(int(pl.down_part_floats), device=DEV)
        part_bytes += (int(pl.up_part_floats) + int(pl.down_part_floats)) * 4
        pk_down, pk_up = torch.empty(int(pl.pack_down_elems), dtype=dt, device=DEV), torch.empty(int(pl.pack_up_elems), dtype=dt, device=DEV)
        packs.append((down, up, pk_down, pk_up))
        by_cls.setdefault((int(pl.lds_class), int(pl.rows_per_block)), []).append((gs[key], xs[key], pk_down, pk_up, up_part, down_part, 1.0,
                                                        None, None, r, pl))
        rows_m += [(up_part, torch.empty(N, r, device=DEV), pl.nparts, pl.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
                   (down_part, torch.empty(r, K, device=DEV), pl.nparts, pl.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)]
    arr, total = _C.factor_pack_table(packs)
    tab_p = _C.table_to_device(arr, DEV)
    t, _ = timeit(lambda: _C.factor_pack(tab_p, len(packs), total, dt), inner=5)
    rec["pack_us"] = round(t * 1e6, 1)
    tot = 0.0
    tabs = []
    for (cls, rpb), ss in sorted(by_cls.items()):   # one launch per (register class, block height)
        arr, grid = _C.factors_mfma_table(ss, dt, cls)
        tab = _C.table_to_device(arr, DEV)
        tabs.append((tab, len(ss), grid, cls, rpb))
        b = sum((s_[0].numel() + s_[1].numel()) * 2 for s_ in ss)
        t, _ = timeit(lambda: _C.linear_bwd_factors_mfma_ragged(tab, len(ss), grid, cls, dt, False, rpb), inner=5)
        rec[f"mfma_class{cls}_rows{rpb}"] = {"sites": len(ss), "blocks": grid, "GX_GB": round(b / 1e9, 4), "us": round(t * 1e6, 1),
                                             "frac8": round(b / 8e12 / t, 3)}
        tot += t
    # the same launches with the block -> site map behind the table (ABI 7): one scalar load instead of the LDS prefix search
    for (cls, rpb), ss in sorted(by_cls.items()):
        arr, grid = _C.factors_mfma_table(ss, dt, cls)
        raw, moff = _C.factors_mfma_table_bytes(arr, grid)
        tabm = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(DEV)
        t, _ = timeit(lambda: _C.linear_bwd_factors_mfma_ragged(tabm, len(ss), grid, cls, dt, False, rpb, moff), inner=5)
        rec[f"mapped_class{cls}_rows{rpb}_us"] = round(t * 1e6, 1)
        t, _ = timeit(lambda: _C.linear_bwd_factors_mfma_ragged(tabm, len(ss), grid, cls, dt, False, rpb), inner=5)
        rec[f"search_class{cls}_rows{rpb}_us"] = round(t * 1e6, 1)
    # class 1 on each of its kernels (lora_amd_factors_mfma_set_tuning): 0 = 10 pairs, two workgroups per CU, 2 groups in flight;
    # 1 / 2 = 6 pairs, three per CU, 1 / 2 groups in flight; 3 / 4 / 5 = 6 pairs, two per CU, 2 / 3 / 4 groups in flight
    for tab, ns, grid, cls, rpb in tabs:
        if (cls, rpb) != (1, 64):
            continue
        prev = _C.factors_mfma_set_tuning(-1)
        for mode in (0, 1, 2, 3, 4, 5):
            _C.factors_mfma_set_tuning(mode)
            t, _ = timeit(lambda: _C.linear_bwd_factors_mfma_ragged(tab, ns, grid, cls, dt, False, rpb), inner=5)
            rec[f"class1_kernel{mode}_us"] = round(t * 1e6, 1)
        _C.factors_mfma_set_tuning(prev)
    # class 1 by site type and the step's own class-2 table (both block heights in one launch).  (Call c16 also swept runs of
    # 2 / 4 / 8 consecutive row blocks per workgroup with a kernel that has since been removed: profiles/r06_kbench_fm_span.log)
    prev = _C.factors_mfma_set_tuning(-1)
    extra = []
    c1 = by_cls.get((1, 64), [])
    for name, pick in (("c1_square", lambda s_: s_[0].shape[0] > 1024 and s_[0].shape[1] == s_[1].shape[1]),
                       ("c1_geglu", lambda s_: s_[0].shape[1] == 8 * s_[1].shape[1]),
                       ("c1_text_kv", lambda s_: s_[0].shape[0] <= 1024)):
        ss = [s_ for s_ in c1 if pick(s_)]
        if ss:
            arr, grid = _C.factors_mfma_table(ss, dt, 1)
            extra.append((name, _C.table_to_device(arr, DEV), len(ss), grid, 1, 64, sum((s_[0].numel() + s_[1].numel()) * 2 for s_ in ss)))
    c2 = [s_ for (cls, rpb), ss in sorted(by_cls.items()) if cls == 2 for s_ in ss]
    if c2:
        arr, grid = _C.factors_mfma_table(c2, dt, 2)
        extra.append(("c2_mixed", _C.table_to_device(arr, DEV), len(c2), grid, 2, 0, sum((s_[0].numel() + s_[1].numel()) * 2 for s_ in c2)))
    for tab, ns, grid, cls, rpb in tabs:
        extra.append((f"class{cls}_rows{rpb}", tab, ns, grid, cls, rpb, 0))
    # longest blocks first (bytes per row block, descending): the table order ops.flush_factors uses since call c20
    cost = lambda s_: -int(s_[10].rows_per_block) * (s_[0].shape[1] + s_[1].shape[1])  # noqa: E731
    for name, ss, cls, rpb in (("c1_longest_first", sorted(c1, key=cost), 1, 64), ("c2_mixed_longest_first", sorted(c2, key=cost), 2, 0)):
        if ss:
            arr, grid = _C.factors_mfma_table(ss, dt, cls)
            extra.append((name, _C.table_to_device(arr, DEV), len(ss), grid, cls, rpb, sum((s_[0].numel() + s_[1].numel()) * 2 for s_ in ss)))
    # class 2 by site type too
    for name, pick in (("c2_square_640", lambda s_: s_[0].shape[0] == 4096 and s_[0].shape[1] == s_[1].shape[1]),
                       ("c2_geglu_640", lambda s_: s_[0].shape[0] == 4096 and s_[0].shape[1] == 8 * s_[1].shape[1]),
                       ("c2_1280", lambda s_: s_[1].shape[1] == 1280), ("c2_text_kv", lambda s_: s_[1].shape[1] == 768)):
        ss = [s_ for s_ in c2 if pick(s_)]
        if ss:
            arr, grid = _C.factors_mfma_table(ss, dt, 2)
            extra.append((name, _C.table_to_device(arr, DEV), len(ss), grid, 2, 0, sum((s_[0].numel() + s_[1].numel()) * 2 for s_ in ss)))
    for name, tab, ns, grid, cls, rpb, b in extra:
        t, _ = timeit(lambda: _C.linear_bwd_factors_mfma_ragged(tab, ns, grid, cls, dt, False, rpb), inner=5)
        row = {"blocks": grid, "GB": round(b / 1e9, 4), "us": round(t * 1e6, 1), **({"frac8": round(b / 8e12 / t, 3)} if b else {})}
        # ring depths (lora_amd_factors_mfma_set_tuning): class 1 on its kernels 2 .. 5, class-2 mixed-height tables on the 10-pair
        # kernel's ring variants 1 .. 4
        if os.environ.get("LORA_AMD_FM_RINGS", "1") == "1" and b:
            if cls == 1:
                for mode in (2, 3, 4, 5):
                    _C.factors_mfma_set_tuning(mode)
                    t, _ = timeit(lambda: _C.linear_bwd_factors_mfma_ragged(tab, ns, grid, cls, dt, False, rpb), inner=5)
                    row[f"kernel{mode}_us"] = round(t * 1e6, 1)
            elif rpb == 0:
                for w in (1, 2, 3, 4):
                    _C.factors_mfma_set_tuning((prev & 15) | (w << 4))
                    t, _ = timeit(lambda: _C.linear_bwd_factors_mfma_ragged(tab, ns, grid, cls, dt, False, rpb), inner=5)
                    row[f"ring{w}_us"] = round(t * 1e6, 1)
            _C.factors_mfma_set_tuning(prev & 15)
        rec["part_" + name] = row
    red_m = _C.make_reduce_table(rows_m, DEV)
    t, _ = timeit(lambda: _C.reduce_batched(*red_m), inner=5)
    rec["mfma_fold_us"], rec["mfma_partial_MB"] = round(t * 1e6, 1), round(part_bytes / 1e6, 1)
    rec["mfma_pass_us"], rec["mfma_frac8"] = round(tot * 1e6, 1), round(byts / 8e12 / tot, 3)
    # numeric cross-check of the two passes on the folded gradients
    worst = 0.0
    for (a, b) in zip(rows_v, rows_m):
        worst = max(worst, float((a[1] - b[1]).abs().max() / (a[1].abs().max() + 1e-30)))
    rec["max_rel_diff_valu_vs_matrix_core_last_run"] = worst   # the slabs hold the register form's result (it ran last)
    print(json.dumps(rec), flush=True)


def bench_fmtrace(args):
    """Where a block of the factor pass spends its life: the pass from scripts/fm_trace/liblora_amd_trace.so (the same source
    under -DFM_TRACE), every wave's 100 MHz wall-clock stamps averaged per stage, per site type.  LORA_AMD_LIB must point at
    that library BEFORE lora_amd._C loads (kbench does it below when --what fmtrace)."""
    import ctypes as C

    import numpy as np
    lib = _C.require()
    lib.lora_amd_fm_trace_set.argtypes = [C.c_void_p, C.c_int64]
    r, dt = args.rank, torch.bfloat16
    sites = sd15_site_list()
    groups = {"c1_square": [s_ for s_ in sites if s_[0] == 16384 and s_[1] == s_[2]],
              "c1_geglu": [s_ for s_ in sites if s_[0] == 16384 and s_[2] == 8 * s_[1]],
              "c2_square_640": [s_ for s_ in sites if s_[0] == 4096 and s_[1] == s_[2]],
              "c2_geglu_640": [s_ for s_ in sites if s_[0] == 4096 and s_[2] == 8 * s_[1]],
              "c2_1280": [s_ for s_ in sites if s_[0] <= 1024 and s_[1] == 1280]}
    names = ["prefix_to_LDS", "search", "site_record+row_offsets", "issue_A", "issue_B0", "A_landed+phase1", "T_handoff", "B_stream",
             "Gt_handoff", "phase2_A+stores"]
    NS = len(names) + 1
    for name, shp in groups.items():
        ss, packs = [], []
        cls = rpb = None
        for (M, K, N) in shp:
            pl = _C.factors_mfma_plan(M, K, N, r, dt)
            g, x = torch.randn(M, N, device=DEV).to(dt), torch.randn(M, K, device=DEV).to(dt)
            down, up = torch.randn(r, K, device=DEV) * 0.25, torch.randn(N, r, device=DEV) * 0.05
            pk_down, pk_up = torch.empty(int(pl.pack_down_elems), dtype=dt, device=DEV), torch.empty(int(pl.pack_up_elems), dtype=dt, device=DEV)
            packs.append((down, up, pk_down, pk_up))
            ss.append((g, x, pk_down, pk_up, torch.empty(int(pl.up_part_floats), device=DEV), torch.empty(int(pl.down_part_floats), device=DEV),
                       1.0, None, None, r, pl))
            cls = int(pl.lds_class)
            rpb = int(pl.rows_per_block) if rpb in (None, int(pl.rows_per_block)) else 0
        arr, total = _C.factor_pack_table(packs)
        _C.factor_pack(_C.table_to_device(arr, DEV), len(packs), total, dt)
        arr, grid = _C.factors_mfma_table(ss, dt, cls)
        tab = _C.table_to_device(arr, DEV)
        buf = torch.zeros(grid * 4 * 16, dtype=torch.int64, device=DEV)
        for _ in range(3):
            _C.linear_bwd_factors_mfma_ragged(tab, len(ss), grid, cls, dt, False, rpb)
        torch.cuda.synchronize()
        lib.lora_amd_fm_trace_set(buf.data_ptr(), buf.numel())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _C.linear_bwd_factors_mfma_ragged(tab, len(ss), grid, cls, dt, False, rpb)
        e1.record()
        torch.cuda.synchronize()
        lib.lora_amd_fm_trace_set(None, 0)
        t = buf.view(grid, 4, 16).cpu().numpy().astype(np.float64)[:, :, :NS] * 0.01    # us (100 MHz)
        d = np.diff(t, axis=2)                                                # [block][wave][stages]
        life = t[:, :, NS - 1] - t[:, :, 0]
        wg_life = t[:, :, NS - 1].max(axis=1) - t[:, :, 0].min(axis=1)
        span = t[:, :, NS - 1].max() - t[:, :, 0].min()
        byts = sum((s_[0].numel() + s_[1].numel()) * 2 for s_ in ss)
        rec = {"group": name, "sites": len(ss), "blocks": grid, "class": cls, "rows": rpb, "GB": round(byts / 1e9, 4),
               "launch_us_events": round(e0.elapsed_time(e1) * 1e3, 1), "first_to_last_stamp_us": round(float(span), 1),
               "wave_life_us_mean": round(float(life.mean()), 2), "workgroup_life_us_mean": round(float(wg_life.mean()), 2),
               "workgroup_life_us_p10_p50_p90": [round(float(v), 2) for v in np.percentile(wg_life, [10, 50, 90])],
               "concurrent_workgroups_mean": round(float(wg_life.sum() / span), 1),
               "stage_us_mean": {n_: round(float(d[:, :, i].mean()), 2) for i, n_ in enumerate(names)},
               "stage_us_p90": {n_: round(float(np.percentile(d[:, :, i], 90)), 2) for i, n_ in enumerate(names)}}
        print(json.dumps(rec), flush=True)


def bench_gemm_layouts(args):
    """The merged-weight path's library GEMMs per site shape, in both storage layouts of the scratch weight: forward
    X W^T as F.linear on W [N, K] (TN) or X @ Wt on Wt [K, N] (NN); input gradient G W as G @ W (NN) or F.linear(G, Wt) (TN).
    Head-padded variants of the 320-wide sites included (N or K = 512)."""
    dt = torch.bfloat16
    shapes = SELF_SHAPES + [(16384, 320, 512), (16384, 512, 320)]
    for (M, K, N) in shapes:
        x = torch.randn(M, K, device=DEV).to(dt)
        g = torch.randn(M, N, device=DEV).to(dt)
        w = (torch.randn(N, K, device=DEV) * 0.03).to(dt)
        wt = w.t().contiguous()
        rec = {"M": M, "K": K, "N": N}
        for name, fn in (("fwd_TN_linear(x,W)", lambda: torch.nn.functional.linear(x, w)),
                         ("fwd_NN_x@Wt", lambda: x @ wt),
                         ("dx_NN_g@W", lambda: g @ w),
                         ("dx_TN_linear(g,Wt)", lambda: torch.nn.functional.linear(g, wt))):
            t, _ = timeit(fn)
            rec[name] = round(t * 1e6, 2)
        print(json.dumps(rec), flush=True)


def bench_conv(args):
    """K4 conv-adapter kernels at SD1.5 ResNet sites (bf16 activations, f32 factors)."""
    from lora_amd import ops

    for (B, Ci, Co, Hh, ks, r) in ((4, 320, 320, 64, 3, 4), (4, 640, 640, 32, 3, 4), (4, 1280, 1280, 16, 3, 4),
                                   (4, 2560, 1280, 8, 3, 4), (4, 640, 320, 64, 1, 4), (1, 320, 320, 96, 3, 16),
                                   (1, 1280, 640, 48, 3, 16), (1, 1920, 1280, 24, 3, 16)):
        plan = _C.conv_plan(B, Ci, Co, Hh, Hh, ks, r)
        x = torch.randn(B, Ci, Hh, Hh, device=DEV).to(torch.bfloat16)
        y = torch.randn(B, Co, Hh, Hh, device=DEV).to(torch.bfloat16)
        g = torch.randn(B, Co, Hh, Hh, device=DEV).to(torch.bfloat16)
        dx = torch.randn(B, Ci, Hh, Hh, device=DEV).to(torch.bfloat16)
        down = torch.randn(r, Ci, ks, ks, device=DEV) * 0.1
        up = torch.randn(Co, r, 1, 1, device=DEV) * 0.05
        bufs = ops.conv_buffers(plan, B, r, Hh * Hh, DEV)
        t_part, gt_part, gt, up_part, down_part = bufs
        t = torch.empty(B, r, Hh, Hh, device=DEV)
        ex, ey = B * Ci * Hh * Hh * 2, B * Co * Hh * Hh * 2
        res = dict(B=B, Ci=Ci, Co=Co, HW=Hh, ks=ks, r=r, split_in=plan.split_in, split_out=plan.split_out,
                   groups_in=plan.ngroups_in)
        med, _ = timeit(lambda: _C.conv_down_fwd(x, down, None, t_part, t, ks), args.iters)
        res["down_us"], res["down_GBs"] = med * 1e6, ex / med / 1e9
        med, _ = timeit(lambda: _C.conv_up_fwd_(y, t, up, 1e-3, 0.0, 0, 0), args.iters)
        res["up_us"], res["up_GBs"] = med * 1e6, 2 * ey / med / 1e9
        med, _ = timeit(lambda: _C.conv_bwd_g(g, t, up, None, gt_part, gt, up_part, 1.0, 0.0, 0, 0), args.iters)
        res["bwd_g_us"], res["bwd_g_GBs"] = med * 1e6, ey / med / 1e9
        med, _ = timeit(lambda: _C.conv_bwd_x(x, dx, gt, down, down_part, ks), args.iters)
        res["bwd_x_us"], res["bwd_x_GBs"] = med * 1e6, 3 * ex / med / 1e9
        W = (torch.randn(Co, Ci, ks, ks, device=DEV) * 0.02).to(torch.bfloat16)
        downb, upb = down.to(torch.bfloat16), up.to(torch.bfloat16)
        pad = (ks - 1) // 2
        F = torch.nn.functional
        med, _ = timeit(lambda: y + F.conv2d(F.conv2d(x, downb, None, 1, pad), upb) * 0.5, args.iters)
        res["aten_branch_fwd_us"] = med * 1e6
        med, _ = timeit(lambda: F.conv2d(x, W, None, 1, pad), args.iters)
        res["frozen_conv_us"] = med * 1e6
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)


def bench_nhwc(args):
    """K4 in both layouts at the same 3x3 sites: csrc/conv_nhwc.hip (MFMA, channels-last) next to csrc/conv.hip (NCHW).
    GB/s = algorithmic bytes (X once; dX read + write; X once) over the kernel time."""
    from lora_amd import ops

    for (B, C, Hh, r) in ((4, 320, 64, 4), (4, 320, 64, 16), (4, 640, 32, 16), (4, 1280, 16, 16), (4, 1280, 8, 16),
                          (1, 320, 96, 16), (1, 640, 48, 16), (1, 1280, 24, 16), (1, 2560, 24, 16), (1, 1280, 12, 16)):
        x = torch.randn(B, C, Hh, Hh, device=DEV).to(torch.bfloat16)
        xl = x.contiguous(memory_format=torch.channels_last)
        dxl = torch.randn(B, C, Hh, Hh, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        down = torch.randn(r, C, 3, 3, device=DEV) * 0.1
        M = B * Hh * Hh
        gt = torch.randn(M, r, device=DEV)
        plan = _C.conv3_nhwc_plan(B, C, Hh, Hh, r)
        pf, pd = _C.conv3_nhwc_pack(down, torch.bfloat16, plan)
        part = torch.empty(int(plan.down_part_floats), device=DEV)
        ex = B * C * Hh * Hh * 2
        res = dict(B=B, C=C, HW=Hh, r=r, pt=plan.pt, nsplit=plan.nsplit)
        med, _ = timeit(lambda: _C.conv3_nhwc_pack(down, torch.bfloat16, plan), args.iters)
        res["pack_us"] = med * 1e6
        med, _ = timeit(lambda: _C.conv3_nhwc_down_fwd(xl, pf, r), args.iters)
        res["T_us"], res["T_GBs"] = med * 1e6, ex / med / 1e9
        med, _ = timeit(lambda: _C.conv3_nhwc_bwd_dx_(dxl, gt, pd), args.iters)
        res["dx_us"], res["dx_GBs"] = med * 1e6, 2 * ex / med / 1e9
        med, _ = timeit(lambda: _C.conv3_nhwc_bwd_down(xl, gt, part), args.iters)
        res["ddown_us"], res["ddown_GBs"] = med * 1e6, ex / med / 1e9
        cplan = _C.conv_plan(B, C, C, Hh, Hh, 3, r)
        if cplan.native:
            t_part, gt_part, gtn, up_part, down_part = ops.conv_buffers(cplan, B, r, Hh * Hh, DEV)
            t = torch.empty(B, r, Hh, Hh, device=DEV)
            dx = torch.randn(B, C, Hh, Hh, device=DEV).to(torch.bfloat16)
            med, _ = timeit(lambda: _C.conv_down_fwd(x, down, None, t_part, t, 3), args.iters)
            res["nchw_T_us"] = med * 1e6
            med, _ = timeit(lambda: _C.conv_bwd_x(x, dx, gtn, down, down_part, 3), args.iters)
            res["nchw_dx_ddown_us"] = med * 1e6
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)


def bench_hostops(args):
    """GroupNorm(+SiLU), LayerNorm and the GEGLU gate of the frozen UNet: HIP passes vs the ATen sequences (bf16)."""
    import torch.nn as nn

    from lora_amd.standin import fused

    F = torch.nn.functional

    def both(tag, shape, run_fused, run_aten, nbytes_fwd, nbytes_bwd):
        res = {"op": tag, "shape": list(shape)}
        for name, run in (("hip", run_fused), ("aten", run_aten)):
            fwd, bwd = run()
            res[f"{name}_fwd_us"] = timeit(fwd, args.iters)[0] * 1e6
            res[f"{name}_fwd_bwd_us"] = timeit(bwd, args.iters)[0] * 1e6
        res["hip_fwd_GBs"] = nbytes_fwd / res["hip_fwd_us"] / 1e3
        res["hip_bwd_GBs"] = nbytes_bwd / max(res["hip_fwd_bwd_us"] - res["hip_fwd_us"], 1e-3) / 1e3
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)

    for B, C_, H in ((4, 320, 64), (4, 640, 32), (4, 1280, 16), (4, 1280, 8), (4, 960, 64), (4, 2560, 16)):
        x = torch.randn(B, C_, H, H, device=DEV, dtype=torch.bfloat16, requires_grad=True)
        go = torch.randn_like(x)
        norm = nn.GroupNorm(32, C_).to(DEV).to(torch.bfloat16).requires_grad_(False)
        e = x.numel() * 2
        both("groupnorm_silu", x.shape,
             lambda: (lambda: fused.group_norm_act(x, norm, True), lambda: fused.group_norm_act(x, norm, True).backward(go)),
             lambda: (lambda: F.silu(norm(x)), lambda: F.silu(norm(x)).backward(go)), 3 * e, 5 * e)
    for M, K in ((16384, 320), (4096, 640), (1024, 1280), (256, 1280)):
        x = torch.randn(4, M // 4, K, device=DEV, dtype=torch.bfloat16, requires_grad=True)
        go = torch.randn_like(x)
        ln = nn.LayerNorm(K).to(DEV).to(torch.bfloat16).requires_grad_(False)
        e = x.numel() * 2
        both("layernorm", x.shape, lambda: (lambda: fused.layer_norm(x, ln), lambda: fused.layer_norm(x, ln).backward(go)),
             lambda: (lambda: ln(x), lambda: ln(x).backward(go)), 2 * e, 3 * e)
    for M, inner in ((16384, 1280), (4096, 2560), (1024, 5120), (256, 5120)):
        y = torch.randn(4, M // 4, 2 * inner, device=DEV, dtype=torch.bfloat16, requires_grad=True)
        go = torch.randn(4, M // 4, inner, device=DEV, dtype=torch.bfloat16)
        e = M * inner * 2

        def aten():
            h, g = y.chunk(2, dim=-1)
            return h * F.gelu(g)

        both("geglu", y.shape, lambda: (lambda: fused.geglu(y), lambda: fused.geglu(y).backward(go)),
             lambda: (aten, lambda: aten().backward(go)), 3 * e, 5 * e)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="merge,linear,ws,conv,hostops")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rank", type=int, default=4)
    a = ap.parse_args()
    print(torch.cuda.get_device_name(0), flush=True)
    if "merge" in a.what:
        bench_merge(a)
    if "r16" in a.what.split(","):
        bench_r16(a)
    if "mstep" in a.what.split(","):
        bench_mstep(a)
    if "linear" in a.what:
        bench_linear(a)
    if "xs" in a.what.split(","):
        bench_xs(a)
    if "ws" in a.what.split(","):
        bench_ws(a)
    if "conv" in a.what:
        bench_conv(a)
    if "nhwc" in a.what:
        bench_nhwc(a)
    if "hostops" in a.what:
        bench_hostops(a)
    if "self" in a.what:
        bench_self(a)
    if "fm" in a.what.split(","):
        bench_fm(a)
    if "fmtrace" in a.what.split(","):
        bench_fmtrace(a)
    if "gemmlayout" in a.what:
        bench_gemm_layouts(a)
