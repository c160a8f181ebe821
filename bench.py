#!/usr/bin/env python
"""bench.py — train steps/sec of the DreamBooth LoRA step (BASELINE.json metric) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one optimiser update of train_lora_dreambooth.py's loop (ref :816-888) on the workload
BASELINE.json's configs[1] names: SD1.5-shaped UNet (859.5 M params, random init — no network for
checkpoints), reference-default LoRA injection (144 Linear sites in CrossAttention/GEGLU), rank 4,
bf16 compute with f32 LoRA masters, batch 4 synthetic 64x64x4 latents (= 512x512 images) per GPU,
DDPM noise, MSE, backward, flat-buffer all-reduce (RCCL), clip(1.0), AdamW(lr 1e-4, wd 1e-2).
Rank 0 prints ONE JSON line (see DESIGN.md §Measurement for `roofline` and `cpu_baseline`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import lora_amd as L  # noqa: E402
from lora_amd import _C, trainer as T  # noqa: E402
from lora_amd.standin import DDPMScheduler, sd15_unet  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_unet(device, dtype, seed=0):
    """Random-init SD1.5-shaped UNet, built on the meta device and filled in place (fast, deterministic)."""
    with torch.device("meta"):
        unet = sd15_unet()
    unet.to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if p.dim() > 1:
                p.normal_(0.0, 0.02, generator=g)
            elif name.endswith("weight"):  # norm scales
                p.fill_(1.0)
            else:
                p.zero_()
    unet.to(dtype)
    unet.requires_grad_(False)
    return unet


def merge_roofline(unet, iters=30):
    """K3 fused merge over all adapter sites of this UNet, timed with HIP events on the launch stream."""
    sites = []
    for m in unet.modules():
        if isinstance(m, L.LoraInjectedLinear):
            w = m.linear.weight.data
            sites.append((w, torch.empty_like(w), m.lora_up.weight.data.contiguous(), m.lora_down.weight.data.contiguous()))
    plan = _C.MergePlan(sites)
    for _ in range(3):
        plan.launch(1.0)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    inner = 4  # back-to-back launches per event pair: the host's launch cost is not what is measured
    for a, b in evs:
        a.record()
        for _ in range(inner):
            plan.launch(1.0)
        b.record()
    torch.cuda.synchronize()
    avg_s = sum(a.elapsed_time(b) for a, b in evs) / (iters * inner) * 1e-3
    ach = plan.bytes_algorithmic / avg_s / 1e9
    traffic, traffic_src = None, None
    pmc = _newest_profile("merge_pmc.json")  # offline rocprofv3 --pmc passes (scripts/r0N_profiles.sh), newest round first
    if os.path.exists(pmc):
        try:
            m = json.load(open(pmc))["merge_per_launch"]
            if m["algorithmic_bytes"] == plan.bytes_algorithmic:
                traffic, traffic_src = m["hbm_total_bytes"], f"profiles/{os.path.basename(pmc)} (FETCH_SIZE/WRITE_SIZE, calibrated)"
        except (KeyError, ValueError):
            pass
    return {"kernel": "lora_amd::merge_co_kernel<bf16,f32,4> (K3 fused W+alpha*up@down, all %d sites, 1 launch)" % plan.n_sites,
            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": plan.bytes_algorithmic, "avg_launch_us": round(avg_s * 1e6, 2),
            "launches_timed": iters * inner}


def _newest_profile(suffix: str) -> str:
    """profiles/rNN_<suffix> of the highest round that has one ('' if none)."""
    import glob

    found = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_" + suffix)))
    return found[-1] if found else ""


HBM_PEAK, MFMA_BF16_PEAK = 8.0e12, 2.5e15  # MI355X_MICROARCH.md: HBM3E spec, dense bf16 MFMA


def _time_launch(fn, iters=20, inner=4) -> float:
    """Average seconds of one call of ``fn`` (HIP events on the launch stream, ``inner`` back-to-back calls per pair)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        for _ in range(inner):
            fn()
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / (iters * inner) * 1e-3


def in_step_rooflines(state, fwd_bwd, latents, ehs) -> dict:
    """The two hand-written launches INSIDE the timed step, timed live on this run's own tables (HIP events on the launch
    stream): the in-step merge (``MergedWeights.refresh``: W_eff and W_eff^T of every site from one read of W) and the
    factor-gradient pass (``MergedWeights.flush_factors``: pack + matrix-core pass per LDS class, re-issued on the G / X of
    one eager step), each against its algorithmic bytes."""
    mw = getattr(state, "merged", None)
    if mw is None or not mw.entries:
        return {}
    out = {}
    sec = _time_launch(mw.refresh)
    byts = mw.bytes_algorithmic
    n_t = sum(1 for e in mw.entries.values() if e["w_eff_t"] is not None)
    out["merge"] = {"kernel": "lora_amd::merge_step_kernel<bf16,4> (W_eff + W_eff^T of %d sites (%d with a transpose), 1 launch, "
                              "rounding %s)" % (len(mw.entries), n_t, "dither" if ops_mod().MERGE_ROUNDING == _C.ROUND_DITHER else "once"),
                    "bound": "hbm", "algorithmic_bytes_per_launch": int(byts), "avg_launch_us": round(sec * 1e6, 2),
                    "achieved": round(byts / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(byts / sec / HBM_PEAK, 4)}
    # factor pass: one eager forward+backward leaves the sites owed; replay the flush on the same tensors
    state.zero_grad()
    fwd_bwd(latents, ehs)
    owed = list(mw._owed)
    if owed:
        gx_rows = sum((st[0].numel() + st[1].numel()) * st[0].element_size() for st in owed)   # rows as laid out (head padding)
        dense = lambda t, hd: t.shape[0] * (hd[0] * hd[1] if hd else t.shape[1])  # noqa: E731
        gx = sum((dense(st[0], st[7]) + dense(st[1], st[8])) * st[0].element_size() for st in owed)   # M (N + K) e

        # record the launches of ONE flush (tables built and uploaded once), then time re-issuing exactly those launches:
        # the host's table building must not sit between the timed launches
        names = ("factor_pack", "linear_bwd_factors_mfma_ragged", "linear_bwd_factors_self_ragged", "reduce_batched")
        orig, calls = {nm: getattr(_C, nm) for nm in names}, []
        try:
            for nm in names:
                setattr(_C, nm, (lambda *a, _n=nm: (calls.append((_n, a)), orig[_n](*a))[1]))
            mw._owed = list(owed)
            state.reduce_pending()   # flush (pack + pass per class) AND the fold of every site's partial slabs into flat_g
        finally:
            for nm in names:
                setattr(_C, nm, orig[nm])
        torch.cuda.synchronize()
        sec = _time_launch(lambda: [orig[nm](*a) for nm, a in calls], iters=10, inner=2)
        nofold = [(nm, a) for nm, a in calls if nm != "reduce_batched"]
        sec_nofold = _time_launch(lambda: [orig[nm](*a) for nm, a in nofold], iters=10, inner=2)
        kinds = sorted({st[9] for st in owed})
        launched = sorted({nm for nm, _ in calls})
        kern = "lora_amd::factors_reg_kernel<bf16> (register-resident matrix-core pass) + factor_pack + reduce_batched" \
            if "linear_bwd_factors_mfma_ragged" in launched else "lora_amd::linear_bwd_factors_self_ragged_kernel<bf16> (VALU pass) + fold"
        out["factor_pass"] = {"kernel": "%s: G and X of %d sites; passes: %s" % (kern, len(owed), "+".join(kinds)),
                              "bound": "hbm", "algorithmic_bytes_per_launch": int(gx), "avg_launch_us": round(sec * 1e6, 2),
                              "achieved": round(gx / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(gx / sec / HBM_PEAK, 4),
                              "launches": [nm for nm, _ in calls],
                              "without_fold_us": round(sec_nofold * 1e6, 2),
                              "frac_without_fold": round(gx / sec_nofold / HBM_PEAK, 4),
                              "bytes_of_rows_as_laid_out": int(gx_rows),
                              "includes": "the pack launch, one pass launch per LDS class AND the fold (reduce_batched) of the "
                                          "partial slabs; bytes = SURVEY 8d's M (N + K) e of the dense sites (the head-padded "
                                          "rows the kernel walks are bytes_of_rows_as_laid_out)"}
    state.reduce_pending()
    state.zero_grad()
    return out


def ops_mod():
    from lora_amd import ops
    return ops


def roofline_entry(flops: float, byts: float, sec: float) -> dict:
    """Which roof binds a launch of `flops` / algorithmic `byts`, and the fraction of it reached in `sec` seconds."""
    t_hbm, t_mfma = byts / HBM_PEAK, flops / MFMA_BF16_PEAK
    e = {"avg_launch_us": round(sec * 1e6, 2), "arithmetic_intensity_flop_per_byte": round(flops / byts, 1),
         "hbm_frac": round(t_hbm / sec, 4), "mfma_frac": round(t_mfma / sec, 4)}
    if t_hbm >= t_mfma:  # below the machine balance (2.5 PF / 8 TB/s = 312 flop/B): the byte roof is the one that binds
        e.update({"bound": "hbm", "achieved": round(byts / sec / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                  "frac": e["hbm_frac"]})
    else:
        e.update({"bound": "mfma", "achieved": round(flops / sec / 1e12, 1), "peak": MFMA_BF16_PEAK / 1e12,
                  "unit": "TFLOP/s", "frac": e["mfma_frac"]})
    return e


def gemm_roofline(iters=20):
    """K1 fully fused MFMA kernel (frozen GEMM + LoRA branch in one launch) at the two largest site shapes of the
    workload, with the kernel the step itself uses there (`_C.gemm_choice`: weight-stationary csrc/gemm_ws.hip or the
    LDS-ring csrc/gemm_fused.hip), timed like the merge: informational second roofline.  With K = 320 both sites sit
    below the machine balance, so the byte roof binds; what the kernels run into first is the per-CU L2 -> CU rate of
    pulling the weight panel (every workgroup needs the whole [N, 320] panel for its 64 rows)."""
    out = []
    for (M, K, N) in ((16384, 320, 320), (16384, 320, 2560)):
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
        b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
        down = torch.randn(4, K, device="cuda", generator=g) * 0.25
        up = torch.randn(N, 4, device="cuda", generator=g) * 0.05
        tile = _C.gemm_choice(x, w, b, down, up, 1.0)
        if tile == _C.WS_TILE:
            y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            site = dict(wp=_C.ws_pack(w), N=N, bias=b, down=down, up=up, scale=1.0, y=y)
            run, name = (lambda: _C.linear_ws(x, [site])), "lora_amd::linear_ws_kernel<bf16> (K1 fused, weight-stationary)"
        elif tile and _C.gemm_supported(x, w, N, 4):
            run = lambda: _C.linear_gemm_fwd(x, w, b, down, up, 1.0, tile)  # noqa: E731
            name = "lora_amd::linear_gemm_fwd_kernel<bf16> (K1 fused, LDS ring, tile %d)" % tile
        else:
            continue
        flops, byts = 2.0 * M * K * N + 2.0 * M * 4 * (K + N), (M * K + N * K + M * N) * 2 + (N + K) * 4 * 4 + M * 4 * 4

        def timed(fn, inner=10):
            """Seconds per call: `inner` back-to-back calls captured into a hipGraph, the replay bracketed by HIP events on
            the launch stream (the host's ctypes / launch cost per call exceeds these kernels' 10-13 us)."""
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(inner):
                    fn()
            graph.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(iters):
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                graph.replay()
                e.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(e) / inner * 1e-3)
            ts.sort()
            return ts[len(ts) // 2]

        out.append({"kernel": name, "site": [M, K, N, 4], **roofline_entry(flops, byts, timed(run))})
        # the library GEMM alone on the same site (round 5's input-stationary experiment left the product library in round
        # 6: scripts/gemm_xs/, `scripts/kbench.py --what xs`)
        if True:
            pb = (M * K + N * K + M * N) * 2
            out.append({"kernel": "library GEMM (hipBLASLt via torch, X W^T + b)", "site": [M, K, N, 0],
                        **roofline_entry(2.0 * M * K * N, pb, timed(lambda: torch.nn.functional.linear(x, w, b)))})
    return out


def usable_cores() -> int:
    """Cores this process may actually use: scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_worker(spec: str) -> None:
    """Child process of cpu_baseline(): one reference-algorithm training step on the host, timed."""
    from oracle import torch_ref as TR

    batch, res, threads, rank_r, n_timed = (int(v) for v in spec.split(","))
    torch.set_num_threads(threads)
    unet = build_unet(torch.device("cpu"), torch.float32, seed=0)
    params = TR.inject(unet, L.UNET_DEFAULT_TARGET_REPLACE, r=rank_r)
    opt = torch.optim.AdamW(params, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    sched = DDPMScheduler()
    unet.train()

    def one(b, hw):
        g = torch.Generator().manual_seed(0)
        lat = torch.randn(b, 4, hw, hw, generator=g) * 0.18215
        ehs = torch.randn(b, 77, 768, generator=g)
        noise = torch.randn(b, 4, hw, hw, generator=g)
        t = torch.randint(0, 1000, (b,), generator=g)
        t0 = time.perf_counter()
        TR.dreambooth_step(lambda x, tt, c: unet(x, tt, c).sample, params, opt, lat, noise, t, ehs, sched.alphas_cumprod)
        return time.perf_counter() - t0

    warm_s = one(batch, res)  # warm-up at the SAME shapes: allocator, oneDNN primitive creation, thread pool
    times = [one(batch, res) for _ in range(n_timed)]
    step_s = sum(times) / len(times)
    # the graded kernel's CPU counterpart: the reference's collapse op sequence over all adapter sites, fp32
    sites = [(m.frozen.weight.detach(), m.up.detach(), m.down.detach()) for m in TR.sites_of(unet)
             if isinstance(m, TR.RefLinearSite)]
    merge_s = float("inf")
    with torch.no_grad():
        for _ in range(2):  # second pass: pages of the 0.77 GB of fp32 weights are resident
            t0 = time.perf_counter()
            for W, up, down in sites:
                TR.collapse(W, up, down, 0.5)
            merge_s = min(merge_s, time.perf_counter() - t0)
    elems = sum(W.numel() for W, _, _ in sites)
    print(json.dumps({"seconds": step_s, "step_seconds": times, "warmup_seconds": warm_s, "threads": threads,
                      "merge_seconds": merge_s, "merge_sites": len(sites), "merge_elems": elems}), flush=True)


def cpu_baseline(rank_r=4):
    """The reference's algorithm (oracle/torch_ref.py restatement: the reference tree does not travel to this
    box) on the host cores: fp32 CPU PyTorch, same UNet, same step.  Bounded: each attempt runs in a child
    process under a timeout; the first sample that finishes is reported, scaled to the batch-4 512^2 step."""
    import subprocess

    cores = usable_cores()
    threads = min(cores, 64)  # CPU GEMM/conv scaling flattens (and oversubscription collapses) beyond this
    # (batch, latent hw, timed steps, timeout s, factor to the batch-4 512^2 step, description)
    attempts = [(4, 64, 3, 240, 1.0, "3 timed steps after 1 same-shape warm-up step, batch 4 at 512x512 (64x64 latents): "
                                     "the workload itself"),
                (1, 64, 3, 120, 4.0, "3 timed steps after 1 same-shape warm-up step, batch 1 at 512x512; x4 for the "
                                     "batch-4 workload (the batch-4 sample did not finish in its 240 s window)")]
    errs = []
    for batch, res, n_timed, timeout_s, factor, what in attempts:
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                                  f"{batch},{res},{threads},{rank_r},{n_timed}"], capture_output=True, text=True,
                                 timeout=timeout_s, env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
            rec = json.loads(line)
            sec = rec["seconds"]
            res_ = {"value": round(1.0 / (sec * factor), 6), "unit": "steps/s", "cores": threads, "kind": "port",
                    "host_cores_usable": cores, "torch": torch.__version__,
                    "step_seconds": [round(t, 3) for t in rec["step_seconds"]],
                    "warmup_step_seconds": round(rec["warmup_seconds"], 3),
                    "sample": f"{what}; mean {sec:.2f} s/step on {threads} threads, fp32, oracle/torch_ref.py "
                              "(the reference's op sequence; the reference tree itself does not travel to this box)"}
            if rec.get("merge_seconds"):  # the merge (roofline kernel) on the same cores: reference op sequence, fp32
                ms = rec["merge_seconds"]
                res_["merge"] = {"ms": round(ms * 1e3, 2), "sites": rec["merge_sites"],
                                 "algorithmic_GBs": round(2 * rec["merge_elems"] * 4 / ms / 1e9, 2), "dtype": "f32",
                                 "what": "reference collapse_lora op sequence (mm, cast, mul, add) over every Linear site"}
            return res_
        except subprocess.TimeoutExpired:
            errs.append(f"batch {batch} {res}x{res} latents: warm-up + {n_timed} steps > {timeout_s} s")
        except Exception as e:  # noqa: BLE001
            errs.append(f"{type(e).__name__}: {e}")
    return {"value": None, "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "no bounded sample finished: " + "; ".join(errs)}


# ----------------------------------------------------------------------------- adapter-path evidence
def _site_bytes(phase: str, path: str, M: int, K: int, N: int, r: int, e: int = 2) -> int:
    """Algorithmic HBM bytes of one adapter site.  A site whose frozen product runs INSIDE our launch is
    priced with the fused formula (W, X, Y / G, X, dX, W once each); a site whose frozen product stays a library GEMM is
    priced with the branch-only bound (read X + read-modify-write Y; backward: read G, read X, read-modify-write dX) —
    the library GEMM's own time and bytes are then not the adapter path's."""
    ab = (N + K) * r * 4
    fused = not (path.startswith("lib") or path.startswith("g+lib") or path == "primitives")
    if phase == "fwd":
        return ((N * K + M * K + M * N) * e + ab) if fused else (M * K * e + 2 * M * N * e + ab)
    return ((2 * M * K + M * N + N * K) * e + 2 * ab) if fused else (M * N * e + 3 * M * K * e + 2 * ab)


PATH_LOG_FILE = None  # --path-log


def adapter_path_profile(step_fn, state) -> dict:
    """One EAGER step under torch.profiler (kineto's device-side kernel records, i.e. kernel durations without launch
    gaps): which kernel ran which adapter site (ops.PATH_LOG), device time of every lora_amd:: kernel, and the adapter
    path's algorithmic bytes -> fraction of the HBM roof.  Adapter kernels = the Linear / Conv2d adapter launches, the
    partial reduce and the flat-buffer optimiser; the frozen-UNet passes of csrc/hostops.hip are listed separately."""
    from collections import Counter

    from torch.profiler import ProfilerActivity, profile

    from lora_amd import ops

    step_fn()  # warm (lazy workspaces, tuning)
    state.zero_grad()
    torch.cuda.synchronize()
    ops.PATH_LOG = []
    try:
        # CPU activity too: the frozen GEMMs of the merged-weight sites are library kernels; their device time is
        # attributed through the record_function range ops.LoraLinearMergedFunction opens around them
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            step_fn()
            torch.cuda.synchronize()
    finally:
        log_, ops.PATH_LOG = ops.PATH_LOG, None
    state.zero_grad()
    host_pref = ("gn_", "ln_", "geglu_", "add_ln", "groupnorm", "layernorm")
    kern, total_us, ours_us, host_us = {}, 0.0, 0.0, 0.0
    for ev in prof.events():
        if getattr(ev, "device_type", None) is None or "cuda" not in str(ev.device_type).lower():
            continue
        dur = float(getattr(ev, "device_time", 0.0) or getattr(ev, "cuda_time", 0.0) or 0.0)
        total_us += dur
        name = ev.name
        if "lora_amd::" not in name or name == "lora_amd::merged_gemm":  # the range itself: added below, once
            continue
        short = name.split("lora_amd::", 1)[1].split("(")[0]
        k = kern.setdefault(short, [0, 0.0])
        k[0] += 1
        k[1] += dur
        if short.startswith(host_pref):
            host_us += dur
        else:
            ours_us += dur
    merged_gemm_us, merged_gemm_calls = 0.0, 0
    for ka in prof.key_averages():
        if ka.key == "lora_amd::merged_gemm":
            merged_gemm_us = float(getattr(ka, "device_time_total", 0.0) or getattr(ka, "cuda_time_total", 0.0) or 0.0)
            merged_gemm_calls = int(ka.count)
    if merged_gemm_calls:
        kern["(library GEMM) X W_eff^T / G W_eff of the merged-weight sites"] = [merged_gemm_calls, merged_gemm_us]
        ours_us += merged_gemm_us
    choices = Counter((ph, path) for ph, path, *_ in log_)
    if PATH_LOG_FILE:
        with open(PATH_LOG_FILE, "w") as f:
            json.dump([list(map(lambda v: v if isinstance(v, (int, float, str)) else str(v), rec)) for rec in log_], f)
    byts = sum(_site_bytes(*rec) for rec in log_)
    mw = getattr(state, "merged", None)
    if mw is not None:
        byts += mw.bytes_algorithmic  # the step's one merge launch: read W, write W_eff, read the factors
    own_sites = sum(1 for ph, path, *_ in log_ if ph == "fwd" and not (path.startswith("lib") or path.startswith("merged")))
    merged_sites = sum(1 for ph, path, *_ in log_ if ph == "fwd" and path.startswith("merged"))
    out = {"how": "one eager step under torch.profiler (device-side kernel durations); bytes per SURVEY 8d: fused "
                  "formula where the frozen product is inside our launch, branch-only bound where it is a library GEMM",
           "gpu_ms_per_step": round(ours_us / 1e3, 3), "algorithmic_bytes_per_step": int(byts),
           "frac": round(byts / (ours_us * 1e-6) / HBM_PEAK, 4) if ours_us else None, "bound": "hbm",
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": round(byts / (ours_us * 1e-6) / 1e9, 1) if ours_us else None,
           "library_gemm_ms": round(merged_gemm_us / 1e3, 3), "library_gemm_calls": merged_gemm_calls,
           "hostops_gpu_ms_per_step": round(host_us / 1e3, 3), "all_kernels_gpu_ms_per_step": round(total_us / 1e3, 3),
           "sites_fwd_frozen_product_in_our_launch": own_sites, "sites_fwd_library_gemm_on_merged_weight": merged_sites,
           "sites_fwd_total": sum(1 for ph, *_ in log_ if ph == "fwd"),
           "kernels": {k: {"calls": v[0], "us": round(v[1], 1)} for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])}}
    return out, {f"{ph}:{path}": c for (ph, path), c in sorted(choices.items())}


class AtenLoraLinear(torch.nn.Module):
    """The reference's op sequence for one adapter (lora.py:53-58: linear, linear, linear, dropout, mul, add) as stock
    ATen launches on the GPU — the `--adapters aten` comparison leg of bench.py only (what the same host model costs
    with the reference's adapters instead of the HIP kernels).  Factors are cast to the compute dtype per call, as
    autocast does."""

    def __init__(self, src):
        super().__init__()
        self.linear, self.lora_down, self.lora_up = src.linear, src.lora_down, src.lora_up
        self.dropout, self.scale = src.dropout, src.scale

    def forward(self, x):
        dt = x.dtype
        low = torch.nn.functional.linear(torch.nn.functional.linear(x, self.lora_down.weight.to(dt)),
                                         self.lora_up.weight.to(dt))
        return self.linear(x) + self.dropout(low) * self.scale


def swap_in_aten_adapters(model) -> int:
    n = 0
    for parent in list(model.modules()):
        for name, child in list(parent.named_children()):
            if isinstance(child, L.LoraInjectedLinear):
                parent._modules[name] = AtenLoraLinear(child)
                n += 1
    return n


SECONDARY = [  # (tag, extra argv, env, timeout s): driver-observed lines for the other BASELINE geometries, short runs
    ("frozen_only", ["--adapters", "none"], {}, 150),
    ("frozen_only configs[2]", ["--adapters", "none", "--text-encoder", "1", "--rank", "8"], {}, 150),
    ("frozen_only configs[3]", ["--adapters", "none", "--extended", "1", "--rank", "16", "--res", "768", "--batch", "1"], {}, 240),
    ("host_options_off", ["--channels-last", "0", "--head-pad", "0", "--conv-find", "0"], {"LORA_AMD_HOSTOPS": "0"}, 150),
    ("aten_adapters", ["--adapters", "aten"], {}, 150),
    ("fused_per_site_kernels (--merged 0)", ["--merged", "0"], {}, 150),
    ("configs[2] UNet+CLIP rank 8", ["--text-encoder", "1", "--rank", "8"], {}, 150),
    ("configs[3] extended rank 16 768^2 batch 1", ["--extended", "1", "--rank", "16", "--res", "768", "--batch", "1"], {}, 240),
    ("prior preservation (8 samples/step)", ["--with-prior-preservation", "1"], {}, 150),
    ("configs[4] cli_svd 224 sites rank 8", ["--svd", "--warmup", "1", "--no-cpu-baseline"], {}, 120),
]


def _attention_choices() -> dict:
    from lora_amd.standin import attention as _att

    return {k: v for k, v in _att.choices().items()}


def run_secondaries(budget_s: float, steps: int) -> list:
    """Each secondary line is its own `python bench.py ...` child (fresh process, GPU shared sequentially), bounded by a
    timeout and by the remaining budget; what does not fit is reported as skipped, never silently dropped."""
    import subprocess

    out, t_begin = [], time.perf_counter()
    for tag, extra, env, tmo in SECONDARY:
        left = budget_s - (time.perf_counter() - t_begin)
        rec = {"tag": tag, "argv": " ".join(extra), "env": env}
        if left < 30:
            rec["skipped"] = "time budget of the default run spent"
            out.append(rec)
            continue
        argv = [sys.executable, os.path.abspath(__file__)] + extra
        if "--svd" not in extra:
            argv += ["--steps", str(steps), "--warmup", "3", "--no-cpu-baseline", "--no-roofline", "--no-secondary"]
        t0 = time.perf_counter()
        dtag = "sec%d" % len(out)
        try:
            r = subprocess.run(argv, capture_output=True, text=True, timeout=min(tmo, left),
                               env={**os.environ, **env, "LORA_AMD_BENCH_DETAIL_TAG": dtag})
            d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            rec.update({"value": d["value"], "unit": d["unit"], "ms_per_step": d.get("ms_per_step"), "steps": d.get("steps"),
                        "workload": d["config"].get("workload"), "wall_s": round(time.perf_counter() - t0, 1),
                        "execution": d["config"].get("execution")})
            try:  # the child's detail record (attention picks, merged-site count): detail only, never the stdout line
                with open(os.path.join(REPO, "gpurun_out", "bench_detail_%s.json" % dtag)) as f:
                    dd = json.load(f)
                rec["attention_kernels"] = dd["config"].get("attention_kernels")
                rec["merged_sites"] = (dd["config"].get("adapter_options", {}).get("merged_weights") or {}).get("sites")
            except (OSError, KeyError, ValueError):
                pass
            if r.stderr and "capture failed" in r.stderr:
                rec["note"] = [ln for ln in r.stderr.splitlines() if "capture failed" in ln][-1][:300]
            if "roofline" in d:
                rec["roofline_frac"] = d["roofline"].get("frac")
        except subprocess.TimeoutExpired:
            rec["skipped"] = f"did not finish in {min(tmo, left):.0f} s"
        except Exception as e:  # noqa: BLE001
            rec["skipped"] = f"{type(e).__name__}: {e}"
        out.append(rec)
    return out


# ----------------------------------------------------------------------------- the ONE stdout line (driver contract)
LINE_LIMIT = 6144  # bytes; the driver keeps a ~12 KB tail of stdout and truncates strings at 128 characters


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _short(s, n=120):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 1] + "~"


_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us")


def _roof(d):
    r = _pick(d, _ROOF_KEYS)
    if "kernel" in r:
        r["kernel"] = _short(r["kernel"].split(" (")[0], 80)
    return r


def compact_record(out: dict, detail_path: str = "") -> dict:
    """The record rank 0 prints as the LAST (and only) stdout line: the contract's fields, ``roofline`` (the graded K3 kernel,
    with the in-step launches nested under ``in_step`` so that they ride in the one object the driver keeps), ``cpu_baseline``,
    ``secondary`` as (tag, value, unit, ms_per_step, steps, execution) only.  Everything else — kernel tables, attention
    picks, long descriptions — is the detail record (``bench_detail.json`` + one stderr line)."""
    cfg = out.get("config", {})
    c = _pick(cfg, ("global_batch", "samples_per_s", "global_steps_per_s", "parallelism", "execution", "allreduce_us", "device", "trainable_params",
                    "allreduce_payload_bytes", "final_loss", "sites", "groups", "weight_elements", "power_iterations"))
    c = {"workload": _short(cfg.get("workload_short") or cfg.get("workload", ""), 126), **c}
    rec = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                               "scaling", "vs_baseline", "dtype", "data") if k in out}
    rec["metric"] = _short(rec.get("metric", ""), 126)
    rec["config"] = c
    ins = {k: _roof(v) for k, v in (out.get("roofline_in_step") or {}).items()}
    if "roofline" in out:
        rec["roofline"] = _roof(out["roofline"])
        if ins:
            rec["roofline"]["in_step"] = ins
    if ins:
        rec["roofline_in_step"] = ins
        for k, v in ins.items():  # scalars the driver's `config` keeps
            c[k + "_frac_in_step"] = v.get("frac")
    if out.get("roofline_fused_gemm"):
        rec["roofline_fused_gemm"] = [dict(_pick(e, ("site", "avg_launch_us", "frac")), kernel=_short(e.get("kernel", ""), 64))
                                      for e in out["roofline_fused_gemm"]]
        c["fused_gemm_frac"] = out["roofline_fused_gemm"][0].get("frac")
    if "step_hbm" in out and "measured_frac_of_peak" in out["step_hbm"]:
        # rocprof counters of the SAME command taken in a separate run (PMC passes cannot ride a timed run): three scalars + the
        # file they come from; the per-kernel tables stay in that file
        rec["step_pmc"] = {"hbm_frac_of_peak": out["step_hbm"]["measured_frac_of_peak"],
                           "hbm_GBs": out["step_hbm"].get("measured_GBs_at_this_step_time"),
                           "mfma_frac_of_peak": (out.get("mfma_util") or {}).get("whole_step_mfma_frac_of_peak"),
                           "static": True, "source": _short(str((out.get("mfma_util") or {}).get("source", "profiles/")), 60)}
    if "adapter_path" in out:
        rec["adapter_path"] = _pick(out["adapter_path"], ("gpu_ms_per_step", "algorithmic_bytes_per_step", "frac",
                                                          "library_gemm_ms", "library_gemm_calls", "frac_of_step_time"))
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        rec["cpu_baseline"] = dict(_pick(cb, ("value", "unit", "cores", "kind")), sample=_short(cb.get("sample", ""), 126))
        if isinstance(cb.get("merge"), dict):
            rec["cpu_baseline"]["merge_ms"] = cb["merge"].get("ms")
    for k in ("lora_overhead_ms", "lora_overhead_ms_cfg2", "lora_overhead_ms_cfg3", "value_frozen_only_cfg2",
              "value_frozen_only_cfg3", "value_host_options_off", "value_frozen_only", "value_aten_adapters",
              "value_fused_per_site_kernels"):
        if k in out:
            rec[k] = out[k]
            c[k] = out[k]
    if "secondary" in out:
        rec["secondary"] = [dict(_pick(r, ("value", "unit", "ms_per_step", "steps", "execution")), tag=_short(r.get("tag", ""), 48),
                                 **({"skipped": _short(r["skipped"], 60)} if "skipped" in r else {}))
                            for r in out["secondary"]]
    if detail_path:
        rec["detail"] = detail_path
    # hard guard: never print a line the driver cannot keep whole
    for drop in ("step_pmc", "adapter_path", "roofline_fused_gemm", "roofline_in_step", "secondary"):
        if len(json.dumps(rec)) <= LINE_LIMIT:
            break
        rec.pop(drop, None)
    return rec


def emit(out: dict) -> None:
    """Detail record -> gpurun_out/bench_detail.json (merged back from the GPU box) + ONE stderr line; compact record -> the
    one stdout line."""
    path = ""
    try:
        d = os.path.join(REPO, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        tag = os.environ.get("LORA_AMD_BENCH_DETAIL_TAG") or ("svd" if "cli_svd" in out.get("metric", "")
                                                              else "n%d" % out.get("n_gpus", 1))
        path = os.path.join("gpurun_out", "bench_detail_%s.json" % tag)
        with open(os.path.join(REPO, path), "w") as f:
            json.dump(out, f, indent=1)
    except OSError as e:
        log(f"[bench] detail file not written: {e}")
        path = ""
    log("[bench-detail] " + json.dumps(out))
    print(json.dumps(compact_record(out, path)), flush=True)


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start one process per GPU ourselves, the way
    the driver's documented command does (torch.distributed.run, 127.0.0.1 rendezvous), and relay rank 0's JSON line."""
    import socket
    import subprocess

    have = torch.cuda.device_count() if args.device == "cuda" else args.gpus
    if have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    log(f"[bench] launching {args.gpus} ranks:", " ".join(cmd))
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    return subprocess.run(cmd, env=env).returncode


def svd_worker(spec: str) -> None:
    """Child process of svd_bench(): the reference recipe (cli_svd.py:30-47: full torch.linalg.svd per site + signed
    quantile clamp) on the host cores for a few sites, timed."""
    from oracle import torch_ref as TR  # noqa: F401  (the recipe itself is three torch calls, restated below)

    threads, rank = (int(v) for v in spec.split(","))
    torch.set_num_threads(threads)
    out = []
    for (N, K) in ((320, 320), (640, 640), (1280, 1280), (1280, 2560)):
        g = torch.Generator().manual_seed(0)
        res = torch.randn(N, K, generator=g)
        t0 = time.perf_counter()
        U, S, Vh = torch.linalg.svd(res)
        U = U[:, :rank] @ torch.diag(S[:rank])
        Vh = Vh[:rank, :]
        hi = torch.quantile(torch.cat([U.flatten(), Vh.flatten()]), 0.99)
        U.clamp(-hi, hi), Vh.clamp(-hi, hi)
        out.append({"shape": [N, K], "seconds": time.perf_counter() - t0})
    print(json.dumps({"sites": out, "threads": threads}), flush=True)


def svd_bench(args) -> dict:
    """BASELINE configs[4]: SVD distillation of a fully fine-tuned SD1.5 UNet to rank-8 LoRA over the 224 sites of the
    extended injection (shapes of lora_amd.standin.sd15_lora_site_shapes(extended=True); W_tuned = W_base + planted
    rank-12 update + noise, synthetic).  One "step" = all 224 sites through lora_amd.cli_svd.distill_group (batched
    randomized subspace iteration on the HIP passes + clamp).  The dominant kernels stream the f32 residuals:
    (2 n_iter + 2) passes over sum(N K) * 4 bytes."""
    from collections import Counter

    from lora_amd import cli_svd as S
    from lora_amd.standin import sd15_lora_site_shapes

    dev = torch.device("cuda", 0)
    rank, n_iter = 8, (int(os.environ["LORA_AMD_SVD_ITERS"]) if os.environ.get("LORA_AMD_SVD_ITERS") else None)  # None = adaptive
    shapes = Counter(sd15_lora_site_shapes(extended=True))
    g = torch.Generator(device=dev).manual_seed(0)
    groups = []
    for (N, K), cnt in sorted(shapes.items()):
        base = torch.randn(cnt, N, K, device=dev, generator=g) * 0.05
        u = torch.randn(cnt, N, 12, device=dev, generator=g)
        v = torch.randn(cnt, 12, K, device=dev, generator=g)
        sv = torch.tensor([3.0 * 0.6 ** i for i in range(12)], device=dev)
        tuned = base + torch.bmm(u / u.norm(dim=1, keepdim=True) * sv, v / v.norm(dim=2, keepdim=True)) * 0.2 \
            + 1e-4 * torch.randn(cnt, N, K, device=dev, generator=g)
        groups.append((list(tuned), list(base)))
    n_sites = sum(shapes.values())
    elems = sum(N * K * c for (N, K), c in shapes.items())

    def step():
        gen = torch.Generator(device=dev).manual_seed(1)
        return S.distill_model(groups, rank, 0.99, gen, n_iter=n_iter)[-1]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        up, down = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    iters = n_iter if n_iter is not None else (S.LAST_ITERATIONS or 4)
    passes = 2 * iters + 2
    # the subspace-steering passes (sketch + power iterations) read the hi plane only (2 bytes / element), the pass that forms
    # the factors both planes (4); + forming the planes: read W_tuned and W_base (f32), write four 16-bit planes
    hi_only = bool(getattr(S, "HI_ONLY_ITERATIONS", False))
    # both planes: b = Q^T dW and the last dW Qz (the adaptive loop decides one iteration ahead and knows its last pass too)
    full = 2 if hi_only else passes
    pass_bytes = ((passes - full) * 2 + full * 4) * elems
    byts = pass_bytes + 2 * elems * 4 + 4 * elems * 2
    out = {"metric": "cli_svd distillation, SD1.5 UNet -> rank-8 LoRA (224 sites, extended injection)",
           "value": round(n_sites / dt, 2), "unit": "sites/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE configs[4]: cli_svd distillation of a fine-tuned SD1.5 UNet to rank-8 LoRA, 224 "
                                  "sites (31 shape groups, 730 M weight elements), randomized subspace iteration, "
                                  "every step ONE ragged launch over all shape groups; the products with dW / dW^T on the "
                                  "matrix cores over (hi, lo) bf16 planes of the residuals (lora_amd_split16_transpose + "
                                  "lora_amd_rowdot16_planes), on-device CholeskyQR3",
                      "workload_short": "BASELINE configs[4]: cli_svd distillation of a fine-tuned SD1.5 UNet to rank-8 LoRA, 224 sites",
                      "sites": n_sites, "groups": len(groups), "weight_elements": elems, "power_iterations": iters,
                      "iteration_count": "adaptive (Ritz energy settled)" if n_iter is None else "fixed"},
           "roofline": {"kernel": "lora_amd::rowdot16_planes_kernel<bf16> (%d passes over the residuals, %s) + split16_residual + "
                                  "everything else of the step" % (passes, "%d of them on the hi plane only" % (passes - full)),
                        "bound": "hbm", "achieved": round(byts / dt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(byts / dt / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                        "algorithmic_bytes_per_step": byts, "passes_over_residuals": passes, "hi_plane_only_passes": passes - full,
                        "note": "WHOLE-STEP figure (passes + forming the planes + ~70 small launches + one host sync per adaptive "
                                "iteration from the 3rd on); per-kernel shares: profiles/r06_svd_kernel_trace_summary.txt.  "
                                "LORA_AMD_SVD_ITERS=4 fixes the iteration count (rounds 2-4's iso-work figure)"}}
    if not args.no_cpu_baseline:
        import subprocess

        threads = min(usable_cores(), 64)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--svd-worker", f"{threads},{rank}"],
                               capture_output=True, text=True, timeout=240, env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
            rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            tot = sum(s_["seconds"] for s_ in rec["sites"])
            out["cpu_baseline"] = {"value": round(len(rec["sites"]) / tot, 3), "unit": "sites/s", "cores": threads,
                                   "kind": "port", "sites": rec["sites"],
                                   "sample": "the reference recipe (full torch.linalg.svd + signed 0.99-quantile clamp, "
                                             "cli_svd.py:30-47) on 4 of the 224 sites (320^2, 640^2, 1280^2, 1280x2560), f32; "
                                             "the largest sites (10240x1280, 1280x23040) cost far more than these"}
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"value": None, "unit": "sites/s", "cores": threads, "kind": "port",
                                   "sample": f"sample did not finish: {type(e).__name__}: {e}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 50 for the training step, 3 for --svd)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch (train_batch_size, ref :285-289)")
    ap.add_argument("--rank", type=int, default=4, dest="lora_rank")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph")
    ap.add_argument("--channels-last", type=int, default=-1, help="NHWC activations/conv weights (the layout MIOpen's "
                    "implicit-GEMM convolutions run in; GroupNorm(+SiLU) has an NHWC HIP pass, and with --extended the "
                    "Conv2d adapters take the channels-last MFMA kernels of csrc/conv_nhwc.hip).  0 = NCHW; -1 = on")
    ap.add_argument("--head-pad", type=int, default=1, help="q/k/v/out projections write / read the padded head layout "
                    "of the chosen attention kernel (LORA_AMD_HEAD_PAD; no pad / slice copies around the attention core)")
    ap.add_argument("--extended", type=int, default=0, help="inject_trainable_lora_extended: + ResnetBlock2D Conv2d "
                    "adapters (BASELINE configs[3] geometry; not the headline workload)")
    ap.add_argument("--text-encoder", type=int, default=0, help="also train CLIP text-encoder LoRA (configs[2] geometry)")
    ap.add_argument("--res", type=int, default=512, help="image resolution (latents are res/8)")
    ap.add_argument("--conv-find", type=int, default=1, help="torch.backends.cudnn.benchmark: MIOpen Find picks the "
                    "frozen convs' kernels by timing them once, in the warm-up steps before the hipGraph capture "
                    "(+5 %% steps/s measured; 0 = MIOpen's immediate-mode heuristic)")
    ap.add_argument("--with-prior-preservation", type=int, default=0, help="instance + class-prior batch (2 x --batch "
                    "samples per step, ref :698-702, 855-875); not the headline workload")
    ap.add_argument("--merged", type=int, default=1, help="1: the maskless Linear adapters run on the step's merged weight "
                    "W + scale up down (one K3 merge launch per step, frozen GEMMs forward / input gradient, one launch for "
                    "both factor gradients); 0: the fused per-site MFMA kernels of rounds 1-2 (A/B)")
    ap.add_argument("--adapters", choices=["hip", "aten", "none"], default="hip", help="none: the same host UNet with NO "
                    "adapters, gradient to the input latents only (the `frozen_only` leg: step - frozen_only = what the LoRA "
                    "path costs); aten: the reference's op sequence "
                    "(lora.py:53-58 as stock ATen launches) in place of the HIP adapter kernels, same host model: the "
                    "comparison leg behind `secondary[aten_adapters]`")
    ap.add_argument("--device", choices=["cuda", "cpu"], default="cuda", help="cpu: BASELINE configs[0]-style plumbing "
                    "run (gloo for --gpus N > 1, f32, eager, no roofline); what the multi-rank CPU test drives")
    ap.add_argument("--standin", choices=["sd15", "tiny"], default="sd15", help="tiny: 4-level miniature UNet (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--path-log", default=None, help="write the adapter-path profile's per-site records (phase, kernel path, "
                    "M, K, N, r) of one eager step to this JSON file")
    ap.add_argument("--frozen-twins", type=int, default=1, help="--adapters none: 1 = every adapter site runs its frozen twin "
                    "(the merged path's GEMM launches on frozen weights, standin/frozen.py: like for like with the adapter "
                    "step); 0 = plain nn.Linear modules (round 5's form of the leg)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short secondary lines (other BASELINE "
                    "geometries, host options off, ATen adapters) the default 1-GPU run appends under `secondary`")
    ap.add_argument("--secondary-budget", type=float, default=420.0, help="seconds all secondary lines may take together")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--svd", action="store_true", help="time BASELINE configs[4] (cli_svd distillation, 224 sites) "
                    "instead of the training step; prints its own JSON line")
    ap.add_argument("--svd-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    global PATH_LOG_FILE
    PATH_LOG_FILE = args.path_log
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)
    if args.svd_worker:
        return svd_worker(args.svd_worker)
    if args.svd:
        assert torch.cuda.is_available(), "bench.py needs a GPU"
        _C.require()
        if args.steps is None:
            args.steps = 3
        emit(svd_bench(args))
        return

    if args.steps is None:
        args.steps = 50 if args.device == "cuda" else 3
    on_gpu = args.device == "cuda"
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py needs a GPU (use --device cpu for the plumbing run)"
    if args.channels_last < 0:
        args.channels_last = 1 if on_gpu else 0
    os.environ.setdefault("LORA_AMD_HEAD_PAD", str(int(bool(args.head_pad and on_gpu))))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # one process per GPU; rank 0 of the child job prints the JSON line
    if on_gpu:
        torch.backends.cudnn.benchmark = bool(args.conv_find)
        if args.conv_find:
            # Find times every applicable solver once per convolution geometry; MIOpen's naive reference solvers
            # (naive_conv_*: 150-260 ms per call at 768^2 NHWC, never the pick) would turn that into minutes of warm-up
            for k in ("FWD", "BWD", "WRW"):
                os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + k, "0")
        _C.require()
    rank, local, world = T.init_distributed(args.device)
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    if on_gpu:
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
    else:
        dev = torch.device("cpu")
        args.mode = "eager"
    torch.manual_seed(0)
    cdt = torch.bfloat16 if on_gpu else torch.float32

    if args.standin == "tiny":
        from lora_amd.standin import tiny_unet

        unet = tiny_unet(cross_attention_dim=768).to(dev).to(cdt)
        unet.requires_grad_(False)
    else:
        unet = build_unet(dev, cdt, seed=0)
    if args.channels_last:
        unet.to(memory_format=torch.channels_last)
    if on_gpu and os.environ.get("LORA_AMD_FM_NARROW"):   # same-box A/B of the factor pass's class-1 kernels (measurement only)
        _C.factors_mfma_set_tuning(int(os.environ["LORA_AMD_FM_NARROW"]))
    if args.adapters == "none":
        # frozen_only leg: no adapter anywhere.  On the device every site an adapter would occupy gets its FROZEN TWIN
        # (standin/frozen.py): the merged path's own GEMM launches on frozen weights in the same layouts (grouped q / k / v,
        # head-padded rows / columns, transposed copy for the input gradient), so that step - frozen_only isolates the LoRA
        # launches (the plain-Linear form of this leg ran 3 GEMMs where the adapter step runs 1)
        if on_gpu and args.frozen_twins:
            from lora_amd.standin.frozen import install_frozen_twins

            install_frozen_twins(unet, L.UNET_EXTENDED_TARGET_REPLACE if args.extended else L.UNET_DEFAULT_TARGET_REPLACE)
    elif args.extended:
        L.inject_trainable_lora_extended(unet, r=args.lora_rank)  # conv adapters keep the constructor's dropout 0.1
    else:
        L.inject_trainable_lora(unet, r=args.lora_rank)  # reference default: dropout 0, scale 1
    T.promote_lora_to_fp32(unet)
    unet.train()
    if args.adapters == "none":  # a one-element stand-in state keeps the step's tail (all-reduce, clip, AdamW) in the region
        groups = [{"params": [torch.nn.Parameter(torch.zeros(4, device=dev))], "lr": 1e-4, "weight_decay": 1e-2}]
    else:
        groups = [{"params": T.lora_params(unet), "lr": 1e-4, "weight_decay": 1e-2}]
    text_encoder = None
    if args.text_encoder:
        from lora_amd.standin import clip_text_model

        text_encoder = clip_text_model().to(dev).to(cdt)
        text_encoder.requires_grad_(False)
        if args.adapters == "none":
            # frozen twin of configs[2]: the text encoder's backward must still run (in the adapter step it reaches the
            # encoder's LoRA factors): the embedding output becomes a gradient leaf
            if on_gpu and args.frozen_twins:
                from lora_amd.standin.frozen import install_frozen_twins

                install_frozen_twins(text_encoder, ["CLIPAttention"])
            emb = next(m for n_, m in text_encoder.named_modules() if n_.endswith("embeddings"))  # transformers 4 / 5 paths
            emb.register_forward_hook(lambda m, i, o: o.detach().requires_grad_(True))
            text_encoder.train()
        else:
            L.inject_trainable_lora(text_encoder, target_replace_module=["CLIPAttention"], r=args.lora_rank)
            T.promote_lora_to_fp32(text_encoder)
            text_encoder.train()
            groups.append({"params": T.lora_params(text_encoder), "lr": 5e-6, "weight_decay": 1e-2})
    state = T.FlatLoraState(groups, max_grad_norm=1.0, device=dev)
    merged = None
    if args.adapters == "none":
        n_sites = 0
    elif args.adapters == "aten":
        n_sites = swap_in_aten_adapters(unet) + (swap_in_aten_adapters(text_encoder) if text_encoder is not None else 0)
    elif on_gpu:
        n_sites = state.attach_direct_grads(unet, *([text_encoder] if text_encoder is not None else []))
        if args.merged:
            merged = state.enable_merged_weights(unet, *([text_encoder] if text_encoder is not None else []))
    else:
        n_sites = sum(isinstance(m, (L.LoraInjectedLinear, L.LoraInjectedConv2d)) for m in unet.modules())
    sched = DDPMScheduler()
    # weights / factors from seed 0 on every rank (and rank 0's factors are broadcast by FlatLoraState); the noise and
    # timestep stream of a rank from seed + rank, its data shard from 1234 + rank
    torch.manual_seed(rank)
    replicas = None
    if world > 1:
        mine = torch.tensor([float(torch.randn(8, device=dev).double().sum()), float(state.flat_p.double().sum())],
                            dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        replicas = {"noise_stream_fingerprints": [round(float(v[0]), 6) for v in allv],
                    "param_checksums": [float(v[1]) for v in allv]}
    cfg = T.StepConfig(with_prior_preservation=bool(args.with_prior_preservation))
    if args.with_prior_preservation:
        args.batch *= 2  # collate_fn concatenates instance and class examples (ref :698-702)

    g = torch.Generator(device=dev).manual_seed(1234 + rank)  # per-rank data shard
    hw = args.res // 8
    latents = (torch.randn(args.batch, 4, hw, hw, device=dev, generator=g) * 0.18215).to(cdt)
    if args.channels_last:
        latents = latents.contiguous(memory_format=torch.channels_last)
    if text_encoder is not None:
        ehs = torch.randint(0, 49408, (args.batch, 77), device=dev, generator=g)
    else:
        ehs = torch.randn(args.batch, 77, 768, device=dev, generator=g).to(cdt)

    def fwd_bwd(lat, cond):
        if args.adapters == "none":  # gradient to the input only: the frozen UNet's forward + input-gradient chain
            lat = lat.detach().requires_grad_(True)
        return T.forward_backward(unet, sched, lat, cond, cfg, text_encoder=text_encoder, merged=merged)

    def barrier():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    # which kernel runs which site, device time of the adapter kernels, adapter-path bytes: one eager profiled step,
    # BEFORE the capture (rank 0 of a 1-GPU run; its attention / MIOpen choices are the ones the graph then bakes in)
    adapter_path, kernel_choices = None, {}
    # world == 1 only: the eager steps below run the step's collectives (the flat-gradient all-reduce, the broadcast of rank
    # 0's attention choices) and must never be entered by one rank of several
    if on_gpu and world == 1 and args.adapters == "hip" and not args.no_roofline:
        def eager_step():
            fwd_bwd(latents, ehs)
            state.step(state.all_reduce())
        try:
            eager_step()  # first touch: attention candidates and MIOpen Find are timed here, not under the profiler
            adapter_path, kernel_choices = adapter_path_profile(eager_step, state)
        except Exception as e:  # noqa: BLE001 - the profiler is evidence, not the product: say so and go on
            log(f"[bench] adapter-path profile failed: {type(e).__name__}: {e}")
    barrier()

    mode = args.mode
    runner = fwd_bwd
    if mode == "graph":
        try:
            runner = T.GraphedForwardBackward(fwd_bwd, latents, ehs, state)
        except Exception as e:  # capture unsupported in this environment: say so, run eager
            log(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager")
            mode, runner = "eager", fwd_bwd
            state.zero_grad()

    tail_host = []  # host seconds spent ENQUEUEING the eager tail of a step (all-reduce + clip + AdamW: three launches)

    def step():
        loss = runner(latents, ehs)
        t_h = time.perf_counter()
        scale = state.all_reduce()
        state.step(scale)
        tail_host.append(time.perf_counter() - t_h)
        return loss

    for _ in range(args.warmup):
        step()
    barrier()
    tail_host.clear()
    if rank == 0 and on_gpu:  # which kernels the per-shape tuning settled on (stderr; the JSON line stays alone on stdout)
        from lora_amd.standin import attention as _att

        log("[bench] attention kernels:", {k: v for k, v in _att.choices().items()})
        log("[bench] adapter kernels per step:", kernel_choices)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    loss_v = float(loss.item())
    # the step's one collective on its real payload, timed on its own (HIP events, RCCL's stream ordering; host clock
    # around gloo on CPU)
    allreduce_us = None
    if world > 1:
        buf = torch.zeros_like(state.flat_g)
        for _ in range(3):
            dist.all_reduce(buf)
        barrier()
        if on_gpu:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                dist.all_reduce(buf)
            b.record()
            torch.cuda.synchronize()
            allreduce_us = round(a.elapsed_time(b) / 20 * 1e3, 1)
        else:
            ta = time.perf_counter()
            for _ in range(20):
                dist.all_reduce(buf)
            allreduce_us = round((time.perf_counter() - ta) / 20 * 1e6, 1)
        log(f"[bench] ranks: {dist.get_world_size()} (backend {dist.get_backend()}); all-reduce of "
            f"{state.payload_bytes} B: {allreduce_us} us")

    if rank == 0:
        headline = not (args.extended or args.text_encoder or args.res != 512 or args.with_prior_preservation
                        or args.standin != "sd15" or not on_gpu or args.adapters != "hip")
        host_model = ("stand-in UNet2DConditionModel (859,520,964 params, random init)" if args.standin == "sd15"
                      else "tiny stand-in UNet (plumbing)")
        out = {
            "metric": "train steps/sec SD1.5 rank-4 512^2 (train_lora_dreambooth.py step)",
            # whole-job aggregate (driver contract): every rank's step consumes its own batch, so N ranks complete N x K
            # batch-steps in the timed region; at N = 1 this is train_lora_dreambooth.py's global step rate, at N > 1 the
            # global (optimiser-update) rate is config.global_steps_per_s = value / N
            "value": round(args.steps * world / dt, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if on_gpu else "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: SD1.5 UNet LoRA rank-%d bf16, batch %d/GPU, 512x512 (64x64x4 "
                                    "latents), reference-default injection (%d Linear sites: Q/K/V/O + GEGLU), f32 LoRA "
                                    "masters, DDPM+MSE+clip(1.0)+AdamW" % (args.lora_rank, args.batch, n_sites))
                       if headline else
                       ("non-headline variant: rank %d, batch %d/%s, %dx%d, extended=%d, text_encoder=%d, "
                        "prior_preservation=%d, adapters=%s, host=%s, %d adapter sites"
                        % (args.lora_rank, args.batch, "GPU" if on_gpu else "CPU rank", args.res, args.res, args.extended,
                           args.text_encoder, args.with_prior_preservation, args.adapters, args.standin, n_sites)),
                       "workload_short": ("BASELINE configs[1]: SD1.5 UNet LoRA rank-%d bf16 batch %d/GPU 512x512, %d Linear sites"
                                          % (args.lora_rank, args.batch, n_sites)) if headline else
                       ("variant: r%d b%d %dpx ext=%d te=%d prior=%d adapters=%s host=%s dev=%s sites=%d"
                        % (args.lora_rank, args.batch, args.res, args.extended, args.text_encoder,
                           args.with_prior_preservation, args.adapters, args.standin, args.device, n_sites)),
                       "global_batch": args.batch * world, "samples_per_s": round(args.steps * args.batch * world / dt, 3),
                       "parallelism": f"dp{world}",
                       "value_counts": "batch-steps per second summed over the ranks (a rank's step = one pass of the hot path "
                       "over its own batch; weak scaling): value = samples_per_s / batch = n_gpus x global_steps_per_s, where "
                       "global_steps_per_s counts train_lora_dreambooth.py's optimiser updates (one consumes batch x n_gpus samples)",
                       "rank_steps_per_s": round(args.steps * world / dt, 4),
                       "global_steps_per_s": round(args.steps / dt, 4),
                       "timed_region": "noise + add_noise + UNet fwd + MSE + bwd + partial reduce + all-reduce + clip + AdamW; "
                       "VAE encode and CLIP forward (ref :818-840) are outside it: latents and text states are the "
                       "synthetic inputs SURVEY 8d prescribes (cached_latents-style)",
                       "allreduce_us": allreduce_us, "replicas": replicas,
                       "eager_tail": {"what": "the part of a step outside the captured hipGraph: ONE all-reduce of the flat "
                                              "gradient + sumsq + clip/AdamW (three launches), enqueued by the host while the "
                                              "graph's kernels still run",
                                      "host_enqueue_us_per_step": round(sum(tail_host) / max(len(tail_host), 1) * 1e6, 1),
                                      "frac_of_step": round(sum(tail_host) / max(len(tail_host), 1) / (dt / args.steps), 5)},
                       "kernel_choices": kernel_choices,
                       "adapter_options": {"ab_overrides": os.environ.get("LORA_AMD_AB", "") or None,
                                           "grouped_qkv_one_launch": os.environ.get("LORA_AMD_GROUP_QKV", "1") != "0",
                                           "adapters": args.adapters,
                                           "merged_weights": None if merged is None else {
                                               "sites": len(merged.entries), "merge_launches_per_step": len(merged._plans or []),
                                               "merge_bytes_per_step": merged.bytes_algorithmic,
                                               "concatenated_groups": len(merged.groups),
                                               "what": "W_eff = W + scale up down (and its transpose, from one read of W) for "
                                                       "every maskless Linear adapter, one merge_step launch per step inside "
                                                       "the timed region; forward / input gradient = frozen GEMM on it (q/k/v "
                                                       "of a block: one GEMM on a concatenated scratch weight), factor "
                                                       "gradients = one matrix-core pass after the backward"}},
                       "host_model_options": {"channels_last": bool(args.channels_last),
                                              "head_padded_projections": os.environ.get("LORA_AMD_HEAD_PAD") == "1",
                                              "fused_hostops": os.environ.get("LORA_AMD_HOSTOPS", "1") != "0",
                                              "miopen_find": bool(args.conv_find and on_gpu),
                                              "scope": "stand-in UNet only (lora_amd/standin); a diffusers host gets the "
                                                       "grouped q/k/v processor of diffusers_glue.py and nothing else"},
                       "execution": mode, "attention_kernels": _attention_choices() if on_gpu else None,
                       "channels_last": bool(args.channels_last), "host_model": host_model,
                       "device": args.device, "trainable_params": state.n,
                       "allreduce_payload_bytes": state.payload_bytes, "final_loss": round(loss_v, 5)},
        }
        out["samples_per_s"] = out["config"]["samples_per_s"]
        if adapter_path is not None:
            adapter_path["frac_of_step_time"] = round(adapter_path["gpu_ms_per_step"] / out["ms_per_step"], 4)
            out["adapter_path"] = adapter_path
            # the step against the byte roof: the time the adapter path's algorithmic bytes need at 8 TB/s over the
            # measured step time (the rest of the step is the frozen UNet's library MFMA work, left to PyTorch)
            out["step_hbm"] = {"adapter_algorithmic_bytes_per_step": adapter_path["algorithmic_bytes_per_step"],
                               "floor_ms_at_peak": round(adapter_path["algorithmic_bytes_per_step"] / HBM_PEAK * 1e3, 4),
                               "frac_of_step": round(adapter_path["algorithmic_bytes_per_step"] / HBM_PEAK
                                                     / (out["ms_per_step"] * 1e-3), 5)}
            pmc = _newest_profile("step_pmc.json")
            if os.path.exists(pmc):
                try:
                    m = json.load(open(pmc))
                    out["step_hbm"].update({"measured_hbm_bytes_per_step": m["hbm_bytes_per_step"],
                                            "measured_GBs_at_this_step_time": round(m["hbm_bytes_per_step"] /
                                                                                    (out["ms_per_step"] * 1e-3) / 1e9, 1),
                                            "measured_frac_of_peak": round(m["hbm_bytes_per_step"] /
                                                                           (out["ms_per_step"] * 1e-3) / HBM_PEAK, 4),
                                            "static": True,
                                            "source": "profiles/%s: a SEPARATE rocprofv3 --pmc run of this command (FETCH_SIZE / "
                                                      "WRITE_SIZE summed over every dispatch of a step, separate passes), "
                                                      "divided by THIS run's step time; not measured in this run"
                                                      % os.path.basename(pmc)})
                    if "mfma" in m:
                        out["mfma_util"] = dict(m["mfma"], static=True, source="profiles/%s (separate counter run)"
                                                % os.path.basename(pmc))
                except (KeyError, ValueError):
                    pass
        if on_gpu and not args.no_roofline:
            out["roofline"] = merge_roofline(unet)
            if merged is not None and world == 1:
                try:
                    out["roofline_in_step"] = in_step_rooflines(state, fwd_bwd, latents, ehs)
                except Exception as e:  # noqa: BLE001 - evidence, not the product
                    log(f"[bench] in-step roofline failed: {type(e).__name__}: {e}")
            out["roofline_fused_gemm"] = gemm_roofline()
        if world == 1 and on_gpu and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.lora_rank)
        if world == 1 and on_gpu and headline and not args.no_secondary:
            out["secondary"] = run_secondaries(args.secondary_budget, steps=20)
            for rec in out["secondary"]:
                if rec["tag"] == "frozen_only" and "value" in rec:
                    # what the hand-written path costs a step: the same UNet, same graph, adapters removed
                    out["value_frozen_only"] = rec["value"]
                    out["lora_overhead_ms"] = round(out["ms_per_step"] - rec["ms_per_step"], 3)
                if rec["tag"] == "host_options_off" and "value" in rec:
                    out["value_host_options_off"] = rec["value"]
                if rec["tag"] == "aten_adapters" and "value" in rec:
                    out["value_aten_adapters"] = rec["value"]
                if rec["tag"].startswith("fused_per_site_kernels") and "value" in rec:
                    out["value_fused_per_site_kernels"] = rec["value"]
            # the same difference for the other two training geometries: adapter step - its frozen twin, same child-run length
            by_tag = {rec["tag"]: rec for rec in out["secondary"] if "ms_per_step" in rec and rec.get("ms_per_step")}
            for cfg, full in (("cfg2", "configs[2] UNet+CLIP rank 8"), ("cfg3", "configs[3] extended rank 16 768^2 batch 1")):
                twin = "frozen_only configs[%s]" % cfg[-1]
                if full in by_tag and twin in by_tag:
                    out["lora_overhead_ms_" + cfg] = round(by_tag[full]["ms_per_step"] - by_tag[twin]["ms_per_step"], 3)
                    out["value_frozen_only_" + cfg] = by_tag[twin]["value"]
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
