"""The ONE stdout line of bench.py (driver contract): short enough for the driver's stdout tail, carries ``roofline`` and
``cpu_baseline``, secondaries reduced to their numbers; the detail record goes to a side file.  Round 4's line was 25 KB and
the driver's record of it (``BENCH_r04.parsed``) was null."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

from tests import helpers as H


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(H.REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _full_record():
    """A record with every field the 1-GPU default run produces, each at a generous size (round 4's own line, padded)."""
    with open(os.path.join(H.REPO, "profiles", "r04_bench_line.json")) as f:
        d = json.load(f)
    d["config"]["workload"] = d["config"]["workload"] * 3
    d["secondary"] = d["secondary"] + [dict(d["secondary"][0], tag="frozen_only " + "x" * 300)] * 3
    d["mfma_util"] = {"tables": ["y" * 100] * 200}
    d["lora_overhead_ms"], d["value_frozen_only"] = 4.321, 39.9
    for e in d["roofline_fused_gemm"]:
        e["kernel"] = e["kernel"] * 5
    return d


def test_compact_line_is_short_and_keeps_the_graded_objects(bench):
    d = _full_record()
    assert len(json.dumps(d)) > 25000
    rec = bench.compact_record(d, "gpurun_out/bench_detail_n1.json")
    line = json.dumps(rec)
    assert len(line) < 8192 and len(line) <= bench.LINE_LIMIT, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_in_step", "cpu_baseline", "secondary"):
        assert k in rec, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rec["roofline"], k
    assert rec["roofline"]["in_step"]["factor_pass"]["frac"] == d["roofline_in_step"]["factor_pass"]["frac"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    assert all(set(s) <= {"tag", "value", "unit", "ms_per_step", "steps", "execution", "skipped"} for s in rec["secondary"])
    assert "workload" in rec["config"] and len(rec["config"]["workload"]) <= 128
    assert all(not isinstance(v, (dict, list)) for v in rec["config"].values())  # scalars only
    assert len(rec["config"]) <= 24
    assert all(len(v) <= 128 for v in _strings(rec)), [v for v in _strings(rec) if len(v) > 128]


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)


def test_hard_guard_drops_optional_objects_rather_than_overflow(bench):
    d = _full_record()
    d["secondary"] = d["secondary"] * 40
    rec = bench.compact_record(d, "")
    assert len(json.dumps(rec)) <= bench.LINE_LIMIT
    assert "roofline" in rec and "cpu_baseline" in rec and "value" in rec


def test_svd_record_compacts_too(bench):
    with open(os.path.join(H.REPO, "profiles", "r04_bench_svd.json")) as f:
        d = json.load(f)
    rec = bench.compact_record(d, "")
    assert len(json.dumps(rec)) < 4096
    assert rec["unit"] == "sites/s" and "roofline" in rec and rec["config"]["sites"] == 224


def test_cpu_run_prints_one_parseable_stdout_line_and_a_detail_file(tmp_path):
    """`bench.py --device cpu --standin tiny`: exactly one stdout line, JSON, < 8 KB; the detail record lands in gpurun_out/."""
    env = {**os.environ, "LORA_AMD_BENCH_DETAIL_TAG": "cputest"}
    r = subprocess.run([sys.executable, os.path.join(H.REPO, "bench.py"), "--device", "cpu", "--standin", "tiny", "--steps", "2",
                        "--warmup", "1", "--batch", "1", "--res", "64"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert len(lines[0]) < 8192 and rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["config"]["device"] == "cpu"
    detail = os.path.join(H.REPO, "gpurun_out", "bench_detail_cputest.json")
    assert os.path.exists(detail)
    with open(detail) as f:
        assert json.load(f)["value"] == rec["value"]
    os.remove(detail)
    assert "[bench-detail] {" in r.stderr


def test_eight_gloo_ranks_print_one_short_line_with_distinct_noise_streams_and_equal_replicas():
    """the 8-rank line (what the driver's SCALE run parses) — driven on the CPU (gloo, tiny stand-in UNet):
    exactly ONE parseable stdout line < 8 KB from rank 0 with n_gpus 8 and the all-reduce timed; every rank's noise stream is
    seeded seed + rank (distinct), every rank's factors are rank 0's (equal checksums)."""
    env = {**os.environ, "OMP_NUM_THREADS": "1", "LORA_AMD_BENCH_DETAIL_TAG": "cputest8"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(H.REPO, "bench.py"), "--gpus", "8", "--device", "cpu", "--standin", "tiny",
                        "--steps", "2", "--warmup", "1", "--res", "64", "--batch", "1"], capture_output=True, text=True,
                       timeout=900, env=env, cwd=H.REPO)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    # (gloo itself writes interleaved "[Gloo] Rank i is connected ..." chatter to stdout on the CPU; RCCL does not)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0], r.stdout[-3000:]
    assert len(lines[0]) < 8192
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "dp8" and d["config"]["global_batch"] == 8
    assert d["config"]["allreduce_us"] and d["config"]["allreduce_us"] > 0
    assert d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d and "secondary" not in d
    detail = os.path.join(H.REPO, "gpurun_out", "bench_detail_cputest8.json")
    with open(detail) as f:
        rep = json.load(f)["config"]["replicas"]
    os.remove(detail)
    assert len(set(rep["noise_stream_fingerprints"])) == 8, rep
    assert len(set(rep["param_checksums"])) == 1, rep


def _bench_line(n_ranks: int, tag: str):
    env = {**os.environ, "OMP_NUM_THREADS": "1", "LORA_AMD_BENCH_DETAIL_TAG": tag}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(H.REPO, "bench.py"), "--gpus", str(n_ranks), "--device", "cpu", "--standin",
                        "tiny", "--steps", "2", "--warmup", "1", "--res", "64", "--batch", "2", "--no-cpu-baseline",
                        "--no-secondary"], capture_output=True, text=True, timeout=900, env=env, cwd=H.REPO)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    detail = os.path.join(H.REPO, "gpurun_out", "bench_detail_%s.json" % tag)
    with open(detail) as f:
        det = json.load(f)
    os.remove(detail)
    return json.loads(lines[0]), det


def test_lines_at_one_and_two_ranks_are_the_same_metric_and_scale_weakly():
    """what the driver's SCALE run compares across N.  The N = 1 and N = 2 lines (gloo on the CPU, tiny
    stand-in UNet) name the same metric / unit / dtype / workload, differ in n_gpus / parallelism / global batch exactly as
    weak scaling says (per-rank batch fixed), and report the WHOLE-JOB aggregate the driver divides by N x value(1):
    value = batch-steps per second summed over the ranks = rank_steps_per_s = n_gpus x global_steps_per_s (optimiser updates
    per second) = samples_per_s / per-rank batch; the N = 2 line carries the timed all-reduce and its payload (the flat LoRA
    gradient: 4 bytes x trainable parameters)."""
    one, det1 = _bench_line(1, "cpuscale1")
    two, det2 = _bench_line(2, "cpuscale2")
    for k in ("metric", "unit", "dtype", "higher_is_better", "scaling", "data", "steps", "warmup"):
        assert one[k] == two[k], k
    assert one["scaling"] == "weak" and one["vs_baseline"] is None and two["vs_baseline"] is None
    assert (one["n_gpus"], two["n_gpus"]) == (1, 2)
    c1, c2 = one["config"], two["config"]
    assert c1["workload"] == c2["workload"]
    assert (c1["parallelism"], c2["parallelism"]) == ("dp1", "dp2") and (c1["global_batch"], c2["global_batch"]) == (2, 4)
    assert c1["allreduce_us"] is None and c2["allreduce_us"] > 0
    assert c1["allreduce_payload_bytes"] == c2["allreduce_payload_bytes"] == 4 * c1["trainable_params"]
    for line, det in ((one, det1), (two, det2)):
        c = line["config"]
        per_rank_batch = c["global_batch"] // line["n_gpus"]
        assert per_rank_batch == 2
        assert abs(c["samples_per_s"] - line["value"] * per_rank_batch) <= 1e-2 * c["samples_per_s"] + 1e-3
        assert abs(det["config"]["rank_steps_per_s"] - line["value"]) <= 1e-6 + 1e-3 * line["value"]
        assert abs(c["global_steps_per_s"] * line["n_gpus"] - line["value"]) <= 1e-3 * line["value"] + 1e-3
        assert abs(line["ms_per_step"] * c["global_steps_per_s"] - 1e3) <= 1.0   # ms_per_step: wall time of one global step
    rep = det2["config"]["replicas"]
    assert len(set(rep["noise_stream_fingerprints"])) == 2 and len(set(rep["param_checksums"])) == 1
