"""Repository contract: the product never touches the oracle; required files exist."""
import os
import re

from tests.helpers import REPO


def _py_files(root):
    for d, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_does_not_import_oracle_or_reference():
    bad = []
    for root in ("lora_amd", "training_scripts"):
        for p in _py_files(os.path.join(REPO, root)):
            src = open(p).read()
            if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "/root/reference" in src.replace(
                    "``/root/reference", ""):
                bad.append(p)
    assert not bad, f"product code references the oracle or the reference tree: {bad}"


def test_gpu_side_code_does_not_read_reference_at_runtime():
    for p in ("bench.py", "__graft_entry__.py"):
        path = os.path.join(REPO, p)
        if os.path.exists(path):
            assert "/root/reference" not in open(path).read(), p


def test_required_layout():
    for p in ("include/lora_amd.h", "oracle/lora_numpy.py", "oracle/torch_ref.py", "tests/golden", "profiles",
              "__graft_entry__.py", "lora_amd/csrc/merge.hip", "lora_amd/csrc/linear.hip", "lora_amd/csrc/optim.hip",
              "lora_amd/csrc/linear_fused.hip", "lora_amd/csrc/gemm_fused.hip", "lora_amd/csrc/conv.hip", "DESIGN.md",
              "INTEGRATION.md", "bench.py", "training_scripts/train_lora_dreambooth.py", "lora_amd/cli_lora_pti.py"):
        assert os.path.exists(os.path.join(REPO, p)), p


def test_bench_roofline_entry_reports_the_binding_roof():
    """bench.roofline_entry: below the machine balance (2.5 PF / 8 TB/s = 312 flop/B) the byte roof binds."""
    import bench

    M, K, N = 16384, 320, 320
    e = bench.roofline_entry(2.0 * M * K * N, (M * K + N * K + M * N) * 2, 15.7e-6)
    assert e["bound"] == "hbm" and e["unit"] == "GB/s" and abs(e["frac"] - e["achieved"] / e["peak"]) < 1e-3
    assert 0.16 < e["frac"] < 0.18 and e["mfma_frac"] < e["hbm_frac"]
    e = bench.roofline_entry(2.0 * 8192 ** 3, 3 * 8192 * 8192 * 2, 1e-3)  # square GEMM: 2731 flop/B
    assert e["bound"] == "mfma" and e["unit"] == "TFLOP/s" and abs(e["frac"] - 0.4398) < 1e-3


def test_bench_self_launch_fails_loudly_without_enough_gpus():
    """`python bench.py --gpus N` must start N ranks itself or refuse; it may never report a 1-GPU run as N."""
    import argparse

    import pytest
    import torch

    import bench

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box could actually launch 2 ranks")
    with pytest.raises(SystemExit, match="--gpus 2 requested"):
        bench.self_launch(argparse.Namespace(gpus=2, device="cuda"))


def test_bench_self_launch_runs_two_gloo_ranks_end_to_end():
    """`python bench.py --gpus 2` with no launcher environment must start its own ranks
    (torch.distributed.run, 127.0.0.1), shard the batch, all-reduce the flat gradient and have rank 0 print ONE JSON line
    with n_gpus 2 — driven here on the CPU (`--device cpu`: gloo, tiny stand-in UNet), the same code path the GPU run
    takes with backend nccl."""
    import json
    import os
    import subprocess
    import sys

    from tests import helpers as H

    env = {**os.environ, "OMP_NUM_THREADS": "2"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(H.REPO, "bench.py"), "--gpus", "2", "--device", "cpu", "--standin", "tiny",
                        "--steps", "2", "--warmup", "1", "--res", "128", "--batch", "1"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=H.REPO)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 2
    assert d["config"]["allreduce_us"] and d["config"]["allreduce_us"] > 0
    assert d["config"]["allreduce_payload_bytes"] == 4 * d["config"]["trainable_params"]
    assert len(lines[0]) < 8192
    assert d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d and "secondary" not in d
    assert "backend gloo" in r.stderr
    # the part of a step outside the graph (one all-reduce + clip + AdamW) is reported with its host cost
    # (detail record: the stdout line carries scalars only)
    det = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("[bench-detail] ")][-1][len("[bench-detail] "):])
    assert det["config"]["eager_tail"]["host_enqueue_us_per_step"] > 0 and 0 < det["config"]["eager_tail"]["frac_of_step"] < 1
