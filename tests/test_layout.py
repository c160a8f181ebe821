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
