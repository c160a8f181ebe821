"""Per-kernel resource usage (VGPRs, spills, LDS, occupancy) of one .hip file, from hipcc's kernel-resource-usage remarks.

usage: python scripts/kres.py lora_amd/csrc/factor_mfma.hip [name substring]   (cross-compiles for gfx950; no GPU needed)"""
import os
import re
import subprocess
import sys


def resources(path: str, extra=()):
    d = os.path.dirname(os.path.abspath(path))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{repo}/include", f"-I{d}",
           f"-I{repo}/lora_amd/csrc", "-c", path, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    recs, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?:\[[^\]]+\])?):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            recs.append(cur)
        elif cur is not None:
            cur[k] = v
    return recs


if __name__ == "__main__":
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for r in resources(sys.argv[1]):
        if pat in r["name"]:
            try:
                name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip() or r["name"]
            except OSError:
                name = r["name"]
            print(f"{name[:110]:110s} vgpr {r.get('VGPRs')} agpr {r.get('AGPRs')} spill v{r.get('VGPRs Spill')} s{r.get('SGPRs Spill')} "
                  f"lds {r.get('LDS Size [bytes/block]')} occ {r.get('Occupancy [waves/SIMD]')}")
