#!/usr/bin/env python3
"""Registers / spills / LDS of every kernel in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
    python tools/kernel_resources.py las_pytorch_amd/csrc/gemm_f32.hip [filter-substring] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
if src.endswith("gemm_f32.hip"):
    flags.insert(0, "-fno-slp-vectorize")
out = subprocess.run(["/opt/rocm/bin/hipcc"] + extra + flags, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?:\[[^\]]*\])?):\s*(\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, r in rows.items():
    if flt in name:
        print(f"{name[:110]:110s} vgpr {r.get('VGPRs')} agpr {r.get('AGPRs')} spill v{r.get('VGPRs Spill')} s{r.get('SGPRs Spill')} "
              f"scratch {r.get('ScratchSize [bytes/lane]')} occ {r.get('Occupancy [waves/SIMD]')} lds {r.get('LDS Size [bytes/block]')}")
