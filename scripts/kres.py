#!/usr/bin/env python3
"""Kernel resource table of one csrc file: python scripts/kres.py ff_pass [filter] [-D...]
(hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel: SGPR / VGPR / AGPR / scratch / spills / occupancy / LDS)"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
flt = [a for a in sys.argv[2:] if not a.startswith("-")]
extra = [a for a in sys.argv[2:] if a.startswith("-")]
src = os.path.join(root, "prego_amd", "csrc", name + (".hip" if not name.endswith((".hip", ".cpp")) else ""))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-inline-asm",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: +(.+?) \[-Rpass", line) or re.search(r"remark: +(.+?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    try:
        dem = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = r["name"]
    if flt and not all(f in dem for f in flt):
        continue
    print(f"{dem[:110]:110s} S{r.get('TotalSGPRs','?'):>4} V{r.get('VGPRs','?'):>4} A{r.get('AGPRs','?'):>4} scr {r.get('ScratchSize [bytes/lane]','?'):>4} "
          f"sspill {r.get('SGPRs Spill','?'):>4} vspill {r.get('VGPRs Spill','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}")
