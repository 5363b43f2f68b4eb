#!/usr/bin/env python3
"""gpurun_out/ (scripts/collect_round.sh <tag>, scripts/collect_split_pmc.sh <tag>) -> profiles/<tag>_*: bench lines, parity report, split A/B
and job statistics, kernel-stats summaries of every profiled path, PMC traffic (chunked kernels, then the two launches of the split
pass replayed alone) and profiles/traffic_latest.json.     python scripts/summarize_round.py r05"""
import glob
import os
import shutil
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
copied = []


def cp(src, name):
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(dst, name))
        copied.append(name)


for f in glob.glob(os.path.join(out, f"{tag}_bench_line*.json")):
    cp(f, os.path.basename(f))
cp(os.path.join(out, f"parity_{tag}.json"), f"parity_{tag}.json")
cp(os.path.join(out, f"{tag}_split_ab.log"), f"{tag}_split_ab.log")
cp(os.path.join(out, f"{tag}_split_job_stats.log"), f"{tag}_split_job_stats.log")
newest = lambda fs: sorted(fs, key=os.path.getmtime)[-1:]
for path, name in (("train", "train"), ("vit", "vit"), ("vit_train", "vit_train"), ("attn", "attn"), ("step", "step"), ("chunked", "chunked"), ("x2", "x2")):
    for f in newest(glob.glob(os.path.join(out, f"prof_{tag}_{path}", "**", "*kernel_stats.csv"), recursive=True)):
        cp(f, f"{tag}_{name}_kernel_stats.csv")
subprocess.run([sys.executable, os.path.join(root, "scripts", "summarize_profiles.py"), tag], check=False)
if os.path.isdir(os.path.join(out, f"prof_{tag}_split")):
    subprocess.run([sys.executable, os.path.join(root, "scripts", "summarize_split_pmc.py"), tag], check=False)
print("copied:", ", ".join(sorted(copied)))
