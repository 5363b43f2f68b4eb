#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_split/ (scripts/collect_split_pmc.sh) -> profiles/<tag>_split_pmc_traffic.csv, profiles/<tag>_split_replay_kernel_stats.csv
and the ff_pass / gru_pass entries of profiles/traffic_latest.json.

Counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are KiB on the memory side of the L2s; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, so the read side is doubled.  One row per kernel: launches, raw counters per launch, corrected bytes per
launch, and - for the two launches of the split pass - the algorithmic bytes per launch of DESIGN.md section 4 and the ratio."""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}_split")
dst = os.path.join(root, "profiles")
newest = lambda fs: sorted(fs, key=os.path.getmtime)[-1:]


def pmc(name, counter):
    out = defaultdict(lambda: [0, 0.0])
    for f in newest(glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                k = r["Kernel_Name"].split("(")[0]
                out[k][0] += 1
                out[k][1] += float(r["Counter_Value"])
    return out


st = newest(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True))
if st:
    shutil.copy(st[0], os.path.join(dst, f"{tag}_split_replay_kernel_stats.csv"))
frames = None
log = os.path.join(src, "trace.log")
if os.path.exists(log):
    m = re.search(r"frames (\d+) clips (\d+)", open(log).read())
    if m:
        frames = int(m.group(1))
    shutil.copy(log, os.path.join(dst, f"{tag}_split_replay.log"))
fe, wr = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
is_pass = lambda k: "gru_recurrence_kernel" in k and ("ELb1ELb1EEv" in k or "true, true>" in k)
# DESIGN 4, bytes per frame: the feed-forward launch reads 16 384 B of fp32 features and writes 8 192 B of X (pack: 24 576), writes and
# reads Y (2 x 4 096) and E (2 x 4 096), writes 6 144 B of GI (X, Y and E are meant to be re-read from the unit's own XCD's L2: what
# exceeds this figure is ring traffic that did reach the fabric, plus the weight slabs); the recurrence reads GI and writes 2 048 B of relu(h)
ALG = {"ff": 24576 + 2 * 4096 + 2 * 4096 + 6144, "rec": 6144 + 2048}
rows = []
for k in sorted(set(fe) | set(wr)):
    nf, vf = fe.get(k, [0, 0.0]); nw, vw = wr.get(k, [0, 0.0])
    rd = 2 * 1024 * vf / nf if nf else 0.0
    wb = 1024 * vw / nw if nw else 0.0
    row = {"kernel": k, "launches": max(nf, nw), "FETCH_SIZE_KiB_per_launch_raw": vf / nf if nf else 0,
           "read_bytes_per_launch_corrected_x2": rd, "WRITE_SIZE_KiB_per_launch": vw / nw if nw else 0, "write_bytes_per_launch": wb,
           "traffic_bytes_per_launch": rd + wb, "algorithmic_bytes_per_launch": "", "traffic_over_algorithmic": ""}
    kind = "ff" if "ff_pass_kernel" in k else "rec" if is_pass(k) else None
    if kind and frames:
        alg = ALG[kind] * frames
        row["algorithmic_bytes_per_launch"] = alg
        row["traffic_over_algorithmic"] = round((rd + wb) / alg, 4)
    rows.append(row)
if not rows:
    sys.exit("no counter files under " + src)
with open(os.path.join(dst, f"{tag}_split_pmc_traffic.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
tp = os.path.join(dst, "traffic_latest.json")
t = json.load(open(tp)) if os.path.exists(tp) else {}
try:
    git = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    git = None
for r in rows:
    if "ff_pass_kernel" in r["kernel"]:
        t["ff_pass_bytes_per_launch"] = r["traffic_bytes_per_launch"]
        t["ff_pass_algorithmic_bytes_per_launch"] = r["algorithmic_bytes_per_launch"]
    if is_pass(r["kernel"]):
        t["gru_pass_bytes_per_launch"] = r["traffic_bytes_per_launch"]
        t["gru_pass_algorithmic_bytes_per_launch"] = r["algorithmic_bytes_per_launch"]
t["split_source"] = f"profiles/{tag}_split_pmc_traffic.csv"
t["split_git"] = git
t["split_workload"] = (f"scripts/split_replay.py: the bench workload of BASELINE configs[1] ({frames} frames), each launch of the split pass "
                       "replayed ALONE (debug library, handshake pre-decided, the partner's counters pre-armed) under two separate rocprofv3 --pmc "
                       "passes (FETCH_SIZE x 2, WRITE_SIZE)")
json.dump(t, open(tp, "w"))
for r in rows:
    if r["algorithmic_bytes_per_launch"] != "":
        print(r)


# ---- matrix-pipe and LDS counters of the same replay (collect_split_pmc.sh passes d, e): one row per kernel and counter, per launch
def pmc_all(name):
    out = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in newest(glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            c = out[k][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    return out


extra = []
SIMDS = 256 * 4                 # SIMD_NUM of an MI355X (256 CUs x 4): what rocprofv3's MfmaUtil expression divides by
for name in ("pmc_mfma", "pmc_lds"):
    for k, cs in pmc_all(name).items():
        if not ("ff_pass_kernel" in k or is_pass(k)):
            continue
        per = {c: v[1] / v[0] for c, v in cs.items() if v[0]}
        row = {"kernel": k, "pass": name, **{c: round(v, 1) for c, v in per.items()}}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in per and per.get("GRBM_GUI_ACTIVE"):
            xcds = 5 if "ff_pass_kernel" in k else 3
            # the CSV holds ONE value per dispatch and counter, summed over the counter's instances: GRBM_GUI_ACTIVE over the 8 XCDs (a
            # 92.2 ms launch reads 1.70e9 = 8 x 2.30 GHz x 92.2 ms), the SQ counters over all SIMDs.  rocprofv3's MfmaUtil expression is
            # busy cycles / (max-over-instances active cycles x SIMD_NUM): the per-XCD active cycles are the sum / 8
            util = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (per["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS) * 100.0
            row["MfmaUtil_pct_of_chip"] = round(util, 2)
            row["MfmaUtil_pct_of_its_xcds"] = round(util * 8.0 / xcds, 2)
            mops = per.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) + per.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
            row["mfma_flop_per_launch"] = mops * 512.0
            t[("ff_pass" if "ff_pass_kernel" in k else "gru_pass") + "_mfma_util_pct_of_its_xcds"] = row["MfmaUtil_pct_of_its_xcds"]
        if "SQ_LDS_BANK_CONFLICT" in per and per.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_frac_of_active"] = round(per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"], 4)
        extra.append(row)
if extra:
    keys = []
    for r in extra:
        for k in r:
            if k not in keys:
                keys.append(k)
    with open(os.path.join(dst, f"{tag}_split_pmc_mfma_lds.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keys); w.writeheader(); w.writerows(extra)
    json.dump(t, open(tp, "w"))
    for r in extra:
        print(r)
