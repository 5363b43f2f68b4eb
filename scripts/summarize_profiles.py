#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ -> profiles/<tag>_kernel_stats.csv + profiles/<tag>_pmc_traffic.csv + profiles/traffic_latest.json
FETCH_SIZE on gfx950 reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section): the read side is
doubled, as that guide prescribes; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Units: KiB."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
newest = lambda fs: sorted(fs, key=os.path.getmtime)[-1:]          # gpurun_out/ keeps the files of earlier collections of the same tag
st = newest(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True))
if st:
    shutil.copy(st[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))


def pmc(name, counter):
    out = defaultdict(lambda: [0, 0.0])
    for f in newest(glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                k = r["Kernel_Name"].split("(")[0]
                out[k][0] += 1
                out[k][1] += float(r["Counter_Value"])
    return out


fe, wr = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
rows = []
for k in sorted(set(fe) | set(wr)):
    nf, vf = fe.get(k, [0, 0.0]); nw, vw = wr.get(k, [0, 0.0])
    rows.append({"kernel": k, "launches": max(nf, nw), "FETCH_SIZE_KiB_per_launch_raw": vf / nf if nf else 0,
                 "read_bytes_per_launch_corrected_x2": 2 * 1024 * vf / nf if nf else 0,
                 "WRITE_SIZE_KiB_per_launch": vw / nw if nw else 0, "write_bytes_per_launch": 1024 * vw / nw if nw else 0})
if rows:
    with open(os.path.join(dst, f"{tag}_pmc_traffic.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    g = [r for r in rows if "gemm_bf16_nt" in r["kernel"]]
    p = [r for r in rows if "pack_rows" in r["kernel"]]
    is_pass = lambda k: "gru_recurrence_kernel" in k and ("ELb1ELb1EEv" in k or "true, true>" in k)     # the PASS instantiation (split pass)
    u = [r for r in rows if "gru_recurrence_kernel" in r["kernel"] and not is_pass(r["kernel"])]
    up = [r for r in rows if is_pass(r["kernel"])]
    ff = [r for r in rows if "ff_pass_kernel" in r["kernel"]]
    tot = lambda rs: sum((r["read_bytes_per_launch_corrected_x2"] + r["write_bytes_per_launch"]) * r["launches"] for r in rs) / max(1, sum(r["launches"] for r in rs))
    import subprocess
    try:
        git = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        git = None
    json.dump({"source": f"profiles/{tag}_pmc_traffic.csv", "git": git, "dtype": os.environ.get("PREGO_PROFILE_DTYPE", "fp16"),
               "workload": "bench.py --clips 64 --len-scale 0.25 (64 clips at a quarter of the bench lengths: the same 49 152-row chunks per launch), "
                           "two separate rocprofv3 --pmc passes (FETCH_SIZE x 2, WRITE_SIZE)",
               "gemm_bytes_per_launch": tot(g), "pack_bytes_per_launch": tot(p), "gru_bytes_per_launch": tot(u),
               "ff_pass_bytes_per_launch": tot(ff) if ff else None, "gru_pass_bytes_per_launch": tot(up) if up else None},
              open(os.path.join(dst, "traffic_latest.json"), "w"))
    for r in rows[:12]:
        print(r)
