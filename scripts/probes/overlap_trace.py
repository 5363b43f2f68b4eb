"""From a rocprofv3 --kernel-trace CSV of one bench run: how much of the layer1 worker GEMM (gemm_bf16_nt_pingpong_kernel<..., true>)
ran INSIDE recurrence launches.  usage: python scripts/probes/overlap_trace.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
name = lambda r: r.get("Kernel_Name", "")
t0 = lambda r: int(r["Start_Timestamp"])
t1 = lambda r: int(r["End_Timestamp"])
rec = sorted([(t0(r), t1(r)) for r in rows if "gru_recurrence" in name(r)])
wrk = sorted([(t0(r), t1(r)) for r in rows if "pingpong" in name(r) and ("Lb1E" in name(r) or ", true>" in name(r))])
stat = sorted([(t0(r), t1(r)) for r in rows if "pingpong" in name(r) and not ("Lb1E" in name(r) or ", true>" in name(r))])


def inside(a, bs):
    tot = 0
    for b in bs:
        lo, hi = max(a[0], b[0]), min(a[1], b[1])
        if hi > lo:
            tot += hi - lo
    return tot


ov = sum(inside(w, rec) for w in wrk)
out = {"recurrence_launches": len(rec), "worker_gemm_launches": len(wrk), "static_gemm_launches": len(stat),
       "worker_gemm_total_ms": sum(b - a for a, b in wrk) / 1e6, "of_which_inside_a_recurrence_launch_ms": ov / 1e6,
       "recurrence_total_ms": sum(b - a for a, b in rec) / 1e6,
       "note": "a worker launch lasts until its workgroups pinned to the recurrence's XCDs have been dispatched, i.e. at least as long as the "
               "recurrence launch beside it; its tiles run on the other XCDs meanwhile"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
