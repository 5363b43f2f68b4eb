"""From a rocprofv3 --kernel-trace CSV of one bench run: the serial gaps of the LAST pass.  For every recurrence launch c: its own
duration, the gap to the next recurrence launch and which kernels ran (wall-clock union per kind) inside that gap; totals per kind.
usage: python scripts/probes/pass_timeline.py <kernel_trace.csv> [launches per pass = 47] [out.json]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per_pass = int(sys.argv[2]) if len(sys.argv) > 2 else 47


def kind(n):
    if "gru_recurrence" in n:
        return "recurrence"
    if "pingpong" in n:
        return "worker_gemm" if ("Lb1E" in n or ", true>" in n) else "gemm"
    for k in ("pack_rows", "head_softmax", "ln_relu_rows", "gru_arm"):
        if k in n:
            return k
    return "other"


ev = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind(r.get("Kernel_Name", ""))) for r in rows])
rec = [e for e in ev if e[2] == "recurrence"]
rec = rec[-per_pass:]
t_lo = rec[0][0]
out, tot = [], {}
for i, (a, b, _) in enumerate(rec):
    nxt = rec[i + 1][0] if i + 1 < len(rec) else None
    row = {"chunk": i, "start_ms": (a - t_lo) / 1e6, "recurrence_ms": (b - a) / 1e6}
    if nxt is not None:
        row["gap_ms"] = (nxt - b) / 1e6
        inside = {}
        for s, e, k in ev:
            lo, hi = max(s, b), min(e, nxt)
            if hi > lo and k != "recurrence":
                inside[k] = inside.get(k, 0) + (hi - lo) / 1e6
        row["in_gap_ms"] = {k: round(v, 4) for k, v in inside.items()}
        for k, v in inside.items():
            tot[k] = tot.get(k, 0) + v
        tot["gap"] = tot.get("gap", 0) + row["gap_ms"]
    tot["recurrence"] = tot.get("recurrence", 0) + row["recurrence_ms"]
    out.append(row)
res = {"pass_ms_first_to_last_recurrence": (rec[-1][1] - t_lo) / 1e6, "totals_ms": {k: round(v, 3) for k, v in tot.items()}, "chunks": out}
print(json.dumps(res["totals_ms"], indent=1), res["pass_ms_first_to_last_recurrence"])
for r in out:
    print(r["chunk"], f"{r['start_ms']:.2f}", f"rec {r['recurrence_ms']:.3f}", f"gap {r.get('gap_ms', 0):.3f}", r.get("in_gap_ms"))
if len(sys.argv) > 3:
    json.dump(res, open(sys.argv[3], "w"), indent=1)
