"""print the HIP API calls (name, start, duration) around the last hipEventSynchronize calls of a rocprofv3 --hip-trace run"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if not r["Function"].startswith("__hip")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Function"] == "hipEventSynchronize"]
lo = idx[-7] - 3
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:idx[-1] + 6]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 8 or r["Function"] in ("hipEventSynchronize", "hipEventRecord", "hipMemcpyAsync", "hipMemsetAsync"):
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {r['Function']:28s} {d:8.1f} us")
