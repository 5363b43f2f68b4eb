"""print the kernel sequence (name, duration us, gap to the previous kernel's end us) of the last N kernels of a rocprofv3 kernel trace"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:64]:64s} {(e - s) / 1e3:8.1f} us  gap {((s - prev) / 1e3) if prev else 0:7.1f}  grid {r.get('Grid_Size_X','')}")
    prev = e
