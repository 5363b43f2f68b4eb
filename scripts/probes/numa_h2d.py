"""H2D bandwidth of pinned buffers allocated while the process is bound to each NUMA node's cores (first-touch placement)"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out

nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
print("nodes:", {k: f"{v[0]}..{v[-1]} ({len(v)})" for k, v in nodes.items() if v})
for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print(f, open(f).read().strip())
print("allowed cpus:", len(os.sched_getaffinity(0)))
allowed = os.sched_getaffinity(0)
dev = torch.device("cuda:0")
dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for node, cpus in nodes.items():
    cpus = [c for c in cpus if c in allowed]
    if not cpus:
        continue
    os.sched_setaffinity(0, cpus)
    src = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    src.fill_(1)
    best = 0
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = max(best, (1 << 30) / dt / 1e9)
    print(f"node {node}: H2D {best:.1f} GB/s")
    del src
os.sched_setaffinity(0, allowed)
