#!/usr/bin/env python3
"""Feed realism for an 8-GPU `--eval` (round-5 verdict, item 8): what does ONE rank's link-fed eval lose when the host DRAM is also
serving the other ranks' feeders?  A one-GPU box cannot run eight ranks; it can run the host side of them: N background processes that
stream through their own pinned-size host buffers (memcpy of 1 GiB blocks: one read + one write stream each, the DRAM traffic of a
57 GB/s H2D source plus its loader filling the next buffer), pinned to the cores of one NUMA node or spread over all, while the
measured process runs `Evaluate` end to end on the whole 182-video set (16-bit pinned features, as scripts/eval_e2e_bench.py).
Reports frames/s of the eval and the aggregate GB/s the background processes moved, for N = 0, 3, 7 (8 ranks = this one + 7).
    python3 scripts/probes/feed_contention.py [n_clips]"""
import glob
import json
import logging
import multiprocessing as mp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
    return out


def hog(cpus, stop, moved, idx):
    import numpy as np
    if cpus:
        os.sched_setaffinity(0, cpus)
    a = np.ones(1 << 28, np.uint8)             # 256 MiB blocks: far past any cache
    b = np.empty_like(a)
    n = 0
    while not stop.value:
        np.copyto(b, a)
        n += 1
    moved[idx] = n * 2 * a.nbytes               # bytes read + written


def main():
    import torch
    from prego_amd import weights as W
    from prego_amd.config import assembly101_cfg
    from prego_amd.registry import build_model, build_eval
    from prego_amd.workloads import assembly101_eval_lengths
    import prego_amd.model, prego_amd.evaluate  # noqa: F401
    n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 182
    nodes = {int(d.rsplit("node", 1)[1]): cpulist(open(d + "/cpulist").read()) for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))}
    allowed = sorted(os.sched_getaffinity(0))
    gpu_node = None
    for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
        try:
            gpu_node = int(open(f).read())
        except Exception:
            pass
    print(f"host: {len(allowed)} allowed cpus, NUMA nodes {{{', '.join(f'{k}: {len(v)} cpus' for k, v in nodes.items())}}}, GPU on node {gpu_node}", flush=True)
    tmp = tempfile.mkdtemp()
    vl = os.path.join(tmp, "vl.json")
    json.dump({"ASSEMBLY101-O": {"class_index": [f"c{i}" for i in range(86)]}}, open(vl, "w"))
    cfg = assembly101_cfg(eval="ckpt.pth", video_list_path=vl, eval_output_dir=os.path.join(tmp, "out"), assume_zero_flow=True)
    model = build_model(cfg, "cuda:0")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20, head_gain=8.0).items()})
    model.eval()
    lens = assembly101_eval_lengths(seed=20)[:n_clips]
    g = torch.Generator().manual_seed(5)
    items = []
    for i, T in enumerate(lens):
        tgt = torch.zeros(T, 86)
        tgt[torch.arange(T), (torch.arange(T) // 97 + i) % 86] = 1
        items.append((torch.randn((1, T, 2048), generator=g).clamp_(min=0).half().pin_memory(), torch.zeros(1, 1, 2048).expand(1, T, 2048),
                      tgt[None].pin_memory(), (f"v{i}",), torch.tensor([0]), torch.tensor([T])))
    frames = sum(lens)
    ev = build_eval(cfg)
    log = logging.getLogger("feed")
    ev(model, items, log, "cuda:0")                                     # warm
    ctx = mp.get_context("spawn")
    local = [c for c in nodes.get(gpu_node if gpu_node is not None and gpu_node >= 0 else 0, allowed) if c in allowed] or allowed
    for n_hogs, where in ((0, "-"), (3, "all nodes"), (7, "all nodes"), (7, "the GPU's node"), (15, "all nodes")):
        stop = ctx.Value("i", 0)
        moved = ctx.Array("q", max(1, n_hogs))
        procs = []
        for k in range(n_hogs):
            cpus = local if where == "the GPU's node" else allowed
            # one core per hog, spread
            cpu = [cpus[(k * max(1, len(cpus) // max(1, n_hogs)) + 1) % len(cpus)]]
            p = ctx.Process(target=hog, args=(cpu, stop, moved, k))
            p.start()
            procs.append(p)
        time.sleep(1.5 if n_hogs else 0.0)
        best = 0.0
        t_bg0 = time.perf_counter()
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev(model, items, log, "cuda:0")
            torch.cuda.synchronize()
            best = max(best, frames / (time.perf_counter() - t0))
        bg_dt = time.perf_counter() - t_bg0 + (1.5 if n_hogs else 0.0)
        stop.value = 1
        for p in procs:
            p.join(30)
        bg = sum(moved[k] for k in range(n_hogs)) / bg_dt / 1e9 if n_hogs else 0.0
        print(f"{n_hogs:2d} background feeders ({where}): eval end to end {best / 1e6:.2f} M frames/s = {best * 4096 / 1e9:.1f} GB/s of 16-bit features over the link; "
              f"background DRAM traffic {bg:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
