"""Does replaying the streaming step (three / four launches per frame, prego_miniroad_step) from a captured HIP graph shorten it?
One stream / 16 streams, zero flow / rgb + flow; static input and state buffers; per-call time of 2 000 direct calls against 2 000
graph replays.      python scripts/probes/step_graph_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.model  # noqa: F401

cfg = assembly101_cfg()
m = build_model(cfg, "cuda:0")
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
m.eval()
eng = m.engine()
N = 2000
for n, with_flow in ((1, False), (16, True)):
    rgb = torch.randn(n, 2048, device="cuda").clamp_(min=0)
    flow = torch.randn(n, 2048, device="cuda").clamp_(min=0) if with_flow else None
    h = torch.zeros(n, 1024, device="cuda")
    out = torch.empty(n, 86, device="cuda")
    arg = torch.empty(n, dtype=torch.int32, device="cuda")
    for _ in range(20):
        eng.step(rgb, flow, h, out=out, argmax=arg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        eng.step(rgb, flow, h, out=out, argmax=arg)
    torch.cuda.synchronize()
    direct = (time.perf_counter() - t0) / N * 1e6
    h_ref = h.clone()
    # capture on a side stream
    h.zero_()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.step(rgb, flow, h, out=out, argmax=arg)
        torch.cuda.synchronize()
        h.zero_()
        with torch.cuda.graph(g, stream=s):
            eng.step(rgb, flow, h, out=out, argmax=arg)
    torch.cuda.synchronize()
    for _ in range(20 + N):
        g.replay()
    torch.cuda.synchronize()
    same = torch.equal(h, h_ref)
    h.zero_()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / N * 1e6
    print(f"{n} stream(s), flow {with_flow}: direct {direct:.2f} us per frame, graph replay {graph:.2f} us per frame, state after {20 + N} frames identical: {same}")
