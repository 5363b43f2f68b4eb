#!/usr/bin/env python3
"""Replay of the two launches of a split pass ONE AT A TIME (include/prego_amd_debug.h: prego_debug_split_fault modes 3 / 4), so that
rocprofv3 --pmc - which serialises kernel dispatches, under which the pair can never run - sees each of them at full length:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- python3 scripts/split_replay.py [--clips N --len-scale S --reps K]

Call 1 is an ordinary chunked pass (it establishes the verified workgroup placement and leaves finite 16-bit rows where the GI ring will
be), then K x (feed-forward launch alone, recurrence launch alone) on the bench workload of BASELINE configs[1].  Outputs are
meaningless by construction; traffic and instruction streams are those of the real pass.  Runs on libprego_amd_debug.so."""
import argparse
import os
import sys
import time

os.environ["PREGO_AMD_DEBUG_LIB"] = "1"
os.environ.setdefault("PREGO_SPLIT_PASS", "3")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from prego_amd import _lib, weights as W  # noqa: E402
from prego_amd.config import assembly101_cfg  # noqa: E402
from prego_amd.registry import build_model  # noqa: E402
from prego_amd.workloads import assembly101_eval_lengths  # noqa: E402
import prego_amd.model  # noqa: F401,E402

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=0)
ap.add_argument("--len-scale", type=float, default=1.0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--no-flow", action="store_true")
args = ap.parse_args()

dev = torch.device("cuda", 0)
cfg = assembly101_cfg(compute_dtype=args.dtype)
sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
model = build_model(cfg, dev)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model.eval()
eng = model.engine()
lib = _lib.load()
lens = assembly101_eval_lengths(seed=20)
if args.clips:
    lens = lens[: args.clips]
lens = [max(1, int(l * args.len_scale)) for l in lens]
gen = torch.Generator(device=dev)
gen.manual_seed(1234)
rgb = [torch.randn((T, 2048), device=dev, generator=gen).clamp_(min=0) for T in lens]
flow = None if args.no_flow else [torch.randn((T, 2048), device=dev, generator=gen).clamp_(min=0) for T in lens]


def call(mode):
    if mode:
        assert lib.prego_debug_split_fault(eng.h, mode) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)
    torch.cuda.synchronize()
    eng.check()
    return (time.perf_counter() - t0) * 1e3, eng.pass_info()


ms, info = call(0)
print(f"chunked pass (placement): {ms:.1f} ms {info}", flush=True)
for k in range(args.reps):
    ms, info = call(3)
    assert info["mode"] > 0, info
    print(f"feed-forward launch alone: {ms:.1f} ms {info}", flush=True)
    ms, info = call(4)
    assert info["mode"] > 0, info
    print(f"recurrence launch alone: {ms:.1f} ms {info}", flush=True)
print(f"frames {sum(lens)} clips {len(lens)}")
