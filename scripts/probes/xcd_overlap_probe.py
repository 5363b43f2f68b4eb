"""Measured version of DESIGN 5c's premise: the recurrence packed onto Gd XCDs, the projection GEMM as a persistent worker on the
OTHER 8 - Gd XCDs, alone and together.

  recurrence  prego_debug_recurrence_only: n_slots = 16 Gd equal slots x n_steps steps, dealt to Gd groups (group := XCD), one launch
  GEMM        prego_debug_gemm_worker: the 256 x 256 ping-pong kernel, 256 persistent workgroups; those on XCDs < Gd leave at once,
              the others claim tiles of C[M, 2048] = A[M, 4096] . W^T from an atomic counter (layer1's shape, M sized to last about
              as long as the recurrence)
  together    the worker is launched FIRST (its workgroups on the recurrence's XCDs find nothing resident yet and leave), the
              recurrence right behind it on a second stream: its live groups take XCDs 0 .. Gd - 1, its other workgroups wait for a
              CU on the worker's XCDs and leave when they get one.

Prints per Gd: recurrence us per step alone / beside the worker, GEMM TFLOP/s on 8 - Gd XCDs alone / beside the recurrence, and
the whole-chip GEMM rate for scale.        python scripts/probes/xcd_overlap_probe.py [steps]
"""
import os
os.environ.setdefault("PREGO_AMD_DEBUG_LIB", "1")      # the prego_debug_* hooks live in libprego_amd_debug.so (include/prego_amd_debug.h)
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from prego_amd import _lib
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.model  # noqa: F401

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _lib.load()
dev = torch.device("cuda:0")
cfg = assembly101_cfg(compute_dtype="bf16")
m = build_model(cfg, dev)
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
m.eval()
eng = m.engine()
H, N, K = 1024, 2048, 4096
Wt = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
bias = torch.randn(N, device=dev)
counter = torch.zeros(1, dtype=torch.int32, device=dev)
s_rec, s_gemm = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def ev():
    return torch.cuda.Event(enable_timing=True)


def run(gd, M, what):
    """returns (recurrence ms, gemm ms); what in {'rec', 'gemm', 'both'}"""
    n_slots = 16 * gd
    rows = n_slots * steps
    gi = (torch.randn(rows, 3 * H, device=dev) * 0.5).to(torch.bfloat16)
    hr = torch.empty(rows, H, dtype=torch.bfloat16, device=dev)
    A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    Cc = torch.empty(M, N, device=dev)
    torch.cuda.synchronize()
    r0, r1, g0, g1 = ev(), ev(), ev(), ev()
    counter.zero_()
    torch.cuda.synchronize()
    if what in ("gemm", "both"):
        with torch.cuda.stream(s_gemm):
            g0.record()
            _lib.check(lib.prego_debug_gemm_worker(C.c_void_p(A.data_ptr()), C.c_void_p(Wt.data_ptr()), C.c_void_p(bias.data_ptr()),
                                                   C.c_void_p(Cc.data_ptr()), M, N, K, gd if what != "gemm_all" else 0,
                                                   C.c_void_p(counter.data_ptr()), 256, C.c_void_p(s_gemm.cuda_stream)))
            g1.record()
    if what in ("rec", "both"):
        with torch.cuda.stream(s_rec):
            r0.record()
            _lib.check(lib.prego_debug_recurrence_only(eng.h, n_slots, steps, gd, C.c_void_p(gi.data_ptr()), C.c_void_p(hr.data_ptr()),
                                                       C.c_void_p(s_rec.cuda_stream)))
            r1.record()
    torch.cuda.synchronize()
    eng.check()
    return (r0.elapsed_time(r1) if what in ("rec", "both") else None, g0.elapsed_time(g1) if what in ("gemm", "both") else None)


def gemm_all(M):
    A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    Cc = torch.empty(M, N, device=dev)
    counter.zero_()
    torch.cuda.synchronize()
    g0, g1 = ev(), ev()
    g0.record()
    _lib.check(lib.prego_debug_gemm_worker(C.c_void_p(A.data_ptr()), C.c_void_p(Wt.data_ptr()), C.c_void_p(bias.data_ptr()),
                                           C.c_void_p(Cc.data_ptr()), M, N, K, 0, C.c_void_p(counter.data_ptr()), 256,
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    g1.record()
    torch.cuda.synchronize()
    return g0.elapsed_time(g1)


out = []
Mfull = 49152 * 4
gemm_all(Mfull)
t_all = min(gemm_all(Mfull) for _ in range(3))
full_tf = 2.0 * Mfull * N * K / t_all / 1e9
print(f"worker on all 8 XCDs: {full_tf:.0f} TFLOP/s (M = {Mfull})")
for gd in (1, 2, 4, 6):
    t_r = min(run(gd, 256, "rec")[0] for _ in range(3))
    # size the GEMM to last about as long as the recurrence on 8 - gd XCDs
    M = int(t_r / t_all * Mfull * (8 - gd) / 8) // 256 * 256
    M = max(M, 2048)
    run(gd, M, "gemm")
    t_g = min(run(gd, M, "gemm")[1] for _ in range(3))
    both = [run(gd, M, "both") for _ in range(3)]
    b_r, b_g = min(b[0] for b in both), min(b[1] for b in both)
    fl = 2.0 * M * N * K
    rec = {"Gd": gd, "steps": steps, "slots": 16 * gd, "gemm_rows": M,
           "recurrence_us_per_step_alone": t_r * 1e3 / steps, "recurrence_us_per_step_beside_gemm": b_r * 1e3 / steps,
           "gemm_tflops_alone_on_other_xcds": fl / t_g / 1e9, "gemm_tflops_beside_recurrence": fl / b_g / 1e9,
           "gemm_share_of_whole_chip_rate_alone": fl / t_g / 1e9 / full_tf, "ideal_share": (8 - gd) / 8,
           "ms": {"recurrence_alone": t_r, "gemm_alone": t_g, "recurrence_together": b_r, "gemm_together": b_g}}
    out.append(rec)
    print(json.dumps(rec))
json.dump({"whole_chip_worker_tflops": full_tf, "runs": out}, open("gpurun_out/xcd_overlap_probe.json", "w"), indent=1)
