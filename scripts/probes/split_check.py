"""Split pass (PREGO_SPLIT_PASS=R) against the chunked pass of the SAME handle: bit-identical probabilities / argmax, and time per pass.
The first forward of a handle is always the chunked pass (it establishes the verified placement); later ones split when eligible.
usage: PREGO_SPLIT_PASS=3 python scripts/probes/split_check.py [n_clips] [min_T] [max_T] [flow 0/1] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.model  # noqa: F401

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
use_flow = (int(sys.argv[4]) if len(sys.argv) > 4 else 0) != 0
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dt = os.environ.get("SPLIT_DTYPE", "fp16")
cfg = assembly101_cfg(compute_dtype=dt)
sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
model = build_model(cfg, "cuda:0"); model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); model.eval()
eng = model.engine()
g = torch.Generator().manual_seed(7)
lens = [int(x) for x in torch.randint(lo, hi + 1, (n,), generator=g)]
dev = torch.device("cuda:0")
rgb = [torch.randn((T, 2048), generator=g).clamp_(min=0).to(dev) for T in lens]
flow = [torch.randn((T, 2048), generator=g).clamp_(min=0).to(dev) for T in lens] if use_flow else None
frames = sum(lens)
print(f"{n} clips, {frames} frames, flow={use_flow}, dtype {dt}, PREGO_SPLIT_PASS={os.environ.get('PREGO_SPLIT_PASS')}", flush=True)


def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o, a, _ = eng.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)
    eng.check()
    torch.cuda.synchronize()
    return o, a, time.perf_counter() - t0


o0, a0, t = run()
info = eng.pass_info()
print(f"call 0: mode {info['mode']} steps {info['steps']} slots {info['slots']}  {t*1e3:.2f} ms", flush=True)
ref_o, ref_a = torch.cat(o0), torch.cat(a0)
eng.timing_enable(True)
import ctypes as C
if (os.environ.get("PREGO_GRU_STAMPS") or os.environ.get("PREGO_SPLIT_STATS")) and hasattr(eng.lib, "prego_miniroad_debug_stamps"):
    _o = (C.c_uint64 * 8)(); eng.lib.prego_miniroad_debug_stamps(eng.h, _o)          # drop call 0's stamps: the report below is about calls 1 .. reps
    print("call 0 stamps:", [int(x) for x in _o], flush=True)
for k in range(1, reps + 1):
    o, a, t = run()
    info = eng.pass_info()
    oo, aa = torch.cat(o), torch.cat(a)
    same = bool(torch.equal(oo, ref_o)) and bool(torch.equal(aa, ref_a))
    dmax = float((oo - ref_o).abs().max())
    nbad = int((oo != ref_o).any(1).sum())
    print(f"call {k}: mode {info['mode']} steps {info['steps']} slots {info['slots']}  {t*1e3:.2f} ms = {frames/t/1e6:.2f} M frames/s   "
          f"bit-identical to call 0: {same} (max |d| {dmax:.3e}, rows differing {nbad}, argmax mismatches {int((aa != ref_a).sum())})", flush=True)
    if nbad:
        bad = (oo != ref_o).any(1).nonzero().flatten()
        print("   first differing frames:", bad[:8].tolist(), " last:", bad[-4:].tolist(), flush=True)
tm = eng.timing_read()
if reps:
    print(f"per pass: recurrence launch(es) {tm['gru_ms']/reps:.2f} ms ({tm['gru_launches']//reps} launches), feed-forward launch / packs {tm['pack_ms']/reps:.2f} ms, "
          f"static GEMM launches {tm['gemm_ms']/reps:.2f} ms", flush=True)
if os.environ.get("PREGO_SPLIT_STATS") and hasattr(eng.lib, "prego_miniroad_debug_stamps"):
    out = (C.c_uint64 * 8)(); eng.lib.prego_miniroad_debug_stamps(eng.h, out)
    v = [x / 1e5 / max(1, reps) for x in out]          # 10 ns ticks -> ms, per pass (summed over the feed-forward workgroups)
    print(f"feed-forward workgroup-ms per pass: pack {v[0]:.1f}  layer1 {v[1]:.1f}  ln {v[2]:.1f}  w_ih {v[3]:.1f}  waits {v[4]:.1f}  tickets {v[5]:.1f}  "
          f"lifetime {v[7]:.1f}  shader clock of the feed-forward XCDs {out[6] / max(1, out[7]) * 100:.0f} MHz", flush=True)
if os.environ.get("PREGO_GRU_STAMPS") and not os.environ.get("PREGO_SPLIT_STATS") and hasattr(eng.lib, "prego_miniroad_debug_stamps"):
    out = (C.c_uint64 * 8)(); eng.lib.prego_miniroad_debug_stamps(eng.h, out)
    steps = max(1, out[6])
    names = ["rest of gather + mfma", "step top -> first segment valid", "reduce+barrier", "gates+publish", "outputs"]
    tot = sum(out[i] for i in range(5))
    print(f"recurrence workgroup 0 / wave 0 (shader cycles per step, {steps} steps): total {tot/steps:.1f}: " +
          ", ".join(f"{names[i]} {out[i]/steps:.1f}" for i in range(5)) + f"; retry rounds/step {out[5]/steps:.2f}; "
          f"loop {out[7] / 100 / steps:.3f} us per step of real time -> shader clock of that XCD {tot / max(1, out[7]) * 100:.0f} MHz", flush=True)
