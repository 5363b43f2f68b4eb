"""which library call leaves a HIP error code behind?  hipPeekAtLastError after every step of a split-pass sequence"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
from prego_amd.workloads import assembly101_eval_lengths
import prego_amd.model  # noqa
hip = None
for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
    try:
        hip = ctypes.CDLL(name); break
    except OSError:
        pass
hip.hipGetErrorName.restype = ctypes.c_char_p
def peek(tag):
    e = hip.hipPeekAtLastError()
    print(f"{tag}: last error {e} {hip.hipGetErrorName(e).decode()}", flush=True)
cfg = assembly101_cfg(compute_dtype="fp16")
sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
def eng(split):
    if split is None: os.environ.pop("PREGO_SPLIT_PASS", None)
    else: os.environ["PREGO_SPLIT_PASS"] = split
    m = build_model(cfg, "cuda:0"); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.eval()
    e = m.engine(); os.environ.pop("PREGO_SPLIT_PASS", None)
    return m, e
lens = assembly101_eval_lengths(seed=20)
def op(tag):
    try:
        torch.empty(64, device="cuda").fill_(1.0); torch.cuda.synchronize(); print(tag, "ok", flush=True)
    except Exception as e:
        print(tag, "STICKY:", str(e).splitlines()[0], flush=True)
g = torch.Generator(device="cuda"); g.manual_seed(1)
rgb = [torch.randn((T, 2048), device="cuda", generator=g).clamp_(min=0) for T in lens]
flow = [torch.randn((T, 2048), device="cuda", generator=g).clamp_(min=0) for T in lens]
op("inputs")
m_c, e_c = eng("0"); op("create chunked")
b, bb, _ = e_c.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True); op("chunked fwd enq")
e_c.check(); op("chunked check")
m_s, e_s = eng("3"); op("create split")
e_s.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True); op("split fwd1")
for k in range(2):
    a, aa, _ = e_s.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True); op(f"split fwd {k+2} enq")
    e_s.check(); op(f"split check {k+2}"); print(e_s.pass_info())
del a, aa, b, bb; op("del outputs")
del e_s, m_s; import gc; gc.collect(); op("del split engine")
del e_c, m_c; gc.collect(); op("del chunked engine")
