import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
from prego_amd.workloads import assembly101_eval_lengths
import prego_amd.model  # noqa
cfg = assembly101_cfg()
m = build_model(cfg, "cuda:0"); m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()}); m.eval()
eng = m.engine()
lens = assembly101_eval_lengths(seed=20)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
rgb = [torch.randn((T, 2048), device="cuda", generator=gen).clamp_(min=0) for T in lens]
flow = [torch.randn((T, 2048), device="cuda", generator=gen).clamp_(min=0) for T in lens]
def run(head, n=8):
    for _ in range(3): eng.forward_ragged(rgb, flow, softmax=True, want_out=head, want_argmax=head)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.forward_ragged(rgb, flow, softmax=True, want_out=head, want_argmax=head)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for i in range(3):
    print("with head %.2f ms, without %.2f ms" % (run(True), run(False)))
