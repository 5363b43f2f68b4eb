"""Race screen of the layer1 overlap: the bench workload through a handle with the overlap and one without
(PREGO_NO_XCD_OVERLAP read at create), N passes each, every output compared bit for bit (probabilities and argmax of all clips)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
from prego_amd.workloads import assembly101_eval_lengths
import prego_amd.model  # noqa: F401

n_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = assembly101_cfg()
sd = {k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20, head_gain=8.0).items()}
lens = assembly101_eval_lengths(seed=20)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
rgb = [torch.randn((T, 2048), device="cuda", generator=gen).clamp_(min=0) for T in lens]
flow = [torch.randn((T, 2048), device="cuda", generator=gen).clamp_(min=0) for T in lens]


def make(no_overlap):
    if no_overlap:
        os.environ["PREGO_NO_XCD_OVERLAP"] = "1"
    else:
        os.environ.pop("PREGO_NO_XCD_OVERLAP", None)
    m = build_model(cfg, "cuda:0"); m.load_state_dict(sd); m.eval()
    return m


ref_m, ov_m = make(True), make(False)
ro, ra, _ = ref_m.engine().forward_ragged(rgb, flow, want_argmax=True)
ref_m.engine().check()
ro = [x.clone() for x in ro]; ra = [x.clone() for x in ra]
bad = 0
copy_src = torch.empty(1 << 28, device="cuda")
for p in range(n_pass):
    if p % 2:                                   # every other pass beside a stream of 1 GiB copies
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(20):
                copy_src.clone()
    o, a, _ = ov_m.engine().forward_ragged(rgb, flow, want_argmax=True)
    ov_m.engine().check()
    torch.cuda.synchronize()
    for i in range(len(lens)):
        if not (torch.equal(o[i], ro[i]) and torch.equal(a[i], ra[i])):
            bad += 1
print(f"overlap race screen: {n_pass} passes x {len(lens)} clips, mismatching clip outputs: {bad}")
assert bad == 0
