#!/usr/bin/env python3
"""Where does a split pass differ from the chunked pass?  (debugging aid for changes to csrc/ff_pass.hip)  Same inputs as
tests/test_gpu_split.py::test_split_pass_equals_chunked_pass_bit_for_bit; prints the size and the frame pattern of the differences."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from prego_amd import weights as W  # noqa: E402
from prego_amd.config import assembly101_cfg  # noqa: E402
from prego_amd.registry import build_model  # noqa: E402
import prego_amd.model  # noqa: F401,E402


def engine(split):
    cfg = assembly101_cfg(compute_dtype="fp16")
    os.environ["PREGO_SPLIT_PASS"] = split
    m = build_model(cfg, "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20, head_gain=8.0).items()})
    m.eval()
    e = m.engine()
    return m, e


g = torch.Generator().manual_seed(11)
lens = [int(x) for x in torch.randint(3000, 6201, (64,), generator=g)]
rgb = []
for i, T in enumerate(lens):
    gg = torch.Generator(device="cuda"); gg.manual_seed(100 + i)
    rgb.append(torch.randn((T, 2048), device="cuda", generator=gg).clamp_(min=0))
m0, e0 = engine("0")
m3, e3 = engine("3")
ref, _, _ = e0.forward_ragged(rgb, None, softmax=False, want_out=True)
for k in range(2):
    out, _, _ = e3.forward_ragged(rgb, None, softmax=False, want_out=True)
    print("pass info", e3.pass_info())
tot = 0
for i in range(len(lens)):
    d = (out[i] - ref[i]).abs().amax(1).cpu().numpy()
    bad = np.nonzero(d > 0)[0]
    tot += len(bad)
    if i < 6 or len(bad) == 0:
        print(f"clip {i} T {lens[i]}: {len(bad)} frames differ, max {d.max():.3e}, first {bad[:12].tolist()}, first-diff size {d[bad[0]] if len(bad) else 0:.3e}")
print("frames differing:", tot, "of", sum(lens))
