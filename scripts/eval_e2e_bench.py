"""End-to-end `Evaluate` (EVAL registry 'OAD': batched forward + argmax on device + output JSON + per-frame mAP) on a synthetic
Assembly101-O-shaped loader held in host memory: what `main.py --eval` costs around the frames/s path.
usage: python scripts/eval_e2e_bench.py [n_clips] [len_scale]"""
import json, logging, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model, build_eval
from prego_amd.workloads import assembly101_eval_lengths
import prego_amd.model, prego_amd.evaluate  # noqa: F401

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 60
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tmp = tempfile.mkdtemp()
vl = os.path.join(tmp, "vl.json")
json.dump({"ASSEMBLY101-O": {"class_index": [f"c{i}" for i in range(86)]}}, open(vl, "w"))
extra = {}
if os.environ.get("E2E_SPLIT"):
    extra["eval_split_fraction"] = float(os.environ["E2E_SPLIT"])
if os.environ.get("E2E_FRAMES_PER_BATCH"):
    extra["eval_frames_per_batch"] = int(os.environ["E2E_FRAMES_PER_BATCH"])
if os.environ.get("E2E_PIECE"):
    extra["eval_piece_frames"] = int(os.environ["E2E_PIECE"])
half = os.environ.get("E2E_FEATURE_DTYPE") == "fp16"       # the feeder's 16-bit features (cfg['feature_dtype'], prego_amd/data.py)
cfg = assembly101_cfg(eval="ckpt.pth", video_list_path=vl, eval_output_dir=os.path.join(tmp, "out"), assume_zero_flow=True, **extra)
sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
model = build_model(cfg, "cuda:0"); model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); model.eval()
lens = [max(8, int(l * scale)) for l in assembly101_eval_lengths(seed=20)[:n_clips]]
g = torch.Generator().manual_seed(5)
items = []
for i, T in enumerate(lens):
    tgt = torch.zeros(T, 86); tgt[torch.arange(T), (torch.arange(T) // 97 + i) % 86] = 1
    items.append(((torch.randn((1, T, 2048), generator=g).clamp_(min=0).half() if half else torch.randn((1, T, 2048), generator=g).clamp_(min=0)).pin_memory(), torch.zeros(1, 1, 2048).expand(1, T, 2048), tgt[None].pin_memory(), (f"v{i}",),
                  torch.tensor([0]), torch.tensor([T])))
frames = sum(lens)
ev = build_eval(cfg)
log = logging.getLogger("e2e")
for rep in range(int(os.environ.get("E2E_REPS", "2"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mAP = ev(model, items, log, "cuda:0")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"Evaluate end to end: {n_clips} videos, {frames} frames: {dt:.2f} s = {frames/dt/1e6:.2f} M frames/s (H2D of pinned features + forward + "
          f"argmax + JSON + device mAP), mAP {mAP:.4f}")
    if os.environ.get("E2E_PHASES"):
        prev = 0.0
        print("   phases (ms since start / delta): " + "  ".join(f"{n} {t*1e3:.1f}/{(t-prev)*1e3:.1f}" for (n, t), prev in zip(ev.phase_log, [0.0] + [t for _, t in ev.phase_log[:-1]])))
