"""Where does the end-to-end Evaluate spend its time?  Phases of prego_amd/evaluate.py timed one by one on the 60-video bench set
(pinned fp16 / fp32 features): zero-copy forward (pack pulls over PCIe), H2D + forward, targets H2D, AP kernel, JSON text.
usage: python scripts/probes/eval_phases.py [fp16|fp32]"""
import json, logging, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model, build_eval
from prego_amd.workloads import assembly101_eval_lengths
from prego_amd.metrics import perframe_average_precision_device
import prego_amd.model, prego_amd.evaluate  # noqa: F401

half = (sys.argv[1] if len(sys.argv) > 1 else "fp16") == "fp16"
tmp = tempfile.mkdtemp()
vl = os.path.join(tmp, "vl.json")
json.dump({"ASSEMBLY101-O": {"class_index": [f"c{i}" for i in range(86)]}}, open(vl, "w"))
cfg = assembly101_cfg(eval="ckpt.pth", video_list_path=vl, eval_output_dir=os.path.join(tmp, "out"))
sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
model = build_model(cfg, "cuda:0"); model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); model.eval()
lens = assembly101_eval_lengths(seed=20)[:60]
g = torch.Generator().manual_seed(5)
feats, tgts = [], []
for i, T in enumerate(lens):
    x = torch.randn((T, 2048), generator=g).clamp_(min=0)
    feats.append((x.half() if half else x).pin_memory())
    t = torch.zeros(T, 86); t[torch.arange(T), (torch.arange(T) // 97 + i) % 86] = 1
    tgts.append(t.pin_memory())
frames = sum(lens)
dev = torch.device("cuda:0")


def timed(name, fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{name:48s} {best*1e3:8.1f} ms  = {frames/best/1e6:6.2f} M frames/s", flush=True)
    return r


timed("forward, features on the device already", (lambda d=[f.cuda() for f in feats]: model.forward_clips(d, None)))
timed("forward, zero-copy from pinned host", lambda: model.forward_clips(feats, None))
timed("H2D of all features (60 copies)", lambda: [f.to(dev, non_blocking=True) for f in feats])
timed("H2D of all targets (60 copies)", lambda: [t.to(dev, non_blocking=True) for t in tgts])
probs, args, _ = model.forward_clips(feats, None)
td = [t.to(dev) for t in tgts]
names = [f"c{i}" for i in range(86)]
timed("cat + device AP", lambda: perframe_average_precision_device(torch.cat(probs), torch.cat(td), names, None, "AP"))
timed("gt argmax on device + D2H of ids", lambda: torch.stack([torch.cat(args), torch.cat([torch.argmax(t, 1) for t in td]).int()]).cpu())
ids = torch.stack([torch.cat(args), torch.cat([torch.argmax(t, 1) for t in td]).int()]).cpu().numpy()
ev = build_eval(cfg)
if os.environ.get("EV_PIECE"):
    type(ev).PIECE_FRAMES = int(os.environ["EV_PIECE"])
if os.environ.get("EV_BYTES"):
    type(ev).EVENT_BYTES = int(os.environ["EV_BYTES"]) << 20
out, o = {}, 0
for i, T in enumerate(lens):
    out[f"v{i}"] = {"pred": ids[0, o:o + T], "gt": ids[1, o:o + T]}; o += T
timed("JSON text + write", lambda: open(os.path.join(tmp, "o.json"), "wb").write(ev._json_int_lists(out)))
items = [(f[None], torch.zeros(1, 1, 2048).expand(1, f.shape[0], 2048), t[None], (f"v{i}",), torch.tensor([0]), torch.tensor([f.shape[0]])) for i, (f, t) in enumerate(zip(feats, tgts))]
timed("Evaluate end to end", lambda: ev(model, items, logging.getLogger("p"), "cuda:0"))
