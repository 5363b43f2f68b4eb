"""GPU tests of the rows either side of the hot path:
 f1  feeder -> DataLoader -> EVAL["OAD"] -> output JSON on a feature tree on disk (the tree of fixture G9, regenerated from
     seeds), per-frame predictions against the numpy oracle;
 f2  the 200-frame majority vote on the device (prego_window_vote) against the reference's shipped known-answer pair G8
     (output_miniRoad/output_miniROAD.json -> data/output/aggregated_data.json) and against the host form."""
import gzip
import json
import logging
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O            # noqa: E402  (checker only)
from prego_amd import weights as W           # noqa: E402
from prego_amd.config import epic_tent_cfg   # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


def _tree(root):
    lens = {"vidA": 300, "vidB": 157}
    for sub in ("target_perframe", "rgb_anet_resnet50"):
        os.makedirs(os.path.join(root, sub))
    for vid, T in lens.items():
        tgt = np.zeros((T, 12), np.float32)
        tgt[np.arange(T), (np.arange(T) // 29) % 12] = 1.0
        np.save(os.path.join(root, "rgb_anet_resnet50", vid + ".npy"), W.tsn_features((T, 2048), 20, f"g9.rgb.{vid}"))
        np.save(os.path.join(root, "target_perframe", vid + ".npy"), tgt)
    vl = os.path.join(root, "video_list.json")
    json.dump({"EPIC-TENT-O": {"train_session_set": ["vidA", "vidB"], "test_session_set": ["vidB", "vidA"],
                               "class_index": [f"c{i}" for i in range(12)]}}, open(vl, "w"))
    return vl, lens


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_feeder_to_evaluate_to_json(tmp_path, dtype):
    from prego_amd.aggregate import aggregate, aggregate_device
    from prego_amd.data import build_data_loader
    from prego_amd.registry import build_eval, build_model
    import prego_amd.evaluate, prego_amd.model  # noqa: F401
    vl, lens = _tree(str(tmp_path))
    cfg = epic_tent_cfg(root_path=str(tmp_path), video_list_path=vl, eval="ckpt.pth", num_workers=0, compute_dtype=dtype,
                        assume_zero_flow=True, eval_output_dir=str(tmp_path / "output_miniRoad"))
    g9 = np.load(os.path.join(G, "g9_feeder.npz"))
    loader = build_data_loader(cfg, "test")
    assert [w[0] for w in loader.dataset.inputs] == list(g9["test.vids"])       # the reference's item order
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    model = build_model(cfg, "cuda:0")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ev = build_eval(cfg)
    mAP = ev(model, loader, logging.getLogger("t"), "cuda:0")
    assert 0.0 <= mAP <= 1.0
    js = json.load(open(tmp_path / "output_miniRoad" / "output_miniROAD.json"))
    assert set(js) == set(lens)
    tol = 1e-2 if dtype == "bf16" else 1e-3
    for vid, T in lens.items():
        rgb = W.tsn_features((T, 2048), 20, f"g9.rgb.{vid}")
        probs = O.miniroad_forward(sd, rgb[None], None)["logits"][0]
        srt = np.sort(probs, 1)
        safe = (srt[:, -1] - srt[:, -2]) > 2 * tol
        assert len(js[vid]["pred"]) == T
        assert not np.any((np.array(js[vid]["pred"]) != probs.argmax(1)) & safe)
        assert js[vid]["gt"] == ((np.arange(T) // 29) % 12).tolist()
    # f2 on the same run: device vote over the int32 argmax still in HBM == host aggregate over the JSON
    dev = aggregate_device(ev.last_device_argmax, {k: v["gt"] for k, v in js.items()}, n_classes=12)
    assert dev == aggregate(js)
    assert ev.aggregate_last() == dev                              # the evaluator's own helper passes the config's class count


def test_window_vote_reproduces_reference_known_answer_g8():
    from prego_amd.aggregate import aggregate_device
    with gzip.open(os.path.join(G, "g8_output_miniROAD.json.gz"), "rt") as f:
        data = json.load(f)
    want = json.load(open(os.path.join(G, "g8_aggregated_data.json")))
    preds = {k: torch.tensor(v["pred"], dtype=torch.int32, device="cuda") for k, v in data.items()}
    got = aggregate_device(preds, {k: v["gt"] for k, v in data.items()}, n_classes=12)
    assert got == want                               # 15 videos, 187 959 frames: identical step sequences and change lists


def test_window_vote_ties_and_ragged_tail():
    from prego_amd.aggregate import aggregate, aggregate_device
    rng = np.random.default_rng(5)
    data = {}
    for i, T in enumerate([1, 199, 200, 201, 1000, 4096 + 37]):
        pred = rng.integers(0, 86, T)
        if T >= 400:
            pred[:200] = np.tile([7, 3], 100)        # exact tie: the lower id (3) must win
        data[f"v{i}"] = {"pred": pred.tolist(), "gt": rng.integers(0, 86, T).tolist()}
    preds = {k: torch.tensor(v["pred"], dtype=torch.int32, device="cuda") for k, v in data.items()}
    got = aggregate_device(preds, {k: v["gt"] for k, v in data.items()}, n_classes=86)
    assert got == aggregate(data)
    assert got["v4"]["pred"][0] == 3
