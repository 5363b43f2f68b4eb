"""GPU: the CLI mirror of step_recognition/main.py end to end on a small feature tree on disk: train branch (TRAINER["OAD"],
OadLoss, FusedAdamW, one epoch, per-epoch eval, best-mAP checkpoint with the reference's state_dict keys) and the --eval branch
loading that checkpoint (main.py:42-57, 59-115)."""
import glob
import json
import os

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from prego_amd import weights as W           # noqa: E402


def _tree(root):
    lens = {"vidA": 300, "vidB": 157, "vidC": 211}
    for sub in ("target_perframe", "rgb_anet_resnet50"):
        os.makedirs(os.path.join(root, sub))
    for vid, T in lens.items():
        tgt = np.zeros((T, 12), np.float32)
        tgt[np.arange(T), (np.arange(T) // 29) % 12] = 1.0
        np.save(os.path.join(root, "rgb_anet_resnet50", vid + ".npy"), W.tsn_features((T, 2048), 20, f"cli.rgb.{vid}"))
        np.save(os.path.join(root, "target_perframe", vid + ".npy"), tgt)
    vl = os.path.join(root, "video_list.json")
    json.dump({"EPIC-TENT-O": {"train_session_set": ["vidA", "vidB"], "test_session_set": ["vidC", "vidB"],
                               "class_index": [f"c{i}" for i in range(12)]}}, open(vl, "w"))
    return vl


@pytest.mark.parametrize("amp", [False, True])
def test_main_train_then_eval(tmp_path, monkeypatch, amp):
    from prego_amd import main as M
    from prego_amd.config import epic_tent_cfg
    vl = _tree(str(tmp_path / "data"))
    cfg = epic_tent_cfg(root_path=str(tmp_path / "data"), video_list_path=vl, output_path=str(tmp_path / "out"), num_epoch=1,
                        num_workers=0, batch_size=16)
    for k in ("eval", "amp", "tensorboard", "lr_scheduler", "no_rgb", "no_flow", "config"):
        cfg.pop(k, None)                              # argparse supplies these (main.py:16-24)
    ypath = tmp_path / "cfg.yaml"
    yaml.safe_dump(cfg, open(ypath, "w"))
    monkeypatch.chdir(tmp_path)
    best = M.main(["--config", str(ypath)] + (["--amp"] if amp else []))
    assert 0.0 <= best <= 1.0
    ck = glob.glob(str(tmp_path / "out" / "*" / "ckpts" / "best_*.pth"))
    assert len(ck) == 1, ck
    sd = torch.load(ck[0], map_location="cpu")
    assert set(sd) == {"gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0", "layer1.0.weight", "layer1.0.bias",
                       "layer1.1.weight", "layer1.1.bias", "f_classification.0.weight", "f_classification.0.bias"}
    assert all(v.dtype == torch.float32 for v in sd.values())
    mAP = M.main(["--config", str(ypath), "--eval", ck[0]])
    assert abs(mAP - best) < 1e-6                     # the checkpoint reproduces the epoch's eval
    js = json.load(open(tmp_path / "output_miniRoad" / "output_miniROAD.json"))
    assert set(js) == {"vidC", "vidB"} and len(js["vidC"]["pred"]) == 211
