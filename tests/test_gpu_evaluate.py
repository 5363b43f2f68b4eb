"""GPU: the EVAL["OAD"] loop end to end against the fixture produced by the reference's own Evaluate
(trainer/eval.py) on the same synthetic videos: identical JSON schema, identical gt, pred identical wherever the
reference's top-1/top-2 margin allows it at bf16 tolerance, mAP within tolerance."""
import json
import logging
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O                         # noqa: E402  (checker only)
from prego_amd import weights as W                        # noqa: E402
from prego_amd.config import epic_tent_cfg                # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


class _Loader:
    """yields what the reference's test DataLoader yields: batch dim 1, vid as a 1-tuple"""
    def __init__(self, lens, C, seed, zero_flow_tensor=True):
        self.items = []
        for i, T in enumerate(lens):
            rgb = W.tsn_features((T, 2048), seed, f"g7.rgb.{i}")
            seg = (np.arange(T) // 37) % C
            tgt = np.zeros((T, C), np.float32)
            tgt[np.arange(T), seg] = 1.0
            self.items.append((torch.from_numpy(rgb)[None], torch.zeros(1, T, 2048), torch.from_numpy(tgt)[None],
                               (f"synth_video_{i}",), torch.tensor([0]), torch.tensor([T])))

    def __iter__(self):
        return iter(self.items)


@pytest.mark.parametrize("dtype,assume_zero", [("fp16", False), ("bf16", False), ("bf16", True), ("fp32", False), ("fp16x2", False)])
def test_evaluate_matches_reference_fixture(tmp_path, dtype, assume_zero):
    from prego_amd.registry import build_model, build_eval
    import prego_amd.model, prego_amd.evaluate  # noqa: F401
    g = json.load(open(os.path.join(G, "g7_evaluate.json")))
    vl = os.path.join(tmp_path, "video_list.json")
    json.dump({"EPIC-TENT-O": {"class_index": [f"c{i}" for i in range(12)]}}, open(vl, "w"))
    cfg = epic_tent_cfg(eval="dummy.pth", video_list_path=vl, compute_dtype=dtype, assume_zero_flow=assume_zero,
                        eval_output_dir=str(tmp_path / "output_miniRoad"))
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    model = build_model(cfg, "cuda:0")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    ev = build_eval(cfg)
    mAP = ev(model, _Loader(g["lens"], 12, 20), logging.getLogger("t"), "cuda:0")
    js = json.load(open(tmp_path / "output_miniRoad" / "output_miniROAD.json"))
    assert set(js.keys()) == set(g["output"].keys())
    tol = {"bf16": 1e-2, "fp16": 3e-3, "fp32": 1e-3, "fp16x2": 1e-4}[dtype]
    exact = dtype in ("fp32", "fp16x2")        # the fp32-class modes reproduce the reference's pred lists entry for entry
    total_mism = 0
    for i, T in enumerate(g["lens"]):
        vid = f"synth_video_{i}"
        assert js[vid]["gt"] == g["output"][vid]["gt"]
        probs = O.miniroad_forward(sd, W.tsn_features((T, 2048), 20, f"g7.rgb.{i}")[None], None)["logits"][0]
        srt = np.sort(probs, 1)
        safe = (srt[:, -1] - srt[:, -2]) > 2 * tol
        mism = np.array(js[vid]["pred"]) != np.array(g["output"][vid]["pred"])
        assert not np.any(mism & safe)
        assert not (exact and mism.any()), (vid, int(mism.sum()))
        total_mism += int(mism.sum())
    assert abs(mAP - g["mAP"]) < 5e-3
    assert ev.last_fps and ev.last_fps > 0
    print(f"evaluate {dtype}: mAP {mAP:.5f} (ref {g['mAP']:.5f}), argmax mismatches {total_mism} of {sum(g['lens'])}")


@pytest.mark.parametrize("dtype,with_flow", [("fp16", False), ("fp16", True), ("fp16x2", False)])
def test_link_fed_eval_equals_the_copy_then_forward_eval(tmp_path, dtype, with_flow):
    """Pinned host features (a DataLoader with pin_memory=True) take the link-fed path: the features are copied in pieces in the order the
    packed pipeline needs them, under ONE forward whose packing stream waits on feed events (prego_miniroad_plan_starts /
    set_feed_events; a slot schedule costed for a link-bound feed).  A clip's result does not depend on the packing, so the output file and
    the mAP must equal, entry for entry and bit for bit, those of the copy-everything-then-forward path (cfg eval_link_fed = False)."""
    from prego_amd.registry import build_model, build_eval
    import prego_amd.model, prego_amd.evaluate  # noqa: F401
    vl = os.path.join(tmp_path, "video_list.json")
    json.dump({"EPIC-TENT-O": {"class_index": [f"c{i}" for i in range(12)]}}, open(vl, "w"))
    lens = [5000, 1, 2500, 777, 3100, 64, 1025, 4000, 19, 2048, 300, 1500]           # ragged, incl. one-frame and piece-edge lengths
    items = []
    for i, T in enumerate(lens):
        rgb = torch.from_numpy(W.tsn_features((T, 2048), 21, f"lf.rgb.{i}"))[None].pin_memory()
        flow = torch.from_numpy(W.tsn_features((T, 2048), 21, f"lf.flow.{i}"))[None].pin_memory() if with_flow else \
            torch.zeros(1, 1, 2048).expand(1, T, 2048)
        tgt = torch.zeros(1, T, 12)
        tgt[0, torch.arange(T), (torch.arange(T) // 41 + i) % 12] = 1
        items.append((rgb, flow, tgt.pin_memory(), (f"v{i}",), torch.tensor([0]), torch.tensor([T])))
    res = {}
    for fed in (True, False):
        out_dir = tmp_path / f"out_{int(fed)}"
        cfg = epic_tent_cfg(eval="dummy.pth", video_list_path=vl, compute_dtype=dtype, eval_output_dir=str(out_dir), eval_link_fed=fed)
        model = build_model(cfg, "cuda:0")
        model.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20, head_gain=8.0).items()})
        ev = build_eval(cfg)
        taken = []
        orig = ev._enqueue_link_fed
        ev._enqueue_link_fed = lambda *a, **k: (taken.append(1), orig(*a, **k))[1]
        mAP = ev(model, items, logging.getLogger("t"), "cuda:0")
        model.check()
        assert bool(taken) == fed                                     # the path under test really ran (and only when asked for)
        res[fed] = (mAP, json.load(open(out_dir / "output_miniROAD.json")))
    assert res[True][1] == res[False][1]
    assert res[True][0] == res[False][0]
    assert all(len(res[True][1][f"v{i}"]["pred"]) == T for i, T in enumerate(lens))


def test_one_hot_targets_travel_as_class_ids(tmp_path):
    """The loader's one-hot target rows are reduced to one class id per frame on the host (prego_onehot_labels) and the metric takes the
    ids (prego_perframe_ap_labels): output file and mAP bit for bit those of the matrix path (cfg eval_label_targets = False) - also when
    one video is NOT one-hot (two positives in some rows, an all-zero row: that video's rows travel as they are and the whole set is
    scored through the matrix entry point)."""
    from prego_amd.registry import build_model, build_eval
    import prego_amd.model, prego_amd.evaluate  # noqa: F401
    vl = os.path.join(tmp_path, "video_list.json")
    json.dump({"EPIC-TENT-O": {"class_index": [f"c{i}" for i in range(12)]}}, open(vl, "w"))
    lens = [900, 1, 2500, 333, 64]
    for multi in (False, True):
        items = []
        for i, T in enumerate(lens):
            rgb = torch.from_numpy(W.tsn_features((T, 2048), 21, f"oh.rgb.{i}"))[None].pin_memory()
            tgt = torch.zeros(1, T, 12)
            tgt[0, torch.arange(T), (torch.arange(T) // 41 + i) % 12] = 1
            if multi and i == 2:
                tgt[0, 100:200, 7] = 1                                   # a second positive
                tgt[0, 300] = 0                                          # a row without any
            items.append((rgb, torch.zeros(1, 1, 2048).expand(1, T, 2048), tgt.pin_memory(), (f"v{i}",), torch.tensor([0]), torch.tensor([T])))
        res = {}
        for by_id in (True, False):
            out_dir = tmp_path / f"out_{int(multi)}_{int(by_id)}"
            cfg = epic_tent_cfg(eval="dummy.pth", video_list_path=vl, compute_dtype="fp16", eval_output_dir=str(out_dir), eval_label_targets=by_id)
            model = build_model(cfg, "cuda:0")
            model.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20, head_gain=8.0).items()})
            ev = build_eval(cfg)
            seen = []
            orig = ev._cat_targets
            ev._cat_targets = lambda ts, matrix=False: (seen.append([t.dim() for t in ts]), orig(ts, matrix))[1]
            mAP = ev(model, items, logging.getLogger("t"), "cuda:0")
            assert seen and seen[0] == ([1 if (by_id and not (multi and i == 2)) else 2 for i in range(len(lens))])
            res[by_id] = (mAP, json.load(open(out_dir / "output_miniROAD.json")))
        assert res[True][1] == res[False][1]
        assert res[True][0] == res[False][0]
        gt2 = np.argmax(items[2][2][0].numpy(), 1)
        assert res[True][1]["v2"]["gt"] == gt2.tolist()


def test_device_average_precision_by_class_id_equals_the_matrix_entry():
    from prego_amd.metrics import perframe_average_precision_device
    rng = np.random.default_rng(5)
    n, C = 70_000, 23
    pr = torch.from_numpy(np.round(rng.random((n, C)), 3).astype(np.float32)).cuda()
    lab = torch.from_numpy(rng.integers(0, C + 2, n).astype(np.int32)).cuda()          # ids C, C + 1: frames without a positive
    gt = torch.zeros(n, C, device="cuda")
    ok = lab < C
    gt[torch.arange(n, device="cuda")[ok], lab[ok].long()] = 1
    names = [f"c{i}" for i in range(C)]
    a = perframe_average_precision_device(pr, gt, names, raw=True)
    b = perframe_average_precision_device(pr, lab, names, raw=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("n,C", [(1, 3), (63, 5), (4097, 12), (150_000, 86), (9_001, 130)])      # 130 classes: two column blocks in ap_extract
def test_device_average_precision_kernel_vs_sklearn(n, C):
    """prego_perframe_ap (csrc/metrics.hip: the positives of every class sorted, every score counted against them) against sklearn's
    average_precision_score and the host implementation: random scores, heavy ties, a constant column, negative scores (raw logits),
    -0.0 vs +0.0, a class without positives, tile edges (n = 1, 63, 4097); multi-label targets with more positives than the count
    kernel's LDS table holds (a class that is positive on half of the frames, one on all of them: the two-level search)."""
    from sklearn.metrics import average_precision_score
    from prego_amd.metrics import average_precision_columns, perframe_average_precision, perframe_average_precision_device
    rng = np.random.default_rng(n + C)
    pr = rng.random((n, C)).astype(np.float32)
    if C > 2:
        pr[:, 2] = np.round(pr[:, 2], 1)                 # heavy ties
    if C > 4:
        pr[:, 4] = 0.5                                   # one threshold
        pr[:, 3] = rng.standard_normal(n).astype(np.float32) * 30          # negative scores too
    if C > 5:
        pr[:, 5] = np.where(rng.random(n) < 0.5, 0.0, -0.0).astype(np.float32)     # signed zeros tie
    gt = np.zeros((n, C), np.float32)
    gt[np.arange(n), rng.integers(0, C, n)] = 1
    if C > 6:
        gt[:, 6] = 0                                     # class without positives
    if C > 9:
        gt[:, 7] = rng.random(n) < 0.5                   # multi-label: 75 000 positives of 150 000 frames
        gt[:, 8] = 1                                     # every frame positive
        pr[:, 9] = np.round(pr[:, 9], 2); gt[:, 9] = rng.random(n) < 0.3      # many positives AND heavy ties
    names = [f"c{i}" for i in range(C)]
    dev = perframe_average_precision_device(torch.from_numpy(pr).cuda(), torch.from_numpy(gt).cuda(), names)
    host = perframe_average_precision(pr, gt, names)
    assert list(dev["per_class_AP"]) == list(host["per_class_AP"])
    for c in range(1, C):
        if gt[:, c].any():
            want = average_precision_score(gt[:, c], pr[:, c])
            assert abs(dev["per_class_AP"][names[c]] - want) < 1e-12, (c, dev["per_class_AP"][names[c]], want)
    if dev["per_class_AP"]:
        assert abs(dev["mean_AP"] - host["mean_AP"]) < 1e-12
    assert dev["num"] == host["num"] or n > 4000          # the "pred:" figure is int(sum of scores): fp64 device sum vs numpy's
    ap = average_precision_columns(pr, gt != 0)
    assert np.isnan(ap[6]) if C > 6 else True


def test_device_average_precision_rejects_host_tensors():
    from prego_amd._lib import PregoError
    from prego_amd.metrics import perframe_average_precision_device
    with pytest.raises(PregoError):
        perframe_average_precision_device(torch.zeros(4, 3), torch.zeros(4, 3), ["a", "b", "c"])


def test_evaluate_runs_the_transformer_entry_per_frame(tmp_path):
    """`model: 'Transformer'` + --eval: EVAL["OAD"] drives ViTEnc through its sliding-window runner (one logit row per frame =
    the window ending there), writes the reference's JSON schema, and scores the raw logits with the device AP kernel."""
    from prego_amd.metrics import perframe_average_precision
    from prego_amd.registry import build_model, build_eval
    import prego_amd.transformer, prego_amd.evaluate  # noqa: F401
    vl = os.path.join(tmp_path, "video_list.json")
    json.dump({"EPIC-TENT-O": {"class_index": [f"c{i}" for i in range(12)]}}, open(vl, "w"))
    cfg = epic_tent_cfg(model="Transformer", window_size=32, patch_dim=1, num_heads=8, attn_dropout_rate=0.0, dropout=0.0,
                        eval="dummy.pth", video_list_path=vl, eval_output_dir=str(tmp_path / "output_miniRoad"))
    model = build_model(cfg, "cuda:0")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.vit_state_dict(cfg, 20).items()})
    lens = [90, 33, 150]
    loader = _Loader(lens, 12, 20)
    ev = build_eval(cfg)
    mAP = ev(model, loader, logging.getLogger("t"), "cuda:0")
    js = json.load(open(tmp_path / "output_miniRoad" / "output_miniROAD.json"))
    scores, gts = [], []
    for i, item in enumerate(loader.items):
        logits, arg = model.forward_frames(item[0][0].cuda(), item[1][0].cuda())
        assert js[f"synth_video_{i}"]["pred"] == arg.cpu().numpy().tolist()
        assert js[f"synth_video_{i}"]["gt"] == item[2][0].numpy().argmax(1).tolist()
        scores.append(logits.cpu().numpy())
        gts.append(item[2][0].numpy())
    want = perframe_average_precision(np.concatenate(scores), np.concatenate(gts), [f"c{i}" for i in range(12)])["mean_AP"]
    assert abs(mAP - want) < 1e-12


def test_format_ids_text_is_the_json_of_the_ids():
    """prego_format_ids: four bytes "%3d," per id in memory order; ids outside 0..999 set the flag (the caller then formats on the host)"""
    import ctypes as C
    from prego_amd import _lib
    lib = _lib.load()
    ids = torch.tensor([0, 7, 10, 85, 99, 100, 101, 999, 512, 3], dtype=torch.int32, device="cuda")
    text = torch.empty(ids.numel(), dtype=torch.int32, device="cuda")
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.prego_format_ids(C.c_void_p(ids.data_ptr()), ids.numel(), C.c_void_p(text.data_ptr()), C.c_void_p(bad.data_ptr()), s) == 0
    raw = text.cpu().numpy().tobytes()
    assert raw == b"".join(b"%3d," % int(v) for v in ids.tolist()) and int(bad.item()) == 0
    assert json.loads(b"[" + raw[:-1] + b"]") == ids.tolist()
    ids[4] = 1000
    assert lib.prego_format_ids(C.c_void_p(ids.data_ptr()), ids.numel(), C.c_void_p(text.data_ptr()), C.c_void_p(bad.data_ptr()), s) == 0
    assert int(bad.item()) == 1
    ids[4] = -1
    bad.zero_()
    assert lib.prego_format_ids(C.c_void_p(ids.data_ptr()), ids.numel(), C.c_void_p(text.data_ptr()), C.c_void_p(bad.data_ptr()), s) == 0
    assert int(bad.item()) == 1
    assert lib.prego_format_ids(None, 4, C.c_void_p(text.data_ptr()), None, s) != 0
