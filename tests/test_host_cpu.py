"""CPU-only tests: the C-ABI library loads and exports every declared symbol, host logic (registries,
config, deterministic generator, feeder windowing, clip sharding, aggregation known answer)."""
import gzip
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_abi_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from prego_amd import _lib
    lib = _lib.load()
    assert lib.prego_abi_version() == 7
    hdr = open(os.path.join(ROOT, "include", "prego_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(prego_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/prego_amd.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared
    # the probe / unit-test entry points are NOT in the product library; the debug library has both sets
    dhdr = open(os.path.join(ROOT, "include", "prego_amd_debug.h")).read()
    ddecl = sorted(set(re.findall(r"\b(prego_[a-z0-9_]+)\s*\(", dhdr)))
    assert ddecl == sorted(_lib.DEBUG_SYMBOLS)
    assert not any("debug" in n for n in declared)
    for name in ddecl:
        assert not hasattr(lib, name), f"{name} is a debug entry point but the product library exports it"
    dbg = _lib.load_debug()
    for name in declared + ddecl:
        assert hasattr(dbg, name), f"{name} missing from libprego_amd_debug.so"


def test_build_rebuilds_when_the_recorded_source_hash_differs_from_the_tree(tmp_path):
    """a prebuilt library that does not belong to this tree must not pass as current: build() compares build_info.json's
    sources_sha256 with the tree and build_info() says so"""
    import json
    from prego_amd import build as B
    B.build()
    info = B.build_info()
    assert info["sources_match_tree"] and "STALE" not in info["build_mode"]
    p = os.path.join(B.LIBDIR, "build_info.json")
    saved = open(p).read()
    try:
        d = json.loads(saved)
        d["sources_sha256"] = "0" * 64
        json.dump(d, open(p, "w"))
        assert not B.build_info()["sources_match_tree"] and "STALE" in B.build_info()["build_mode"]
    finally:
        open(p, "w").write(saved)


def test_product_path_fails_loudly_without_gpu():
    """no CPU fallback: asking for a CPU device is an error, not a silent torch path"""
    from prego_amd._lib import PregoError
    from prego_amd.config import assembly101_cfg
    from prego_amd.registry import build_model
    import prego_amd.model  # noqa: F401
    m = build_model(assembly101_cfg(), "cpu").eval()
    x = torch.zeros(1, 4, 2048)
    with pytest.raises(PregoError):
        m(x, x)


def test_state_dict_keys_match_reference_checkpoint_layout():
    from prego_amd.config import assembly101_cfg
    from prego_amd.registry import build_model
    from prego_amd import weights as W
    import prego_amd.model  # noqa: F401
    cfg = assembly101_cfg()
    m = build_model(cfg, "cpu")
    sd = m.state_dict()
    ref = W.miniroad_state_dict(cfg, 20)
    assert set(sd.keys()) == set(ref.keys())          # gru.*, layer1.*, f_classification.* 
    for k in ref:
        assert tuple(sd[k].shape) == ref[k].shape and sd[k].dtype == torch.float32
    assert sum(p.numel() for p in m.parameters()) == 17926230
    m.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()})


def test_weight_generator_is_deterministic_and_bounded():
    from prego_amd import weights as W
    a = W.uniform((1000,), -0.5, 0.5, 20, "x")
    b = W.uniform((1000,), -0.5, 0.5, 20, "x")
    c = W.uniform((1000,), -0.5, 0.5, 21, "x")
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.min() >= -0.5 and a.max() < 0.5 and abs(a.mean()) < 0.05
    f = W.tsn_features((64, 2048), 20, "f")
    assert f.min() == 0.0 and 0.3 < (f > 0).mean() < 0.7


def test_shard_clips_is_a_balanced_partition():
    from prego_amd.data import shard_clips
    from prego_amd.workloads import assembly101_eval_lengths
    lens = assembly101_eval_lengths()
    assert len(lens) == 182 and min(lens) > 3000 and max(lens) < 35000
    for world in (1, 2, 4, 8):
        parts = [shard_clips(lens, world, r) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(lens)))
        loads = [sum(lens[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(lens)


def test_feeder_windows_and_zero_flow(tmp_path):
    from prego_amd.config import epic_tent_cfg
    from prego_amd.data import StepRecognitionDataset
    root = tmp_path / "Epic-tent-O"
    (root / "target_perframe").mkdir(parents=True)
    (root / "rgb_anet_resnet50").mkdir()
    vl = {"EPIC-TENT-O": {"train_session_set": ["a", "missing"], "test_session_set": ["a"], "class_index": list("abcdefghijkl")}}
    (tmp_path / "vl.json").write_text(json.dumps(vl))
    T = 300
    np.save(root / "rgb_anet_resnet50" / "a.npy", np.random.rand(T, 2048).astype(np.float32))
    tgt = np.zeros((T, 12), np.float32); tgt[np.arange(T), np.arange(T) % 12] = 1
    np.save(root / "target_perframe" / "a.npy", tgt)
    cfg = epic_tent_cfg(root_path=str(root), video_list_path=str(tmp_path / "vl.json"))
    tr = StepRecognitionDataset(cfg, "train")
    assert tr.removed == 1 and tr.vids == ["a"]
    n = T + 127
    assert len(tr) in {len(range(s + 128, n + 1, 4)) for s in range(4)}
    rgb, flow, target, vid, start, end = tr[0]
    assert rgb.shape == (128, 2048) and flow.shape == (128, 2048) and target.shape == (128, 12) and end - start == 128
    assert float(flow.abs().sum()) == 0.0 and flow.stride() == (0, 1)
    if start == 0:
        assert float(rgb[:127].abs().sum()) == 0.0       # front padding rows (dataset.py:53-55)
    te = StepRecognitionDataset(cfg, "test")
    rgb, flow, target, vid, start, end = te[0]
    assert rgb.shape == (T, 2048) and (start, end) == (0, T)


def test_aggregate_reproduces_reference_known_answer():
    from prego_amd.aggregate import aggregate
    with gzip.open(os.path.join(G, "g8_output_miniROAD.json.gz"), "rt") as f:
        data = json.load(f)
    want = json.load(open(os.path.join(G, "g8_aggregated_data.json")))
    assert aggregate(data) == want


def test_metrics_ignore_class_zero_and_match_sklearn():
    from prego_amd.metrics import perframe_average_precision
    from sklearn.metrics import average_precision_score
    rng = np.random.default_rng(0)
    gt = np.zeros((500, 5)); gt[np.arange(500), rng.integers(0, 5, 500)] = 1
    pr = rng.random((500, 5))
    res = perframe_average_precision(pr, gt, ["bg", "a", "b", "c", "d"])
    assert "bg" not in res["per_class_AP"]
    want = np.mean([average_precision_score(gt[:, i], pr[:, i]) for i in range(1, 5)])
    assert abs(res["mean_AP"] - want) < 1e-12


def test_host_average_precision_matches_sklearn_with_ties():
    """the vectorised host AP (metrics.average_precision_columns) against sklearn, incl. heavy ties, a constant column and a
    class without positives"""
    from prego_amd.metrics import average_precision_columns, perframe_average_precision
    from sklearn.metrics import average_precision_score
    rng = np.random.default_rng(3)
    T, Cn = 4000, 9
    pr = rng.random((T, Cn)).astype(np.float32)
    pr[:, 2] = np.round(pr[:, 2], 1)                 # heavy ties
    pr[:, 4] = 0.5                                   # all equal
    gt = np.zeros((T, Cn), np.float32)
    gt[np.arange(T), rng.integers(0, Cn, T)] = 1
    gt[:, 6] = 0                                     # class without positives
    ap = average_precision_columns(pr, gt != 0)
    for c in range(Cn):
        if c == 6:
            assert np.isnan(ap[c])
        else:
            assert abs(ap[c] - average_precision_score(gt[:, c], pr[:, c])) < 1e-12, c
    names = [f"c{i}" for i in range(Cn)]
    res = perframe_average_precision(pr, gt, names)
    assert "c0" not in res["per_class_AP"] and "c6" not in res["per_class_AP"] and len(res["per_class_AP"]) == Cn - 2
    with pytest.raises(RuntimeError):
        perframe_average_precision(pr, gt, names, metrics="mAP@k")


def test_calibrated_ap_matches_the_reference_formula_and_empty_sets_report_nan():
    """metric 'cAP' (utils/metrics.py:10-22: TVSeries' calibrated AP) restated for all classes at once, against the formula
    evaluated class by class on tie-free scores; an empty eval set gives an empty report with a NaN mean (np.mean([]) in the
    reference), not an exception."""
    from prego_amd.metrics import perframe_average_precision
    rng = np.random.default_rng(4)
    T, Cn = 3000, 7
    pr = rng.random((T, Cn))
    gt = np.zeros((T, Cn))
    gt[np.arange(T), rng.integers(0, Cn, T)] = 1
    names = [f"c{i}" for i in range(Cn)]
    res = perframe_average_precision(pr, gt, names, metrics="cAP")
    eps = np.finfo(float).eps
    for c in range(1, Cn):
        y = gt[np.argsort(-pr[:, c]), c]
        tps, fps = np.cumsum(y), np.cumsum(1 - y)
        ratio = np.sum(y == 0) / np.sum(y)
        want = np.sum((tps / (tps + fps / (ratio + eps) + eps))[y == 1]) / np.sum(y)
        assert abs(res["per_class_AP"][f"c{c}"] - want) < 1e-12
    for m in ("AP", "cAP"):
        empty = perframe_average_precision(np.zeros((0, Cn)), np.zeros((0, Cn)), names, metrics=m)
        assert empty["per_class_AP"] == {} and np.isnan(empty["mean_AP"])


def test_bench_gpus_n_spawns_n_ranks_dry_run():
    """`python bench.py --gpus 2` without a launcher must start 2 ranks itself (one per GPU) and say n_gpus = 2; the dry run
    rendezvouses over gloo and does no GPU work, so it runs on the CPU box."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2
    assert "rank 0: dist.get_world_size() = 2" in r.stderr and "rank 1: dist.get_world_size() = 2" in r.stderr
    # a rank count that contradicts --gpus is an error, not a silent 1-GPU run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=120, env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_train_mode_dry_run_reduces_a_real_size_bucket():
    """`bench.py --mode train --gpus 2 --dry-run`: two ranks over gloo push a 71.7 MB gradient bucket through the trainer's
    bucketed all-reduce (fp32 and bf16-compressed) and rank 0 reports the batch split"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for extra in ([], ["--grad-compress", "bf16"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--mode", "train"] + extra,
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["mode"] == "train" and line["local_batch"] == 8 and line["grad_bucket_bytes"] == 71704920
        assert line["grad_compress"] == (extra[1] if extra else None)


def test_bench_strong_scaling_dry_run_shards_the_one_eval_set_over_eight_ranks():
    """`bench.py --gpus 8 --scaling strong --dry-run` (gloo): the ONE 182-clip eval set is sharded over the ranks by data.shard_clips -
    every clip on exactly one rank, loads within one clip of each other, the longest clip's frames a lower bound of some rank's
    sequential steps - and the line carries `scaling`, per-rank frames / sequential steps and the cost model's N = 1/2/4/8 table;
    `--scaling weak` gives every rank the whole list."""
    import subprocess
    import sys
    from prego_amd.workloads import assembly101_eval_lengths
    lens = assembly101_eval_lengths(seed=20)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for scaling in ("strong", "weak"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--scaling", scaling],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        pr = line["per_rank"]
        assert line["n_gpus"] == 8 and line["scaling"] == scaling and len(pr["frames"]) == 8
        if scaling == "strong":
            assert sum(pr["frames"]) == sum(lens) == line["frames_per_step_all_ranks"] and sum(pr["clips"]) == len(lens)
            assert max(pr["frames"]) - min(pr["frames"]) <= max(lens)
            assert max(pr["longest_clip"]) == max(lens) and max(pr["predicted_sequential_steps"]) >= max(lens)
        else:
            assert pr["frames"] == [sum(lens)] * 8 and line["frames_per_step_all_ranks"] == 8 * sum(lens)
        ev = line["predicted"]["eval"]
        assert [row["n_gpus"] for row in ev["strong"]] == [1, 2, 4, 8] == [row["n_gpus"] for row in ev["weak"]]
        # the documented bound: strong scaling of one eval set cannot beat one-GPU time / the longest clip's sequential floor
        bound = ev["strong"][0]["ms_per_step"] / ev["strong_bound"]["floor_ms"]
        assert all(row["speedup"] <= bound + 1e-9 for row in ev["strong"]) and ev["strong"][-1]["speedup"] < 2.0
        assert ev["weak"][-1]["speedup"] > 6.5


def _g9_tree(tmp_path):
    from prego_amd import weights as W
    lens = {"vidA": 300, "vidB": 157}
    for sub in ("target_perframe", "rgb_anet_resnet50"):
        os.makedirs(os.path.join(tmp_path, sub))
    for vid, T in lens.items():
        tgt = np.zeros((T, 12), np.float32)
        tgt[np.arange(T), (np.arange(T) // 29) % 12] = 1.0
        np.save(os.path.join(tmp_path, "rgb_anet_resnet50", vid + ".npy"), W.tsn_features((T, 2048), 20, f"g9.rgb.{vid}"))
        np.save(os.path.join(tmp_path, "target_perframe", vid + ".npy"), tgt)
    vl = os.path.join(tmp_path, "video_list.json")
    json.dump({"EPIC-TENT-O": {"train_session_set": ["vidA", "vidB"], "test_session_set": ["vidB", "vidA"]}}, open(vl, "w"))
    return vl


def test_feeder_matches_reference_dataset_fixture(tmp_path):
    """G9: the reference's THUMOSDataset (datasets/dataset.py:24-135) on the same synthetic 2-video tree: identical window list
    in train mode (np.random phase, seed 20), identical whole-video items in test mode, identical sample items."""
    from prego_amd.config import epic_tent_cfg
    from prego_amd.data import StepRecognitionDataset
    g = np.load(os.path.join(G, "g9_feeder.npz"))
    vl = _g9_tree(str(tmp_path))
    cfg = epic_tent_cfg(root_path=str(tmp_path), video_list_path=vl)
    for mode in ("train", "test"):
        np.random.seed(20)
        ds = StepRecognitionDataset(cfg, mode)
        assert len(ds) == int(g[f"{mode}.len"])
        assert [w[0] for w in ds.inputs] == list(g[f"{mode}.vids"])
        assert [w[1] for w in ds.inputs] == list(g[f"{mode}.start"]) and [w[2] for w in ds.inputs] == list(g[f"{mode}.end"])
        for j in (0, len(ds) - 1):
            r, f, t, vid, s_, e_ = ds[j]
            meta = list(g[f"{mode}.item{j}.meta"])
            assert [str(vid), str(int(s_)), str(int(e_)), str(tuple(r.shape)), str(r.dtype), str(f.dtype), str(t.dtype)] == meta
            assert abs(float(r.double().sum()) - float(g[f"{mode}.item{j}.rgb_sum"])) < 1e-6
            assert np.array_equal(r.numpy()[[0, -1]][:, :16], g[f"{mode}.item{j}.rgb_rows"])
            assert float(f.double().abs().sum()) == float(g[f"{mode}.item{j}.flow_abs_sum"]) == 0.0
            assert np.array_equal(t.numpy().argmax(1), g[f"{mode}.item{j}.target_argmax"])
            assert np.array_equal(t.numpy().sum(1), g[f"{mode}.item{j}.target_rowsum"])


def test_feeder_drops_the_video_the_reference_drops(tmp_path):
    """datasets/dataset.py:100-107 removes one Assembly101-O session from every split: 181 test videos, not 182."""
    from prego_amd.config import assembly101_cfg
    from prego_amd.data import REFERENCE_EXCLUDED_VIDEOS, StepRecognitionDataset
    bad = REFERENCE_EXCLUDED_VIDEOS[0]
    vids = ["v0", bad, "v1"]
    for sub in ("target_perframe", "rgb_anet_resnet50"):
        os.makedirs(os.path.join(tmp_path, sub))
    for v in vids:
        np.save(os.path.join(tmp_path, "rgb_anet_resnet50", v + ".npy"), np.ones((20, 2048), np.float32))
        np.save(os.path.join(tmp_path, "target_perframe", v + ".npy"), np.eye(86, dtype=np.float32)[np.arange(20) % 86])
    vl = os.path.join(tmp_path, "vl.json")
    json.dump({"ASSEMBLY101-O": {"train_session_set": vids, "test_session_set": vids}}, open(vl, "w"))
    ds = StepRecognitionDataset(assembly101_cfg(root_path=str(tmp_path), video_list_path=vl), "test")
    assert ds.vids == ["v0", "v1"] and [w[0] for w in ds.inputs] == ["v0", "v1"]
    ds = StepRecognitionDataset(assembly101_cfg(root_path=str(tmp_path), video_list_path=vl, exclude_videos=()), "test")
    assert ds.vids == vids


def test_feeder_feature_dtype_holds_16_bit_test_features(tmp_path):
    """cfg['feature_dtype'] = 'fp16' / 'bf16': test-mode items carry that dtype (converted once at load, round to nearest even:
    what the pack kernel would do on the device), training items stay fp32"""
    from prego_amd.config import epic_tent_cfg
    from prego_amd.data import StepRecognitionDataset
    vl = _g9_tree(str(tmp_path))
    ref = StepRecognitionDataset(epic_tent_cfg(root_path=str(tmp_path), video_list_path=vl), "test")
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        cfg = epic_tent_cfg(root_path=str(tmp_path), video_list_path=vl, feature_dtype=name)
        ds = StepRecognitionDataset(cfg, "test")
        rgb, flow, tgt, vid, s, e = ds[0]
        r0 = ref[0][0]
        assert rgb.dtype == dt and flow.dtype == dt and tgt.dtype == torch.float32 and rgb.shape == r0.shape
        assert torch.equal(rgb, r0.to(dt)) and float(flow.abs().sum()) == 0.0
        np.random.seed(1)
        tr = StepRecognitionDataset(cfg, "train")
        assert tr[0][0].dtype == torch.float32
    with pytest.raises(ValueError):
        StepRecognitionDataset(epic_tent_cfg(root_path=str(tmp_path), video_list_path=vl, feature_dtype="int8"), "test")


def test_build_provenance_record_matches_the_tree():
    """prego_amd/lib/build_info.json (written by the in-tree build, shipped with the .so): bench.py quotes it, so a run can tell a
    library built on its own box from a prebuilt one, and whether it was built from the sources in this tree"""
    from prego_amd import build as B
    B.build(force=False)
    info = B.build_info()
    assert info["sources_match_tree"] is True and info["arch"] == "gfx950" and info["build_mode"] in (
        "built on this host", "prebuilt elsewhere, shipped with the tree")


def test_onehot_labels_host_reducer():
    """prego_onehot_labels (a HOST function of the library, no device work): np.argmax of every target row, and per video whether the
    ids say everything the rows do (exactly one nonzero entry per row, positive)."""
    import ctypes as C
    import numpy as np
    from prego_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    ncls = 86
    vids = []
    for T in (1, 0, 40_000, 777):
        t = np.zeros((T, ncls), np.float32)
        t[np.arange(T), rng.integers(0, ncls, T)] = 1
        vids.append(t)
    soft = rng.random((500, ncls)).astype(np.float32)                     # not one-hot: plain argmax
    neg = np.zeros((300, ncls), np.float32); neg[np.arange(300), 5] = -1   # one nonzero per row, but negative: argmax is column 0
    two = vids[2][:1000].copy(); two[17, 3] = 1; two[17, 9] = 1
    zero = vids[2][:1000].copy(); zero[400] = 0
    vids += [soft, neg, two, zero]
    nv = len(vids)
    ptrs = (C.c_void_p * nv)(*[v.ctypes.data for v in vids])
    rows = (C.c_int64 * nv)(*[v.shape[0] for v in vids])
    total = sum(v.shape[0] for v in vids)
    labels = np.full(total, -7, np.int32)
    flags = (C.c_int32 * nv)()
    assert lib.prego_onehot_labels(nv, ptrs, rows, ncls, C.c_void_p(labels.ctypes.data), flags) == 0
    assert list(flags) == [1, 1, 1, 1, 0, 0, 0, 0]
    o = 0
    for v in vids:
        assert np.array_equal(labels[o:o + v.shape[0]], np.argmax(v, 1).astype(np.int32) if v.shape[0] else np.zeros(0, np.int32))
        o += v.shape[0]
    assert lib.prego_onehot_labels(1, None, rows, ncls, C.c_void_p(labels.ctypes.data), flags) != 0
