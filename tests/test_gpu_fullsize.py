"""Full-size GPU parity cases for the BASELINE configs that round 1 only ran in bench.py:

 * configs[4] per-GPU share: 512 clips x 512 frames x 2048-d rgb, zero flow, C = 86 (4 tiles per recurrence group);
 * configs[1] at the bench's real packing: 182 ragged clips (3.3 k ... 34 k frames) into 128 slots, 47 pipeline chunks;
 * configs[2] training at its real shape: B = 16 windows x T = 128 frames against the reference fixture G4c.

Size-independent property used at full size: a clip's result does not depend on what else is in the batch - the packing,
the slot it lands in, the chunking - so every sampled clip must be BIT-identical to the same clip run alone; a handful of
clips are also held to the numpy oracle at the north-star tolerance (1e-2 for bf16 operands).  The bf16 argmax bookkeeping
the north star asks for ("identical argmax action sequences") is written to profiles/parity_r02.json by
scripts/parity_report.py; the assertions here are the hard gates."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O            # noqa: E402  (checker only)
from prego_amd import weights as W           # noqa: E402
from prego_amd.config import assembly101_cfg  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


def _model(cfg, sd, dtype="bf16"):
    from prego_amd.registry import build_model
    import prego_amd.model  # noqa: F401
    m = build_model(dict(cfg, compute_dtype=dtype), "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.eval()


def _feat(shape, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randn(shape, device="cuda", generator=g).clamp_(min=0)


OTOL = {"bf16": 1e-2, "fp16": 3e-3, "fp16x2": 1e-4}


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_config4_synth512_zero_flow(dtype):
    """BASELINE configs[4]: 512 clips x 512 frames per GPU, flow zeros (never materialised); fp16 = the shipped default."""
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, dtype)
    eng = m.engine()
    n, T = 512, 512
    rgb = [_feat((T, 2048), 1000 + i) for i in range(n)]
    outs, args, _ = eng.forward_ragged(rgb, None, softmax=True, want_out=True, want_argmax=True)
    eng.check()
    assert len(outs) == n and all(o.shape == (T, 86) for o in outs)
    big = torch.stack(outs)
    assert torch.isfinite(big).all()
    assert torch.allclose(big.sum(-1), torch.ones_like(big[..., 0]), atol=1e-4)
    assert torch.equal(torch.stack(args).long(), big.argmax(-1))
    # (1) batch independence, bit for bit: sampled clips run alone (1 slot, 1 tile) vs inside the 512-clip batch (4 tiles)
    for i in (0, 1, 15, 16, 127, 128, 255, 300, 511):
        o1, a1, _ = eng.forward_ragged([rgb[i]], None, softmax=True, want_out=True, want_argmax=True)
        eng.check()
        assert torch.equal(o1[0], outs[i]), f"clip {i}: result depends on the batch"
        assert torch.equal(a1[0], args[i])
    # (2) against the oracle (fp64 restatement of rnn.py:51-71), north-star tolerance for bf16 operands
    for i in (0, 129, 511):
        ref = O.miniroad_forward(sd, rgb[i].cpu().numpy()[None], None)["logits"][0]
        got = outs[i].cpu().numpy()
        err = np.abs(got - ref).max()
        assert err < OTOL[dtype], (i, err)
        srt = np.sort(ref, 1)
        safe = (srt[:, -1] - srt[:, -2]) > 2 * OTOL[dtype]
        assert not np.any((got.argmax(1) != ref.argmax(1)) & safe)


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp16x2"])
def test_config1_full_eval_set_packing(dtype):
    """BASELINE configs[1] at full size: the bench's 182-clip workload (continuous batching into 128 slots - 64 for fp16x2 -,
    ~34 k steps, 47 chunks; fp16 = the shipped default with the layer1 overlap worker on).  Six sampled clips - the longest one among
    them - must equal the same clip run alone, bit for bit."""
    from prego_amd.workloads import assembly101_eval_lengths
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, dtype)
    eng = m.engine()
    lens = assembly101_eval_lengths(seed=20)
    assert len(lens) == 182
    rgb = [_feat((T, 2048), 50 + i) for i, T in enumerate(lens)]
    flow = [_feat((T, 2048), 5000 + i) for i, T in enumerate(lens)]
    outs, args, _ = eng.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)
    eng.check()
    order = np.argsort(lens)
    sample = [int(order[-1]), int(order[0]), int(order[len(order) // 2]), 0, 97, 181]
    for i in sample:
        assert outs[i].shape == (lens[i], 86)
        o1, a1, _ = eng.forward_ragged([rgb[i]], [flow[i]], softmax=True, want_out=True, want_argmax=True)
        eng.check()
        assert torch.equal(o1[0], outs[i]), f"clip {i} (T={lens[i]}): packed result differs from the clip run alone"
        assert torch.equal(a1[0], args[i])
        assert torch.equal(args[i].long(), outs[i].argmax(-1))
    # the shortest clip against the oracle (the others are pinned to it through bit-identity + the G1/G2 fixtures)
    i = int(order[0])
    ref = O.miniroad_forward(sd, rgb[i].cpu().numpy()[None], flow[i].cpu().numpy()[None])["logits"][0]
    assert np.abs(outs[i].cpu().numpy() - ref).max() < OTOL[dtype]


def _with_env(name, value, fn):
    """run fn() with an environment variable set (the library reads its A/B knobs once per handle, at create)"""
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


def test_config1_overlap_worker_is_bit_identical_to_the_serial_pass():
    """the shipped default runs layer1 of chunk c + 1 as a persistent tile-queue worker on the XCDs the compacted recurrence of chunk c
    has left.  Same tiles, same K order: every output of the 182-clip workload must equal, bit for bit, the serial pass
    of a handle created under PREGO_NO_XCD_OVERLAP=1 (round 3 screened this with builder-run scripts only)."""
    from prego_amd.workloads import assembly101_eval_lengths
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    lens = assembly101_eval_lengths(seed=20)
    rgb = [_feat((T, 2048), 50 + i) for i, T in enumerate(lens)]
    flow = [_feat((T, 2048), 5000 + i) for i, T in enumerate(lens)]
    m_on = _model(cfg, sd, "fp16")
    e_on = _with_env("PREGO_SPLIT_PASS", "0", m_on.engine)        # the chunked pass is what this test is about (tests/test_gpu_split.py: the split pass)
    m_off = _model(cfg, sd, "fp16")
    e_off = _with_env("PREGO_NO_XCD_OVERLAP", "1", lambda: _with_env("PREGO_SPLIT_PASS", "0", m_off.engine))
    a, aa, _ = e_on.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)     # first pass verifies the placement
    a, aa, _ = e_on.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)     # second pass runs compacted + worker
    e_on.check()
    b, bb, _ = e_off.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)
    e_off.check()
    for i in range(len(lens)):
        assert torch.equal(a[i], b[i]), f"clip {i}: overlap worker changed the result"
        assert torch.equal(aa[i], bb[i])


def test_config1_split_pass_is_bit_identical_to_the_chunked_pass():
    """BASELINE configs[1] at the bench's size (182 ragged clips, 2.3 M frames, rgb + flow): the split pass (recurrence of the whole
    call on 3 XCDs beside one persistent feed-forward launch on the other 5; what `python bench.py` runs) against the chunked pass of a
    handle created under PREGO_SPLIT_PASS=0, bit for bit, probabilities and argmax, two split passes in a row."""
    from prego_amd.workloads import assembly101_eval_lengths
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    lens = assembly101_eval_lengths(seed=20)
    rgb = [_feat((T, 2048), 50 + i) for i, T in enumerate(lens)]
    flow = [_feat((T, 2048), 5000 + i) for i, T in enumerate(lens)]
    m_c = _model(cfg, sd, "fp16")
    e_c = _with_env("PREGO_SPLIT_PASS", "0", m_c.engine)
    m_s = _model(cfg, sd, "fp16")
    e_s = _with_env("PREGO_SPLIT_PASS", "3", m_s.engine)
    b, bb, _ = e_c.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)
    e_c.check()
    assert e_c.pass_info()["mode"] == 0
    e_s.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)                 # first call: chunked, verifies the placement
    for _ in range(2):
        a, aa, _ = e_s.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)
        e_s.check()
        info = e_s.pass_info()
        assert info["mode"] == 3 and info["slots"] == 48 and info["steps"] <= int(1.01 * sum(lens) / 48) + 1, info   # packed to within 1 % of frames / slots
        for i in range(len(lens)):
            assert torch.equal(a[i], b[i]), f"clip {i}: the split pass changed the result"
            assert torch.equal(aa[i], bb[i])


def test_overlap_stays_off_without_a_verified_placement():
    """a handle whose recurrence never verifies the one-group-per-XCD placement (PREGO_GRU_NO_LOCAL=1: no rendezvous at all) must not
    launch the layer1 worker beside a recurrence that then runs full width (round-3 advisor): the host mirrors the kernel's
    verified-placement word and keeps the serial pass.  Three passes of a thinned-out workload: no timeout, results identical to
    the default handle's."""
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    lens = [9000, 6000] + [700] * 100                # 102 clips: the tail of the pass has two live slots (compaction + worker territory)
    rgb = [_feat((T, 2048), 300 + i) for i, T in enumerate(lens)]
    m_def = _model(cfg, sd, "fp16")
    e_def = m_def.engine()
    m_nl = _model(cfg, sd, "fp16")
    e_nl = _with_env("PREGO_GRU_NO_LOCAL", "1", m_nl.engine)
    for _ in range(3):
        a, _, _ = e_def.forward_ragged(rgb, None)
        b, _, _ = e_nl.forward_ragged(rgb, None)
    e_def.check()
    e_nl.check()
    for i in range(len(lens)):
        assert torch.equal(a[i], b[i]), i


def test_config4_multitile_kernel_is_bit_identical_to_the_classic_kernel():
    """512 clips x 512 frames = four clip tiles per group: the software-pipelined multi-tile recurrence (HISTORY 5d, default) against the
    classic one-tile-at-a-time kernel (handle created under PREGO_GRU_NO_MT=1), bit for bit, in the shipped default dtype."""
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    n, T = 512, 512
    rgb = [_feat((T, 2048), 1000 + i) for i in range(n)]
    m_mt = _model(cfg, sd, "fp16")
    e_mt = m_mt.engine()
    m_cl = _model(cfg, sd, "fp16")
    e_cl = _with_env("PREGO_GRU_NO_MT", "1", m_cl.engine)
    a, aa, _ = e_mt.forward_ragged(rgb, None, softmax=True, want_out=True, want_argmax=True)
    e_mt.check()
    b, bb, _ = e_cl.forward_ragged(rgb, None, softmax=True, want_out=True, want_argmax=True)
    e_cl.check()
    assert torch.equal(torch.stack(a), torch.stack(b))
    assert torch.equal(torch.stack(aa), torch.stack(bb))


def _targets(B, T, C, seed, name):
    cls = (W.uniform01((B, T), seed, name) * C).astype(np.int64)
    tgt = np.zeros((B, T, C), dtype=np.float32)
    bi, ti = np.meshgrid(np.arange(B), np.arange(T), indexing="ij")
    tgt[bi, ti, cls] = 1.0
    return tgt


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_g4c_train_step_16x128(dtype):
    """configs/miniroad_assembly101-O.yaml's real training shape (B = 16, T = 128: 128 BPTT steps) against the reference
    (fixture G4c: loss, last-frame logits, per-tensor gradient norms, 256 sampled entries per tensor)."""
    from prego_amd.registry import build_criterion, build_model
    import prego_amd.loss  # noqa: F401
    import prego_amd.model  # noqa: F401
    g = np.load(os.path.join(G, "g4c_miniroad_train_16x128.npz"))
    cfg = assembly101_cfg(dropout=0.0, compute_dtype=dtype)
    sd = W.miniroad_state_dict(cfg, 20)
    model = build_model(cfg, "cuda:0")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.train()
    crit = build_criterion(cfg, "cuda:0")
    B, T = 16, 128
    rgb = torch.from_numpy(W.tsn_features((B, T, 2048), 20, "g4c.rgb")).cuda()
    flow = torch.from_numpy(W.tsn_features((B, T, 2048), 20, "g4c.flow")).cuda()
    tgt = torch.from_numpy(_targets(B, T, 86, 20, "g4c.tgt")).cuda()
    out = model(rgb, flow)
    loss = crit(out, tgt)
    loss.backward()
    model.engine().check()
    ltol = 2e-2 if dtype == "bf16" else 2e-4
    assert abs(float(loss.detach()) - float(g["loss"])) < ltol, (float(loss.detach()), float(g["loss"]))
    lerr = np.abs(out["logits"][:, -1, :].detach().cpu().numpy() - g["logits_last"]).max()
    assert lerr < (3e-2 if dtype == "bf16" else 1e-3), lerr
    # fp32 operands: 0.5 % of every tensor's norm.  bf16 operands (dGH and the wgrad operands are rounded to bf16 at each of the
    # 128 steps): cosine > 0.995 on the sampled entries and norms within 10 %.
    for k, p in model.named_parameters():
        gr = p.grad.detach().cpu().numpy().reshape(-1)
        ref_norm = float(g["norm." + k])
        got_norm = float(np.linalg.norm(gr.astype(np.float64)))
        ref = g["val." + k].astype(np.float64)
        got = gr[g["idx." + k]].astype(np.float64)
        if dtype == "fp32":
            assert abs(got_norm - ref_norm) < 5e-3 * ref_norm + 1e-9, (k, got_norm, ref_norm)
            scale = max(np.abs(ref).max(), ref_norm / np.sqrt(gr.size))
            assert np.abs(got - ref).max() < 2e-2 * scale + 1e-9, (k, np.abs(got - ref).max(), scale)
        else:
            assert abs(got_norm - ref_norm) < 0.10 * ref_norm + 1e-9, (k, got_norm, ref_norm)
            cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
            assert cos > 0.995, (k, cos)
