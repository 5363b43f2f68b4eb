"""GPU parity tests of the MiniROAD eval path: HIP kernels (through the C ABI) vs the numpy oracle and
the golden vectors generated from the reference.  Tolerances follow BASELINE.json north_star:
per-frame probabilities within 1e-3 (fp32 operands) / 1e-2 (bf16 operands); argmax identical wherever
the reference's top-1/top-2 margin exceeds twice that tolerance (a smaller margin cannot be resolved
at that tolerance by construction; such frames are counted and reported, not hidden)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle_np as O            # noqa: E402  (checker only)
from prego_amd import weights as W           # noqa: E402
from prego_amd.config import assembly101_cfg, epic_tent_cfg  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
# fp16 operands: not a north-star tier; held to 3e-3 (measured 1.7e-3).  fp16x2 (split operands, round 4) is the fp32-class mode
# on the 16-bit matrix pipe: held to 1e-4 (measured <= 5.2e-6 on every fixture)
TOL = {"bf16": 1e-2, "fp16": 3e-3, "fp32": 1e-3, "fp16x2": 1e-4}
DT16 = ("bf16", "fp16")
EXACT = ("fp32", "fp16x2")        # modes held to the reference's argmax on EVERY frame of a reference fixture (north star: identical sequences)
ALL_DT = ["bf16", "fp16", "fp32", "fp16x2"]
HL_TOL = {"bf16": 3e-2, "fp16": 4e-3, "fp32": 1e-3, "fp16x2": 1e-4}


def _model(cfg, sd, dtype):
    from prego_amd.registry import build_model
    import prego_amd.model  # noqa: F401  (registers "MiniROAD")
    cfg = dict(cfg, compute_dtype=dtype)
    m = build_model(cfg, "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.eval()


def _check_probs(got, ref, dtype, what="", exact=None):
    """exact: argmax must equal the reference's on EVERY frame (default for fp32 / fp16x2 against a fixture written by the reference
    itself; the numpy-oracle comparisons pass exact=False and keep a 1e-5 margin: the oracle is fp64, the reference fp32)"""
    tol = TOL[dtype]
    err = np.abs(got - ref).max()
    assert err < tol, f"{what}: max |dprob| {err:.3e} >= {tol}"
    srt = np.sort(ref, 1)
    margin = srt[:, -1] - srt[:, -2]
    if exact is None:
        exact = dtype in EXACT
    safe = margin > (0.0 if exact else (1e-5 if dtype in EXACT else 2 * tol))
    if exact:
        safe = np.ones_like(margin, dtype=bool)
    mism = (got.argmax(1) != ref.argmax(1))
    assert not np.any(mism & safe), f"{what}: argmax differs on {int(np.sum(mism & safe))} frames (largest margin {float(margin[mism].max()):.2e})"
    return err, int(mism.sum()), int((~safe).sum())


@pytest.mark.parametrize("dtype", ALL_DT)
@pytest.mark.parametrize("tag,gain", [("plain", 1.0), ("peaky", 8.0)])
def test_g1_cfg1_golden(dtype, tag, gain):
    """BASELINE config 1 shape (1 clip x 256 frames x 2048-d, zero flow) against the reference's output."""
    g = np.load(os.path.join(G, f"g1_miniroad_eval_{tag}.npz"))
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=gain)
    m = _model(cfg, sd, dtype)
    rgb = torch.from_numpy(W.tsn_features((1, 256, 2048), 20, "g1.rgb")).cuda()
    flow = torch.zeros_like(rgb)
    with torch.no_grad():
        out = m(rgb, flow)["logits"]
    m.engine().check()
    assert out.shape == (1, 256, 86)
    err, mism, unsafe = _check_probs(out[0].cpu().numpy(), g["probs"], dtype, f"g1-{tag}")
    print(f"g1 {tag} {dtype}: max|dprob|={err:.2e} argmax mismatches={mism} (frames under margin: {unsafe})")
    # zero-flow fast path (flow half of K skipped) must give the same numbers
    eng = m.engine()
    outs, args, hl = eng.forward_ragged([rgb[0]], None, want_argmax=True, want_h_last=True)
    eng.check()
    o2 = outs[0].cpu().numpy()
    assert np.abs(o2 - out[0].cpu().numpy()).max() < (2e-3 if dtype in DT16 else 1e-5)
    assert np.array_equal(args[0].cpu().numpy(), o2.argmax(1))
    assert np.abs(hl[0].cpu().numpy() - g["h_last"]).max() < HL_TOL[dtype]


@pytest.mark.parametrize("dtype", ALL_DT)
def test_g3_nonzero_flow_T8(dtype):
    g = np.load(os.path.join(G, "g3_miniroad_intermediates.npz"))
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20)
    m = _model(cfg, sd, dtype)
    rgb = torch.from_numpy(W.tsn_features((1, 8, 2048), 20, "g3.rgb")).cuda()
    flow = torch.from_numpy(W.tsn_features((1, 8, 2048), 20, "g3.flow")).cuda()
    with torch.no_grad():
        out = m(rgb, flow)["logits"][0].cpu().numpy()
    m.engine().check()
    _check_probs(out, g["probs"], dtype, "g3")
    # raw logits (training-mode output convention) through the ragged API
    outs, _, _ = m.engine().forward_ragged([rgb[0]], [flow[0]], softmax=False)
    tol = {"bf16": 5e-2, "fp16": 8e-3, "fp32": 2e-3, "fp16x2": 2e-4}[dtype]
    assert np.abs(outs[0].cpu().numpy() - g["raw_logits"]).max() < tol


@pytest.mark.parametrize("dtype", ALL_DT)
def test_g2_long_T_4096(dtype):
    g = np.load(os.path.join(G, "g2_miniroad_longT_4096.npz"))
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, dtype)
    T = 4096
    rgb = torch.from_numpy(W.tsn_features((T, 2048), 20, f"g2.rgb.{T}")).cuda()
    flow = torch.from_numpy(W.tsn_features((T, 2048), 20, f"g2.flow.{T}")).cuda()
    outs, args, _ = m.engine().forward_ragged([rgb], [flow], want_argmax=True)
    m.engine().check()
    got = outs[0].cpu().numpy()
    tol = TOL[dtype]
    assert np.abs(got[g["sample_idx"]] - g["sample_probs"]).max() < tol
    safe = g["margin"] > (-1.0 if dtype in EXACT else 2 * tol)          # fp32 / fp16x2: every frame
    mism = args[0].cpu().numpy() != g["argmax"].astype(np.int32)
    assert not np.any(mism & safe)
    print(f"g2 T=4096 {dtype}: argmax mismatches {int(mism.sum())} of {T}, all under margin; unsafe frames {int((~safe).sum())}")


def _run_g1c(dtype):
    g = np.load(os.path.join(G, "g1c_miniroad_eval_gain32.npz"))
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=32.0)
    m = _model(cfg, sd, dtype)
    rgb = torch.from_numpy(W.tsn_features((1024, 2048), 20, "g1c.rgb")).cuda()
    flow = torch.from_numpy(W.tsn_features((1024, 2048), 20, "g1c.flow")).cuda()
    outs, args, _ = m.engine().forward_ragged([rgb], [flow], want_argmax=True)
    m.engine().check()
    got = outs[0].cpu().numpy()
    err = float(np.abs(got - g["probs"]).max())
    mism = args[0].cpu().numpy() != g["argmax"]
    worst = float(g["margin"][mism].max()) if mism.any() else 0.0
    print(f"g1c gain32 {dtype}: max|dprob| {err:.2e}, argmax mismatches {int(mism.sum())}, largest violated margin {worst:.2e}")
    return g, err, mism, worst


@pytest.mark.parametrize("dtype", ["fp16", "fp32", "fp16x2"])
def test_g1c_trained_like_head_gain32(dtype):
    """head gain 32 (the reference's top-1 probability is ~0.8 in the median): fp32 and fp16x2 operands reproduce EVERY argmax and
    stay within 1e-3 / 1e-4; fp16 operands stay within 5e-3 and keep every argmax above a 1e-3 margin."""
    g, err, mism, worst = _run_g1c(dtype)
    if dtype in EXACT:
        assert err < TOL[dtype] and not mism.any(), (err, int(mism.sum()), worst)
    else:
        assert err < 5e-3
        assert not np.any(mism & (g["margin"] > 1e-3))


def test_g1c_bf16_operands_are_outside_the_1e2_tier_on_a_gain32_head():
    """NOT a parity claim: a record of where bf16 operands stand on the trained-like fixture.  A x32 head multiplies every upstream
    rounding by 32; the CPU emulation with one precision switch per stage (profiles/precision_study_r04.json) puts bf16 at 1.7e-2
    with ten stages of 3.5e-3 .. 8.7e-3 each, and at 1.12e-2 even with the classifier on split (bf16 hi + lo) or fp32 operands - the fix
    the round-3 verdict proposed - so no single-stage repair brings bf16 inside the north star's 1e-2 here.  The modes that are
    inside a tier on this fixture are fp16 (2e-3, the default), fp16x2 and fp32 (test above).  This test pins the measured envelope
    (2.3e-2 on the GPU) so that a regression of the bf16 path still shows."""
    g, err, mism, worst = _run_g1c("bf16")
    assert err < 3e-2 and worst < 3e-2


def test_g2_long_T_31114_bf16():
    """Longest Epic-tent-O video length: recurrent rounding drift over 31k steps stays inside tolerance."""
    g = np.load(os.path.join(G, "g2_miniroad_longT_31114.npz"))
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, "bf16")
    T = 31114
    rgb = torch.from_numpy(W.tsn_features((T, 2048), 20, f"g2.rgb.{T}")).cuda()
    outs, args, _ = m.engine().forward_ragged([rgb], None, want_argmax=True)
    m.engine().check()
    got = outs[0].cpu().numpy()
    assert np.abs(got[g["sample_idx"]] - g["sample_probs"]).max() < 1e-2
    safe = g["margin"] > 2e-2
    mism = args[0].cpu().numpy() != g["argmax"].astype(np.int32)
    assert not np.any(mism & safe)
    print(f"g2 T=31114 bf16: argmax mismatches {int(mism.sum())} of {T} (frames under margin {int((~safe).sum())})")


@pytest.mark.parametrize("dtype", ["fp32", "fp16x2"])
def test_g2_long_T_31114_exact_argmax(dtype):
    """the longest Epic-tent-O video length in the two fp32-class modes: all 31 114 argmaxes equal the reference's (smallest
    reference margin on this fixture: 2.8e-6) and the sampled probabilities agree to 1e-3 / 1e-4 (round-3 verdict items 1, 3)."""
    g = np.load(os.path.join(G, "g2_miniroad_longT_31114.npz"))
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, dtype)
    T = 31114
    rgb = torch.from_numpy(W.tsn_features((T, 2048), 20, f"g2.rgb.{T}")).cuda()
    outs, args, _ = m.engine().forward_ragged([rgb], None, want_argmax=True)
    m.engine().check()
    got = outs[0].cpu().numpy()
    err = float(np.abs(got[g["sample_idx"]] - g["sample_probs"]).max())
    mism = args[0].cpu().numpy() != g["argmax"].astype(np.int32)
    print(f"g2 T=31114 {dtype}: max|dprob| {err:.2e}, argmax mismatches {int(mism.sum())}")
    assert err < TOL[dtype] and not mism.any()


@pytest.mark.parametrize("dtype", ALL_DT)
def test_ragged_vs_oracle(dtype):
    """ragged clips incl. T=1, lengths around tile edges; some with flow, some zero-flow"""
    cfg = epic_tent_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, dtype)
    lens = [1, 5, 37, 64, 129, 200, 16, 17]
    rgb = [W.tsn_features((T, 2048), 7, f"rag.rgb.{i}") for i, T in enumerate(lens)]
    flow = [W.tsn_features((T, 2048), 7, f"rag.flow.{i}") if i % 2 == 0 else None for i, T in enumerate(lens)]
    outs, args, hl = m.engine().forward_ragged([torch.from_numpy(r).cuda() for r in rgb],
                                                [None if f is None else torch.from_numpy(f).cuda() for f in flow],
                                                want_argmax=True, want_h_last=True)
    m.engine().check()
    for i, T in enumerate(lens):
        f = flow[i] if flow[i] is not None else np.zeros_like(rgb[i])
        ref = O.miniroad_forward(sd, rgb[i][None], f[None], keep=True)
        _check_probs(outs[i].cpu().numpy(), ref["logits"][0], dtype, f"clip{i}", exact=False)
        assert np.array_equal(args[i].cpu().numpy(), outs[i].cpu().numpy().argmax(1))
        assert np.abs(hl[i].cpu().numpy() - ref["h_last"][0]).max() < HL_TOL[dtype]


def test_g2_long_T_31114_fp16_argmax_above_1e3_margin():
    """fp16 operands (the default compute_dtype): over 31 114 recurrent steps every frame whose reference top-1/top-2 margin
    exceeds 1e-3 has the reference's argmax, and the sampled probabilities are within 3e-3 (round-3 verdict item 2;
    profiles/precision_study_r03.json predicts 0 mismatches above a 1.4e-4 margin)."""
    g = np.load(os.path.join(G, "g2_miniroad_longT_31114.npz"))
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, "fp16")
    T = 31114
    rgb = torch.from_numpy(W.tsn_features((T, 2048), 20, f"g2.rgb.{T}")).cuda()
    outs, args, _ = m.engine().forward_ragged([rgb], None, want_argmax=True)
    m.engine().check()
    got = outs[0].cpu().numpy()
    assert np.abs(got[g["sample_idx"]] - g["sample_probs"]).max() < 3e-3
    mism = args[0].cpu().numpy() != g["argmax"].astype(np.int32)
    assert not np.any(mism & (g["margin"] > 1e-3)), float(g["margin"][mism].max())
    print(f"g2 T=31114 fp16: argmax mismatches {int(mism.sum())} of {T}, largest violated margin {float(g['margin'][mism].max()) if mism.any() else 0.0:.2e}")


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_16_bit_features_in_are_bit_identical_to_fp32_features(dtype):
    """PREGO_FWD_IN16: features that already hold the engine's operand type (a feeder with cfg['feature_dtype']) give bit-identical
    results to the fp32 features they were rounded from (the pack kernel's own conversion is the same round-to-nearest-even);
    an fp32 engine and a mismatching 16-bit type are rejected"""
    from prego_amd._lib import PregoError
    cfg = epic_tent_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, dtype)
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    lens = [130, 17, 301]
    rgb = [torch.from_numpy(W.tsn_features((T, 2048), 13, f"in16.r{i}")).cuda() for i, T in enumerate(lens)]
    flow = [torch.from_numpy(W.tsn_features((T, 2048), 13, f"in16.f{i}")).cuda() for i, T in enumerate(lens)]
    eng = m.engine()
    for fl32, fl16 in ((None, None), (flow, [f.to(tdt) for f in flow])):
        o32, a32, _ = eng.forward_ragged(rgb, fl32, want_argmax=True)
        o16, a16, _ = eng.forward_ragged([r.to(tdt) for r in rgb], fl16, want_argmax=True)
        eng.check()
        for x, y, p, q in zip(o32, o16, a32, a16):
            assert torch.equal(x, y) and torch.equal(p, q)
    other = torch.bfloat16 if dtype == "fp16" else torch.float16
    with pytest.raises(PregoError):
        eng.forward_ragged([r.to(other) for r in rgb], None)
    m32 = _model(cfg, sd, "fp32")
    with pytest.raises(PregoError):
        m32.engine().forward_ragged([r.half() for r in rgb], None)


def test_default_compute_dtype_is_fp16_and_training_uses_bf16_engine():
    from prego_amd.registry import build_model
    import prego_amd.model  # noqa: F401
    m = build_model(assembly101_cfg(), "cuda:0")
    assert m.compute_dtype == "fp16"
    assert m.engine().compute_dtype == "fp16" and m.engine(train=True).compute_dtype == "bf16"
    from prego_amd._lib import PregoError
    x = torch.zeros((1, 4, 2048), device="cuda")
    with pytest.raises(PregoError):
        m.engine().forward_train(x, None)                      # fp16 handles are inference-only, and say so


@pytest.mark.parametrize("nclips", [40, 200, 400, 700])
def test_many_clips_all_tile_counts(nclips):
    """clip-tile counts 1/2/4 of the recurrence kernel and the >max_clips multi-pass path (bf16)"""
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, "bf16")
    T = 6
    rgb = W.tsn_features((nclips, T, 2048), 3, f"many.{nclips}")
    lens = [T - (i % 3) for i in range(nclips)]
    outs, _, _ = m.engine().forward_ragged([torch.from_numpy(rgb[i, :lens[i]]).cuda() for i in range(nclips)], None)
    m.engine().check()
    ref = O.miniroad_forward(sd, rgb, None, dt=np.float32)["logits"]
    for i in range(nclips):
        assert np.abs(outs[i].cpu().numpy() - ref[i, :lens[i]]).max() < 1e-2, i


@pytest.mark.parametrize("dtype", ALL_DT)
def test_chunking_and_streaming_are_bit_exact(dtype):
    """properties: (a) the result does not depend on the chunk size of the packed pipeline;
    (b) two half-clips chained through h_last -> h0 equal one full pass (streaming mode)."""
    cfg = epic_tent_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, dtype)
    eng = m.engine()
    lens = [300, 150, 77, 301]
    rgb = [torch.from_numpy(W.tsn_features((T, 2048), 11, f"chk.{i}")).cuda() for i, T in enumerate(lens)]
    eng.rows_per_chunk = 65536
    eng._ws = None
    a, _, ha = eng.forward_ragged(rgb, None, want_h_last=True)
    eng.rows_per_chunk = 128
    eng._ws = None
    b, _, hb = eng.forward_ragged(rgb, None, want_h_last=True)
    eng.check()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(ha, hb)
    eng.rows_per_chunk = 65536
    eng._ws = None
    first = [r[: r.shape[0] // 2] for r in rgb]
    second = [r[r.shape[0] // 2:] for r in rgb]
    o1, _, h1 = eng.forward_ragged(first, None, want_h_last=True)
    o2, _, h2 = eng.forward_ragged(second, None, h0=h1, want_h_last=True)
    eng.check()
    for i in range(len(lens)):
        assert torch.equal(torch.cat([o1[i], o2[i]]), a[i])
    assert torch.equal(h2, ha)


@pytest.mark.parametrize("dtype", ALL_DT)
def test_continuous_batching_matches_per_clip_results(dtype):
    """more clips than recurrence slots: several clips share a slot back to back (h restarts at 0 at every clip
    boundary, chunk boundaries fall anywhere).  Every clip must come out exactly as when it is run alone."""
    cfg = epic_tent_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, dtype)
    eng = m.engine()
    n = 300 if dtype in DT16 else 150                      # > 128 (bf16) / 64 (fp32, fp16x2) slots of one tile layer
    lens = [3 + (i * 7) % 40 for i in range(n)]
    lens[5] = 200                                            # one long clip sets the number of sequential steps
    rgb = [torch.from_numpy(W.tsn_features((T, 2048), 13, f"cb.{i}")).cuda() for i, T in enumerate(lens)]
    eng.rows_per_chunk = 1000                                # several chunks: restarts also land on launch boundaries
    eng._ws = None
    outs, args, _ = eng.forward_ragged(rgb, None, want_argmax=True)
    eng.check()
    for i in (0, 5, 17, 128, 129, n - 1):
        alone, _, _ = eng.forward_ragged([rgb[i]], None)
        assert torch.equal(outs[i], alone[0]), i
    ref = O.miniroad_forward(sd, rgb[n - 1].cpu().numpy()[None], None)["logits"][0]
    assert np.abs(outs[n - 1].cpu().numpy() - ref).max() < TOL[dtype]
    for i in range(n):
        assert outs[i].shape == (lens[i], 12) and bool(torch.isfinite(outs[i]).all())
        assert np.array_equal(args[i].cpu().numpy(), outs[i].cpu().numpy().argmax(1))
    eng.rows_per_chunk = 65536
    eng._ws = None


@pytest.mark.parametrize("which", ["no_rgb", "no_flow"])
def test_single_stream_models_no_rgb_no_flow(which):
    """--no_rgb / --no_flow (main.py:23-24, rnn.py:23-29,54-57): input_dim = one feature stream; eval and one training step"""
    from prego_amd.registry import build_criterion
    import prego_amd.loss  # noqa: F401
    cfg = epic_tent_cfg(**{which: True}, dropout=0.0)
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    assert sd["layer1.0.weight"].shape == (2048, 2048)
    m = _model(cfg, sd, "fp32")
    B, T = 3, 21
    feat = W.tsn_features((B, T, 2048), 31, f"single.{which}")
    dummy = torch.zeros(B, T, 0, device="cuda")
    x = torch.from_numpy(feat).cuda()
    rgb, flow = (dummy, x) if which == "no_rgb" else (x, dummy)
    with torch.no_grad():
        out = m(rgb, flow)["logits"].cpu().numpy()
    m.engine().check()
    ref = O.miniroad_forward(sd, feat, None)["logits"]
    assert np.abs(out - ref).max() < 1e-3
    # training step (fp32 operands): loss and gradients against the oracle's BPTT
    tgt = np.zeros((B, T, 12), np.float32)
    tgt[np.arange(B)[:, None], np.arange(T)[None, :], (np.arange(T)[None, :] + np.arange(B)[:, None]) % 12] = 1.0
    crit = build_criterion(cfg, "cuda:0")
    m.train()
    loss = crit(m(rgb, flow), torch.from_numpy(tgt).cuda())
    loss.backward()
    m.engine().check()
    ref_loss, ref_g = O.miniroad_loss_and_grads(sd, feat, None, tgt)
    assert abs(float(loss.detach()) - ref_loss) < 1e-4
    for k, p in m.named_parameters():
        g = p.grad.cpu().numpy()
        assert np.abs(g - ref_g[k]).max() < 2e-3 * max(np.abs(ref_g[k]).max(), 1e-6) + 1e-7, k


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("n,zero_flow", [(1, False), (3, False), (4, False), (5, False), (16, False), (2, True)])
def test_streaming_step_fast_path_vs_oracle_and_batched(n, zero_flow, dtype):
    """prego_miniroad_step (three launches per frame up to 4 streams, four above; state carried by the caller): n streams fed frame by frame for T frames
    equal MROAD.forward on the whole sequences - against the numpy oracle at the north-star tolerance for bf16 operands, and
    against the batched GPU path (same operand rounding, different summation order and fp32 instead of bf16 projection outputs)."""
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, dtype)
    eng = m.engine()
    T = 24
    rgb = np.stack([W.tsn_features((T, 2048), 31, f"st.rgb{i}") for i in range(n)])
    flow = None if zero_flow else np.stack([W.tsn_features((T, 2048), 31, f"st.flow{i}") for i in range(n)])
    trgb = torch.from_numpy(rgb).cuda()
    tflow = None if flow is None else torch.from_numpy(flow).cuda()
    h = torch.zeros((n, 1024), device="cuda")
    probs, args = [], []
    for t in range(T):
        p, a = m.step(trgb[:, t].contiguous(), None if tflow is None else tflow[:, t].contiguous(), h)
        probs.append(p.clone())
        args.append(a.clone())
    eng.check()
    got = torch.stack(probs, 1).cpu().numpy()                     # [n, T, C]
    garg = torch.stack(args, 1).cpu().numpy()
    ref = O.miniroad_forward(sd, rgb, flow, keep=True)
    assert np.abs(got - ref["logits"]).max() < TOL[dtype]
    assert np.allclose(got.sum(-1), 1.0, atol=1e-4)
    assert np.array_equal(garg, got.argmax(-1))
    # batched path on the same sequences, incl. the final state
    outs, _, hl = eng.forward_ragged([trgb[i] for i in range(n)], None if tflow is None else [tflow[i] for i in range(n)],
                                     softmax=True, want_out=True, want_argmax=False, want_h_last=True)
    eng.check()
    assert np.abs(got - torch.stack(outs).cpu().numpy()).max() < 5e-3
    assert (h - hl).abs().max().item() < 5e-3
    assert np.abs(h.cpu().numpy() - ref["h_last"]).max() < 1e-2


def test_streaming_step_rejects_misuse():
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20)
    m = _model(cfg, sd, "bf16")
    h = torch.zeros((17, 1024), device="cuda")
    x = torch.zeros((17, 2048), device="cuda")
    from prego_amd._lib import PregoError
    with pytest.raises(PregoError):
        m.step(x, None, h)                 # 17 streams: 16 per call
    with pytest.raises(PregoError):
        m.step(x[:2], None, h[:3])         # state rows != frames
    # an fp32 engine serves step() through the general forward (h0 / h_last)
    m32 = _model(cfg, sd, "fp32")
    h2 = torch.zeros((2, 1024), device="cuda")
    r = torch.from_numpy(W.tsn_features((2, 2048), 32, "st32")).cuda()
    p, a = m32.step(r, None, h2)
    m32.engine().check()
    ref = O.miniroad_forward(sd, r.cpu().numpy()[:, None], None, keep=True)
    assert np.abs(p.cpu().numpy() - ref["logits"][:, 0]).max() < 1e-3
    assert np.abs(h2.cpu().numpy() - ref["h_last"]).max() < 1e-3


@pytest.mark.parametrize("which", ["no_rgb", "no_flow"])
def test_streaming_step_single_stream_models(which):
    """the fast path on a --no_rgb / --no_flow model (Epic-tent dims: 12 classes = one class tile, K = 2048 for layer1)"""
    cfg = epic_tent_cfg(**{which: True}, dropout=0.0)
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = _model(cfg, sd, "bf16")
    n, T = 2, 12
    feat = W.tsn_features((n, T, 2048), 33, f"st.single.{which}")
    x = torch.from_numpy(feat).cuda()
    h = torch.zeros((n, 1024), device="cuda")
    got = []
    for t in range(T):
        fr = x[:, t].contiguous()
        p, _ = m.step(None, fr, h) if which == "no_rgb" else m.step(fr, None, h)
        got.append(p.clone())
    m.engine().check()
    ref = O.miniroad_forward(sd, feat, None)["logits"]
    assert np.abs(torch.stack(got, 1).cpu().numpy() - ref).max() < 1e-2


# ---- dimensions the shipped yamls do not use (rnn.py:31-38 takes any hidden_dim / num_layers) ---------------------------------------
G10 = [("h512", 512, 1), ("h2048", 2048, 1), ("h1024_l2", 1024, 2), ("h512_l2", 512, 2)]


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("tag,hid,layers", G10)
def test_g10_other_hidden_sizes_and_two_gru_layers(tag, hid, layers, dtype):
    """hidden_dim 512 / 2048 and nn.GRU(num_layers=2) against the reference's own outputs (fixture G10, two ragged clips with flow): the
    probabilities, the argmax (every frame for fp32 operands) and h_n [layers, H]; then the same two clips streamed in two halves through
    h_last -> h0 ([layers, n, H] for the stacked GRU), which must reproduce the one-shot result bit for bit"""
    if dtype == "fp32" and hid == 2048:
        from prego_amd._lib import PregoError
        with pytest.raises(PregoError, match="hidden_dim 2048 unsupported"):
            _model(assembly101_cfg(hidden_dim=hid, num_layers=layers), W.miniroad_state_dict(assembly101_cfg(hidden_dim=hid, num_layers=layers), 20), dtype).engine()
        return
    g = np.load(os.path.join(G, f"g10_miniroad_eval_{tag}.npz"))
    cfg = assembly101_cfg(hidden_dim=hid, num_layers=layers)
    sd = W.miniroad_state_dict(cfg, seed=20, head_gain=8.0)
    m = _model(cfg, sd, dtype)
    eng = m.engine()
    lens = (96, 40)
    rgb = [torch.from_numpy(W.tsn_features((1, T, 2048), 20, f"g10.{tag}.rgb.{i}")[0]).cuda() for i, T in enumerate(lens)]
    flow = [torch.from_numpy(W.tsn_features((1, T, 2048), 20, f"g10.{tag}.flow.{i}")[0]).cuda() for i, T in enumerate(lens)]
    outs, args, hl = eng.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True, want_h_last=True)
    eng.check()
    assert tuple(hl.shape) == ((2, hid) if layers == 1 else (layers, 2, hid))
    for i in range(2):
        got = outs[i].cpu().numpy()
        _check_probs(got, g[f"probs{i}"], dtype, f"g10 {tag} clip {i}")
        assert np.array_equal(args[i].cpu().numpy(), got.argmax(1))
        hn = (hl[i][None] if layers == 1 else hl[:, i]).cpu().numpy()
        assert np.abs(hn - g[f"h_n{i}"]).max() < HL_TOL[dtype], (tag, i)
    # streaming: first halves, then the rest from the returned state
    cut = [48, 20]
    o1, _, h1 = eng.forward_ragged([r[:c] for r, c in zip(rgb, cut)], [f[:c] for f, c in zip(flow, cut)], want_h_last=True)
    o2, _, h2 = eng.forward_ragged([r[c:] for r, c in zip(rgb, cut)], [f[c:] for f, c in zip(flow, cut)], h0=h1, want_h_last=True)
    eng.check()
    for i in range(2):
        assert torch.equal(torch.cat([o1[i], o2[i]]), outs[i]), (tag, dtype, i)
    assert torch.equal(h2, hl)


def test_other_dimensions_many_clips_and_rejections():
    """hidden_dim 512 with 300 ragged clips (three clip tiles per group: the multi-tile launch of the classic kernel), two GRU layers
    with 150 clips and hidden_dim 2048 with 70 clips against the numpy oracle on sampled clips; what the kernels are not built for is refused with a message: the
    streaming step and fp16x2 at those dimensions (training at these dimensions: tests/test_gpu_train.py, fixtures G4d / G4e)"""
    from prego_amd._lib import PregoError
    rng = np.random.RandomState(3)
    for hid, layers, n in ((512, 1, 300), (1024, 2, 150), (2048, 1, 70)):      # 2048: two groups of 128 workgroups, 70 clips = three tiles
        cfg = assembly101_cfg(hidden_dim=hid, num_layers=layers)
        sd = W.miniroad_state_dict(cfg, seed=20, head_gain=8.0)
        m = _model(cfg, sd, "fp16")
        lens = [int(x) for x in rng.randint(5, 60, size=n)]
        feats = [W.tsn_features((T, 2048), 7, f"odim.{hid}.{layers}.{i}") for i, T in enumerate(lens)]
        outs, args, _ = m.forward_clips([torch.from_numpy(f).cuda() for f in feats])
        m.check()
        for i in (0, n // 2, n - 1, int(np.argmax(lens))):
            ref = O.miniroad_forward(sd, feats[i][None], None)["logits"][0]
            _check_probs(outs[i].cpu().numpy(), ref, "fp16", f"hid {hid} layers {layers} clip {i}", exact=False)
        # online stepping at these dimensions runs the general forward with h0 / h_last: three frames of clip 0 one by one = its first rows
        hst = torch.zeros((1, hid) if layers == 1 else (layers, 1, hid), device="cuda")
        for t in range(3):
            pr, am = m.step(torch.from_numpy(feats[0][t:t + 1]).cuda(), None, hst)
            assert torch.equal(pr[0], outs[0][t]) and int(am[0]) == int(args[0][t]), (hid, layers, t)
        with pytest.raises(PregoError, match="built for hidden_dim 1024"):     # the C entry point itself says what it is built for
            import ctypes as C
            from prego_amd._lib import check
            e = m.engine()
            check(e.lib.prego_miniroad_step(e.h, 1, C.c_void_p(torch.zeros(1, 2048).cuda().data_ptr()), None,
                                            C.c_void_p(torch.zeros(2, 1, 2048).cuda().data_ptr()), None, None, 1, None))
    with pytest.raises(PregoError, match="fp16x2"):
        _model(assembly101_cfg(num_layers=2), W.miniroad_state_dict(assembly101_cfg(num_layers=2), 20), "fp16x2").engine()
    with pytest.raises(PregoError, match="hidden_dim 768"):
        _model(assembly101_cfg(hidden_dim=768), W.miniroad_state_dict(assembly101_cfg(hidden_dim=768), 20), "fp16").engine()
