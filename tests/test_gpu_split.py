"""The split pass (csrc/ff_pass.hip + the PASS instantiation of csrc/gru_recurrence.hip): the recurrence of a whole call as one
launch on R XCDs beside ONE persistent feed-forward launch on the others, instead of a chain of launches per chunk.

Same tiles, same K order, same step arithmetic: every output must equal, BIT FOR BIT, the chunked pass of a handle created under
PREGO_SPLIT_PASS=0 - for ragged clips (partial last 256-row unit), with and without the flow half, fp32 and 16-bit features, fp16 and bf16
operands, probabilities / raw logits / argmax only, and on repeated calls (the counters, rings and rendezvous words are re-armed per
pass).  Sampled clips are also held to the numpy oracle (model/rnn/rnn.py:51-71).  A handle's first call is always chunked (it
establishes the verified workgroup placement), so every test runs its split handle at least twice and asks prego_miniroad_pass_info
what actually ran."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O            # noqa: E402  (checker only)
from prego_amd import weights as W           # noqa: E402
from prego_amd.config import assembly101_cfg  # noqa: E402
from prego_amd._lib import PregoError         # noqa: E402


def _with_env(name, value, fn):
    old = os.environ.get(name)
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = value
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


def _engine(sd, cfg, dtype, split):
    """split: '0' = never, 'R' = whenever eligible, None = the library decides per call (cost model)"""
    from prego_amd.registry import build_model
    import prego_amd.model  # noqa: F401
    m = build_model(dict(cfg, compute_dtype=dtype), "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    return m, _with_env("PREGO_SPLIT_PASS", split, m.engine)


def _feat(shape, seed, dtype=torch.float32):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randn(shape, device="cuda", generator=g).clamp_(min=0).to(dtype)


def _lens(n, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return [int(x) for x in torch.randint(lo, hi + 1, (n,), generator=g)]


@pytest.fixture(scope="module")
def weights():
    cfg = assembly101_cfg()
    return cfg, W.miniroad_state_dict(cfg, 20, head_gain=8.0)


def _run(eng, rgb, flow, **kw):
    o, a, _ = eng.forward_ragged(rgb, flow, **kw)
    eng.check()
    return o, a, eng.pass_info()


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_split_pass_equals_chunked_pass_bit_for_bit(weights, dtype):
    cfg, sd = weights
    lens = _lens(64, 3000, 6200, 11)               # 64 ragged clips, ~295 k frames: the last unit of the pass is partial
    assert sum(lens) >= 262144 and sum(lens) % 256 != 0
    rgb = [_feat((T, 2048), 100 + i) for i, T in enumerate(lens)]
    _, e0 = _engine(sd, cfg, dtype, "0")
    _, e3 = _engine(sd, cfg, dtype, "3")
    ref_o, ref_a, info = _run(e0, rgb, None, softmax=True, want_out=True, want_argmax=True)
    assert info["mode"] == 0
    o, a, info = _run(e3, rgb, None, softmax=True, want_out=True, want_argmax=True)
    assert info["mode"] == 0, "the first call of a handle establishes the placement: chunked"
    for k in range(3):                                                    # repeated split passes: everything is re-armed per pass
        o, a, info = _run(e3, rgb, None, softmax=True, want_out=True, want_argmax=True)
        assert info["mode"] == 3 and info["slots"] == 48, info
        assert info["steps"] >= max(lens)
        for i in range(len(lens)):
            assert torch.equal(o[i], ref_o[i]), f"pass {k}, clip {i}: the split pass changed the probabilities"
            assert torch.equal(a[i], ref_a[i])
    # two clips against the oracle (every other clip is pinned to them through bit-identity with the chunked pass and its own gates)
    tol = {"fp16": 3e-3, "bf16": 1e-2}[dtype]
    for i in (int(np.argmin(lens)), 17):
        ref = O.miniroad_forward(sd, rgb[i].cpu().numpy()[None], None)["logits"][0]
        assert np.abs(o[i].cpu().numpy() - ref).max() < tol


def test_split_pass_with_flow_16bit_features_logits_and_argmax_only(weights):
    cfg, sd = weights
    lens = _lens(50, 4200, 6400, 12)                # 50 clips >= 48 slots: two slots run two clips back to back
    rgb = [_feat((T, 2048), 300 + i, torch.float16) for i, T in enumerate(lens)]
    flow = [_feat((T, 2048), 900 + i, torch.float16) for i, T in enumerate(lens)]
    _, e0 = _engine(sd, cfg, "fp16", "0")
    _, e3 = _engine(sd, cfg, "fp16", "3")
    ref_o, _, _ = _run(e0, rgb, flow, softmax=False, want_out=True, want_argmax=False)         # raw logits (training branch of MROAD.forward)
    _, ref_a, _ = _run(e0, rgb, flow, softmax=True, want_out=False, want_argmax=True)
    _run(e3, rgb, flow, softmax=False, want_out=True, want_argmax=False)
    o, _, info = _run(e3, rgb, flow, softmax=False, want_out=True, want_argmax=False)
    assert info["mode"] == 3
    _, a, info = _run(e3, rgb, flow, softmax=True, want_out=False, want_argmax=True)
    assert info["mode"] == 3
    for i in range(len(lens)):
        assert torch.equal(o[i], ref_o[i]), i
        assert torch.equal(a[i], ref_a[i]), i


@pytest.mark.parametrize("dims", [(1024, 1024, 1024), (1088, 576, 512), (2048, 0, 1536), (0, 1024, 2048)],
                         ids=["bninception_e1024", "odd_64s_e512", "rgb_only_e1536", "flow_only_e2048"])
def test_split_pass_other_feature_and_embedding_sizes(dims):
    """The row jobs of the feed-forward launch (pack, LayerNorm) at sizes other than the shipped 2 048 + 2 048 -> 2 048: sources that fill
    part of a 2 048-column round or straddle a 512-column chunk (buffer accesses past the row's end return zeros / are dropped), a
    missing stream, embedding_dim 512 / 1 024 (LDS gamma / beta) and 1 536 (the generic LayerNorm job) - bit for bit the chunked pass,
    fp32 and 16-bit feature arrays, and a clip against the oracle."""
    from prego_amd.engine import MiniRoadEngine
    d_rgb, d_flow, emb = dims
    C, H = 12, 1024
    rng = np.random.default_rng(5)
    din = d_rgb + d_flow
    u = lambda shape, b: rng.uniform(-b, b, shape).astype(np.float32)
    sd = {"gru.weight_ih_l0": u((3 * H, emb), 1 / 32), "gru.weight_hh_l0": u((3 * H, H), 1 / 32), "gru.bias_ih_l0": u((3 * H,), 1 / 32),
          "gru.bias_hh_l0": u((3 * H,), 1 / 32), "layer1.0.weight": u((emb, din), din ** -0.5), "layer1.0.bias": u((emb,), din ** -0.5),
          "layer1.1.weight": (1.0 + u((emb,), 0.25)), "layer1.1.bias": u((emb,), 0.1),
          "f_classification.0.weight": 8.0 * u((C, H), 1 / 32), "f_classification.0.bias": u((C,), 1 / 32)}
    lens = _lens(52, 5100, 5600, 31)
    assert sum(lens) >= 262144 and sum(lens) % 256 != 0

    def make(split):
        def mk():
            e = MiniRoadEngine(d_rgb, d_flow, emb, H, C, "cuda:0", "fp16")
            e.set_weights({k: torch.from_numpy(v).cuda() for k, v in sd.items()})
            return e
        return _with_env("PREGO_SPLIT_PASS", split, mk)
    e0, e3 = make("0"), make("3")
    for fdt in (torch.float32, torch.float16):
        rgb = [_feat((T, d_rgb), 3100 + i, fdt) for i, T in enumerate(lens)] if d_rgb else None
        flow = [_feat((T, d_flow), 3900 + i, fdt) for i, T in enumerate(lens)] if d_flow else None
        if flow is not None and d_rgb and dims[2] == 1024:                 # some clips without a flow array (= zeros, datasets/dataset.py:69):
            flow = [None if i % 5 == 2 else f for i, f in enumerate(flow)]  # a missing row is a buffer resource of length 0
        if rgb is None:                                     # --no_rgb models: the engine takes the flow stream alone
            ref_o, ref_a, info = _run(e0, None, flow, softmax=True, want_out=True, want_argmax=True)
        else:
            ref_o, ref_a, info = _run(e0, rgb, flow, softmax=True, want_out=True, want_argmax=True)
        assert info["mode"] == 0
        for k in range(2):
            o, a, info = _run(e3, rgb, flow, softmax=True, want_out=True, want_argmax=True)
        assert info["mode"] == 3, info
        for i in range(len(lens)):
            assert torch.equal(o[i], ref_o[i]), (str(fdt), i)
            assert torch.equal(a[i], ref_a[i]), (str(fdt), i)
        if fdt == torch.float32:
            i = int(np.argmin(lens))
            streams = [(x[i].cpu().numpy() if x[i] is not None else np.zeros((lens[i], d_flow), np.float32))[None] for x in (rgb, flow) if x is not None]
            ref = O.miniroad_forward(sd, streams[0], streams[1] if len(streams) > 1 else None)["logits"][0]
            assert np.abs(o[i].cpu().numpy() - ref).max() < 3e-3


def test_split_pass_is_chosen_per_call_and_falls_back(weights):
    """default handle (no PREGO_SPLIT_PASS): the library picks the pass per call - a cost model, corrected by what passes of either kind
    took on this device - so WHICH pass runs is not asserted for eligible calls (devices of the pool differ), only that every call agrees
    bit for bit with the never-split handle; a call that needs h_last, one with too few clips and one with too few frames must stay
    on the chunked pass"""
    cfg, sd = weights
    lens = _lens(60, 3000, 5000, 13) + [12000] * 4     # four long videos: the chunked pass is bound by their 12 000 sequential steps
    rgb = [_feat((T, 2048), 500 + i) for i, T in enumerate(lens)]
    _, e0 = _engine(sd, cfg, "fp16", "0")
    _, ea = _engine(sd, cfg, "fp16", None)
    ref_o, ref_a, _ = _run(e0, rgb, None, softmax=True, want_out=True, want_argmax=True)
    modes = []
    for _ in range(5):
        o, a, info = _run(ea, rgb, None, softmax=True, want_out=True, want_argmax=True)
        modes.append(info["mode"])
        for i in range(len(lens)):
            assert torch.equal(o[i], ref_o[i]) and torch.equal(a[i], ref_a[i]), (modes, i)
    assert modes[0] == 0 and set(modes) <= {0, 3}, modes
    print("passes chosen by the default handle:", modes)
    o, _, hl = ea.forward_ragged(rgb, None, want_h_last=True)            # one clip per slot, state handed back: chunked
    ea.check()
    assert ea.pass_info()["mode"] == 0 and hl.shape == (64, 1024)
    for i in range(len(lens)):
        assert torch.equal(o[i], ref_o[i])
    o, _, _ = ea.forward_ragged(rgb[:40], None)                           # fewer clips than slots
    ea.check()
    assert ea.pass_info()["mode"] == 0
    for i in range(40):
        assert torch.equal(o[i], ref_o[i])
    short = [r[:512] for r in rgb]                                         # 64 x 512 frames: too little work to fill the pipeline
    o, _, _ = ea.forward_ragged(short, None)
    ea.check()
    assert ea.pass_info()["mode"] == 0
    o, a, info = _run(ea, rgb, None, softmax=True, want_out=True, want_argmax=True)      # and back to the long clips
    assert info["mode"] in (0, 3)
    for i in range(len(lens)):
        assert torch.equal(o[i], ref_o[i]) and torch.equal(a[i], ref_a[i])


def test_two_handles_on_two_streams_and_a_bystander_kernel(weights):
    """split passes of two handles enqueued back to back on two streams (the library orders them one behind the other on the device: the
    four persistent launches of two interleaved passes would wait for each other), while a third stream runs ordinary kernels that have
    to wait for the pass's XCDs: everything completes, nothing times out, results equal the chunked pass"""
    cfg, sd = weights
    lens = _lens(56, 4300, 5600, 21)
    rgb = [_feat((T, 2048), 700 + i) for i, T in enumerate(lens)]
    _, e0 = _engine(sd, cfg, "fp16", "0")
    ref_o, ref_a, _ = _run(e0, rgb, None, softmax=True, want_out=True, want_argmax=True)
    _, ea = _engine(sd, cfg, "fp16", "3")
    _, eb = _engine(sd, cfg, "fp16", "3")
    _run(ea, rgb, None, softmax=True, want_out=True, want_argmax=True)      # first calls: chunked, placement
    _run(eb, rgb, None, softmax=True, want_out=True, want_argmax=True)
    sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    x = torch.ones(1 << 20, device="cuda")
    res = {}
    for rep in range(2):
        with torch.cuda.stream(sa):
            res["a"] = ea.forward_ragged(rgb, None, softmax=True, want_out=True, want_argmax=True)
        with torch.cuda.stream(sb):
            res["b"] = eb.forward_ragged(rgb, None, softmax=True, want_out=True, want_argmax=True)
        with torch.cuda.stream(sc):
            y = (x * 2.0 + 1.0).sum()
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        ea.check()
        assert ea.pass_info()["mode"] == 3
    with torch.cuda.stream(sb):
        eb.check()
        assert eb.pass_info()["mode"] == 3
    assert float(y) == 3.0 * (1 << 20)
    for k in ("a", "b"):
        o, a, _ = res[k]
        for i in range(len(lens)):
            assert torch.equal(o[i], ref_o[i]) and torch.equal(a[i], ref_a[i]), (k, i)


def _debug_engine(sd, cfg, dtype, split):
    """an engine on libprego_amd_debug.so (the product sources + the fault-injection entry points of include/prego_amd_debug.h)"""
    import ctypes as C
    from prego_amd import _lib
    from prego_amd.engine import MiniRoadEngine
    from prego_amd.config import FEATURE_SIZES
    dbg = _lib.load_debug()

    def make():
        e = MiniRoadEngine(FEATURE_SIZES[cfg["rgb_type"]], FEATURE_SIZES[cfg["flow_type"]], cfg["embedding_dim"], cfg["hidden_dim"],
                           cfg["num_classes"], "cuda:0", dtype, lib=dbg)
        e.set_weights({k: torch.from_numpy(v).cuda() for k, v in sd.items()})
        return e
    eng = _with_env("PREGO_SPLIT_PASS", split, make)

    def state():
        fb, sk = C.c_int64(), C.c_int64()
        fl, env = C.c_int32(), C.c_int32()
        assert dbg.prego_debug_split_state(eng.h, C.byref(fb), C.byref(fl), C.byref(sk), C.byref(env)) == 0
        return dict(fallbacks=fb.value, fails=fl.value, skip=sk.value, env=env.value)
    return eng, dbg, state


@pytest.mark.parametrize("withheld", [1, 2], ids=["recurrence_withheld", "feed_forward_withheld"])
def test_a_split_pass_that_cannot_run_side_by_side_loses_no_call(weights, withheld):
    """The split pass needs its two persistent launches resident TOGETHER.  When they are not (a profiler that serialises dispatches,
    another tenant on the XCDs - here: the debug library withholds one of the two launches), the start handshake of the launch that did
    start runs out, it leaves without having written anything, and prego_miniroad_forward runs THE SAME CALL as a chunked pass: the
    outputs equal the never-split handle bit for bit, check() reports nothing, pass_info says mode 0, the detour costs well under half a
    second, the next calls are fine (chunked while the back-off lasts, split again afterwards)."""
    import time
    cfg, sd = weights
    lens = _lens(64, 3000, 6200, 31)
    rgb = [_feat((T, 2048), 1300 + i) for i, T in enumerate(lens)]
    _, e0 = _engine(sd, cfg, "fp16", "0")
    ref_o, ref_a, _ = _run(e0, rgb, None, softmax=True, want_out=True, want_argmax=True)
    eng, dbg, state = _debug_engine(sd, cfg, "fp16", "3")
    _run(eng, rgb, None, softmax=True, want_out=True, want_argmax=True)                      # placement
    o, a, info = _run(eng, rgb, None, softmax=True, want_out=True, want_argmax=True)
    assert info["mode"] == 3 and state()["fallbacks"] == 0
    torch.cuda.synchronize()
    t_ok = time.perf_counter()
    _run(eng, rgb, None, softmax=True, want_out=True, want_argmax=True)
    t_ok = time.perf_counter() - t_ok
    # poison the outputs' storage so that a call that wrote nothing cannot pass
    assert dbg.prego_debug_split_fault(eng.h, withheld) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    o, a, info = _run(eng, rgb, None, softmax=True, want_out=True, want_argmax=True)         # check() inside: no error
    dt = time.perf_counter() - t0
    st = state()
    assert info["mode"] == 0, info
    assert st["fallbacks"] == 1 and st["fails"] == 1 and st["skip"] > 0, st
    assert dt < 0.5, f"the detour took {dt:.3f} s (a healthy split pass: {t_ok:.3f} s)"
    for i in range(len(lens)):
        assert torch.equal(o[i], ref_o[i]) and torch.equal(a[i], ref_a[i]), i
    o, a, info = _run(eng, rgb, None, softmax=True, want_out=True, want_argmax=True)         # next call: backing off, chunked, correct
    assert info["mode"] == 0 and state()["fallbacks"] == 1
    for i in range(len(lens)):
        assert torch.equal(o[i], ref_o[i]) and torch.equal(a[i], ref_a[i]), i
    for _ in range(st["skip"] - 1):                                                          # the back-off ends: split again
        eng.forward_ragged(rgb, None, softmax=True, want_out=False, want_argmax=True)
    eng.check()
    o, a, info = _run(eng, rgb, None, softmax=True, want_out=True, want_argmax=True)
    assert info["mode"] == 3, (info, state())
    for i in range(len(lens)):
        assert torch.equal(o[i], ref_o[i]) and torch.equal(a[i], ref_a[i]), i
    print(f"handshake failure ({withheld}): detour {dt * 1e3:.1f} ms, healthy split pass {t_ok * 1e3:.1f} ms")


def test_three_failed_handshakes_keep_the_handle_chunked(weights):
    cfg, sd = weights
    lens = _lens(64, 4100, 4200, 33)
    rgb = [_feat((T, 2048), 1500 + i) for i, T in enumerate(lens)]
    _, e0 = _engine(sd, cfg, "fp16", "0")
    _, ref_a, _ = _run(e0, rgb, None, softmax=True, want_out=False, want_argmax=True)
    eng, dbg, state = _debug_engine(sd, cfg, "fp16", "3")
    _run(eng, rgb, None, softmax=True, want_out=False, want_argmax=True)
    for k in range(3):
        while state()["skip"] > 0:
            eng.forward_ragged(rgb, None, softmax=True, want_out=False, want_argmax=True)
        assert dbg.prego_debug_split_fault(eng.h, 1) == 0
        _, a, info = _run(eng, rgb, None, softmax=True, want_out=False, want_argmax=True)
        assert info["mode"] == 0 and state()["fails"] == k + 1
        for i in range(len(lens)):
            assert torch.equal(a[i], ref_a[i])
    assert state()["env"] == 0
    _, a, info = _run(eng, rgb, None, softmax=True, want_out=False, want_argmax=True)
    assert info["mode"] == 0


def test_forward_allocates_nothing_for_the_whole_call_buffer(weights):
    """ABI 7 ("no allocation of caller-visible memory, workspace sized by a query and passed in; no hidden sync"): relu(h) of
    the whole call - what the once-per-pass classifier and the split pass need - is the CALLER's buffer (prego_miniroad_resident_bytes /
    _set_resident; the engine allocates it through torch).  On the debug library, which counts every hipMalloc of the MiniROAD host code:
    a second, LARGER call and a repeat perform no device allocation inside prego_miniroad_forward, the passes still run split, and a
    handle whose buffer is too small for a call runs the chunked pass with the per-chunk classifier - the same bits."""
    import ctypes as C
    cfg, sd = weights
    lens_a = _lens(64, 3900, 4400, 41)
    lens_b = [T + 700 for T in lens_a]                          # the larger call
    assert sum(lens_a) >= 262144
    rgb_b = [_feat((T, 2048), 2100 + i, torch.float16) for i, T in enumerate(lens_b)]
    rgb_a = [r[:T] for r, T in zip(rgb_b, lens_a)]
    _, e0 = _engine(sd, cfg, "fp16", "0")
    ref_a, _, _ = _run(e0, rgb_a, None, softmax=True, want_out=True, want_argmax=False)
    ref_b, _, _ = _run(e0, rgb_b, None, softmax=True, want_out=True, want_argmax=False)
    eng, dbg, _ = _debug_engine(sd, cfg, "fp16", "3")

    def counts():
        m, w = C.c_int64(), C.c_int64()
        assert dbg.prego_debug_alloc_count(C.byref(m), C.byref(w)) == 0
        return m.value, w.value
    _run(eng, rgb_a, None, softmax=True, want_out=True, want_argmax=False)           # placement (chunked), resident buffer of call A registered
    need_a = dbg.prego_miniroad_resident_bytes(eng.h, len(lens_a), (C.c_int32 * len(lens_a))(*lens_a), 1 | 4)
    need_b = dbg.prego_miniroad_resident_bytes(eng.h, len(lens_b), (C.c_int32 * len(lens_b))(*lens_b), 1 | 4)
    assert need_a >= sum(lens_a) * 2048 and need_b > need_a and eng._res.numel() >= need_a
    o, _, info = _run(eng, rgb_a, None, softmax=True, want_out=True, want_argmax=False)
    assert info["mode"] == 3
    m0, w0 = counts()
    torch.cuda.synchronize()
    o, _, info = _run(eng, rgb_b, None, softmax=True, want_out=True, want_argmax=False)     # larger: the ENGINE grows the buffer (torch), not the library
    m1, w1 = counts()
    assert info["mode"] == 3 and eng._res.numel() >= need_b
    assert m1 == m0, f"prego_miniroad_forward allocated device memory {m1 - m0} times on a larger call"
    assert all(torch.equal(x, y) for x, y in zip(o, ref_b))
    o, _, info = _run(eng, rgb_b, None, softmax=True, want_out=True, want_argmax=False)
    m2, w2 = counts()
    assert m2 == m1 and info["mode"] == 3
    # host waits of a steady-state forward + check: check()'s own synchronisation, and the event of the PREVIOUS call's pointer-table
    # copy out of the handle's pinned staging buffer (long complete; include/prego_amd.h, conventions) - nothing that waits for this call
    assert w2 - w1 <= 2, f"a steady-state forward + check waited {w2 - w1} times on the host"
    # a buffer that is too small: chunked pass, per-chunk classifier, same bits; none at all: likewise
    small = torch.empty(need_a // 2 // 256 * 256, dtype=torch.uint8, device="cuda")
    for buf in (small, None):
        assert dbg.prego_miniroad_set_resident(eng.h, None if buf is None else C.c_void_p(buf.data_ptr()), 0 if buf is None else buf.numel()) == 0
        eng._res = torch.empty(1 << 40, dtype=torch.uint8, device="meta")            # the engine believes it is large enough: it will not re-register
        o, _, info = _run(eng, rgb_a, None, softmax=True, want_out=True, want_argmax=False)
        assert info["mode"] == 0
        assert all(torch.equal(x, y) for x, y in zip(o, ref_a))
    assert counts()[0] == m2
    # misuse
    assert dbg.prego_miniroad_set_resident(eng.h, C.c_void_p(small.data_ptr() + 64), 1024) < 0          # not 256-byte aligned
    assert dbg.prego_miniroad_set_resident(eng.h, None, 1024) < 0
    eng._res = None
