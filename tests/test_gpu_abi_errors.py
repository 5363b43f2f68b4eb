"""Error behaviour of the C ABI (include/prego_amd.h), through raw ctypes as a foreign host would call it: every misuse
returns its PREGO_E* code, leaves a message on the handle (prego_miniroad_last_error) and does not poison the handle - the
next valid call works.  The reference raises Python exceptions at the same points (nn.GRU on an empty sequence, a Linear
fed the wrong feature size, load_state_dict before forward)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from prego_amd import _lib           # noqa: E402

EINVAL, EWORKSPACE = -1, -3


def _err(lib, h):
    lib.prego_miniroad_last_error.restype = C.c_char_p
    return lib.prego_miniroad_last_error(h).decode()


def _weights(din, emb, hid, ncls):
    g = torch.Generator(device="cuda")
    g.manual_seed(3)

    def r(*s):
        return (torch.rand(s, device="cuda", generator=g) - 0.5) * 0.05
    return [r(emb, din), r(emb), r(emb) + 1.0, r(emb), r(3 * hid, emb), r(3 * hid, hid), r(3 * hid), r(3 * hid), r(ncls, hid), r(ncls)]


def test_create_rejects_unsupported_dimensions():
    lib = _lib.load()
    h = C.c_void_p()
    for args in [(2048, 2048, 2048, 768, 86, 1),      # hidden_dim outside {512, 1024, 2048}: the recurrence keeps its W_hh slice in registers
                 (2048, 2048, 2048, 2048, 86, 0),     # ... 2048 with exact-fp32 operands: a 16-row slice is 384 registers per lane
                 (2048, 2048, 2048, 512, 86, 3),      # ... fp16x2 operands: 1024 only
                 (2048, 2048, 1000, 1024, 86, 1),     # embedding_dim not a multiple of 512
                 (2000, 2048, 2048, 1024, 86, 1),     # feature size not a multiple of 64
                 (0, 0, 2048, 1024, 86, 1),           # --no_rgb and --no_flow together: no input at all
                 (2048, 2048, 2048, 1024, 200, 1),    # more classes than the head kernel holds
                 (2048, 2048, 2048, 1024, 86, 7)]:    # unknown compute dtype
        rc = lib.prego_miniroad_create(C.byref(h), *args)
        assert rc == EINVAL, args
        lib.prego_last_error.restype = C.c_char_p
        assert lib.prego_last_error()      # a text exists for the handle-free failure
    for layers, dtype in ((0, 1), (3, 1), (2, 3)):         # num_layers outside {1, 2}; two layers with fp16x2 operands
        assert lib.prego_miniroad_create_layers(C.byref(h), 2048, 2048, 2048, 1024, 86, layers, dtype) == EINVAL, (layers, dtype)
    # what round 5 added IS accepted: hidden_dim 512 / 2048 with 16-bit operands, 512 with fp32, two layers
    for args in [(2048, 2048, 2048, 512, 86, 1, 2), (2048, 2048, 2048, 2048, 86, 1, 1), (2048, 2048, 2048, 512, 86, 1, 0), (2048, 2048, 2048, 1024, 86, 2, 2)]:
        assert lib.prego_miniroad_create_layers(C.byref(h), *args) == 0, args
        lib.prego_miniroad_destroy(h)


def test_forward_misuse_returns_codes_and_handle_survives():
    lib = _lib.load()
    h = C.c_void_p()
    din, emb, hid, ncls = 4096, 2048, 1024, 86
    assert lib.prego_miniroad_create(C.byref(h), 2048, 2048, emb, hid, ncls, 1) == 0
    try:
        T = 40
        rgb = torch.rand((T, 2048), device="cuda")
        flow = torch.rand((T, 2048), device="cuda")
        out = torch.full((T, ncls), float("nan"), device="cuda")
        lens = (C.c_int32 * 1)(T)
        p_rgb = (C.c_void_p * 1)(rgb.data_ptr())
        p_flow = (C.c_void_p * 1)(flow.data_ptr())
        p_out = (C.c_void_p * 1)(out.data_ptr())
        need = lib.prego_miniroad_workspace_bytes(h, 1, lens, 128, 0)
        assert need > 0
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")

        def fwd(n=1, lens_=lens, rgb_=p_rgb, flow_=p_flow, ws_ptr=ws.data_ptr(), ws_bytes=need):
            return lib.prego_miniroad_forward(h, n, lens_, rgb_, flow_, p_out, None, None, None, 1, ws_ptr, ws_bytes, None)
        # forward before load_state_dict
        assert fwd() == EINVAL and "set_weights" in _err(lib, h)
        w = _weights(din, emb, hid, ncls)
        assert lib.prego_miniroad_set_weights(h, *[t.data_ptr() for t in w], None) == 0
        # no clips / NULL tables / a clip without frames (nn.GRU raises on an empty sequence too)
        assert fwd(n=0) == EINVAL
        assert fwd(lens_=None) == EINVAL
        assert fwd(rgb_=None) == EINVAL and "rgb" in _err(lib, h)
        assert fwd(lens_=(C.c_int32 * 1)(0)) == EINVAL and "frames" in _err(lib, h)
        assert fwd(n=lib.prego_miniroad_max_clips(h) + 1) == EINVAL
        # workspace: NULL, and too small for one time step
        assert fwd(ws_ptr=None) == EINVAL
        assert fwd(ws_bytes=4096) == EWORKSPACE and "workspace" in _err(lib, h)
        # h0 / h_last calls take one clip per recurrence slot: 513 clips do not fit a bf16 handle
        n_big = 513
        lens_big = (C.c_int32 * n_big)(*([2] * n_big))
        big = torch.rand((2, 2048), device="cuda")
        tabs = (C.c_void_p * n_big)(*([big.data_ptr()] * n_big))
        hl = torch.empty((n_big, hid), device="cuda")
        need_big = lib.prego_miniroad_workspace_bytes(h, n_big, lens_big, 2048, 0)
        ws_big = torch.empty(need_big, dtype=torch.uint8, device="cuda")
        rc = lib.prego_miniroad_forward(h, n_big, lens_big, tabs, None, None, None, None, hl.data_ptr(), 0, ws_big.data_ptr(), need_big, None)
        assert rc == EINVAL and _err(lib, h)
        # nothing above poisoned the handle: the valid call still runs and produces probabilities
        assert fwd() == 0
        assert lib.prego_miniroad_check(h, None) == 0
        o = out.cpu().numpy()
        assert np.isfinite(o).all() and np.allclose(o.sum(1), 1.0, atol=1e-4)
    finally:
        lib.prego_miniroad_destroy(h)


def test_loss_and_optimizer_entry_points_validate_arguments():
    lib = _lib.load()
    lens = (C.c_int32 * 1)(4)
    lg = torch.rand((4, 86), device="cuda")
    tg = torch.zeros((4, 86), device="cuda")
    tg[:, 3] = 1
    loss = torch.zeros((), device="cuda")
    p_lg, p_tg = (C.c_void_p * 1)(lg.data_ptr()), (C.c_void_p * 1)(tg.data_ptr())
    assert lib.prego_oad_loss(0, lens, p_lg, p_tg, 86, loss.data_ptr(), None, C.c_float(1.0), None) == EINVAL
    assert lib.prego_oad_loss(1, lens, p_lg, p_tg, 500, loss.data_ptr(), None, C.c_float(1.0), None) == EINVAL
    assert lib.prego_oad_loss(1, lens, (C.c_void_p * 1)(None), p_tg, 86, loss.data_ptr(), None, C.c_float(1.0), None) == EINVAL
    assert lib.prego_oad_loss(1, lens, p_lg, p_tg, 86, loss.data_ptr(), None, C.c_float(1.0), None) == 0
    torch.cuda.synchronize()
    assert np.isfinite(float(loss))


def test_split_operand_handle_and_feed_event_misuse():
    """PREGO_F16X2 (ABI 4) is an inference mode: KEEP (training forward), backward, the fused AdamW step and the streaming fast path refuse
    such a handle with a message; feed events (link-fed inference) refuse h0 / h_last calls and events that do not cover the call; and a
    plain forward on the same handle still works afterwards, raw through the C ABI."""
    lib = _lib.load()
    h = C.c_void_p()
    din, emb, hid, ncls = 4096, 2048, 1024, 86
    assert lib.prego_miniroad_create(C.byref(h), 2048, 2048, emb, hid, ncls, _lib.PREGO_F16X2) == 0
    try:
        w = _weights(din, emb, hid, ncls)
        assert lib.prego_miniroad_set_weights(h, *[t.data_ptr() for t in w], None) == 0
        T = 33
        rgb = torch.rand((T, 2048), device="cuda")
        out = torch.full((T, ncls), float("nan"), device="cuda")
        arg = torch.full((T,), -1, dtype=torch.int32, device="cuda")
        hs = torch.zeros((1, hid), device="cuda")
        lens = (C.c_int32 * 1)(T)
        p_rgb, p_out, p_arg = (C.c_void_p * 1)(rgb.data_ptr()), (C.c_void_p * 1)(out.data_ptr()), (C.c_void_p * 1)(arg.data_ptr())
        need = lib.prego_miniroad_workspace_bytes(h, 1, lens, 128, 0)
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")

        def fwd(flags=1, h_last=None):
            return lib.prego_miniroad_forward(h, 1, lens, p_rgb, None, p_out, p_arg, None, h_last, flags, ws.data_ptr(), need, None)
        assert fwd(flags=1 | _lib.FWD_KEEP) == EINVAL and "fp16x2" in _err(lib, h)
        assert fwd(flags=1 | _lib.FWD_IN16) == EINVAL
        assert lib.prego_miniroad_step(h, 1, rgb.data_ptr(), None, hs.data_ptr(), out.data_ptr(), arg.data_ptr(), 1, None) == EINVAL
        # link-fed calls: the schedule query validates its arguments, events must be non-NULL / ascending, and a call with h_last refuses them
        st = (C.c_int32 * 1)()
        ns = C.c_int32(0)
        assert lib.prego_miniroad_plan_starts(h, 0, lens, 4096, st, C.byref(ns)) == EINVAL
        assert lib.prego_miniroad_plan_starts(h, 1, lens, 4096, st, C.byref(ns)) == 0 and st[0] == 0 and ns.value == T
        ev = torch.cuda.Event()
        ev.record()
        up = (C.c_int32 * 1)(2 ** 31 - 1)
        assert lib.prego_miniroad_set_feed_events(h, 1, up, (C.c_void_p * 1)(None), 4096) == EINVAL
        assert lib.prego_miniroad_set_feed_events(h, 1, up, (C.c_void_p * 1)(ev.cuda_event), 0) == EINVAL
        assert lib.prego_miniroad_set_feed_events(h, 1, up, (C.c_void_p * 1)(ev.cuda_event), 4096) == 0
        assert fwd(h_last=hs.data_ptr()) == EINVAL and "feed" in _err(lib, h)
        # events that stop short of the call's last step: the call says so (and the events are dropped: the next call is a plain one)
        short = (C.c_int32 * 1)(5)
        assert lib.prego_miniroad_set_feed_events(h, 1, short, (C.c_void_p * 1)(ev.cuda_event), 4096) == 0
        assert fwd() == EINVAL and "feed events cover" in _err(lib, h)
        assert fwd() == 0 and lib.prego_miniroad_check(h, None) == 0
        o = out.cpu().numpy()
        assert np.isfinite(o).all() and np.allclose(o.sum(1), 1.0, atol=1e-4)
        assert np.array_equal(arg.cpu().numpy(), o.argmax(1))
    finally:
        lib.prego_miniroad_destroy(h)


def test_pass_info_reports_the_last_forward_and_validates_its_handle():
    """prego_miniroad_pass_info (ABI 5): NULL handle -> PREGO_EINVAL; after a small forward: chunked pass (mode 0), the plan's steps and slots;
    NULL out-pointers are allowed"""
    lib = _lib.load()
    m, st, sl = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
    assert lib.prego_miniroad_pass_info(None, C.byref(m), C.byref(st), C.byref(sl)) == -1
    h = C.c_void_p()
    din, emb, hid, ncls = 4096, 2048, 1024, 86
    assert lib.prego_miniroad_create(C.byref(h), 2048, 2048, emb, hid, ncls, _lib.PREGO_F16) == 0
    try:
        w = _weights(din, emb, hid, ncls)
        assert lib.prego_miniroad_set_weights(h, *[t.data_ptr() for t in w], None) == 0
        lens_l = [40, 25, 33]
        rgb = [torch.rand((T, 2048), device="cuda") for T in lens_l]
        out = [torch.empty((T, ncls), device="cuda") for T in lens_l]
        lens = (C.c_int32 * 3)(*lens_l)
        p_rgb = (C.c_void_p * 3)(*[t.data_ptr() for t in rgb])
        p_out = (C.c_void_p * 3)(*[t.data_ptr() for t in out])
        need = lib.prego_miniroad_workspace_bytes(h, 3, lens, 128, 0)
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        assert lib.prego_miniroad_forward(h, 3, lens, p_rgb, None, p_out, None, None, None, 1, ws.data_ptr(), need, None) == 0
        assert lib.prego_miniroad_check(h, None) == 0
        assert lib.prego_miniroad_pass_info(h, C.byref(m), C.byref(st), C.byref(sl)) == 0
        assert (m.value, st.value, sl.value) == (0, 40, 3)
        assert lib.prego_miniroad_pass_info(h, None, None, None) == 0
    finally:
        lib.prego_miniroad_destroy(h)
