"""GPU parity of the TRAINING step of the `Transformer` registry entry (SURVEY section 8 rows a11-a14 backward; north star
"trainer forward/backward" over the attention / LayerNorm / GELU-FFN / CE-head kernels): loss and every parameter gradient of
ViTEnc (1 and 2 layers, window 128, B = 2) against the fixture the reference's autograd produced (G5b), and the attention
backward kernel alone against the numpy oracle (non-causal and causal, ragged N, three head dims).  bf16 MFMA operands with
fp32 accumulation: per-tensor cosine > 0.995 and norm within 10 % (the bar VERDICT r1 set), loss within 2e-2."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O            # noqa: E402  (checker only)
from prego_amd import weights as W           # noqa: E402
from prego_amd.config import assembly101_cfg  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


def test_vit_train_one_epoch_deferred_losses_equal_the_per_step_loop():
    """TRAINER["OAD"] on the `Transformer` entry: the loop without a host synchronisation per step (losses summed at the end of the
    epoch as train.py:26 sums them; ViTEnc has no kernel that could time out) gives the per-step loss.item() loop's epoch losses and
    weights, bit for bit - with the fused AdamW and with torch's."""
    from prego_amd.optim import FusedAdamW
    from prego_amd.registry import build_criterion, build_model, build_trainer
    import prego_amd.loss, prego_amd.transformer, prego_amd.trainer as TR  # noqa: F401
    cfg = _vit_cfg(num_layers=1)
    sd = W.vit_state_dict(cfg, 20)
    batches = []
    for i in range(4):
        rgb = torch.from_numpy(W.tsn_features((3, 128, 2048), 30 + i, "vl.rgb")).pin_memory()
        flow = torch.from_numpy(W.tsn_features((3, 128, 2048), 30 + i, "vl.flow")).pin_memory()
        tgt = torch.from_numpy(_targets(3, 128, 86, 30 + i, "vl.tgt")).pin_memory()
        batches.append((rgb, flow, tgt, ["v"] * 3, torch.zeros(3), torch.full((3,), 128)))
    for fused in (True, False):
        res = {}
        try:
            for deferred in (True, False):
                TR.GUARDED_LOOP = deferred
                model = build_model(cfg, "cuda:0")
                model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
                crit = build_criterion(cfg, "cuda:0")
                opt = FusedAdamW([{"params": list(model.parameters())}], lr=1e-3, weight_decay=0.05, model=model) if fused else \
                    torch.optim.AdamW([{"params": list(model.parameters())}], lr=1e-3, weight_decay=0.05)
                tr = build_trainer(cfg)
                losses = [tr(batches, model, crit, opt, None, e, "cuda:0", None, scheduler=None) for e in (1, 2)]
                res[deferred] = (losses, {k: p.detach().cpu().numpy() for k, p in model.named_parameters()})
        finally:
            TR.GUARDED_LOOP = True
        assert res[True][0] == res[False][0] and np.isfinite(res[True][0]).all(), (fused, res[True][0], res[False][0])
        for k in res[True][1]:
            assert np.array_equal(res[True][1][k], res[False][1][k]), (fused, k)


def _vit_cfg(**over):
    # compute_dtype bf16: the training tests compare eval-mode and training-mode forwards of the SAME (bf16) handle bit for bit
    cfg = dict(model="Transformer", window_size=128, patch_dim=1, num_heads=8, attn_dropout_rate=0.0, dropout=0.0, compute_dtype="bf16")
    cfg.update(over)
    return assembly101_cfg(**cfg)


def _targets(B, T, Cn, seed, name):
    cls = (W.uniform01((B, T), seed, name) * Cn).astype(np.int64)
    tgt = np.zeros((B, T, Cn), dtype=np.float32)
    bi, ti = np.meshgrid(np.arange(B), np.arange(T), indexing="ij")
    tgt[bi, ti, cls] = 1.0
    return tgt


@pytest.mark.parametrize("layers", [1, 2])
def test_g5b_vit_train_step_matches_reference(layers):
    from prego_amd.registry import build_criterion, build_model
    import prego_amd.loss, prego_amd.transformer  # noqa: F401
    g = np.load(os.path.join(G, f"g5b_vit_train_L{layers}.npz"))
    cfg = _vit_cfg(num_layers=layers)
    sd = W.vit_state_dict(cfg, 20)
    model = build_model(cfg, "cuda:0")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    crit = build_criterion(cfg, "cuda:0")
    rgb = torch.from_numpy(W.tsn_features((2, 128, 2048), 20, "g5.rgb")).cuda()
    flow = torch.from_numpy(W.tsn_features((2, 128, 2048), 20, "g5.flow")).cuda()
    tgt = torch.from_numpy(_targets(2, 128, 86, 20, "g5b.tgt")).cuda()
    model.eval()
    with torch.no_grad():
        ev = model(rgb, flow)["logits"]
    assert np.abs(ev.cpu().numpy() - g["logits"]).max() < 2e-2        # 2-layer forward: every row of layer 1 reaches the logits
    model.train()
    out = model(rgb, flow)
    assert out["logits"].shape == (2, 1, 86)
    assert torch.equal(out["logits"].detach(), ev)                  # the keeping forward is the same arithmetic
    loss = crit(out, tgt)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-2, (float(loss.detach()), float(g["loss"]))
    worst = (1.0, "")
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        gr = p.grad.detach().cpu().numpy().reshape(-1).astype(np.float64)
        assert np.isfinite(gr).all(), k
        ref_norm = float(g["norm." + k])
        got_norm = float(np.linalg.norm(gr))
        ref = g["val." + k].astype(np.float64)
        got = gr[g["idx." + k]]
        cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
        worst = min(worst, (cos, k))
        assert abs(got_norm - ref_norm) < 0.10 * ref_norm + 1e-9, (k, got_norm, ref_norm)
        assert cos > 0.995, (k, cos)
    print(f"vit train L{layers}: loss {float(loss.detach()):.5f} (ref {float(g['loss']):.5f}), worst cosine {worst[0]:.5f} ({worst[1]})")


def test_vit_train_is_deterministic():
    from prego_amd.registry import build_criterion, build_model
    import prego_amd.loss, prego_amd.transformer  # noqa: F401
    cfg = _vit_cfg(num_layers=1)
    sd = W.vit_state_dict(cfg, 20)
    rgb = torch.from_numpy(W.tsn_features((3, 128, 2048), 21, "vt.rgb")).cuda()
    flow = torch.from_numpy(W.tsn_features((3, 128, 2048), 21, "vt.flow")).cuda()
    tgt = torch.from_numpy(_targets(3, 128, 86, 21, "vt.tgt")).cuda()
    runs = []
    for _ in range(2):
        model = build_model(cfg, "cuda:0")
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        crit = build_criterion(cfg, "cuda:0")
        model.train()
        crit(model(rgb, flow), tgt).backward()
        runs.append({k: p.grad.detach().clone() for k, p in model.named_parameters()})
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k               # fixed-order sums, no atomics: bit-reproducible


@pytest.mark.parametrize("N,dh,causal", [(129, 256, 0), (129, 256, 1), (70, 128, 0), (200, 64, 1), (64, 256, 0)])
def test_attention_backward_kernel_vs_oracle(N, dh, causal):
    """dq, dk, dv of softmax(q k^T dh^-0.5 [+ causal mask]) v against the fp64 formulas (oracle_np.vit_loss_and_grads uses the
    same ones); inputs bf16-exact so that only the kernel's own rounding shows."""
    from prego_amd import _lib
    lib = _lib.load_debug()          # prego_debug_attention_bwd: only in libprego_amd_debug.so
    lib.prego_debug_attention_bwd.argtypes = [C.c_int] * 5 + [C.c_void_p] * 7 + [C.c_void_p]
    B, h = 2, 2
    E = h * dh
    rng = np.random.default_rng(N + dh)

    def bf(a):
        return torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)
    q, k, v = bf(rng.standard_normal((B, h, N, dh))), bf(rng.standard_normal((B, h, N, dh))), bf(rng.standard_normal((B, h, N, dh)))
    do = bf(rng.standard_normal((B, N, E)) * 0.1)
    scale = dh ** -0.5
    qs = bf(q * scale)                                                  # what the QKV epilogue stores
    s = np.einsum("bhid,bhjd->bhij", qs, k)
    if causal:
        s = np.where(np.triu(np.ones((N, N), dtype=bool), 1), -np.inf, s)
    mx = s.max(-1, keepdims=True)
    p = np.exp(s - mx)
    lse = (mx + np.log(p.sum(-1, keepdims=True)))[..., 0]
    a = p / p.sum(-1, keepdims=True)
    o = np.einsum("bhij,bhjd->bhid", a, v).transpose(0, 2, 1, 3).reshape(B, N, E)
    doh = do.reshape(B, N, h, dh).transpose(0, 2, 1, 3)
    dv = np.einsum("bhij,bhid->bhjd", a, doh)
    da = np.einsum("bhid,bhjd->bhij", doh, v)
    ds = a * (da - (da * a).sum(-1, keepdims=True))
    dq = np.einsum("bhij,bhjd->bhid", ds, k) * scale                    # wrt the UNSCALED q: dq = scale * dq'
    dk = np.einsum("bhij,bhid->bhjd", ds, qs)
    want = np.stack([dq, dk, dv], 0).transpose(1, 3, 0, 2, 4).reshape(B, N, 3 * E)

    def dev(a, dt=torch.bfloat16):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    tq, tk, tv, to, tdo = dev(qs), dev(k), dev(v), dev(o), dev(do)
    tl = dev(lse, torch.float32)
    out = torch.zeros((B, N, 3 * E), dtype=torch.bfloat16, device="cuda")
    rc = lib.prego_debug_attention_bwd(B, N, h, dh, causal, tq.data_ptr(), tk.data_ptr(), tv.data_ptr(), to.data_ptr(), tdo.data_ptr(),
                                       tl.data_ptr(), out.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    got = out.float().cpu().numpy().astype(np.float64)
    for i, name in enumerate(("dq", "dk", "dv")):
        gsl, wsl = got[..., i * E:(i + 1) * E], want[..., i * E:(i + 1) * E]
        err = np.abs(gsl - wsl).max()
        assert err < 2e-2 * max(np.abs(wsl).max(), 1e-3) + 1e-4, (name, err, np.abs(wsl).max())
        cos = float((gsl * wsl).sum() / (np.linalg.norm(gsl) * np.linalg.norm(wsl) + 1e-30))
        assert cos > 0.9995, (name, cos)


@pytest.mark.parametrize("p_drop,p_attn", [(0.2, 0.0), (0.1, 0.15), (0.0, 0.25)])
def test_vit_train_with_dropout_matches_oracle_with_the_same_masks(p_drop, p_attn):
    """cfg['dropout'] = 0.2 on the Transformer entry: the HIP path draws stateless hash masks (seed, site, element); the oracle
    replays EXACTLY those masks (oracle_np.hash_dropout_mask) through its hand-written forward/backward: loss and all gradients
    must agree, which pins (a) one mask per site shared by forward and backward, (b) the 1/(1-p) scaling, (c) the four sites."""
    from prego_amd.registry import build_criterion, build_model
    import prego_amd.loss, prego_amd.transformer  # noqa: F401
    seed, layers, B, T, E, mlp, heads = 123456789, 2, 2, 128, 2048, 1024, 8
    cfg = _vit_cfg(num_layers=layers, dropout=p_drop, attn_dropout_rate=p_attn)
    sd = W.vit_state_dict(cfg, 20)
    model = build_model(cfg, "cuda:0")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.debug_dropout_seed = seed
    crit = build_criterion(cfg, "cuda:0")
    rgb = W.tsn_features((B, T, 2048), 20, "g5.rgb")
    flow = W.tsn_features((B, T, 2048), 20, "g5.flow")
    tgt = _targets(B, T, 86, 20, "g5b.tgt")
    model.train()
    loss = crit(model(torch.from_numpy(rgb).cuda(), torch.from_numpy(flow).cuda()), torch.from_numpy(tgt).cuda())
    loss.backward()
    torch.cuda.synchronize()
    N = T + 1

    def site(layer, s_):
        return seed + 0x1000 * (layer + 1) * (1 if s_ else 0) + s_
    masks = {"pe": O.hash_dropout_mask(site(0, 0), B * N * E, p_drop).reshape(B, N, E)}
    for l in range(layers):
        masks[(l, "attn")] = O.hash_dropout_mask(site(l, 1), B * N * E, p_drop).reshape(B, N, E)
        masks[(l, "gelu")] = O.hash_dropout_mask(site(l, 2), B * N * mlp, p_drop).reshape(B, N, mlp)
        masks[(l, "ffn")] = O.hash_dropout_mask(site(l, 3), B * N * E, p_drop).reshape(B, N, E)
        masks[(l, "prob")] = O.hash_dropout_mask(site(l, 4), B * heads * N * N, p_attn).reshape(B, heads, N, N)
        masks[(l, "proj")] = O.hash_dropout_mask(site(l, 5), B * N * E, p_attn).reshape(B, N, E)
    assert abs(float((masks["pe"] > 0).mean()) - (1 - p_drop)) < 0.02 and abs(float((masks[(0, "prob")] > 0).mean()) - (1 - p_attn)) < 0.02
    ref_loss, _, ref_g = O.vit_loss_and_grads(sd, rgb, flow, tgt, heads=8, num_layers=layers, masks=masks)
    ref_loss0, _, _ = O.vit_loss_and_grads(sd, rgb, flow, tgt, heads=8, num_layers=layers)
    assert abs(ref_loss - ref_loss0) > 1e-3                      # the masks do change the function
    assert abs(float(loss.detach()) - ref_loss) < 2e-2, (float(loss.detach()), ref_loss)
    for k, pr in model.named_parameters():
        g = pr.grad.detach().cpu().numpy().reshape(-1).astype(np.float64)
        r = ref_g[k].reshape(-1)
        cos = float(g @ r / (np.linalg.norm(g) * np.linalg.norm(r) + 1e-30))
        assert cos > 0.995, (k, cos)
        assert abs(np.linalg.norm(g) - np.linalg.norm(r)) < 0.10 * np.linalg.norm(r) + 1e-9, k


@pytest.mark.parametrize("layers", [1, 2])
def test_fused_adamw_refreshes_vit_handle(layers):
    """train.py:20-24 on the `Transformer` entry with FusedAdamW(model=...): prego_vit_adamw_step rewrites the handle's bf16 / fp32
    copies from the updated values, so prego_vit_set_weights is never called again - and the trajectory (losses, parameters,
    eval logits after three steps) is bit-identical to the generic fused step + re-ingest (same kernel arithmetic, same values
    rounded to bf16 once)."""
    from prego_amd import _lib
    from prego_amd.optim import FusedAdamW
    from prego_amd.registry import build_criterion, build_model
    import prego_amd.loss, prego_amd.transformer  # noqa: F401
    cfg = _vit_cfg(num_layers=layers, compute_dtype="bf16")      # eval on the SAME (bf16) handle whose copies the fused step refreshes
    sd = W.vit_state_dict(cfg, 20)
    rgb = torch.from_numpy(W.tsn_features((3, 128, 2048), 22, "va.rgb")).cuda()
    flow = torch.from_numpy(W.tsn_features((3, 128, 2048), 22, "va.flow")).cuda()
    tgt = torch.from_numpy(_targets(3, 128, 86, 22, "va.tgt")).cuda()
    lib = _lib.load()
    res = {}
    real_set_weights = lib.prego_vit_set_weights
    for kind in ("handle", "generic"):
        ingests = []

        def counting(*a, _n=ingests):
            _n.append(1)
            return real_set_weights(*a)
        lib.prego_vit_set_weights = counting
        model = build_model(cfg, "cuda:0")
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        crit = build_criterion(cfg, "cuda:0")
        opt = FusedAdamW([{"params": list(model.parameters())}], lr=1e-3, weight_decay=0.05, model=model if kind == "handle" else None)
        losses = []
        vers = []
        for _ in range(3):
            model.train()
            loss = crit(model(rgb, flow), tgt)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
            vers.append(model._ver)
        model.eval()
        with torch.no_grad():
            logits = model(rgb, flow)["logits"]
        torch.cuda.synchronize()
        lib.prego_vit_set_weights = real_set_weights
        res[kind] = (losses, logits.clone(), {k: p.detach().clone() for k, p in model.named_parameters()}, vers, len(ingests))
    # the handle path ingested the weights ONCE (the fused step rewrites the handle's copies and records the bumped parameter
    # versions as ingested); the generic path re-ingested after every step: before forwards 2 and 3 and before the eval forward
    assert res["handle"][4] == 1 and res["generic"][4] == 4, (res["handle"][4], res["generic"][4])
    assert len(set(res["handle"][3])) == 3          # the parameter versions DO move now (raw-pointer update made visible)
    assert res["handle"][0] == res["generic"][0] and res["handle"][0][2] < res["handle"][0][0]
    for k in res["handle"][2]:
        assert torch.equal(res["handle"][2][k], res["generic"][2][k]), k
    assert torch.equal(res["handle"][1], res["generic"][1])
