"""world_size-2 gloo tests (CPU) of the N>1 path: clip sharding + result gather, and gradient averaging.
The per-rank compute here is the CPU oracle port (test infrastructure standing in for the HIP engine, which
needs a GPU); what is under test is the partition/gather/all-reduce logic that the GPU ranks run unchanged."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from prego_amd import distributed as D
    from prego_amd import weights as W
    from prego_amd.config import epic_tent_cfg
    from oracle.oracle_torch import TorchPort
    r, lr, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    cfg = epic_tent_cfg()
    port_model = TorchPort(W.miniroad_state_dict(cfg, 20, head_gain=8.0), 1024)
    lens = [40, 7, 25, 33, 12, 19, 5]
    clips = [torch.from_numpy(W.tsn_features((T, 2048), 9, f"dist.{i}")) for i, T in enumerate(lens)]

    def run(idxs):
        return [port_model.forward(clips[i][None], torch.zeros_like(clips[i])[None])[0].argmax(1).tolist() for i in idxs]

    out = D.sharded_predict(run, lens, rank, world)
    # gradient averaging
    g = torch.full((8,), float(rank + 1))
    D.allreduce_mean_(g, world)
    if rank == 0:
        q.put((out, g.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_predict_and_grad_average_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, g = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert g == [1.5] * 8
    # single-process reference
    from prego_amd import weights as W
    from prego_amd.config import epic_tent_cfg
    from oracle.oracle_torch import TorchPort
    cfg = epic_tent_cfg()
    m = TorchPort(W.miniroad_state_dict(cfg, 20, head_gain=8.0), 1024)
    lens = [40, 7, 25, 33, 12, 19, 5]
    for i, T in enumerate(lens):
        x = torch.from_numpy(W.tsn_features((T, 2048), 9, f"dist.{i}"))
        assert out[i] == m.forward(x[None], torch.zeros_like(x)[None])[0].argmax(1).tolist()


class _FakeModel:
    """stands in for the HIP MROAD in the CPU test: same forward_clips / max_clips / check contract, oracle port underneath"""
    assume_zero_flow = True
    max_clips = 4

    def __init__(self, port):
        self.port = port

    def eval(self):
        return self

    def check(self):
        pass

    def forward_clips(self, rgb, flow, want_probs=True, want_argmax=True):
        probs = [self.port.forward(r[None].cpu(), torch.zeros_like(r)[None].cpu())[0] for r in rgb]
        return probs, [p.argmax(1).int() for p in probs], None


def _eval_worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import json, logging
    from prego_amd import distributed as D
    from prego_amd import weights as W
    from prego_amd.config import epic_tent_cfg
    from prego_amd.evaluate import Evaluate
    from oracle.oracle_torch import TorchPort
    if world > 1:
        D.init_from_env("gloo")
    vl = os.path.join(tmp, "vl.json")
    if rank == 0 and not os.path.exists(vl):
        json.dump({"EPIC-TENT-O": {"class_index": [f"c{i}" for i in range(12)]}}, open(vl, "w"))
    if world > 1:
        dist.barrier()
    cfg = epic_tent_cfg(eval="x.pth", video_list_path=vl, eval_output_dir=os.path.join(tmp, f"out_w{world}"))
    model = _FakeModel(TorchPort(W.miniroad_state_dict(cfg, 20, head_gain=8.0), 1024))
    lens = [30, 11, 25, 18, 9]
    items = []
    for i, T in enumerate(lens):
        tgt = np.zeros((T, 12), np.float32); tgt[np.arange(T), (np.arange(T) // 5) % 12] = 1
        items.append((torch.from_numpy(W.tsn_features((T, 2048), 9, f"ev.{i}"))[None], torch.zeros(1, T, 2048),
                      torch.from_numpy(tgt)[None], (f"v{i}",), torch.tensor([0]), torch.tensor([T])))
    mAP = Evaluate(cfg)(model, items, logging.getLogger("t"), "cpu")
    if rank == 0:
        q.put((float(mAP), json.load(open(os.path.join(tmp, f"out_w{world}", "output_miniROAD.json")))))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_evaluate_shards_videos_across_ranks_and_matches_single_process(tmp_path):
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_eval_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = q.get(timeout=300)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    assert res[1][1] == res[2][1]                       # identical JSON (all videos, same pred/gt)
    assert abs(res[1][0] - res[2][0]) < 1e-12           # identical mAP (same frame order on rank 0)


class _TinyMROAD(torch.nn.Module):
    """CPU stand-in with the MROAD call contract (forward(rgb, flow) -> {'logits'}) for the trainer test: what is under test is
    train_one_epoch's data-parallel step (grad bucket all-reduce-mean, sampler epoch), not the arithmetic"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.l1 = torch.nn.Linear(12, 8)
        self.gru = torch.nn.GRU(8, 6, batch_first=True)
        self.fc = torch.nn.Linear(6, 5)

    def forward(self, rgb, flow):
        h, _ = self.gru(torch.relu(self.l1(torch.cat((rgb, flow), 2))))
        return {"logits": self.fc(torch.relu(h))}


class _BucketMROAD(_TinyMROAD):
    """+ the gradient layout of the HIP engine (engine.backward): every gradient is a view of ONE flat tensor, cut into
    sub-buckets listed in the order the backward finishes them; the trainer reduces those in place (allreduce_mean_buckets_)"""

    def __init__(self, compress=None):
        super().__init__()
        ps = list(self.parameters())
        offs = [0]
        for p in ps:
            offs.append(offs[-1] + p.numel())
        flat = torch.zeros(offs[-1])

        class _E:
            def check(self):
                pass
        self._engine = _E()
        self._engine._grad_flat = flat
        mid = offs[len(ps) // 2]
        self._engine._grad_bounds = [(mid, offs[-1]), (0, mid)]          # later tensors first, as the backward finishes them
        self._engine._grad_events = None
        self.grad_compress = compress
        for p, o in zip(ps, offs):
            def hook(param, o=o):
                v = flat[o:o + param.numel()].view_as(param)
                v.copy_(param.grad)
                param.grad = v
            p.register_post_accumulate_grad_hook(hook)


class _EarlyHookMROAD(_TinyMROAD):
    """the HIP engine's early bucket hook (trainer._arm_early_allreduce) in the situation the round-4 advisor described: the backward
    writes the gradients into one flat tensor and calls the hook for bucket 0 from inside the backward (the trainer reduces that slice IN
    PLACE there), but autograd ends up with COPIES of the flat tensor's slices as .grad, not views.  _allreduce_grads must hand the
    reduced slice to the copies and reduce only the rest - not average bucket 0 twice"""

    def __init__(self):
        super().__init__()
        names = [k for k, _ in self.named_parameters()]
        ps = list(self.parameters())
        offs = [0]
        for p in ps:
            offs.append(offs[-1] + p.numel())
        mid = offs[len(ps) // 2]
        model = self

        class _E:
            _bucket_hook = None

            def check(self):
                pass

            def set_bucket_hook(self, fn):
                self._bucket_hook = fn
        e = self._engine = _E()
        e._grad_flat = torch.zeros(offs[-1])
        e._grad_bounds = [(mid, offs[-1]), (0, mid)]
        e._grad_offsets = {k: (o, p.numel()) for k, o, p in zip(names, offs, ps)}
        e._grad_events = [None, None]
        e._early_done = set()
        self.hook_calls = 0
        last_of_bucket0 = ps[len(ps) // 2]           # autograd reaches the LATER tensors first; bucket 0 = [mid, end) is final once
        first_seen = {"n": 0}                        # every tensor of the second half has its gradient

        for p, o in zip(ps, offs):
            def hook(param, o=o):
                e._grad_flat[o:o + param.numel()].copy_(param.grad.reshape(-1))          # .grad stays a separate tensor (a copy)
                if o >= mid:
                    first_seen["n"] += 1
                    if first_seen["n"] == len(ps) - len(ps) // 2 and e._bucket_hook is not None:
                        e._bucket_hook(e, 0)
                        model.hook_calls += 1
            p.register_post_accumulate_grad_hook(hook)

    def engine(self, train=False):
        self._engine._early_done = set()
        return self._engine


def _oad_loss_torch(out, target):          # criterions/loss.py:15-34
    lg, tg = out["logits"][:, -1, :], target[:, -1, :]
    return torch.mean(torch.sum(-torch.nn.functional.normalize(tg) * torch.log_softmax(lg, -1), dim=1))


def _train_batches():
    g = torch.Generator().manual_seed(3)
    B, T = 8, 7
    rgb, flow = torch.randn(B, T, 6, generator=g), torch.randn(B, T, 6, generator=g)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (B, T), generator=g), 5).float()
    return rgb, flow, tgt


def _train_worker(rank, world, port, q, kind="tiny"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from prego_amd import distributed as D
    from prego_amd.trainer import train_one_epoch
    if world > 1:
        D.init_from_env("gloo")
    rgb, flow, tgt = _train_batches()
    per = rgb.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)          # every rank its own windows (different data per rank)

    class _Loader(list):
        class _S:
            epoch = None

            def set_epoch(self, e):
                self.epoch = e
        sampler = _S()

    loader = _Loader([(rgb[sl], flow[sl], tgt[sl], ("v",) * per, torch.zeros(per), torch.zeros(per))])
    model = _TinyMROAD() if kind == "tiny" else _EarlyHookMROAD() if kind == "early_copy" else _BucketMROAD("bf16" if kind == "bucket_bf16" else None)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    loss = train_one_epoch(loader, model, _oad_loss_torch, opt, None, 3, "cpu")
    assert loader.sampler.epoch == 3                  # DistributedSampler-style samplers get the epoch
    if kind == "early_copy" and world > 1:
        assert model.hook_calls == 1 and model._engine._early_done == {0}      # the hook did fire from inside the backward
    if rank == 0:
        q.put((float(loss), [p.detach().numpy().tolist() for p in model.parameters()]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_train_one_epoch_allreduces_grads_like_the_global_batch():
    """two ranks with DIFFERENT local batches through TRAINER["OAD"] (-> _allreduce_grads): the parameters after the step equal
    the single-process step over the concatenated (global) batch - the loss is a batch mean (loss.py:30-31)"""
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = q.get(timeout=300)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    for a, b in zip(res[1][1], res[2][1]):
        a, b = torch.tensor(a), torch.tensor(b)
        assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
    assert not torch.allclose(torch.tensor(res[1][1][0]), _TinyMROAD().l1.weight)      # the step did move the weights


@pytest.mark.parametrize("kind", ["bucket", "bucket_bf16", "early_copy"])
def test_train_one_epoch_bucketed_allreduce_matches_global_batch(kind):
    """the flat-bucket path of _allreduce_grads (what the HIP engine's backward feeds): sub-buckets reduced in place, in fp32
    (exactly the global-batch step) and bf16-compressed (within bf16 rounding of it)"""
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, kind)) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = q.get(timeout=300)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    init = [p.detach() for p in _TinyMROAD().parameters()]
    for a, b, p0 in zip(res[1][1], res[2][1], init):
        a, b = torch.tensor(a), torch.tensor(b)
        if kind in ("bucket", "early_copy"):
            assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
        else:       # the update (lr * mean gradient) carries bf16 rounding: 2^-8 relative on the step, not on the weight
            step = (a - p0).abs().max().item()
            assert (a - b).abs().max().item() <= 1e-2 * step + 1e-7, ((a - b).abs().max().item(), step)


class _GuardMROAD(_BucketMROAD):
    """+ the engine's guard slot (engine.backward, ABI 7): one extra element at the end of the last sub-bucket that carries the rank's
    "my kernels gave up" flag through the all-reduce; check() raises on the rank that gave up (its own timeout word)"""

    def __init__(self, gave_up):
        super().__init__()
        e = self._engine
        n = e._grad_flat.numel()
        flat = torch.zeros(n + 4)
        lo = e._grad_bounds[1]
        e._grad_bounds = [(e._grad_bounds[0][0] + 4, n + 4), (0, lo[1] + 4)]      # [0, mid) + the slot | the rest, shifted
        e._guard_off = lo[1]
        e._grad_events = [None, None]
        e._grad_flat = flat
        ps = list(self.parameters())
        offs, o = [], 0
        for p in ps:
            if o == lo[1]:
                o += 4
            offs.append(o)
            o += p.numel()
        for p in ps:
            p._post_accumulate_grad_hooks.clear()
        for p, o in zip(ps, offs):
            def hook(param, o=o):
                v = flat[o:o + param.numel()].view_as(param)
                v.copy_(param.grad)
                param.grad = v
                flat[e._guard_off] = 1.0 if gave_up else 0.0          # what prego_miniroad_guard_publish enqueues behind the backward
            p.register_post_accumulate_grad_hook(hook)

        def check():
            if gave_up:
                from prego_amd._lib import PregoError
                raise PregoError("GRU recurrence kernel timed out")
        e.check = check


def _guard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from prego_amd import distributed as D
    from prego_amd._lib import PregoError
    from prego_amd.trainer import train_one_epoch
    D.init_from_env("gloo")
    rgb, flow, tgt = _train_batches()
    per = rgb.shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    loader = [(rgb[sl], flow[sl], tgt[sl], ("v",) * per, torch.zeros(per), torch.zeros(per))] * 2
    model = _GuardMROAD(gave_up=(rank == 1))
    before = [p.detach().clone() for p in model.parameters()]
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    try:
        train_one_epoch(loader, model, _oad_loss_torch, opt, None, 1, "cpu")
        res = "no error"
    except PregoError as e:
        res = str(e)
    moved = any(not torch.equal(a, b.detach()) for a, b in zip(before, model.parameters()))
    q.put((rank, res, moved))
    dist.barrier()
    dist.destroy_process_group()


def test_a_timeout_on_one_rank_raises_on_every_rank_before_any_step():
    """advisor, round 5: a rank whose recurrence / BPTT gave up used to raise ALONE (the others applied garbage-averaged gradients
    and hung in the next collective).  The flag now rides in the gradient bucket: rank 1 gives up, BOTH ranks raise PregoError at that
    step's check - rank 1 through its own word, rank 0 through the reduced flag - and neither has moved a weight."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (res, moved)) for r, res, moved in (q.get(timeout=300) for _ in range(2)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert "timed out" in got[1][0] and not got[1][1]
    assert "another rank" in got[0][0] and not got[0][1], got[0]


def test_epoch_window_sampler_partitions_and_reshuffles():
    from prego_amd.data import EpochWindowSampler

    class _DS(list):
        pass

    ds = _DS(range(103))
    parts = {}
    for rank in range(4):
        s = EpochWindowSampler.__new__(EpochWindowSampler)
        s.dataset, s.world, s.rank, s.batch_size, s.seed, s.epoch = ds, 4, rank, 2, 0, 0
        parts[rank] = list(iter(s))
        assert len(parts[rank]) == len(s) == 26                       # 13 global steps of 8 windows, 2 per rank
    flat = [i for p in parts.values() for i in p]
    real = sorted(i for i in flat if i < 103)
    pads = [i for i in flat if i >= 103]
    assert real == list(range(103)) and len(pads) == 1 and 0 <= pads[0] - 103 < 103      # every window once; one PAD (index + len)
    # the pad sits in the LAST global step, whose weight makes the weighted mean a mean over the 7 real windows
    assert all(s.step_weight(j) == 1.0 for j in range(12)) and s.step_weight(12) == 8 / 7
    last = [i for r in range(4) for i in parts[r][24:26]]
    assert pads[0] in last
    s.set_epoch(1)
    assert list(iter(s)) != parts[3]
    ds.extend(range(5))                                               # _init_features() changed the window count
    assert len(s) == 28 and len(list(iter(s))) == 28                  # 108 windows: 14 global steps


def test_epoch_window_sampler_pads_a_tiny_dataset_cyclically():
    """3 windows against a global batch of 8 (2 ranks x 4): five pads are needed, more than there are windows - every rank must still get
    the same number of entries (a short order would leave one rank a step short and hang the gradient all-reduce)"""
    from prego_amd.data import EpochWindowSampler

    class _DS(list):
        pass

    ds = _DS(range(3))
    got = []
    for rank in range(2):
        s = EpochWindowSampler.__new__(EpochWindowSampler)
        s.dataset, s.world, s.rank, s.batch_size, s.seed, s.epoch = ds, 2, rank, 4, 0, 0
        got.append(list(iter(s)))
        assert len(got[-1]) == len(s) == 4
        assert s.step_weight(0) == 8 / 3
    flat = got[0] + got[1]
    assert sorted(i for i in flat if i < 3) == [0, 1, 2] and sum(i >= 3 for i in flat) == 5 and all(3 <= i < 6 for i in flat if i >= 3)


def _short_batch_worker(rank, world, port, q):
    """n = 11 windows, local batch 2: world 2 -> 3 global steps of 4 (the last one has 3 real windows + 1 pad); world 1 with batch 4
    is the reference loop (DataLoader without drop_last: batches of 4, 4, 3).  Same permutation in both."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from prego_amd import distributed as D
    from prego_amd.data import EpochWindowSampler
    from prego_amd.trainer import train_one_epoch
    if world > 1:
        D.init_from_env("gloo")
    g = torch.Generator().manual_seed(11)
    n, T = 11, 5
    rgb, flow = torch.randn(n, T, 6, generator=g), torch.randn(n, T, 6, generator=g)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 5, (n, T), generator=g), 5).float()

    class _DS(torch.utils.data.Dataset):
        def __len__(self):
            return n

        def __getitem__(self, i):                       # the StepRecognitionDataset contract for PAD entries: zero target
            pad = i >= n
            i = i - n if pad else i
            return rgb[i], flow[i], (torch.zeros_like(tgt[i]) if pad else tgt[i]), "v", 0, T

    ds = _DS()
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0 + 3)).tolist()      # seed + epoch of the sampler below
    if world > 1:
        sampler = EpochWindowSampler(ds, batch_size=2, seed=0)
        loader = torch.utils.data.DataLoader(ds, batch_size=2, sampler=sampler)
    else:
        loader = torch.utils.data.DataLoader(ds, batch_size=4, sampler=perm)               # the reference loop on the same order
    model = _TinyMROAD()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    loss = train_one_epoch(loader, model, _oad_loss_torch, opt, None, 3, "cpu")
    if world > 1:
        t = torch.tensor([loss], dtype=torch.float64)
        dist.all_reduce(t)
        loss = float(t.item()) / world                   # mean over ranks of the weighted local losses = the reference's epoch loss
    if rank == 0:
        q.put((float(loss), [p.detach().numpy().tolist() for p in model.parameters()]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_short_last_batch_is_weighted_not_duplicated():
    """the reference's DataLoader has no drop_last; a data-parallel epoch must reproduce its short last batch (mean over the
    windows that exist), not count wrapped-around windows twice.  Three optimizer steps on two ranks == the single-process epoch."""
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_short_batch_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = q.get(timeout=300)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    assert abs(res[1][0] - res[2][0]) < 1e-5, (res[1][0], res[2][0])
    for a, b in zip(res[1][1], res[2][1]):
        a, b = torch.tensor(a), torch.tensor(b)
        assert torch.allclose(a, b, atol=2e-6), (a - b).abs().max()
