"""world_size-2 gloo tests (CPU) of the N>1 path: clip sharding + result gather, and gradient averaging.
The per-rank compute here is the CPU oracle port (test infrastructure standing in for the HIP engine, which
needs a GPU); what is under test is the partition/gather/all-reduce logic that the GPU ranks run unchanged."""
import os
import socket

import numpy as np
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


class _FakeEngine:
    max_clips = 4

    def check(self):
        pass


class _FakeModel:
    """stands in for the HIP MROAD in the CPU test: same forward_clips contract, oracle port underneath"""
    assume_zero_flow = True

    def __init__(self, port):
        self.port = port

    def eval(self):
        return self

    def engine(self):
        return _FakeEngine()

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
