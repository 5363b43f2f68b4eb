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
