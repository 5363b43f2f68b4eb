"""One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).

Inference is embarrassingly data parallel: clips are independent (h0 = 0 per clip, rnn.py:49,60), so the clip
list is partitioned across ranks (length-balanced, data.shard_clips), weights are replicated, and there is NO
collective on the data path; rank 0 only gathers the per-clip results at the end.
Training averages gradients with ONE all-reduce per step over a flat fp32 bucket (see autograd/trainer)."""
from __future__ import annotations

import os
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist

from .data import shard_clips


def init_from_env(backend: str | None = None):
    """torchrun-style env (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # dmabuf IPC (what RCCL needs on this pool).  ROCr reads the flag at hsa_init, i.e. at the first HIP call of the process,
    # so it is set before anything below can touch the GPU (prego_amd/__init__.py sets it at import time as well);
    # torch.cuda.device_count() does not initialise HIP, torch.cuda.is_available() does.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.device_count() > 0 else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def sharded_predict(run_clips: Callable[[List[int]], List], lengths: Sequence[int], rank: int, world: int):
    """Each rank runs `run_clips(my_clip_indices) -> list of per-clip results` (any picklable objects, e.g. the
    per-frame argmax lists of eval.py:51-56); rank 0 gets the results of ALL clips in the original order,
    other ranks get None.  No data-path collective: a single gather of the (small) results."""
    mine = shard_clips(lengths, world, rank)
    res = run_clips(mine)
    assert len(res) == len(mine)
    if world == 1:
        out = [None] * len(lengths)
        for i, r in zip(mine, res):
            out[i] = r
        return out
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((mine, res), gathered, dst=0)
    if rank != 0:
        return None
    out = [None] * len(lengths)
    for idxs, rs in gathered:
        for i, r in zip(idxs, rs):
            out[i] = r
    assert all(o is not None for o in out)
    return out


def _scale_(view: torch.Tensor, world: int, weight: float):
    """sum over ranks -> weighted mean: / world (bit-compatible with the unweighted path) or * weight / world"""
    if weight == 1.0:
        view.div_(world)
    else:
        view.mul_(weight / world)


def allreduce_mean_(flat: torch.Tensor, world: int, weight: float = 1.0, force: bool = False):
    """gradient averaging for clip-sharded data-parallel training: loss is a mean over the batch
    (criterions/loss.py:30-31), so grads are summed over ranks and divided by the world size.  `weight` (>= 1) re-weights a step
    whose global batch is short (the last batch of an epoch: the reference's DataLoader has no drop_last, dataset_builder.py:17-23;
    data.EpochWindowSampler pads it with zero-loss windows and says how many were real).  force: run the collective for world == 1
    too (a one-rank process group: the GPU test of this path on a one-GPU box)."""
    if world > 1 or force:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        _scale_(flat, world, weight)
    return flat


_COMM_STREAMS = {}


def allreduce_bucket_(flat: torch.Tensor, lo: int, hi: int, world: int, event=None, compress: str | None = None, weight: float = 1.0):
    """ENQUEUE the averaging all-reduce of flat[lo:hi] (no host wait).  GPU tensors: on the device's comm stream, behind `event` (the
    moment the backward made this sub-bucket final) or, with event = None, behind everything enqueued on the current stream so far;
    the caller joins with allreduce_join_ before it reads the gradients.  CPU tensors (gloo tests): the collective runs here."""
    if compress not in (None, "bf16"):
        raise ValueError(f"compress {compress!r}: expected None or 'bf16'")

    def one():
        view = flat[lo:hi]
        if compress == "bf16":
            buf = view.to(torch.bfloat16)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            view.copy_(buf)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM)
        _scale_(view, world, weight)

    if not flat.is_cuda:
        one()
        return
    dev = flat.device
    comm = _COMM_STREAMS.get(dev)
    if comm is None:
        # high priority: a hardware queue of its own (streams of equal priority may share one and then run in submission order),
        # and the collective wins arbitration against the backward it runs under
        comm = _COMM_STREAMS[dev] = torch.cuda.Stream(dev, priority=-1)
    flat.record_stream(comm)
    if event is not None:
        comm.wait_event(event)           # recorded inside prego_miniroad_backward when this sub-bucket became final
    else:
        comm.wait_stream(torch.cuda.current_stream(dev))      # final once everything enqueued so far has run
    with torch.cuda.stream(comm):
        one()


def allreduce_join_(flat: torch.Tensor):
    """the current stream waits for every sub-bucket enqueued on the comm stream (before optimizer.step reads the gradients)"""
    if flat.is_cuda:
        comm = _COMM_STREAMS.get(flat.device)
        if comm is not None:
            torch.cuda.current_stream(flat.device).wait_stream(comm)


def allreduce_mean_buckets_(flat: torch.Tensor, bounds, world: int, events=None, compress: str | None = None, weight: float = 1.0,
                            force: bool = False, skip=()):
    """Gradient averaging of the flat bucket in sub-buckets `bounds` = [(lo, hi), ...] given in the order the backward finishes
    them.  On a GPU every sub-bucket is reduced on a side stream as soon as its event (events[i]; None = "final in stream order")
    has fired, i.e. under the rest of the backward; the caller's stream waits for the side stream at the end.  compress='bf16'
    sends bf16 (half the bytes over the xGMI ring; the sum runs in bf16 on the wire, the average and everything after it in fp32).
    On CPU tensors (gloo tests) the same sub-buckets are reduced one after the other.  weight / force: see allreduce_mean_.
    skip: indices of sub-buckets that were already enqueued from inside the backward (engine bucket hook)."""
    if world <= 1 and not force:
        return flat
    for i, (lo, hi) in enumerate(bounds):
        if i in skip:
            continue
        allreduce_bucket_(flat, lo, hi, world, events[i] if events is not None else None, compress, weight)
    allreduce_join_(flat)
    return flat
