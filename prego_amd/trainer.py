"""`train_one_epoch` ("OAD") behind the reference's TRAINER registry (trainer/train.py:5-29): same signature and
return value (sum of per-step losses).  `scheduler` is accepted and never stepped, as in the reference.
Data-parallel training: when torch.distributed is initialised, gradients are averaged with ONE all-reduce over a flat
fp32 bucket per step (clip-sharded DP); with a single process this is the reference loop."""
from __future__ import annotations

import torch
import torch.distributed as dist

from .distributed import allreduce_mean_, allreduce_mean_buckets_
from .registry import TRAINER


def _allreduce_grads(model):
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return
    ps = [p for p in model.parameters() if p.grad is not None]
    eng = getattr(model, "_engine", None)
    flat = getattr(eng, "_grad_flat", None) if eng is not None else None
    if flat is not None and all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in ps):
        # the HIP backward wrote every gradient into one flat bucket: reduce it in place, in the three sub-buckets the backward
        # finishes one after the other (head, GRU, layer1 + LayerNorm) - the first two start under the rest of the backward
        bounds = getattr(eng, "_grad_bounds", None)
        if bounds:
            allreduce_mean_buckets_(flat, bounds, world, getattr(eng, "_grad_events", None), getattr(model, "grad_compress", None))
        else:
            allreduce_mean_(flat, world)
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])        # generic modules: gather, reduce, scatter
    allreduce_mean_(flat, world)
    o = 0
    for p in ps:
        n = p.numel()
        p.grad.copy_(flat[o:o + n].view_as(p))
        o += n


def _check_engine(model):
    """A recurrence / BPTT spin timeout only sets the handle's abort word and makes the kernels return early, so the
    logits and gradients of that step are garbage: surface it (PregoError, PREGO_ETIMEOUT) BEFORE optimizer.step().
    `engine().check()` synchronises the stream - the reference loop syncs here anyway (`loss.item()`, train.py:26)."""
    eng = getattr(model, "_engine", None)
    if eng is not None:
        eng.check()


@TRAINER.register("OAD")
def train_one_epoch(trainloader, model, criterion, optimizer, scaler, epoch, device, writer=None, scheduler=None):
    epoch_loss = 0
    sampler = getattr(trainloader, "sampler", None)
    if hasattr(sampler, "set_epoch"):          # data-parallel runs: a different permutation every epoch
        sampler.set_epoch(epoch)
    for it, (rgb_input, flow_input, target, vid, start, end) in enumerate(trainloader):
        rgb_input, flow_input, target = rgb_input.to(device), flow_input.to(device), target.to(device)
        model.train()
        if scaler is not None:
            # --amp (train.py:10-18): the reference's loss-scaling protocol runs unchanged - scaled loss, scaled gradients through
            # the HIP backward (fp32 accumulation; the MFMA operands are bf16 whatever autocast says: there is no fp16 path on
            # this hardware route, and bf16's exponent range makes the scale harmless), unscale + inf check + step by GradScaler
            with torch.autocast(device_type="cuda", enabled=torch.cuda.is_available()):
                out_dict = model(rgb_input, flow_input)
                loss = criterion(out_dict, target)
            optimizer.zero_grad(set_to_none=True)
            scaler.scale(loss).backward()
            _check_engine(model)
            _allreduce_grads(model)
            scaler.step(optimizer)
            scaler.update()
        else:
            out_dict = model(rgb_input, flow_input)
            loss = criterion(out_dict, target)
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            _check_engine(model)
            _allreduce_grads(model)
            optimizer.step()
        epoch_loss += loss.item()
        if writer is not None:
            writer.add_scalar("Train Loss", loss.item(), it + epoch * len(trainloader))
    return epoch_loss
