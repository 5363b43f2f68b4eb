"""`train_one_epoch` ("OAD") behind the reference's TRAINER registry (trainer/train.py:5-29): same signature and
return value (sum of per-step losses).  `scheduler` is accepted and never stepped, as in the reference.
Data-parallel training: when torch.distributed is initialised, gradients are averaged with ONE all-reduce over a flat
fp32 bucket per step (clip-sharded DP); with a single process this is the reference loop."""
from __future__ import annotations

import torch
import torch.distributed as dist

from .distributed import allreduce_bucket_, allreduce_mean_, allreduce_mean_buckets_
from .registry import TRAINER


def _dp_state():
    import os
    init = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if init else 1
    force = init and os.environ.get("PREGO_DP_FORCE_COLLECTIVE") == "1"
    return world, force


def _arm_early_allreduce(model, weight: float = 1.0):
    """Before loss.backward(): let the HIP backward hand every sub-bucket to the collective AT THE MOMENT ITS LAUNCHES ARE ENQUEUED
    (prego_miniroad_backward_callback -> engine bucket hook), not after backward() has returned on the host.  The backward is ~50
    launches; enqueuing the head / GRU all-reduce only behind all of them left them 30-70 us of the backward to hide under (rocprof
    trace of round 4, profiles/r04_train_overlap_trace.json), from inside the backward the GRU bucket gets the whole layer1 / LayerNorm
    tail and the head bucket the BPTT as well."""
    world, force = _dp_state()
    eng_of = getattr(model, "engine", None)
    if (world == 1 and not force) or eng_of is None:
        return
    try:
        eng = model.engine(train=True)
    except TypeError:
        return
    if not hasattr(eng, "set_bucket_hook"):
        return
    if any(p.grad is not None for p in model.parameters()):
        # gradients are being ACCUMULATED (no zero_grad(set_to_none=True) in front of this backward): autograd adds this backward's
        # result into the existing .grad tensors instead of adopting the views of the flat bucket, so an in-place reduce of the bucket
        # from inside the backward would reduce the wrong thing (and race autograd's read of it).  _allreduce_grads takes the generic
        # gather / reduce / scatter path behind the backward instead
        eng.set_bucket_hook(None)
        return
    compress = getattr(model, "grad_compress", None)

    def hook(e, i):
        lo, hi = e._grad_bounds[i]
        allreduce_bucket_(e._grad_flat, lo, hi, world, e._grad_events[i], compress, weight)
        e._early_done.add(i)
    eng.set_bucket_hook(hook)


def _allreduce_grads(model, weight: float = 1.0):
    """sum the gradients over the ranks and average them; `weight` re-weights a short global batch (EpochWindowSampler.step_weight).
    Everything here is ENQUEUED (collectives on a side stream behind the backward's events): no host synchronisation, so the head and
    GRU sub-buckets travel under the rest of the backward.  PREGO_DP_FORCE_COLLECTIVE=1 runs the collective path in a one-rank
    process group as well (GPU test of this path on a one-GPU box)."""
    world, force = _dp_state()
    if world == 1 and not force:
        return
    ps = [p for p in model.parameters() if p.grad is not None]
    eng = getattr(model, "_engine", None)
    flat = getattr(eng, "_grad_flat", None) if eng is not None else None
    if flat is not None and all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in ps):
        # the HIP backward wrote every gradient into one flat bucket: reduce it in place, in the three sub-buckets the backward
        # finishes one after the other (head, GRU, layer1 + LayerNorm) - the first two start under the rest of the backward
        bounds = getattr(eng, "_grad_bounds", None)
        if bounds:
            allreduce_mean_buckets_(flat, bounds, world, getattr(eng, "_grad_events", None), getattr(model, "grad_compress", None),
                                    weight=weight, force=force, skip=getattr(eng, "_early_done", ()))
        else:
            allreduce_mean_(flat, world, weight, force)
        return
    early = set(getattr(eng, "_early_done", ()) or ()) if eng is not None else set()
    if early and flat is not None:
        # The bucket hook already reduced (and scaled) sub-buckets of the flat tensor in place from inside the backward, but autograd
        # did not adopt the flat tensor's views as .grad (AccumulateGrad copied them): wait for those collectives, hand their result to
        # the copies - the hook is only ever armed when .grad was None, so the copy holds exactly this backward's gradient - and reduce
        # only what is left.  Reducing everything again would average those buckets twice (and apply `weight` twice)
        from .distributed import allreduce_join_
        allreduce_join_(flat)
        bounds, offsets = getattr(eng, "_grad_bounds", None), getattr(eng, "_grad_offsets", None)
        if not bounds or not offsets:
            raise RuntimeError("early gradient buckets were reduced in place but the engine does not say where its tensors live "
                               "(_grad_offsets): cannot finish the all-reduce")
        named = dict(model.named_parameters())
        done = set()
        for k, (o, n) in offsets.items():
            p = named.get(k)
            if p is None or p.grad is None:
                continue
            if any(i in early and lo <= o < hi for i, (lo, hi) in enumerate(bounds)):
                p.grad.copy_(flat[o:o + n].view_as(p.grad))
                done.add(id(p))
        ps = [p for p in ps if id(p) not in done]
        if not ps:
            return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])        # generic modules: gather, reduce, scatter
    allreduce_mean_(flat, world, weight, force)
    o = 0
    for p in ps:
        n = p.numel()
        p.grad.copy_(flat[o:o + n].view_as(p))
        o += n


def _check_engine(model):
    """A recurrence / BPTT spin timeout only sets the handle's abort word and makes the kernels return early, so the
    logits and gradients of that step are garbage: surface it (PregoError, PREGO_ETIMEOUT) BEFORE optimizer.step() - or, in the guarded
    loop (_guarded_epoch), behind a step the device skipped for the same reason.
    `engine().check()` synchronises the stream - the reference loop syncs here anyway (`loss.item()`, train.py:26).

    Data-parallel runs: the check is COLLECTIVE.  Every rank's timeout flag travels in the last gradient sub-bucket (engine.backward:
    guard slot, summed by the all-reduce), so after the reduction every rank knows whether ANY rank gave up: the fused AdamW step is a
    no-op on all of them (prego_miniroad_set_peer_guard) and every rank raises here at the same step - the failing rank through its own
    word, the others through the reduced flag - instead of one rank leaving and the rest hanging in the next collective with
    garbage-averaged weights (advisor, round 5)."""
    eng = getattr(model, "_engine", None)
    if eng is None:
        return
    eng.check()
    off = getattr(eng, "_guard_off", None)
    flat = getattr(eng, "_grad_flat", None)
    if off is not None and flat is not None and getattr(eng, "_grad_events", None) is not None:
        if float(flat[off]) != 0.0:        # the stream is synchronised: the reduced flag of this step (or of the step that tripped the guard)
            from ._lib import PregoError
            raise PregoError("data-parallel training: a recurrence / BPTT kernel of another rank timed out (PREGO_ETIMEOUT); "
                             "no rank applied that step")


PREFETCH = True        # train_one_epoch copies batch k + 1 to the device while step k runs (False: the reference's three blocking .to(device))
_SIDE = {}             # device -> (copy stream, pinned scalar for the loss value): created once (a stream costs ~6 ms, a pinned buffer ~1 ms)


def _side_stream(dev):
    ent = _SIDE.get(dev)
    if ent is None:
        ent = _SIDE[dev] = [torch.cuda.Stream(dev), None]
    return ent[0]


def _loss_host(dev, dtype):
    _side_stream(dev)
    ent = _SIDE[dev]
    if ent[1] is None or ent[1].dtype != dtype:
        ent[1] = torch.empty((), dtype=dtype, pin_memory=True)
    return ent[1]


def _device_batches(trainloader, model, device):
    """The loader's batches with rgb / flow / target on `device` (train.py:9: three `.to(device)` per step, in front of the step).
    On a GPU the copies of batch k + 1 are enqueued on a side stream BEFORE step k's kernels, from the loader's pinned tensors
    (`pin_memory=True`, data.build_data_loader), so they run under step k: a 16 x 128 x 4096 fp32 batch is 33.5 MB = 0.6 ms of link time
    per modality against a 1.2 ms step.  A model that is told the flow half is zero (`assume_zero_flow`, dataset.py:69) never reads
    `flow_input`: it is not copied at all.  Same values, same order as the reference loop."""
    dev = torch.device(device)
    skip_flow = bool(getattr(model, "assume_zero_flow", False) and getattr(model, "use_rgb", True))
    if dev.type != "cuda" or not PREFETCH:
        for rgb, flow, target, vid, start, end in trainloader:
            yield rgb.to(device), (flow if skip_flow else flow.to(device)), target.to(device), vid, start, end
        return
    side = _side_stream(dev)

    def move(batch):
        rgb, flow, target, vid, start, end = batch
        with torch.cuda.stream(side):
            moved = (rgb.to(dev, non_blocking=True), flow if skip_flow else flow.to(dev, non_blocking=True), target.to(dev, non_blocking=True))
            ev = torch.cuda.Event()
            ev.record(side)
        return moved + (vid, start, end), ev, batch          # `batch`: the pinned sources stay alive until their copies are done

    it = iter(trainloader)
    nxt = None
    for first in it:
        nxt = move(first)
        break
    while nxt is not None:
        cur, ev, _src = nxt
        main = torch.cuda.current_stream(dev)
        main.wait_event(ev)
        for t in cur[:3]:
            if t.is_cuda:
                t.record_stream(main)             # allocated on the side stream, read by this step's kernels
        nxt = None
        for b in it:                              # batch k + 1's copies go out before step k is enqueued
            nxt = move(b)
            break
        yield cur


CHECK_EVERY = 32       # guarded loop: steps between two timeout checks (each one synchronises)
GUARDED_LOOP = True     # False: always the per-step check + loss.item() loop (tests compare the two)


def _guarded_epoch(trainloader, model, criterion, optimizer, device, step_weight):
    losses, weights, n = None, [], 0
    for it, (rgb_input, flow_input, target, vid, start, end) in enumerate(_device_batches(trainloader, model, device)):
        w = float(step_weight(it)) if step_weight is not None else 1.0
        model.train()
        out_dict = model(rgb_input, flow_input)
        loss = criterion(out_dict, target)
        optimizer.zero_grad(set_to_none=True)
        _arm_early_allreduce(model, w)
        loss.backward()
        _allreduce_grads(model, w)
        optimizer.step()                   # a no-op on the device if this step's recurrence / BPTT gave up
        if losses is None or n == losses.numel():
            grown = torch.empty(max(1024, 2 * n), dtype=loss.dtype, device=loss.device)
            if n:
                grown[:n].copy_(losses[:n])
            losses = grown
        losses[n:n + 1].copy_(loss.detach().reshape(1))
        weights.append(w)
        n += 1
        if n % CHECK_EVERY == 0:
            _check_engine(model)
    _check_engine(model)                   # synchronises; raises PregoError (PREGO_ETIMEOUT) if any step since the last check gave up
    epoch_loss = 0
    if n:
        for v, w in zip(losses[:n].cpu().tolist(), weights):
            epoch_loss += v * w
    return epoch_loss


@TRAINER.register("OAD")
def train_one_epoch(trainloader, model, criterion, optimizer, scaler, epoch, device, writer=None, scheduler=None):
    epoch_loss = 0
    sampler = getattr(trainloader, "sampler", None)
    if hasattr(sampler, "set_epoch"):          # data-parallel runs: a different permutation every epoch
        sampler.set_epoch(epoch)
    step_weight = getattr(sampler, "step_weight", None)       # data.EpochWindowSampler: global batch / real windows of a step (1.0 but for a short last batch)
    # A FusedAdamW bound to this model steps through prego_miniroad_adamw_step, which the DEVICE skips while the engine's timeout word is
    # set: nothing has to be checked on the host in front of optimizer.step().  The loop then runs without a synchronisation per step -
    # the per-step losses go to a device buffer and are summed at the end exactly as train.py:26 sums them (fp64, in step order), the
    # timeout check runs every CHECK_EVERY steps and at the end of the epoch (a step that gave up leaves the weights as they were; so do
    # the steps behind it, the word stays set until the check reports it).  With a tensorboard writer, --amp or any other optimizer the
    # loop below keeps the reference's per-step loss.item().
    guarded = bool(GUARDED_LOOP and scaler is None and writer is None and torch.device(device).type == "cuda" and
                   ((getattr(optimizer, "is_guarded_for", None) is not None and optimizer.is_guarded_for(model))
                    # the `Transformer` entry (ViTEnc) has no persistent kernel that could give up: nothing to check, any optimizer
                    or (hasattr(model, "pre_head_ln") and getattr(model, "_engine", None) is None)))
    if guarded:
        return _guarded_epoch(trainloader, model, criterion, optimizer, device, step_weight)
    for it, (rgb_input, flow_input, target, vid, start, end) in enumerate(_device_batches(trainloader, model, device)):
        w = float(step_weight(it)) if step_weight is not None else 1.0
        loss_value = None
        model.train()
        if scaler is not None:
            # --amp (train.py:10-18): the reference's loss-scaling protocol runs unchanged - scaled loss, scaled gradients through
            # the HIP backward (fp32 accumulation; the MFMA operands are bf16 whatever autocast says: there is no fp16 path on
            # this hardware route, and bf16's exponent range makes the scale harmless), unscale + inf check + step by GradScaler
            with torch.autocast(device_type="cuda", enabled=torch.cuda.is_available()):
                out_dict = model(rgb_input, flow_input)
                loss = criterion(out_dict, target)
            optimizer.zero_grad(set_to_none=True)
            _arm_early_allreduce(model, w)
            scaler.scale(loss).backward()
            _allreduce_grads(model, w)         # enqueued behind the backward's events, BEFORE the host waits for anything
            _check_engine(model)               # synchronises; only gates optimizer.step()
            scaler.step(optimizer)
            scaler.update()
        else:
            out_dict = model(rgb_input, flow_input)
            loss = criterion(out_dict, target)
            optimizer.zero_grad(set_to_none=True)
            _arm_early_allreduce(model, w)
            loss.backward()
            _allreduce_grads(model, w)         # enqueued behind the backward's events, BEFORE the host waits for anything
            # train.py:26's loss.item(): the value travels to pinned host memory in front of the check's synchronisation and is read
            # behind it - optimizer.step() is then enqueued with nothing waiting for it, and the host walks into the next step while the
            # AdamW kernel runs (a second synchronisation behind optimizer.step() cost its launch + 0.09 ms of idle host per step)
            if loss.is_cuda and getattr(model, "_engine", None) is not None:
                loss_host = _loss_host(loss.device, loss.dtype)
                loss_host.copy_(loss.detach(), non_blocking=True)
                _check_engine(model)           # synchronises; only gates optimizer.step()
                optimizer.step()
                loss_value = loss_host.item()
        if loss_value is None:
            if scaler is None:
                _check_engine(model)
                optimizer.step()
            loss_value = loss.item()
        epoch_loss += loss_value * w
        if writer is not None:
            writer.add_scalar("Train Loss", loss_value * w, it + epoch * len(trainloader))
    return epoch_loss
