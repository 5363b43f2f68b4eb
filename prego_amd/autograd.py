"""torch.autograd glue so that the reference's training loop runs unchanged (trainer/train.py:20-24:
`out = model(rgb, flow); loss = criterion(out, target); optimizer.zero_grad(); loss.backward(); optimizer.step()`):
the forward/backward arithmetic is the HIP path, autograd only carries the tensors between the two calls."""
from __future__ import annotations

import torch

from ._lib import PregoError
from .engine import oad_loss, param_order


class _MiniRoadTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, rgb, flow, *params):
        eng = model.engine(train=True)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if model.layer1[3].p > 0 else 0   # torch RNG drives the mask seed
        eng.set_dropout(model.layer1[3].p, seed)
        out = eng.forward_train(rgb, flow)
        # the kept activations and the dropout seed live in the engine (one workspace per model): a second training forward
        # before this one's backward overwrites them, so every forward takes a generation number and backward checks it
        eng._train_gen = getattr(eng, "_train_gen", 0) + 1
        ctx.eng, ctx.gen = eng, eng._train_gen
        return out

    @staticmethod
    def backward(ctx, dout):
        if getattr(ctx.eng, "_train_gen", 0) != ctx.gen:
            raise PregoError("MiniROAD backward: another training forward ran on this model since the forward of this graph; its kept "
                             "activations were overwritten (run backward before the next forward, or use torch.no_grad() / eval() for it)")
        grads = ctx.eng.backward(dout)
        return (None, None, None) + tuple(grads[k] for k in param_order(ctx.eng.num_layers))


def miniroad_train_forward(model, rgb_input, flow_input):
    named = dict(model.named_parameters())
    params = [named[k] for k in param_order(model.num_layers)]           # one GRU layer or two (rnn.py:32,38)
    flow = flow_input if (model.use_flow and (not model.assume_zero_flow or not model.use_rgb)) else None
    return _MiniRoadTrainFn.apply(model, rgb_input if model.use_rgb else None, flow, *params)


class _OadLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, reduction="mean"):
        loss, dl = oad_loss(logits, target, want_grad=True, reduction=reduction)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


def oad_loss_autograd(logits, target, reduction="mean"):
    return _OadLossFn.apply(logits, target, reduction)
