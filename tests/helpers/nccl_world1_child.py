"""Child process of tests/test_gpu_train.py::test_rccl_allreduce_of_the_real_gradient_bucket_in_a_one_rank_group.

A FRESH interpreter (the pytest process has initialised HIP and must not be re-exec'ed; RCCL wants its own process group):
rank 0 of a one-rank `nccl` (= RCCL) process group on cuda:0.  With PREGO_DP_FORCE_COLLECTIVE=1 the trainer's data-parallel path
runs although the world is 1, so RCCL init, the comm side stream, the backward's milestone events and ncclAllReduce over the REAL
gradient bucket (17 926 230 fp32 = 71.7 MB, three sub-buckets) all execute on hardware.  A one-rank sum is the identity, hence:
fp32 wire format -> gradients bit-identical to the plain backward; bf16 wire format -> within bf16 rounding.  Then two optimizer
steps through TRAINER["OAD"] (all-reduce enqueued BEFORE the engine check) against the same steps without a process group.
Prints one JSON line."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from prego_amd import weights as W
    from prego_amd.config import assembly101_cfg
    from prego_amd.optim import FusedAdamW
    from prego_amd.registry import build_criterion, build_model, build_trainer
    from prego_amd.trainer import _allreduce_grads
    import prego_amd.loss, prego_amd.model, prego_amd.trainer  # noqa: F401,E401

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    B, T = 16, 128                                     # configs/miniroad_assembly101-O.yaml: the real training shape
    rgb = torch.from_numpy(W.tsn_features((B, T, 2048), 20, "g4c.rgb")).cuda()
    flow = torch.from_numpy(W.tsn_features((B, T, 2048), 20, "g4c.flow")).cuda()
    cls = (W.uniform01((B, T), 20, "g4c.tgt") * 86).astype(np.int64)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(cls), 86).float().cuda()
    out = {"bucket_bytes": None}

    def build(compress):
        cfg = assembly101_cfg(dropout=0.0, compute_dtype="bf16", grad_compress=compress)
        m = build_model(cfg, "cuda:0")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
        return cfg, m, build_criterion(cfg, "cuda:0")

    def grads(m, crit, reduce):
        m.train()
        loss = crit(m(rgb, flow), tgt)
        loss.backward()
        if reduce:
            _allreduce_grads(m)
        torch.cuda.synchronize()
        m.engine(train=True).check()
        return {k: p.grad.detach().clone() for k, p in m.named_parameters()}

    # (1) no process group: the plain backward
    _, m0, c0 = build(None)
    ref = grads(m0, c0, False)
    # (2) one-rank RCCL group, collective forced
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    os.environ["PREGO_DP_FORCE_COLLECTIVE"] = "1"
    calls = []
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):               # counts what goes through; the REAL collective still runs
        calls.append((int(t.numel()), str(t.dtype), bool(t.is_cuda)))
        return real_all_reduce(t, *a, **k)
    dist.all_reduce = counting_all_reduce
    for compress in (None, "bf16"):
        calls.clear()
        _, m1, c1 = build(compress)
        got = grads(m1, c1, True)
        eng = m1.engine(train=True)
        assert eng._grad_events is not None and len(eng._grad_bounds) == 3
        assert len(calls) == 3 and all(c[2] for c in calls), calls
        assert sum(c[0] for c in calls) == eng._grad_flat.numel()
        out["bucket_bytes"] = eng._grad_flat.numel() * 4
        want_dt = "torch.bfloat16" if compress else "torch.float32"
        assert all(c[1] == want_dt for c in calls), calls
        worst = 0.0
        for k in ref:
            if compress is None:
                assert torch.equal(got[k], ref[k]), k           # a one-rank sum / 1 is exact
            else:
                err = (got[k] - ref[k]).abs().max().item()
                lim = 2.0 ** -8 * ref[k].abs().max().item() + 1e-12
                assert err <= lim, (k, err, lim)
                worst = max(worst, err / (ref[k].abs().max().item() + 1e-30))
        out[f"allreduce_{compress or 'fp32'}"] = {"sub_buckets": [c[0] for c in calls], "wire_dtype": want_dt, "worst_rel_err": worst}
    dist.all_reduce = real_all_reduce
    # (3) two optimizer steps through the trainer, collective path on vs the reference loop (fp32 wire: identical trajectories)
    train = build_trainer(dict(assembly101_cfg(), task="OAD"))
    loader = [(rgb, flow, tgt, ("v",) * B, torch.zeros(B), torch.zeros(B))] * 2
    params = {}
    for forced in (True, False):
        os.environ["PREGO_DP_FORCE_COLLECTIVE"] = "1" if forced else "0"
        cfg, m, crit = build(None)
        opt = FusedAdamW([{"params": list(m.parameters())}], lr=1e-4, weight_decay=0.05, model=m)
        loss = train(loader, m, crit, opt, None, 1, "cuda:0")
        params[forced] = (loss, {k: p.detach().clone() for k, p in m.named_parameters()})
    assert abs(params[True][0] - params[False][0]) < 1e-6
    for k in params[True][1]:
        assert torch.equal(params[True][1][k], params[False][1][k]), k
    out["trainer_two_steps_identical"] = True
    # (4) the class-sharded AP of a multi-rank eval (Evaluate._sharded_ap: all_gather + all_to_all_single + device AP kernel) over RCCL,
    # against the single-process device AP on the same matrices
    import tempfile
    from prego_amd.evaluate import Evaluate
    from prego_amd.metrics import perframe_average_precision_device
    tmp = tempfile.mkdtemp()
    vl = os.path.join(tmp, "vl.json")
    json.dump({"ASSEMBLY101-O": {"class_index": [f"c{i}" for i in range(86)]}}, open(vl, "w"))
    ev = Evaluate(assembly101_cfg(eval=None, video_list_path=vl))
    g = torch.Generator(device="cuda").manual_seed(3)
    n = 50_000
    pred = torch.rand((n, 86), device="cuda", generator=g)
    gt = torch.nn.functional.one_hot(torch.randint(0, 86, (n,), device="cuda", generator=g), 86).float()
    sharded = ev._sharded_ap(pred, gt, 1, 0)
    single = perframe_average_precision_device(pred, gt, ev.all_class_names)
    assert sharded["mean_AP"] == single["mean_AP"] and sharded["per_class_AP"] == single["per_class_AP"]
    out["sharded_ap_equals_single_process"] = True
    out["backend"] = dist.get_backend()
    out["nccl_version"] = list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None
    dist.barrier()
    dist.destroy_process_group()
    out["ok"] = True
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
