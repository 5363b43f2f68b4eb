#!/usr/bin/env python3
"""CLI mirror of step_recognition/main.py (flags, yaml (+) argparse merge, seed 20, eval branch, train branch, best-mAP
checkpointing) on the MI355X path.  Differences on purpose: the device is `cuda:$LOCAL_RANK` (the reference hard-codes
"cuda:1", main.py:33), TensorBoard and the never-stepped lr scheduler are not wired, `--amp` keeps the reference's GradScaler
protocol (train.py:10-18) around a path whose MFMA operands are bf16 either way, and `torchrun` launches give clip-sharded
data-parallel training.

    python -m prego_amd.main --config step_recognition/configs/miniroad_assembly101-O.yaml --eval ckpt.pth
"""
from __future__ import annotations

import argparse
import logging
import os
import os.path as osp
import random

import numpy as np
import torch
import yaml


def set_seed(seed):                      # utils/util.py:25-35
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def create_outdir(result_path):          # utils/util.py:16-24
    i, new = 1, result_path
    while osp.exists(new):
        new = f"{result_path}_{i}"
        i += 1
    os.makedirs(osp.join(new, "ckpts"))
    os.makedirs(osp.join(new, "runs"))
    return new


def get_logger(output_path):             # utils/logger.py:4-16
    logger = logging.getLogger("prego_amd")
    logger.setLevel(logging.DEBUG)
    ch = logging.StreamHandler()
    ch.setLevel(logging.INFO)
    logger.addHandler(ch)
    fh = logging.FileHandler(os.path.join(output_path, "log.txt"))
    fh.setLevel(logging.INFO)
    logger.addHandler(fh)
    return logger


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default="./configs/miniroad_thumos_kinetics.yaml")
    parser.add_argument("--eval", type=str, default=None)
    parser.add_argument("--amp", action="store_true")
    parser.add_argument("--tensorboard", action="store_true")
    parser.add_argument("--lr_scheduler", action="store_true")
    parser.add_argument("--no_rgb", action="store_true")
    parser.add_argument("--no_flow", action="store_true")
    parser.add_argument("--compute_dtype", default="fp16", choices=["fp16", "bf16", "fp32"])
    args = parser.parse_args(argv)
    cfg = yaml.load(open(args.config), Loader=yaml.FullLoader)
    cfg.update(vars(args))                                           # main.py:28-30

    from . import distributed as D
    from .data import build_data_loader
    from .registry import build_criterion, build_eval, build_model, build_trainer
    from . import evaluate, loss, model, trainer  # noqa: F401  (register)

    rank, local_rank, world = D.init_from_env()
    set_seed(20)                                                     # main.py:32
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    cfg["assume_zero_flow"] = cfg["flow_type"] == "flow_anet_resnet50" and not cfg["no_flow"]   # dataset.py:63-69
    identifier = f'{cfg["model"]}_{cfg["data_name"]}_{cfg["feature_pretrained"]}_flow{not cfg["no_flow"]}'
    result_path = create_outdir(osp.join(cfg["output_path"], identifier)) if rank == 0 else None
    logger = get_logger(result_path) if rank == 0 else logging.getLogger("prego_amd.null")
    logger.info(cfg)
    testloader = build_data_loader(cfg, mode="test")
    net = build_model(cfg, device)
    evaluate_fn = build_eval(cfg)
    if args.eval is not None:
        net.load_state_dict(torch.load(args.eval, map_location=device))
        mAP = evaluate_fn(net, testloader, logger, device)
        logger.info(f'{cfg["task"]} result: {mAP * 100:.2f} m{cfg["metric"]}')
        return mAP
    trainloader = build_data_loader(cfg, mode="train")
    criterion = build_criterion(cfg, device)
    train_one_epoch = build_trainer(cfg)
    if cfg["optimizer"] == "AdamW":          # main.py:62-67, arithmetic on the HIP path (one fused launch, csrc/optim.hip)
        from .optim import FusedAdamW
        optimizer = FusedAdamW([{"params": list(net.parameters()), "initial_lr": cfg["lr"]}], lr=cfg["lr"],
                               weight_decay=cfg["weight_decay"], model=net)
    else:
        optimizer = torch.optim.Adam([{"params": net.parameters(), "initial_lr": cfg["lr"]}], lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    scaler = torch.amp.GradScaler("cuda") if args.amp else None          # main.py:75
    best_mAP, best_epoch = 0, 0
    for epoch in range(1, cfg["num_epoch"] + 1):
        epoch_loss = train_one_epoch(trainloader, net, criterion, optimizer, scaler, epoch, device, None, scheduler=None)
        trainloader.dataset._init_features()
        mAP = evaluate_fn(net, testloader, logger, device)
        if rank == 0 and mAP > best_mAP:
            best_mAP, best_epoch = mAP, epoch
            torch.save(net.state_dict(), osp.join(result_path, "ckpts", "best.pth"))
            logger.info(f'Epoch {epoch} mAP: {mAP * 100:.2f} | Best mAP: {best_mAP * 100:.2f} at epoch {best_epoch} | '
                        f'train_loss: {epoch_loss / len(trainloader):.4f}')
    if rank == 0 and best_epoch:
        os.rename(osp.join(result_path, "ckpts", "best.pth"), osp.join(result_path, "ckpts", f"best_{best_mAP * 100:.2f}.pth"))
    return best_mAP


if __name__ == "__main__":
    main()
