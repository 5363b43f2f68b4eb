"""cfg-dict contract of the reference (step_recognition/configs/*.yaml merged with
argparse at main.py:28-30).  Key names are kept verbatim."""
from __future__ import annotations

# model/rnn/rnn.py:6-16 (same table in ViT.py:13-23 and datasets/dataset.py:11-21)
FEATURE_SIZES = {
    "rgb_anet_resnet50": 2048,
    "flow_anet_resnet50": 2048,
    "rgb_kinetics_bninception": 1024,
    "flow_kinetics_bninception": 1024,
    "rgb_kinetics_resnet50": 2048,
    "flow_kinetics_resnet50": 2048,
    "flow_nv_kinetics_bninception": 1024,
    "rgb_kinetics_i3d": 2048,
    "flow_kinetics_i3d": 2048,
}


def rgb_dim(cfg: dict) -> int:
    return 0 if cfg.get("no_rgb", False) else FEATURE_SIZES[cfg["rgb_type"]]


def flow_dim(cfg: dict) -> int:
    return 0 if cfg.get("no_flow", False) else FEATURE_SIZES[cfg["flow_type"]]


def input_dim(cfg: dict) -> int:
    """rnn.py:23-29: input_dim = rgb (+ flow) feature sizes."""
    return rgb_dim(cfg) + flow_dim(cfg)


def assembly101_cfg(**over) -> dict:
    """configs/miniroad_assembly101-O.yaml:1-27 + argparse defaults (main.py:16-24)."""
    cfg = dict(
        model="MiniROAD", data_name="ASSEMBLY101-O", task="OAD", loss="NONUNIFORM",
        metric="AP", optimizer="AdamW", device="cuda:0", feature_pretrained="kinetics",
        root_path="Assembly101-O", rgb_type="rgb_anet_resnet50", flow_type="flow_anet_resnet50",
        annotation_type="target_perframe",
        video_list_path="step_recognition/data_info/video_list.json",
        output_path="step_recognition/checkpoint_miniROAD/Assembly101-O",
        window_size=128, batch_size=16, test_batch_size=1, num_epoch=10, lr=0.0001,
        weight_decay=0.05, num_workers=4, dropout=0.20, num_classes=86,
        embedding_dim=2048, hidden_dim=1024, num_layers=1, stride=4,
        eval=None, amp=False, tensorboard=False, lr_scheduler=False,
        no_rgb=False, no_flow=False, config=None,
    )
    cfg.update(over)
    return cfg


def epic_tent_cfg(**over) -> dict:
    """configs/miniroad_epic-tent-O.yaml: differs in data_name/root/output/num_classes."""
    cfg = assembly101_cfg(
        data_name="EPIC-TENT-O", root_path="Epic-tent-O", num_classes=12,
        output_path="step_recognition/checkpoint_miniROAD/Epic-tent-O",
    )
    cfg.update(over)
    return cfg
