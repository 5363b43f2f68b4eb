"""String-keyed plug-in registries: the drop-in boundary of the reference
(step_recognition/utils/registry.py:6-20 and the five *_builder.py files).
These are this package's OWN registries (the reference asserts unique names,
registry.py:2, so a replacement registers here and `build_*` are used instead
of the reference's)."""
from __future__ import annotations


class Registry(dict):
    def register(self, module_name, module=None):
        if module is not None:
            assert module_name not in self, module_name
            self[module_name] = module
            return module

        def register_fn(fn):
            assert module_name not in self, module_name
            self[module_name] = fn
            return fn

        return register_fn


META_ARCHITECTURES = Registry()   # model/model_builder.py:5
CRITERIONS = Registry()           # criterions/loss_builder.py:7
TRAINER = Registry()              # trainer/train_builder.py:7
EVAL = Registry()                 # trainer/eval_builder.py:7
DATA_LAYERS = Registry()          # datasets/dataset_builder.py:9


def build_model(cfg, device=None):
    """model/model_builder.py:7-9"""
    model = META_ARCHITECTURES[cfg["model"]](cfg)
    return model.to(device)


def build_criterion(cfg, device=None):
    """criterions/loss_builder.py:9-11"""
    return CRITERIONS[cfg["loss"]](cfg).to(device)


def build_trainer(cfg):
    """trainer/train_builder.py:9-11"""
    return TRAINER[cfg["task"]]


def build_eval(cfg):
    """trainer/eval_builder.py:9-11"""
    return EVAL[cfg["task"]](cfg)
