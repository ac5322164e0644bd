"""load_module / move_to — the reference's class-plugin mechanism (magicdrive/misc/common.py:11-15, 18-40):
config strings such as `magicdrive_amd.pipeline.pipeline_bev_controlnet.StableDiffusionBEVControlNetPipeline`
resolve to classes exactly as `cfg.model.pipe_module` does in tools/test.py."""
import importlib

import torch


def load_module(name: str):
    p, m = name.rsplit(".", 1)
    mod = importlib.import_module(p)
    return getattr(mod, m)


def move_to(obj, device, filter=lambda x: True):
    if torch.is_tensor(obj):
        return obj.to(device) if filter(obj) else obj
    if isinstance(obj, dict):
        return {k: move_to(v, device, filter) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(move_to(v, device, filter) for v in obj)
    if obj is None:
        return obj
    raise TypeError(f"Invalid type {obj.__class__} for move_to.")
