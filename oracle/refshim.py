"""TEST INFRASTRUCTURE ONLY — loader shim for the *real* reference stack.

Imports cure-lab/MagicDrive's own `magicdrive.networks.*` / `magicdrive.pipeline.*`
and its vendored diffusers 0.17.1 straight from `/root/reference` (read-only), with
the handful of import shims SURVEY.md §8c lists, so that

  * `oracle/` (our CPU restatement) can be validated against the reference itself, and
  * `tools/make_golden.py` can generate the committed fixtures under `tests/golden/`.

`/root/reference` exists only in the authoring container: nothing in the product
(`magicdrive_amd/`), in `bench.py`, in `__graft_entry__.smoke()` or in the `-m gpu`
tests may import this module.  `available()` is False on the GPU box.
No reference source is copied here; this file only arranges `sys.modules` so the
reference imports under torch 2.10 / transformers 5.x / huggingface_hub 1.x.
"""
from __future__ import annotations

import os
import sys
import types

REF_ROOT = os.environ.get("MAGICDRIVE_REFERENCE", "/root/reference")
DIF_SRC = os.path.join(REF_ROOT, "third_party", "diffusers", "src")
DIF_PKG = os.path.join(DIF_SRC, "diffusers")

_loaded = False


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "magicdrive")) and os.path.isdir(DIF_PKG)


def _stub(name: str, path: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def load():
    """Make `import diffusers...` / `import magicdrive...` resolve to the reference.

    Returns a namespace with the classes the oracle checks need.
    """
    global _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    import importlib

    if not _loaded:
        # 1. names that newer huggingface_hub / transformers dropped
        import huggingface_hub
        import huggingface_hub.constants as hc
        if not hasattr(hc, "hf_cache_home"):
            hc.hf_cache_home = os.path.expanduser("~/.cache/huggingface")
        if not hasattr(huggingface_hub, "HfFolder"):
            class HfFolder:  # noqa: D401 - placeholder, never used offline
                @staticmethod
                def get_token():
                    return None
            huggingface_hub.HfFolder = HfFolder
        if not hasattr(huggingface_hub, "cached_download"):
            huggingface_hub.cached_download = lambda *a, **k: (_ for _ in ()).throw(
                RuntimeError("offline"))
        import transformers
        import transformers.utils as tu
        if not hasattr(tu, "FLAX_WEIGHTS_NAME"):
            tu.FLAX_WEIGHTS_NAME = "flax_model.msgpack"
        try:
            transformers.CLIPFeatureExtractor  # noqa: B018
        except Exception:
            try:
                transformers.CLIPFeatureExtractor = transformers.CLIPImageProcessor
            except Exception:
                transformers.CLIPFeatureExtractor = object

        # 2. namespace stubs so the heavy __init__ files never run
        d = _stub("diffusers", DIF_PKG)
        d.__version__ = "0.17.1"
        _stub("diffusers.pipelines", os.path.join(DIF_PKG, "pipelines"))
        sd = _stub("diffusers.pipelines.stable_diffusion",
                   os.path.join(DIF_PKG, "pipelines", "stable_diffusion"))
        cn = _stub("diffusers.pipelines.controlnet",
                   os.path.join(DIF_PKG, "pipelines", "controlnet"))

        # models first (avoids the loaders <-> unet_2d_condition cycle)
        importlib.import_module("diffusers.models")
        from diffusers.models.unet_2d_condition import UNet2DConditionModel
        from diffusers.models.modeling_utils import ModelMixin
        d.UNet2DConditionModel = UNet2DConditionModel
        d.ModelMixin = ModelMixin
        from diffusers.models import AutoencoderKL
        d.AutoencoderKL = AutoencoderKL
        import diffusers.schedulers as _sch
        d.schedulers = _sch
        for n in ("DDIMScheduler", "UniPCMultistepScheduler", "DDPMScheduler"):
            if hasattr(_sch, n):
                setattr(d, n, getattr(_sch, n))

        # pipeline output + safety checker names used by `from . import ...`
        from dataclasses import dataclass
        from typing import List, Optional, Union
        import numpy as np
        from diffusers.utils import BaseOutput

        @dataclass
        class StableDiffusionPipelineOutput(BaseOutput):
            images: Union[List, np.ndarray]
            nsfw_content_detected: Optional[List[bool]]

        sd.StableDiffusionPipelineOutput = StableDiffusionPipelineOutput

        class StableDiffusionSafetyChecker:  # placeholder type, never instantiated
            pass

        sd.StableDiffusionSafetyChecker = StableDiffusionSafetyChecker
        sc = types.ModuleType("diffusers.pipelines.stable_diffusion.safety_checker")
        sc.StableDiffusionSafetyChecker = StableDiffusionSafetyChecker
        sys.modules[sc.__name__] = sc
        sd.safety_checker = sc

        pc = importlib.import_module("diffusers.pipelines.controlnet.pipeline_controlnet")
        cn.StableDiffusionControlNetPipeline = pc.StableDiffusionControlNetPipeline
        d.StableDiffusionControlNetPipeline = pc.StableDiffusionControlNetPipeline

        if REF_ROOT not in sys.path:
            sys.path.insert(0, REF_ROOT)
        _loaded = True

    ns = types.SimpleNamespace()
    import diffusers
    ns.diffusers = diffusers
    ns.UNet2DConditionModel = diffusers.UNet2DConditionModel
    ns.DDIMScheduler = diffusers.schedulers.DDIMScheduler
    ns.UniPCMultistepScheduler = diffusers.schedulers.UniPCMultistepScheduler
    ns.unet_mv = importlib.import_module("magicdrive.networks.unet_2d_condition_multiview")
    ns.controlnet = importlib.import_module("magicdrive.networks.unet_addon_rawbox")
    ns.blocks = importlib.import_module("magicdrive.networks.blocks")
    ns.bbox_embedder = importlib.import_module("magicdrive.networks.bbox_embedder")
    ns.map_embedder = importlib.import_module("magicdrive.networks.map_embedder")
    ns.pipeline = importlib.import_module("magicdrive.pipeline.pipeline_bev_controlnet")
    return ns
