"""Common host-side behaviour of the two network classes: config + reference-layout state dict,
`from_pretrained` / `save_pretrained` in the diffusers directory layout the reference uses
(`<dir>/config.json` + `diffusion_pytorch_model.(safetensors|bin)`), `.to()`, `.dtype`.

The classes hold weights as a plain state dict (reference key names); the arithmetic lives in libmdx
programs built by magicdrive_amd.denoiser.  There is deliberately no nn.Module forward to fall back to.
"""
from __future__ import annotations

import copy
import json
import os
from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from ..engine import PackedNet
from ..denoiser import PlanCache
from . import spec

WEIGHTS_BIN = "diffusion_pytorch_model.bin"
WEIGHTS_ST = "diffusion_pytorch_model.safetensors"
CONFIG_NAME = "config.json"

_ARCH_KEYS = ("in_channels", "out_channels", "flip_sin_to_cos", "freq_shift", "down_block_types", "up_block_types",
              "block_out_channels", "layers_per_block", "norm_num_groups", "norm_eps", "cross_attention_dim", "attention_head_dim")


def arch_config_from_json(js: Dict, base: Optional[Dict] = None) -> Dict:
    """diffusers config.json -> our cfg dict (unknown keys ignored, missing keys = SD-1.5 defaults)."""
    cfg = copy.deepcopy(base or spec.SD15_CONFIG)
    for k in _ARCH_KEYS:
        if k in js and js[k] is not None:
            v = js[k]
            cfg[k] = tuple(v) if isinstance(v, list) else v
    if js.get("neighboring_view_pair"):
        cfg["neighboring_view_pair"] = {int(k): [int(x) for x in v] for k, v in js["neighboring_view_pair"].items()}
    for k in ("neighboring_attn_type", "zero_module_type"):
        if k in js:
            cfg[k] = js[k]
    unsupported = {"use_linear_projection": False, "only_cross_attention": False, "dual_cross_attention": False,
                   "class_embed_type": None, "addition_embed_type": None, "resnet_time_scale_shift": "default",
                   "upcast_attention": False, "center_input_sample": False}
    for k, ok in unsupported.items():
        if k in js and js[k] not in (ok, None) and js[k] != ok:
            raise NotImplementedError(f"config option {k}={js[k]!r} is outside the built hot path (SD-1.5 MagicDrive uses {ok!r})")
    cn = cfg["controlnet"]
    for k in ("camera_in_dim", "camera_out_dim", "map_size", "conditioning_embedding_out_channels", "uncond_cam_in_dim"):
        if k in js and js[k] is not None:
            cn[k] = tuple(js[k]) if isinstance(js[k], list) else js[k]
    if js.get("map_embedder_cls"):          # configs/exp/272x736.yaml:15-22 (BEVControlNetConditioningEmbeddingPlus)
        mp = js.get("map_embedder_param") or {}
        cn["map_embedder_cls"] = js["map_embedder_cls"]
        cn["map_embedder_param"] = {k: tuple(v) for k, v in mp.items()}
        if "conditioning_size" in mp:
            cn["map_size"] = tuple(mp["conditioning_size"])
        if "block_out_channels" in mp:
            cn["conditioning_embedding_out_channels"] = tuple(mp["block_out_channels"])
    if "cam_embedder_param" in js and js["cam_embedder_param"]:
        cn["cam_embedder_num_freqs"] = js["cam_embedder_param"].get("num_freqs", cn["cam_embedder_num_freqs"])
    if "bbox_embedder_param" in js and js["bbox_embedder_param"]:
        bp = js["bbox_embedder_param"]
        for k in ("n_classes", "class_token_dim", "embedder_num_freq"):
            if k in bp:
                cn["bbox"][k] = bp[k]
        if "proj_dims" in bp:
            cn["bbox"]["proj_dims"] = tuple(bp["proj_dims"])
        # ContinuousBBoxWithTextEmbedding's class default is minmax_normalize=True (bbox_embedder.py:42); the shipped config sets
        # false (configs/model/SDv1.5mv_rawbox.yaml:56).  A config that omits the key means the class default.
        cn["bbox"]["minmax_normalize"] = bool(bp.get("minmax_normalize", True))
        # the class default is mode='cxyz' (4 points, bbox_embedder.py:41); the shipped config sets all-xyz (8 corners,
        # configs/model/SDv1.5mv_rawbox.yaml:54); both are the same prologue ops with a different point count (bbox_embedder.py:52-57)
        mode = bp.get("mode", "cxyz")
        if mode not in ("all-xyz", "cxyz"):
            raise NotImplementedError(f"bbox_embedder mode={mode!r}")        # 'owhr' raises in the reference too (bbox_embedder.py:58-59)
        cn["bbox"]["mode"] = mode
        cn["bbox"]["n_corners"] = 8 if mode == "all-xyz" else 4
    # classifier-free-guidance map substitution (unet_addon_rawbox.py:188-202, 674-677): an `uncond_map` buffer exists in the
    # checkpoint only when use_uncond_map is set AND drop_cond_ratio > 0
    um, dr = js.get("use_uncond_map"), js.get("drop_cond_ratio", 0.0) or 0.0
    if um is not None:
        if um not in ("negative1", "random", "learnable"):
            raise TypeError(f"Unknown map type: {um}.")                 # the reference's error (:200)
        cn["use_uncond_map"] = um if dr > 0 else None
        cn["drop_cond_ratio"] = float(dr)                             # kept so that save_pretrained writes the checkpoint's own value back
    for k in ("guess_mode",):
        if js.get(k):
            raise NotImplementedError(f"config option {k}={js[k]!r} is outside the built hot path")
    return cfg


class MdxModel:
    """Base of UNet2DConditionModelMultiview / BEVControlNetModel."""

    _shape_fn = None          # staticmethod(cfg) -> OrderedDict name -> shape

    def __init__(self, cfg: Dict, state_dict: Dict[str, torch.Tensor], torch_dtype=torch.bfloat16):
        self.cfg = copy.deepcopy(cfg)
        shapes = type(self)._shape_fn(self.cfg)
        missing = [k for k in shapes if k not in state_dict]
        if missing:
            raise KeyError(f"{type(self).__name__}: state dict lacks {len(missing)} tensors, e.g. {missing[:4]}")
        for k, shp in shapes.items():
            if tuple(state_dict[k].shape) != tuple(shp):
                raise ValueError(f"{type(self).__name__}: {k} has shape {tuple(state_dict[k].shape)}, config implies {tuple(shp)}")
        self._sd = OrderedDict((k, state_dict[k].detach()) for k in shapes)
        self._dtype = torch_dtype
        self._device = torch.device("cpu")
        self._packed: Optional[PackedNet] = None
        self._plans = PlanCache()          # LRU: module-API plans are keyed by (batch, padded box count, size, ...)
        self.config = SimpleNamespace(**{k: v for k, v in self.cfg.items() if k != "controlnet"})
        self.training = False

    # ---- construction ----
    @classmethod
    def from_config(cls, cfg: Dict, seed: int = 0, torch_dtype=torch.bfloat16):
        """Seeded random weights of the given architecture (no pretrained weights exist offline)."""
        return cls(cfg, spec.random_state_dict(cls._shape_fn(cfg), seed), torch_dtype)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=torch.bfloat16, subfolder: Optional[str] = None, **unused):
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, CONFIG_NAME)) as f:
            js = json.load(f)
        cfg = arch_config_from_json(js)
        if os.path.exists(os.path.join(d, WEIGHTS_ST)):
            from safetensors.torch import load_file
            sd = load_file(os.path.join(d, WEIGHTS_ST))
        elif os.path.exists(os.path.join(d, WEIGHTS_BIN)):
            sd = torch.load(os.path.join(d, WEIGHTS_BIN), map_location="cpu")
        else:
            raise FileNotFoundError(f"no {WEIGHTS_ST} / {WEIGHTS_BIN} under {d}")
        return cls(cfg, sd, torch_dtype)

    def save_pretrained(self, path: str, safe_serialization: bool = True):
        os.makedirs(path, exist_ok=True)
        js = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.cfg.items() if k != "controlnet"}
        js["neighboring_view_pair"] = {str(k): v for k, v in self.cfg["neighboring_view_pair"].items()}
        js["_class_name"] = type(self).__name__
        self._extra_config(js)
        with open(os.path.join(path, CONFIG_NAME), "w") as f:
            json.dump(js, f, indent=2)
        sd = {k: v.contiguous() for k, v in self._sd.items()}
        sd.update({k: v.contiguous() for k, v in self._extra_tensors().items()})
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(path, WEIGHTS_ST))
        else:
            torch.save(sd, os.path.join(path, WEIGHTS_BIN))

    def _extra_config(self, js):
        pass

    def _extra_tensors(self) -> Dict[str, torch.Tensor]:
        """Tensors outside the architecture's shape table that belong in the checkpoint (e.g. the ControlNet's uncond_map)."""
        return {}

    # ---- torch-module-like surface the reference's callers touch ----
    def state_dict(self):
        return self._sd

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("magicdrive_amd builds the inference hot path only")
        return self

    def requires_grad_(self, flag: bool = False):
        return self

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    def to(self, *args, **kw):
        for a in list(args) + list(kw.values()):
            if isinstance(a, torch.dtype):
                if a not in (torch.bfloat16, torch.float16, torch.float32):
                    raise ValueError(a)
                if (a == torch.float16) != (self._dtype == torch.float16):
                    self._packed = None  # the 16-bit arithmetic type follows the requested dtype: repack (fp16 <-> bf16)
                    self._plans.clear()
                self._dtype = a          # API dtype of inputs/outputs; fp16 -> fp16 kernels, anything else -> bf16 kernels (fp32 accumulate)
            elif isinstance(a, (str, torch.device)):
                dev = torch.device(a)
                if dev != self._device:
                    self._device = dev
                    self._packed = None
                    self._plans.clear()
        return self

    def cuda(self, index: int = 0):
        return self.to(torch.device("cuda", index))

    def packed(self) -> PackedNet:
        if self._packed is None:
            # arithmetic type: fp16 when the model was asked for in fp16 (what the reference samples in, misc/test_utils.py:95), bf16
            # otherwise (incl. fp32 requests: operands are 16-bit on the MFMA path either way, accumulation is fp32)
            self._packed = PackedNet(self._sd, self._device, torch.float16 if self._dtype == torch.float16 else torch.bfloat16)
        return self._packed

    def num_parameters(self) -> int:
        return sum(v.numel() for v in self._sd.values())
