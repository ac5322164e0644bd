"""BEVControlNetModel — drop-in for magicdrive/networks/unet_addon_rawbox.py (inference path).

Keeps `forward(sample, timestep, camera_param, bboxes_3d_data, encoder_hidden_states, controlnet_cond, ...)`
(:707-724) returning (12 down residuals, mid residual, encoder_hidden_states_with_cam) (:921-932), and the CFG
helpers `uncond_cam_param` (:307-315) / `add_uncond_to_kwargs` (:625-682).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Union

import numpy as np
import torch

from ..denoiser import ControlNetPlan
from . import spec
from .base import MdxModel
from .output_cls import BEVControlNetOutput


class BEVControlNetModel(MdxModel):
    _shape_fn = staticmethod(spec.controlnet_param_shapes)

    def __init__(self, cfg, state_dict, torch_dtype=torch.bfloat16):
        super().__init__(cfg, state_dict, torch_dtype)
        # `uncond_map` (C, H, W): present in a checkpoint trained with use_uncond_map + drop_cond_ratio > 0
        # (unet_addon_rawbox.py:188-202); it replaces the BEV map of the unconditional half under CFG (:674-677).
        self._uncond_map = None
        if self.cfg["controlnet"].get("use_uncond_map"):
            if "uncond_map" not in state_dict:
                raise KeyError("BEVControlNetModel: config sets use_uncond_map but the state dict has no 'uncond_map'")
            um = state_dict["uncond_map"].detach().float()
            if tuple(um.shape) != tuple(self.cfg["controlnet"]["map_size"]):
                raise ValueError(f"uncond_map has shape {tuple(um.shape)}, config map_size is {tuple(self.cfg['controlnet']['map_size'])}")
            self._uncond_map = um

    def _extra_config(self, js):
        cn = self.cfg["controlnet"]
        js.update(camera_in_dim=cn["camera_in_dim"], camera_out_dim=cn["camera_out_dim"], map_size=list(cn["map_size"]),
                  conditioning_embedding_out_channels=list(cn["conditioning_embedding_out_channels"]),
                  uncond_cam_in_dim=list(cn["uncond_cam_in_dim"]),
                  cam_embedder_param=dict(input_dims=3, num_freqs=cn["cam_embedder_num_freqs"], include_input=True, log_sampling=True),
                  bbox_embedder_param=dict(n_classes=cn["bbox"]["n_classes"], class_token_dim=cn["bbox"]["class_token_dim"],
                                           embedder_num_freq=cn["bbox"]["embedder_num_freq"], proj_dims=list(cn["bbox"]["proj_dims"]),
                                           minmax_normalize=bool(cn["bbox"].get("minmax_normalize", False)),
                                           mode=cn["bbox"].get("mode", "all-xyz")))
        if cn.get("use_uncond_map"):
            js.update(use_uncond_map=cn["use_uncond_map"], drop_cond_ratio=float(cn.get("drop_cond_ratio") or 0.25))
        if cn.get("map_embedder_cls"):
            js.update(map_embedder_cls=cn["map_embedder_cls"], map_embedder_param={k: list(v) for k, v in cn["map_embedder_param"].items()})

    def _extra_tensors(self):
        return {} if self._uncond_map is None else {"uncond_map": self._uncond_map}

    @property
    def uncond_cam_num(self) -> int:
        return self.cfg["controlnet"]["uncond_cam_in_dim"][1]

    def uncond_cam_param(self, repeat_size: Union[List[int], int] = 1) -> torch.Tensor:
        """Learned unconditional camera, (…repeat_size, 3, 7) (unet_addon_rawbox.py:307-315)."""
        if isinstance(repeat_size, int):
            repeat_size = [1, repeat_size]
        n = int(np.prod(repeat_size))
        w = self._sd["uncond_cam.weight"].to(self._device, torch.float32)
        return w.expand(n, -1).reshape(*repeat_size, -1, self.uncond_cam_num)

    def add_uncond_to_kwargs(self, camera_param, bboxes_3d_data: Optional[Dict], image, max_len=None, **kwargs):
        """uncond in the front, cond in the tail (unet_addon_rawbox.py:625-682)."""
        batch_size, n_cam = camera_param.shape[:2]
        ret = dict()
        ret["camera_param"] = torch.cat([self.uncond_cam_param([batch_size, n_cam]).to(camera_param.device), camera_param.float()])
        if bboxes_3d_data is None:
            if max_len is not None:
                dev = camera_param.device
                ret["bboxes_3d_data"] = {
                    "bboxes": torch.zeros([batch_size * 2, n_cam, max_len, self.cfg["controlnet"]["bbox"].get("n_corners", 8), 3], device=dev),
                    "classes": torch.zeros([batch_size * 2, n_cam, max_len], device=dev, dtype=torch.long),
                    "masks": torch.zeros([batch_size * 2, n_cam, max_len], device=dev, dtype=torch.bool)}
            else:
                ret["bboxes_3d_data"] = None
        else:
            ret["bboxes_3d_data"] = dict()
            for key in ["bboxes", "classes", "masks"]:
                v = torch.cat([torch.zeros_like(bboxes_3d_data[key]), bboxes_3d_data[key]])
                if max_len is not None:
                    token_num = max_len - v.shape[2]
                    assert token_num >= 0
                    pad = torch.zeros_like(v[:, :, :1]).expand(-1, -1, token_num, *v.shape[3:])
                    v = torch.cat([v, pad], dim=2)
                ret["bboxes_3d_data"][key] = v
        if self._uncond_map is None:  # use_uncond_map is null in the shipped config (SDv1.5mv_rawbox.yaml:36): the map is kept
            ret["image"] = image
        else:                         # substitute_with_uncond_map(image, None): every sample's map <- uncond_map (:378-395, :677)
            ret["image"] = self._uncond_map.to(image.device, image.dtype)[None].expand_as(image).clone()
        for k, v in kwargs.items():
            ret[k] = v
        return ret

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep, camera_param: torch.Tensor, bboxes_3d_data: Optional[Dict],
                encoder_hidden_states: torch.Tensor, controlnet_cond: torch.Tensor, encoder_hidden_states_uncond=None,
                conditioning_scale: float = 1.0, class_labels=None, timestep_cond=None, attention_mask=None,
                cross_attention_kwargs=None, guess_mode: bool = False, return_dict: bool = True, **kwargs):
        if guess_mode:
            raise NotImplementedError("guess_mode is outside the built hot path")
        if self._device.type != "cuda":
            raise RuntimeError("BEVControlNetModel.forward needs .to('cuda'): there is no CPU path")
        b, n_cam, _, h, w = sample.shape
        L = 0 if bboxes_3d_data is None else bboxes_3d_data["bboxes"].shape[2]
        key = (b, L, h, w, float(conditioning_scale))
        plan = self._plans.get(key)
        if plan is None:
            with torch.cuda.device(self._device):
                plan = ControlNetPlan(self.cfg, self.packed(), self._device, b, L, (h, w), conditioning_scale,
                                      n_text=encoder_hidden_states.shape[1])
            self._plans.put(key, plan)
        down, mid, ctx = plan.run(sample, timestep, camera_param, bboxes_3d_data, encoder_hidden_states, controlnet_cond)
        odt = sample.dtype if sample.is_floating_point() else self._dtype
        down = [d.to(odt).clone() for d in down]
        mid = mid.to(odt).clone()
        ctx = ctx.to(odt).clone()
        if not return_dict:
            return down, mid, ctx
        return BEVControlNetOutput(down_block_res_samples=down, mid_block_res_sample=mid, encoder_hidden_states_with_cam=ctx)

    __call__ = forward
