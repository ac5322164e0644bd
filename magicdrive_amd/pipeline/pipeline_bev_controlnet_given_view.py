"""StableDiffusionBEVControlNetGivenViewPipeline — drop-in for magicdrive/pipeline/pipeline_bev_controlnet_given_view.py.

Same sampler as StableDiffusionBEVControlNetPipeline with some views' clean latents known (`conditional_latents`, a
B x N_cam list of (C, h, w) tensors or None).  The reference re-noises the known views on the host at the top of every
iteration (:280-291, `conditional_latents_change_every_input=True`) or overrides their noise prediction (:380-390, False);
here both are branches of the fused CFG + DDIM kernel (MdxDdimDesc.gv_*), so the loop stays one hipGraph replay per step.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Union

import torch

from .pipeline_bev_controlnet import BEVStableDiffusionPipelineOutput, StableDiffusionBEVControlNetPipeline  # noqa: F401


class StableDiffusionBEVControlNetGivenViewPipeline(StableDiffusionBEVControlNetPipeline):
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None], image: torch.Tensor, camera_param: Optional[torch.Tensor],
                 height: int, width: int,
                 conditional_latents: List[List[Optional[torch.Tensor]]], conditional_latents_change_every_input=True,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt: Optional[Union[str, List[str]]] = None, num_images_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 output_type: Optional[str] = "pil", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: int = 1,
                 cross_attention_kwargs: Optional[Dict[str, Any]] = None, controlnet_conditioning_scale: float = 1,
                 guess_mode: bool = False, use_zero_map_as_unconditional: bool = False, bev_controlnet_kwargs={},
                 bbox_max_length=None):
        """Signature of pipeline_bev_controlnet_given_view.py:26-58."""
        return super().__call__(
            prompt, image, camera_param, height, width, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
            negative_prompt=negative_prompt, num_images_per_prompt=num_images_per_prompt, eta=eta, generator=generator, latents=latents,
            prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, output_type=output_type, return_dict=return_dict,
            callback=callback, callback_steps=callback_steps, cross_attention_kwargs=cross_attention_kwargs,
            controlnet_conditioning_scale=controlnet_conditioning_scale, guess_mode=guess_mode,
            use_zero_map_as_unconditional=use_zero_map_as_unconditional, bev_controlnet_kwargs=bev_controlnet_kwargs,
            bbox_max_length=bbox_max_length, _given_view=(conditional_latents, bool(conditional_latents_change_every_input)))
