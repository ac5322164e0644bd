"""StableDiffusionBEVControlNetPipeline — drop-in for magicdrive/pipeline/pipeline_bev_controlnet.py.

`__call__` keeps the reference signature (:115-141) and return contract (:466-498: `.images` is
List[List[PIL]] (B x N_cam) for output_type="pil", ndarray (B,N,H,W,3) for "np", latents (B,N,4,h,w) for
"latent").  The 50-step loop (:349-451) — BEV-ControlNet, multi-view UNet, CFG combine, scheduler step — is
one libmdx op program per step (magicdrive_amd.denoiser.SamplerPlan), replayed as a hipGraph; everything
timestep-independent runs once in a prologue program.

Scope notes: CLIP text encoding is outside the built hot path (an ordinary torch module when the caller supplies
it; `prompt_embeds=` bypasses it).  The VAE decode runs on the HIP kernels: `from_pretrained` attaches
`magicdrive_amd.networks.autoencoder_kl.AutoencoderKL` from `<sd15>/vae` exactly where the reference's `from_pretrained` attaches
diffusers' (misc/test_utils.py:118-126); a caller-supplied torch VAE is still accepted.
Fused schedulers: DDIM eta=0 (the north-star config) and UniPC (what tools/test.py installs); others raise.
"""
from __future__ import annotations

import inspect
import json
import os
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np
import torch

from .. import _lib
from ..denoiser import PlanCache, SamplerPlan, device_stream
from ..schedulers import DDIMScheduler, UniPCMultistepScheduler


@dataclass
class BEVStableDiffusionPipelineOutput:
    images: Union[List[List[Any]], np.ndarray, torch.Tensor]
    nsfw_content_detected: Optional[List[bool]]


class _NullBar:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, n=1):
        pass


class StableDiffusionBEVControlNetPipeline:
    def __init__(self, vae=None, text_encoder=None, unet=None, controlnet=None, scheduler=None, tokenizer=None,
                 safety_checker=None, feature_extractor=None, requires_safety_checker: bool = False):
        assert safety_checker is None, "Please do not use safety_checker."          # pipeline_bev_controlnet.py:63
        self.vae, self.text_encoder, self.unet, self.controlnet = vae, text_encoder, unet, controlnet
        self.scheduler = scheduler if scheduler is not None else DDIMScheduler()
        self.tokenizer = tokenizer
        self.vae_scale_factor = 8
        self._progress_bar_config: Dict[str, Any] = {}
        self._device = torch.device("cpu")
        # Plans (op programs + activation pool + captured hipGraph) are cached per batch geometry.  The reference's eval flow pads
        # boxes to the per-batch maximum (configs/runner/default.yaml:61 bbox_max_length null), so L_box changes from batch to batch:
        # the cache is a small LRU and an evicted plan releases its graph and buffers.
        self._plans = PlanCache()
        self.use_graph = True
        # Scene chunks replayed concurrently on separate HIP streams (see __call__): the library option STREAMS (csrc/options.h; `pipe.streams = n`
        # overrides per pipeline); chunks hold at least `min_scenes_per_stream` scenes (below that a launch has too few tiles to fill the chip
        # even alone).  Measured on one box (profiles/r04_streams_ab.log, configs[1]): 128 scenes 7.02 -> 7.21 scenes/s, 192 scenes 7.14 -> 7.23.
        self.streams = None                                   # None: follow the option
        self.min_scenes_per_stream = 16
        # The small-batch operating point (the reference's own flows run bs = 1...4): up to this many scenes per call the ControlNet and the UNet
        # encoder of every step — independent until the zero-convs — replay side by side on two streams (SamplerPlan.launch_step).  Measured
        # (profiles/r05_fork_ab.log, 50-step DDIM, s per call without / with): 1 scene 0.656 / 0.559 (-15 %), 2: 0.849 / 0.747, 4: 1.196 / 1.071,
        # 8: 1.810 / 1.661, 16: 2.749 / 2.582 (-6 %).  From 32 scenes a call is split into scene chunks on two streams instead (above).
        self.fork_max_scenes = 31
        self.fork_chunks = False                              # also fork inside every scene chunk of a multi-stream call (A/B switch: tools/streams_ab.py)
        self._side: Dict[Any, List[Any]] = {}

    # ---- construction / housekeeping the reference's callers use (misc/test_utils.py:94-138) ----
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, controlnet=None, unet=None, safety_checker=None,
                        feature_extractor=None, torch_dtype=torch.bfloat16, vae=None, text_encoder=None, tokenizer=None, **kw):
        """`pipe_cls.from_pretrained(<sd15 dir>, controlnet=, unet=, safety_checker=None, feature_extractor=None, torch_dtype=)` —
        the call `build_pipe` makes (magicdrive/misc/test_utils.py:118-126).  Reads the SD-1.5 directory layout: `scheduler/
        scheduler_config.json`, `vae/` (decoded on the HIP kernels), `text_encoder/` + `tokenizer/` (transformers; optional)."""
        root = pretrained_model_name_or_path
        sch = DDIMScheduler()
        p = os.path.join(root, "scheduler", "scheduler_config.json")
        if os.path.exists(p):
            with open(p) as f:
                sch = DDIMScheduler.from_config(json.load(f))
        if vae is None and os.path.exists(os.path.join(root, "vae", "config.json")):
            from ..networks.autoencoder_kl import AutoencoderKL
            vae = AutoencoderKL.from_pretrained(os.path.join(root, "vae"), torch_dtype=torch_dtype)
        if text_encoder is None and os.path.isdir(os.path.join(root, "text_encoder")):
            import transformers
            text_encoder = transformers.CLIPTextModel.from_pretrained(os.path.join(root, "text_encoder")).eval()
            if torch_dtype is not None:
                text_encoder = text_encoder.to(dtype=torch_dtype)
        if tokenizer is None and os.path.isdir(os.path.join(root, "tokenizer")):
            import transformers
            tokenizer = transformers.CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer"))
        for m in (unet, controlnet):
            if m is not None and torch_dtype is not None:
                m.to(torch_dtype)
        return cls(vae=vae, text_encoder=text_encoder, unet=unet, controlnet=controlnet, scheduler=sch, tokenizer=tokenizer)

    def to(self, device):
        self._device = torch.device(device)
        for m in (self.unet, self.controlnet):
            if m is not None:
                m.to(self._device)
        for m in (self.vae, self.text_encoder):
            if m is not None and hasattr(m, "to"):
                m.to(self._device)
        self._plans.clear()
        return self

    def _side_streams(self, device, n: int):
        """`n` extra HIP streams on `device`, created once per pipeline."""
        have = self._side.setdefault(str(device), [])
        while len(have) < n:
            have.append(torch.cuda.Stream(device=device))
        return have[:n]

    @property
    def device(self):
        return self._device

    @property
    def _execution_device(self):
        return self._device

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        """Accepted for call-site compatibility (misc/test_utils.py:131-132); attention is always the fused HIP kernel."""

    def enable_vae_slicing(self):
        if self.vae is not None and hasattr(self.vae, "enable_slicing"):
            self.vae.enable_slicing()

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    def progress_bar(self, iterable=None, total=None):
        if self._progress_bar_config.get("disable", True):
            return _NullBar() if iterable is None else iterable
        from tqdm.auto import tqdm
        return tqdm(iterable, **self._progress_bar_config) if iterable is not None else tqdm(total=total, **self._progress_bar_config)

    # ---- helpers with the behaviour of pipeline_controlnet.py:285-437, 634-680 ----
    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_cfg, negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None):
        if prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise ValueError("pass prompt_embeds= (and negative_prompt_embeds=) or construct the pipeline with text_encoder + tokenizer")
            prompts = [prompt] if isinstance(prompt, str) else list(prompt)
            tok = self.tokenizer(prompts, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True, return_tensors="pt")
            prompt_embeds = self.text_encoder(tok.input_ids.to(device))[0]
        bs = prompt_embeds.shape[0]
        prompt_embeds = prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
        if do_cfg:
            if negative_prompt_embeds is None:
                if self.text_encoder is None or self.tokenizer is None:
                    raise ValueError("classifier-free guidance needs negative_prompt_embeds= when no text encoder is attached")
                neg = [""] * bs if negative_prompt is None else ([negative_prompt] * bs if isinstance(negative_prompt, str) else list(negative_prompt))
                tok = self.tokenizer(neg, padding="max_length", max_length=prompt_embeds.shape[1], truncation=True, return_tensors="pt")
                negative_prompt_embeds = self.text_encoder(tok.input_ids.to(device))[0]
            negative_prompt_embeds = negative_prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
        return prompt_embeds, negative_prompt_embeds

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        """pipeline_controlnet.py:665-680 + utils/torch_utils.py:36-77: drawn on the generator's device (CPU by
        default, so seeds are device independent), then moved."""
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, (list, tuple)) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            if isinstance(generator, (list, tuple)):      # one generator per scene (`fix_seed_within_batch`, misc/test_utils.py:224-237):
                gdev = generator[0].device                 # randn_tensor draws (1, ...) from each in turn (torch_utils.py:64-71)
                latents = torch.cat([torch.randn((1,) + shape[1:], generator=g, device=gdev, dtype=dtype) for g in generator], dim=0)
            else:
                gdev = generator.device if isinstance(generator, torch.Generator) else torch.device("cpu")
                latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype)
        else:
            assert tuple(latents.shape) == shape, f"latents {tuple(latents.shape)} != {shape}"
        return latents.to(device) * self.scheduler.init_noise_sigma

    def prepare_extra_step_kwargs(self, generator, eta):
        """Same guard as the reference (:82-98): a scheduler whose step() takes `generator` is refused."""
        params = set(inspect.signature(self.scheduler.step).parameters.keys())
        if "generator" in params:
            raise RuntimeError("If you fixed the logic for generator, please remove this. Otherwise, please use other sampler.")
        return {"eta": eta} if "eta" in params else {}

    def decode_latents(self, latents):
        """pipeline_bev_controlnet.py:100-112 (5-D latents); VAE is a caller-supplied torch module."""
        if self.vae is None:
            raise ValueError("no VAE attached: use output_type='latent' or pass vae= to the pipeline")
        sf = getattr(getattr(self.vae, "config", None), "scaling_factor", 0.18215)
        bs = latents.shape[0]
        x = (latents / sf).reshape(-1, *latents.shape[2:])
        vdt = next(self.vae.parameters()).dtype
        image = self.vae.decode(x.to(vdt)).sample
        image = image.reshape(bs, -1, *image.shape[1:])
        image = (image / 2 + 0.5).clamp(0, 1)
        return image.cpu().permute(0, 1, 3, 4, 2).float().numpy()

    @staticmethod
    def numpy_to_pil_double(images):
        from PIL import Image
        out = []
        for imgs in images:
            arr = (imgs * 255).round().astype("uint8")
            out.append([Image.fromarray(a) for a in arr])
        return out

    # ---- the sampler ----
    def _plan_config(self) -> Dict[str, Any]:
        """Architecture config of the sampler plan: the UNet's, with the conditioning encoders' geometry taken from the ControlNet."""
        cfg = dict(self.unet.cfg)
        cfg["controlnet"] = self.controlnet.cfg["controlnet"]
        return cfg

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None], image: torch.Tensor, camera_param: Optional[torch.Tensor],
                 height: int, width: int, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt: Optional[Union[str, List[str]]] = None, num_images_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 output_type: Optional[str] = "pil", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: int = 1,
                 cross_attention_kwargs: Optional[Dict[str, Any]] = None, controlnet_conditioning_scale: float = 1,
                 guess_mode: bool = False, use_zero_map_as_unconditional: bool = False, bev_controlnet_kwargs={},
                 bbox_max_length=None, _given_view=None):
        """`_given_view` = (conditional_latents, change_every_input) is how StableDiffusionBEVControlNetGivenViewPipeline reuses
        this body; not part of the reference signature."""
        if guess_mode:
            # the reference's own guess path cannot run either: under CFG it feeds the ControlNet 5-D `latents` with a 2x-batched camera_param
            # and the un-repeated prompt (pipeline_bev_controlnet.py:366-373) — refused rather than guessed at
            raise NotImplementedError("guess_mode is outside the built hot path")
        if cross_attention_kwargs:
            # the reference forwards these to the UNet's attention processors (:420); the fused attention has no such hooks — refuse,
            # like attention_mask, instead of silently ignoring them
            raise NotImplementedError(f"cross_attention_kwargs={cross_attention_kwargs!r}: the fused attention kernels take no processor kwargs")
        if eta != 0.0:
            raise NotImplementedError("the fused samplers are deterministic (DDIM eta = 0, UniPC)")
        if isinstance(self.scheduler, DDIMScheduler):
            sched_kind = "ddim"
        elif isinstance(self.scheduler, UniPCMultistepScheduler):
            sched_kind = "unipc"
        else:
            raise NotImplementedError(f"{type(self.scheduler).__name__}: fused steps exist for magicdrive_amd.schedulers.DDIMScheduler and UniPCMultistepScheduler")
        if self._device.type != "cuda":
            raise RuntimeError("pipeline.to('cuda') first: the sampler has no CPU path")
        if output_type != "latent" and self.vae is None:      # fail BEFORE the sampling loop, not after it
            raise ValueError("no VAE attached: use output_type='latent', pass vae= to the pipeline, or build it with from_pretrained(<sd15 dir>)")
        device = self._device
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        do_cfg = guidance_scale > 1.0
        n_cam = len(self.unet.cfg["neighboring_view_pair"])
        if camera_param is None:                                       # :260-264
            camera_param = self.controlnet.uncond_cam_param((batch_size, n_cam))
            do_cfg = False
        prompt_embeds, negative_prompt_embeds = self._encode_prompt(prompt, device, num_images_per_prompt, do_cfg, negative_prompt,
                                                                    prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds)
        b = batch_size * num_images_per_prompt
        image = image.to(device, torch.float32)
        if image.shape[0] == 1 and b > 1:
            image = image.expand(b, *image.shape[1:])
        assert image.shape[0] == b, f"image batch {image.shape[0]} != {b}"
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        num_channels_latents = self.unet.config.in_channels
        latents = self.prepare_latents(b, num_channels_latents, height, width, prompt_embeds.dtype, device, generator, latents)
        self.prepare_extra_step_kwargs(generator, eta)
        assert camera_param.shape[0] == batch_size, f"Except {batch_size} camera params, but you have bs={len(camera_param)}"
        n_cam = camera_param.shape[1]
        latents = torch.stack([latents] * n_cam, dim=1)                 # the SAME noise for every view (:326)
        camera_param = camera_param.to(device)
        boxes = dict(bev_controlnet_kwargs).get("bboxes_3d_data", None)
        text = prompt_embeds
        if do_cfg:
            kw = self.controlnet.add_uncond_to_kwargs(camera_param=camera_param, image=image, max_len=bbox_max_length,
                                                      bboxes_3d_data=boxes)
            camera_param, boxes = kw["camera_param"], kw["bboxes_3d_data"]
            uncond_image = torch.zeros_like(image) if use_zero_map_as_unconditional else kw["image"]
            image = torch.cat([uncond_image, image])
            text = torch.cat([negative_prompt_embeds.to(device), prompt_embeds.to(device)])
        # without CFG the reference never looks at bbox_max_length: the padding lives inside add_uncond_to_kwargs, which only the CFG branch
        # calls (pipeline_bev_controlnet.py:330-343) — ignored here too
        L_box = 0 if boxes is None else int(boxes["bboxes"].shape[2])
        h, w = latents.shape[-2:]
        n_steps = len(timesteps)
        gv_mode, gv_mask, gv_lat = 0, None, None
        if _given_view is not None:
            cond_lat, every_input = _given_view
            assert len(cond_lat) == b and all(len(r) == n_cam for r in cond_lat), "conditional_latents must be a B x N_cam list"
            gv_mode = 1 if every_input else 2
            gv_mask = torch.tensor([[v is not None for v in r] for r in cond_lat], dtype=torch.bool)
            gv_lat = torch.zeros(b, n_cam, *latents.shape[2:], dtype=torch.float32, device=device)
            for i, r in enumerate(cond_lat):
                for j, v in enumerate(r):
                    if v is not None:
                        gv_lat[i, j] = v.to(device, torch.float32)
        # ---- scene chunks: `self.streams` HIP streams, each replaying its own plan over a contiguous share of the scenes -------------
        # Scenes are independent (cross-view attention couples only the cameras of one scene, the CFG halves of a scene stay together),
        # and every kernel of a step owns whole CUs: on ONE stream the tail of each launch (the last partial round of tiles) and the
        # boundary between two launches leave CUs idle ~620 times per step.  Two half-batch graph replays in flight on two streams
        # fill those gaps with the other half's kernels (profiles/r04_streams_ab.log).  Callbacks need the whole
        # batch at a step boundary, so they keep the single-stream path.
        n_chunk = 1
        n_streams = max(1, int(self.streams if self.streams is not None else _lib.get_option("STREAMS")))
        if n_streams > 1 and callback is None and b >= 2 * self.min_scenes_per_stream:
            # every chunk's plan must stay cached for the whole call: the LRU evicts (and RELEASES) the oldest plan when it overflows, which
            # with a cache smaller than the chunk count would be chunk 0's, still held in `plans` below
            n_chunk = min(n_streams, b // self.min_scenes_per_stream, self._plans.maxsize)
        bounds = [(b * i) // n_chunk for i in range(n_chunk + 1)]
        c_halves = 2 if do_cfg else 1

        def rows(t, s0, s1):      # rows of scenes [s0, s1) of a tensor that holds the [uncond | cond] halves when CFG is on
            if t is None or n_chunk == 1:
                return t
            return torch.cat([t[hf * b + s0:hf * b + s1] for hf in range(c_halves)]) if c_halves == 2 else t[s0:s1]

        pdt = self.unet.packed().dtype
        fork = (n_chunk == 1 and 0 < b <= int(self.fork_max_scenes)) or (n_chunk > 1 and bool(self.fork_chunks))
        plans = []
        for ci in range(n_chunk):
            s0, s1 = bounds[ci], bounds[ci + 1]
            # the packed nets' identity and 16-bit type are part of the key: `pipe.unet.to(torch.float16)` re-packs the weights, and a plan
            # built on the old PackedNet would keep running (and keep alive) the old ones
            key = (s1 - s0, ci, do_cfg, L_box, h, w, n_steps, float(guidance_scale), float(controlnet_conditioning_scale), text.shape[1], sched_kind,
                   gv_mode, pdt, id(self.unet.packed()), id(self.controlnet.packed()), fork)
            plan = self._plans.get(key)
            if plan is None:
                with torch.cuda.device(device):
                    # the UNet's config.json knows nothing about the conditioning encoders: their geometry (box MLP widths, map embedder
                    # class / size, camera frequencies) is the ControlNet checkpoint's (a tiny or a 272x736 `...Plus` checkpoint loaded with
                    # from_pretrained would otherwise be planned with the SD-1.5 defaults)
                    plan_cfg = self._plan_config()
                    plan = SamplerPlan(plan_cfg, self.unet.packed(), self.controlnet.packed(), device, s1 - s0, do_cfg, L_box, (h, w),
                                       num_steps=n_steps, guidance_scale=guidance_scale,
                                       conditioning_scale=float(controlnet_conditioning_scale), n_text=text.shape[1], scheduler_kind=sched_kind,
                                       given_view_mode=gv_mode, fork=fork)
                    plan.compile()
                self._plans.put(key, plan)
            plan.load_inputs(latents[s0:s1], rows(camera_param, s0, s1), rows(text, s0, s1), rows(image, s0, s1),
                             None if boxes is None else {k: rows(v, s0, s1) for k, v in boxes.items()}, timesteps,
                             self.scheduler.coefficient_table(), given_mask=None if gv_mask is None else gv_mask[s0:s1],
                             given_latents=None if gv_lat is None else gv_lat[s0:s1])
            plans.append(plan)
        with torch.cuda.device(device):                       # launches, graph replays and torch copies all target the pipeline's device
            main = torch.cuda.current_stream(device)
            side_all = self._side_streams(device, (n_chunk - 1) + (n_chunk if fork else 0))
            side, fside = side_all[:n_chunk - 1], side_all[n_chunk - 1:]      # chunk streams beside the caller's; one fork stream per chunk
            if side:
                ready = torch.cuda.Event()
                ready.record(main)                            # the inputs were loaded on the caller's stream
                for s_ in side:
                    s_.wait_event(ready)
            sts = [main.cuda_stream] + [s_.cuda_stream for s_ in side]
            for plan, st in zip(plans, sts):
                plan.prologue.run(st)
            with self.progress_bar(total=num_inference_steps) as bar:
                for i, t in enumerate(timesteps):
                    if fork:                                  # two branches per step and chunk (ControlNet on the chunk's fork stream)
                        for plan, cst, fst in zip(plans, [main] + list(side), fside):
                            plan.launch_step(cst, fst, self.use_graph)
                    else:
                        for plan, st in zip(plans, sts):
                            if self.use_graph:
                                plan.step.launch(st)
                            else:
                                plan.step.run(st)
                    bar.update()
                    if callback is not None and i % callback_steps == 0:
                        callback(i, t, plans[0].latents().to(prompt_embeds.dtype))
            for s_ in side_all:                               # the caller's stream owns the result: join the side streams into it
                done = torch.cuda.Event()
                done.record(s_)
                main.wait_event(done)
            latents = (plans[0].latents() if n_chunk == 1 else torch.cat([pl.latents_on(main) for pl in plans])).to(prompt_embeds.dtype)
        if output_type == "latent":
            out, nsfw = latents, None
        else:
            out = self.decode_latents(latents)
            nsfw = None
            if output_type == "pil":
                out = self.numpy_to_pil_double(out)
        if not return_dict:
            return (out, nsfw)
        return BEVStableDiffusionPipelineOutput(images=out, nsfw_content_detected=nsfw)
