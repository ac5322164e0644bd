#!/usr/bin/env python
"""tools/sample.py — the mm-free equivalent of the reference's sampling entry points (tools/test.py:38-72, demo/run.py:33-90):
load a checkpoint in the reference layout, read the training run's hydra overrides, feed preprocessed `.pth` samples
(demo/readme.md:3-22) through StableDiffusionBEVControlNetPipeline on the MI355X path and save the six views per scene.

  python tools/sample.py --ckpt <ckpt dir with unet/ controlnet/ hydra/overrides.yaml> --sd15 <stable-diffusion-v1-5 dir> \
         --data <folder of *.pth> --out <dir> [key=value overrides as for tools/test.py] [--scheduler unipc|ddim] [--prompt-embeds]
         [--cond-on-view]

--cond-on-view is demo/run_cond_on_view.py:34-120: the pipe class becomes StableDiffusionBEVControlNetGivenViewPipeline, the ground-truth
views of every sample (`img` of the `.pth`) are encoded with the pipeline's VAE (`vae.encode(x).latent_dist.mean * scaling_factor`, on
the HIP kernels), and generation `ti` of the `validation_times - 1` generations is sampled with view `ti` given.

No hydra / omegaconf / mmdet3d: the config tree of the reference is not rebuilt (SURVEY.md §8 out of scope) — only the handful of
keys the sampling loop reads are resolved, in the reference's order (checkpoint overrides first, command line last, tools/test.py:46-54):
  seed, runner.validation_times, runner.pipeline_param.{guidance_scale,num_inference_steps,...}, dataset.image_size / +exp=HxW,
  fix_seed_within_batch, runner.bbox_max_length.
Without a text encoder in --sd15 the prompts cannot be embedded; pass --prompt-embeds to sample with zero embeddings (plumbing /
throughput runs) instead of failing.
"""
import argparse
import os
import sys
from typing import Dict, Iterator, List, Sequence

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEFAULTS = dict(seed=42, validation_times=4, guidance_scale=2.0, num_inference_steps=20, image_size=(224, 400), fix_seed_within_batch=False,
                bbox_max_length=None)          # configs/test_config.yaml + configs/runner/default.yaml:54-61


def _parse(v: str):
    lo = v.strip().lower()
    if lo in ("null", "none", "~"):
        return None
    if lo in ("true", "false"):
        return lo == "true"
    try:
        return int(v)
    except ValueError:
        try:
            return float(v)
        except ValueError:
            return v


def resolve_run_config(ckpt_dir: str, cli_overrides: Sequence[str]) -> Dict:
    """Checkpoint overrides, then the command line's (tools/test.py:46-54), reduced to the keys the sampling loop reads."""
    from magicdrive_amd.dataset import load_overrides
    ov = load_overrides(ckpt_dir)
    for it in cli_overrides:
        k, _, v = it.partition("=")
        ov[k.lstrip("+")] = v
    run = dict(DEFAULTS)
    for k, v in ov.items():
        if k == "exp" and "x" in v:                                  # +exp=224x400 / 272x736 / 424x800abox0.1_nockpt (configs/exp/*.yaml)
            h, w = v.split("x")[:2]
            w = "".join(ch for ch in w if ch.isdigit() or ch == "_").split("_")[0]
            digits = "".join(c for c in w if c.isdigit())
            run["image_size"] = (int(h), int(digits))
        elif k == "dataset.image_size":
            run["image_size"] = tuple(int(x) for x in v.strip("[]() ").split(","))
        elif k == "seed":
            run["seed"] = _parse(v)
        elif k == "fix_seed_within_batch":
            run["fix_seed_within_batch"] = bool(_parse(v))
        elif k == "runner.validation_times":
            run["validation_times"] = int(v)
        elif k == "runner.bbox_max_length":
            run["bbox_max_length"] = _parse(v)
        elif k.startswith("runner.pipeline_param."):
            run[k[len("runner.pipeline_param."):]] = _parse(v)
    return run


def iter_pipe_kwargs(dataset, run: Dict, batch_size: int = 1, with_pixels: bool = False) -> Iterator[Dict]:
    """Batches of samples -> keyword arguments of StableDiffusionBEVControlNetPipeline.__call__, as run_one_batch /
    run_one_batch_pipe assemble them (magicdrive/misc/test_utils.py:191-255, :258-330).  with_pixels: (kwargs, pixel_values) pairs."""
    from magicdrive_amd.dataset import collate_samples, preprocess_fn
    extra = {k: v for k, v in run.items() if k not in DEFAULTS or k in ("guidance_scale", "num_inference_steps")}
    for i0 in range(0, len(dataset), batch_size):
        batch = collate_samples([preprocess_fn(dataset[i]) for i in range(i0, min(i0 + batch_size, len(dataset)))])
        kw = dict(prompt=batch["captions"], image=batch["bev_map_with_aux"], camera_param=batch["camera_param"],
                  height=run["image_size"][0], width=run["image_size"][1], bev_controlnet_kwargs=batch["kwargs"],
                  bbox_max_length=run["bbox_max_length"], **extra)
        if with_pixels:
            yield kw, batch["pixel_values"]
        else:
            yield kw


def encode_given_views(pipe, pixel_values: torch.Tensor) -> torch.Tensor:
    """pixel_values (b, n, 3, H, W) in [-1, 1] -> scaled latents (b, n, 4, H / 8, W / 8): demo/run_cond_on_view.py:77-86."""
    b, n = pixel_values.shape[:2]
    x = pixel_values.reshape(b * n, *pixel_values.shape[2:]).to(device=pipe._execution_device, dtype=pipe.vae.dtype)
    lat = pipe.vae.encode(x).latent_dist.mean * pipe.vae.config.scaling_factor
    return lat.reshape(b, n, *lat.shape[1:])


def cond_on_view_runs(pipe, kw: Dict, pixel_values: torch.Tensor, run: Dict, generator=None, fix_seed_for_every_generation: bool = False):
    """run_one_batch_pipe_given_view (demo/run_cond_on_view.py:34-120): `validation_times - 1` generations, generation `ti` with the
    encoded ground-truth view `ti` of every scene given and the other views sampled.  Yields (ti, images: List[List[PIL]])."""
    if pixel_values is None:
        raise ValueError("--cond-on-view needs the ground-truth views (`img`) in the samples")
    latents = encode_given_views(pipe, pixel_values)
    bs, n_cam = latents.shape[:2]
    for ti in range(run["validation_times"] - 1):
        conditional_latents = [[None] * n_cam for _ in range(bs)]
        for b in range(bs):
            conditional_latents[b][ti] = latents[b, ti]
        if run["seed"] is not None and fix_seed_for_every_generation:
            generator = torch.Generator().manual_seed(run["seed"])         # :97-99
        yield ti, pipe(conditional_latents=conditional_latents, generator=generator, **kw).images


def build_pipe(ckpt: str, sd15: str, scheduler: str, device, given_view: bool = False):
    """The reference's build_pipe sequence (magicdrive/misc/test_utils.py:94-138) with the magicdrive_amd config strings."""
    from magicdrive_amd import schedulers
    from magicdrive_amd.misc.common import load_module
    model_cls = load_module("magicdrive_amd.networks.unet_addon_rawbox.BEVControlNetModel")
    unet_cls = load_module("magicdrive_amd.networks.unet_2d_condition_multiview.UNet2DConditionModelMultiview")
    pipe_cls = load_module("magicdrive_amd.pipeline.pipeline_bev_controlnet.StableDiffusionBEVControlNetPipeline")
    if given_view:                                                         # the swap demo/run_cond_on_view.py:138-140 makes
        pipe_cls = load_module("magicdrive_amd.pipeline.pipeline_bev_controlnet_given_view.StableDiffusionBEVControlNetGivenViewPipeline")
    ckpt = ckpt[:-1] if ckpt.endswith("/") else ckpt
    controlnet = model_cls.from_pretrained(os.path.join(ckpt, "controlnet"), torch_dtype=torch.float16).eval()
    unet = unet_cls.from_pretrained(os.path.join(ckpt, "unet"), torch_dtype=torch.float16).eval()
    pipe = pipe_cls.from_pretrained(sd15, controlnet=controlnet, unet=unet, safety_checker=None, feature_extractor=None, torch_dtype=torch.float16)
    if scheduler == "unipc":
        pipe.scheduler = schedulers.UniPCMultistepScheduler.from_config(pipe.scheduler.config)
    pipe.enable_xformers_memory_efficient_attention()
    return pipe.to(device)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ckpt", required=True); ap.add_argument("--sd15", required=True)
    ap.add_argument("--data", required=True); ap.add_argument("--out", required=True)
    ap.add_argument("--scheduler", choices=["unipc", "ddim"], default="unipc")
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--prompt-embeds", action="store_true", help="no text encoder available: sample with zero prompt embeddings")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--reseed-per-run", action="store_true", help="deviation from the reference: an independent seed per (batch, validation run)")
    ap.add_argument("--cond-on-view", action="store_true", help="demo/run_cond_on_view.py: generation ti is sampled with the encoded ground-truth view ti given")
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args(argv)
    from magicdrive_amd.dataset import FolderSet
    run = resolve_run_config(a.ckpt, a.overrides)
    pipe = build_pipe(a.ckpt, a.sd15, a.scheduler, a.device, given_view=a.cond_on_view)
    data = FolderSet(a.data)
    os.makedirs(a.out, exist_ok=True)
    total = 0
    for item in iter_pipe_kwargs(data, run, a.batch_size, with_pixels=a.cond_on_view):
        kw, pixel_values = item if a.cond_on_view else (item, None)
        bs = kw["image"].shape[0]
        if pipe.text_encoder is None:
            if not a.prompt_embeds:
                raise SystemExit(f"{a.sd15} has no text_encoder/: pass --prompt-embeds to sample with zero embeddings")
            D = pipe.unet.cfg["cross_attention_dim"]
            kw.update(prompt=None, prompt_embeds=torch.zeros(bs, 77, D), negative_prompt_embeds=torch.zeros(bs, 77, D))
        # Seeding as the reference entry point does it (tools/test.py passes no global_generator, so run_one_batch_pipe,
        # magicdrive/misc/test_utils.py:221-237, builds torch.manual_seed(cfg.seed) ONCE per batch, before the validation_times loop: its
        # state carries across the iterations; with fix_seed_within_batch every scene of the batch gets that same generator object) —
        # the same seed reproduces the reference's initial latents.  --reseed-per-run (a deviation) draws a fresh seed per (batch, run).
        if run["seed"] is None:
            base_gen = None
        else:
            base_gen = torch.Generator().manual_seed(run["seed"])
        if a.cond_on_view:
            for ti, images in cond_on_view_runs(pipe, kw, pixel_values, run, generator=base_gen):
                for bi, views in enumerate(images):
                    for vi, im in enumerate(views):
                        im.save(os.path.join(a.out, f"{total + bi}_gen{ti}_view{vi}.png"))
            total += bs
            continue
        for ti in range(run["validation_times"]):
            g = base_gen
            if a.reseed_per_run and base_gen is not None:
                g = torch.Generator().manual_seed(int(torch.randint(0x7ffffffffffffff0, [1], generator=base_gen)))
            gen = None if g is None else ([g] * bs if run["fix_seed_within_batch"] else g)
            images = pipe(generator=gen, **kw).images                      # List[List[PIL]]: scene x view
            for bi, views in enumerate(images):
                for vi, im in enumerate(views):
                    im.save(os.path.join(a.out, f"{total + bi}_gen{ti}_view{vi}.png"))
        total += bs
    print(f"sampled {total} scenes x {run['validation_times'] - (1 if a.cond_on_view else 0)} -> {a.out}")


if __name__ == "__main__":
    main()
