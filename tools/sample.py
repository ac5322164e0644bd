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

Multi-GPU = the reference's FID generator flow (perception/data_prepare/val_set_gen.py:71-161), one process per GPU under torchrun:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 tools/sample.py --ckpt ... --out ...

batch j of the data goes to rank j mod N (what `accelerator.prepare(dataloader)` does, :79), every rank drives GPU LOCAL_RANK; seeding follows
the reference's two branches — default: `manual_seed(cfg.seed)` per batch on every rank (test_utils.py:233-238: a scene's latents do not depend on N);
runner.validation_seed_global=true: one generator per rank seeded `seed + rank` (:83-87), a local seed drawn per batch, nothing is exchanged inside the sampling loop, and
per batch the ranks either write their own scenes' files and gather the labels on rank 0 (the reference's single-node branch, :147-150) or —
`--gather-images`, its multi-node branch :141-146 — gather the uint8 images on rank 0, which writes everything.  File names carry the GLOBAL
scene index, so a file is written exactly once whatever N is; rank 0 writes `index.json` (scene -> files, rank, seed) at the end
(perception/common/ddp_utils.py:5-16: one all_gather_object per batch).
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
                bbox_max_length=None, validation_seed_global=False)          # configs/test_config.yaml + configs/runner/default.yaml:54-61


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
        elif k == "runner.validation_seed_global":
            run["validation_seed_global"] = bool(_parse(v))
        elif k == "runner.validation_times":
            run["validation_times"] = int(v)
        elif k == "runner.bbox_max_length":
            run["bbox_max_length"] = _parse(v)
        elif k.startswith("runner.pipeline_param."):
            run[k[len("runner.pipeline_param."):]] = _parse(v)
    return run


def iter_batches_index(n: int, batch_size: int) -> Iterator[List[int]]:
    """Sample indices of batch 0, 1, ... (the order of the reference's sequential val dataloader)."""
    for i0 in range(0, n, batch_size):
        yield list(range(i0, min(i0 + batch_size, n)))


def iter_pipe_kwargs(dataset, run: Dict, batch_size: int = 1, with_pixels: bool = False, only=None) -> Iterator[Dict]:
    """Batches of samples -> keyword arguments of StableDiffusionBEVControlNetPipeline.__call__, as run_one_batch /
    run_one_batch_pipe assemble them (magicdrive/misc/test_utils.py:191-255, :258-330).  with_pixels: (kwargs, pixel_values) pairs.
    only: the batch indices to produce (a rank's share: rank_batches)."""
    from magicdrive_amd.dataset import collate_samples, preprocess_fn
    extra = {k: v for k, v in run.items() if k not in DEFAULTS or k in ("guidance_scale", "num_inference_steps")}
    for j, i0 in enumerate(range(0, len(dataset), batch_size)):
        if only is not None and j not in only:                       # another rank's batch: not even loaded
            continue
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


def rank_batches(n_batches: int, rank: int, world: int) -> List[int]:
    """Batch indices of this rank: j -> rank j mod world (accelerate's dataloader sharding, val_set_gen.py:79)."""
    return list(range(rank, n_batches, world))


def rank_seed(seed, rank: int, world: int, seed_global: bool = False):
    """The seed a rank's generator starts from.  Default (runner.validation_seed_global false, configs/runner/default.yaml): run_one_batch_pipe
    seeds `torch.manual_seed(cfg.seed)` per batch on EVERY rank, no rank offset (magicdrive/misc/test_utils.py:233-238) — a scene's initial
    latents do not depend on the world size or on the rank its batch lands on.  validation_seed_global: ONE generator per rank, seeded
    `cfg.seed + accelerator.process_index` before the loop (perception/data_prepare/val_set_gen.py:83-87)."""
    if seed is None:
        return None
    return int(seed) + rank if seed_global else int(seed)


def new_local_seed(global_generator) -> int:
    """magicdrive/misc/test_utils.py:184-188."""
    return int(torch.randint(0x7ffffffffffffff0, [1], generator=global_generator).item())


def batch_generator(seed, bs: int, fix_seed_within_batch: bool, global_generator=None):
    """The `generator` argument of one batch's pipe() calls, as run_one_batch_pipe builds it (test_utils.py:221-238).  The reference's
    `torch.manual_seed(s)` re-seeds and returns THE default generator, so its per-scene list holds one object `bs` times — seeded with the
    last local seed drawn when a global generator is given — and its state carries over the validation_times loop; reproduced with one
    private generator."""
    if seed is None:
        return None
    if global_generator is not None:
        local = [new_local_seed(global_generator) for _ in range(bs if fix_seed_within_batch else 1)][-1]
        g = torch.Generator().manual_seed(local)
    else:
        g = torch.Generator().manual_seed(int(seed))
    return [g] * bs if fix_seed_within_batch else g


def save_views(out_dir: str, scene: int, gen: int, views) -> List[str]:
    """One scene's views of one generation -> <scene>_gen<gen>_view<v>.png; `views`: PIL images or uint8 HxWx3 arrays.  Returns the file names."""
    names = []
    for vi, im in enumerate(views):
        if not hasattr(im, "save"):
            from PIL import Image
            import numpy as np
            im = Image.fromarray(np.asarray(im, dtype="uint8"))
        name = f"{scene}_gen{gen}_view{vi}.png"
        im.save(os.path.join(out_dir, name))
        names.append(name)
    return names


def exchange_batch(records: List[Dict], images, rank: int, world: int, gather_images: bool, out_dir: str) -> List[Dict]:
    """End of a batch on every rank (val_set_gen.py:137-151).  records: this rank's [{scene, gen, rank, seed}], images: the matching
    [views] lists.  Single-node branch: the rank writes its own files, then the labels are gathered; multi-node branch (gather_images): labels
    AND uint8 images are gathered and rank 0 writes.  Returns the records with their file names — on rank 0 those of every rank, elsewhere []
    (ddp_utils.concat_from_everyone).  One all_gather_object per batch; ranks whose shard ran out contribute an empty list."""
    import numpy as np
    import torch.distributed as dist
    if not gather_images:
        for r_, views in zip(records, images):
            r_["files"] = save_views(out_dir, r_["scene"], r_["gen"], views)
        payload = records
    else:
        payload = [dict(r_, pixels=[np.asarray(v, dtype="uint8") for v in views]) for r_, views in zip(records, images)]
    if world == 1:
        gathered = [payload]
    else:
        gathered = [None] * world
        dist.all_gather_object(gathered, payload)
    if rank != 0:
        return []
    out = []
    for part in gathered:
        for r_ in part:
            if gather_images:
                r_["files"] = save_views(out_dir, r_["scene"], r_["gen"], r_.pop("pixels"))
            out.append(r_)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ckpt", required=True); ap.add_argument("--sd15", required=True)
    ap.add_argument("--data", required=True); ap.add_argument("--out", required=True)
    ap.add_argument("--scheduler", choices=["unipc", "ddim"], default="unipc")
    ap.add_argument("--batch-size", type=int, default=1)
    ap.add_argument("--prompt-embeds", action="store_true", help="no text encoder available: sample with zero prompt embeddings")
    ap.add_argument("--device", default=None, help="default: cuda:<LOCAL_RANK>")
    ap.add_argument("--gather-images", action="store_true", help="multi-rank: rank 0 writes every file from gathered uint8 images (the reference's multi-node branch); "
                                                                 "default: each rank writes its own scenes, labels are gathered (its single-node branch)")
    ap.add_argument("--pipe-factory", default="", help="TESTS: module:function(ckpt, sd15, scheduler, device, given_view) returning the pipeline instead of build_pipe")
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend under torchrun (default: nccl = RCCL with a GPU, else gloo)")
    ap.add_argument("--reseed-per-run", action="store_true", help="deviation from the reference: an independent seed per (batch, validation run)")
    ap.add_argument("--cond-on-view", action="store_true", help="demo/run_cond_on_view.py: generation ti is sampled with the encoded ground-truth view ti given")
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args(argv)
    from magicdrive_amd.dataset import FolderSet
    from magicdrive_amd import distributed as DD
    import json
    rank, world, local = DD.init_from_env(backend=a.dist_backend)
    device = a.device or (f"cuda:{local}" if torch.cuda.is_available() else "cpu")
    run = resolve_run_config(a.ckpt, a.overrides)
    if a.pipe_factory:
        import importlib
        mod, fn = a.pipe_factory.split(":")
        pipe = getattr(importlib.import_module(mod), fn)(a.ckpt, a.sd15, a.scheduler, device, a.cond_on_view)
    else:
        pipe = build_pipe(a.ckpt, a.sd15, a.scheduler, device, given_view=a.cond_on_view)
    data = FolderSet(a.data)
    if rank == 0:
        os.makedirs(a.out, exist_ok=True)
    DD.barrier()
    seed = rank_seed(run["seed"], rank, world, run["validation_seed_global"])
    global_gen = torch.Generator().manual_seed(seed) if (run["validation_seed_global"] and seed is not None) else None   # carried across batches
    batches = list(iter_batches_index(len(data), a.batch_size))
    mine = set(rank_batches(len(batches), rank, world))
    n_rounds = (len(batches) + world - 1) // world                  # every rank takes part in every per-batch exchange (a collective)
    index: List[Dict] = []
    n_local = 0
    it = iter_pipe_kwargs(data, run, a.batch_size, with_pixels=a.cond_on_view, only=mine)
    for rnd in range(n_rounds):
        j = rnd * world + rank
        records, images_out = [], []
        if j < len(batches):
            item = next(it)
            kw, pixel_values = item if a.cond_on_view else (item, None)
            bs = kw["image"].shape[0]
            scene0 = batches[j][0]
            if getattr(pipe, "text_encoder", None) is None:
                if not a.prompt_embeds:
                    raise SystemExit(f"{a.sd15} has no text_encoder/: pass --prompt-embeds to sample with zero embeddings")
                D = pipe.unet.cfg["cross_attention_dim"]
                kw.update(prompt=None, prompt_embeds=torch.zeros(bs, 77, D), negative_prompt_embeds=torch.zeros(bs, 77, D))
            # Seeding as run_one_batch_pipe does it (magicdrive/misc/test_utils.py:221-238): the generator is built ONCE per batch, before the
            # validation_times loop (its state carries across the iterations).  Default: manual_seed(cfg.seed) on every rank; with
            # runner.validation_seed_global=true the rank's global generator (seed + rank, val_set_gen.py:83-87) hands out a local seed per batch.
            # --reseed-per-run (a deviation) draws a fresh seed per (batch, run).
            gen = batch_generator(seed, bs, run["fix_seed_within_batch"], global_gen)
            base_gen = gen[0] if isinstance(gen, list) else gen
            if a.cond_on_view:
                for ti, images in cond_on_view_runs(pipe, kw, pixel_values, run, generator=base_gen):
                    for bi, views in enumerate(images):
                        records.append(dict(scene=scene0 + bi, gen=ti, rank=rank, seed=seed)); images_out.append(views)
            else:
                for ti in range(run["validation_times"]):
                    g = gen
                    if a.reseed_per_run and base_gen is not None:
                        g1 = torch.Generator().manual_seed(new_local_seed(base_gen))
                        g = [g1] * bs if run["fix_seed_within_batch"] else g1
                    images = pipe(generator=g, **kw).images                      # List[List[PIL]]: scene x view
                    for bi, views in enumerate(images):
                        records.append(dict(scene=scene0 + bi, gen=ti, rank=rank, seed=seed)); images_out.append(views)
            n_local += bs
        index += exchange_batch(records, images_out, rank, world, a.gather_images, a.out)
    if rank == 0:
        index.sort(key=lambda r_: (r_["scene"], r_["gen"]))
        with open(os.path.join(a.out, "index.json"), "w") as f:
            json.dump(dict(world=world, seed=run["seed"], gather_images=bool(a.gather_images), generations=index), f, indent=1)
        print(f"sampled {len(data)} scenes x {run['validation_times'] - (1 if a.cond_on_view else 0)} on {world} rank(s) -> {a.out}")
    DD.shutdown()


if __name__ == "__main__":
    main()
