#!/usr/bin/env python
"""Latency operating point: ONE scene per call (the reference's own flows run bs = 1...4).  Times pipe() and prints the per-kernel table of the
1-scene step program (HIP events per launch).  Usage: python tools/lat1.py [--scenes 1] [--steps 50]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from magicdrive_amd import synthetic  # noqa: E402
from magicdrive_amd.networks import spec  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=1); ap.add_argument("--steps", type=int, default=50); ap.add_argument("--fork-max", type=int, default=-1, help="pipe.fork_max_scenes (0 = never fork; default: the pipeline's)"); ap.add_argument("--no-ops", action="store_true"); ap.add_argument("--rows-json", default="", help="write every launch of the step program (name, kernel, median / best ms, GFLOP) to this file")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = spec.SD15_CONFIG
pipe, unet, cn = bench.build_pipeline(cfg, dev, "ddim", torch.bfloat16)
if a.fork_max >= 0:
    pipe.fork_max_scenes = a.fork_max
sc = [synthetic.make_scene_batch(1, seed=1234 + i, max_len=None, zero_map=True) for i in range(a.scenes)]
cat = lambda k: torch.cat([s[k] for s in sc]).to(dev)
kw = dict(prompt=None, image=cat("bev_map"), camera_param=None, height=224, width=400, num_inference_steps=a.steps, guidance_scale=1.0, latents=cat("latents"),
          prompt_embeds=cat("prompt_embeds"), negative_prompt_embeds=cat("negative_prompt_embeds"), output_type="latent")
pipe(**kw); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); pipe(**kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{a.scenes} scene(s), {a.steps} steps, fork_max_scenes {pipe.fork_max_scenes}: {min(ts):.4f} s per call = {1e3 * min(ts) / a.steps:.2f} ms per step (graph replay)", flush=True)
if a.no_ops:
    sys.exit(0)
plan = next(iter(pipe._plans.values()))
fam, kern, rows = bench.per_op_profile(plan, reps=3)
tot = sum(v["ms"] for v in kern.values())
if a.rows_json:
    import json
    with open(a.rows_json, "w") as f:
        json.dump(rows, f)
print(f"step program op by op: {tot:.2f} ms, {len(rows)} launches")
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:16]:
    print(f"  {k:52s} {v['ms']:7.3f} ms {v['launches']:4d} launches {1e3 * v['ms'] / v['launches']:7.1f} us avg")
