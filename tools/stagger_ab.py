#!/usr/bin/env python
"""Does a half-step offset between the two scene-chunk streams of the bench workload help?  Both chunks replay the same step program at the same speed, so they stay
phase-aligned (conv beside conv, GroupNorm beside GroupNorm); a delay queued on the side stream before the call (torch.cuda._sleep) shifts chunk 2 by `--offsets`
seconds.  Prints scenes/s per offset (one warm-up + `--reps` timed calls each, same box).  Usage: python tools/stagger_ab.py [--offsets 0,0.06,0.13,0.2]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from magicdrive_amd import synthetic  # noqa: E402
from magicdrive_amd.networks import spec  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--offsets", default="0,0.06,0.13,0.2,0"); ap.add_argument("--scenes", type=int, default=192); ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
pipe, unet, cn = bench.build_pipeline(spec.SD15_CONFIG, dev, "ddim", torch.bfloat16)
sc = [synthetic.make_scene_batch(1, seed=1234 + i, max_len=None, zero_map=True) for i in range(a.scenes)]
cat = lambda k: torch.cat([s[k] for s in sc]).to(dev)
kw = dict(prompt=None, image=cat("bev_map"), camera_param=None, height=224, width=400, num_inference_steps=50, guidance_scale=1.0, latents=cat("latents"),
          prompt_embeds=cat("prompt_embeds"), negative_prompt_embeds=cat("negative_prompt_embeds"), output_type="latent")
pipe(**kw); torch.cuda.synchronize()
side = pipe._side_streams(dev, 1)[0]
# calibrate torch.cuda._sleep: cycles per second
t0 = time.perf_counter(); torch.cuda._sleep(200_000_000); torch.cuda.synchronize(); cps = 200_000_000 / (time.perf_counter() - t0)
print(f"_sleep: {cps / 1e6:.1f} M cycles per second", flush=True)
for off in [float(x) for x in a.offsets.split(",")]:
    ts = []
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if off > 0:
            with torch.cuda.stream(side):
                torch.cuda._sleep(int(off * cps))
        pipe(**kw); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"offset {off:5.2f} s: {a.scenes / min(ts):.3f} scenes/s (calls {[round(t, 2) for t in ts]})", flush=True)
