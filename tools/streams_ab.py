#!/usr/bin/env python
"""Same-box A/B of the scene-chunk streams (pipeline.streams) and the scenes per call: ONE model build, then for every
(scenes, streams) pair one warm-up call (plan build + graph capture) and `--reps` timed calls of bench.py's workload
(configs[1]: text-only, 50-step DDIM, bf16).  Prints one line per pair; profiles/r04_streams_ab.log is this script's output."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=str, default="128:1,128:2,192:1,192:2,256:2")
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--ddim-steps", type=int, default=50)
    args = ap.parse_args()
    import bench
    from magicdrive_amd import synthetic
    from magicdrive_amd.networks import spec
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    pipe, unet, cn = bench.build_pipeline(spec.SD15_CONFIG, dev)
    nmax = max(int(p.split(":")[0]) for p in args.pairs.split(","))
    scenes = [synthetic.make_scene_batch(1, seed=1234 + i, max_len=None, zero_map=True) for i in range(nmax)]
    cat = lambda k, n: torch.cat([s[k] for s in scenes[:n]]).to(dev)
    ref = None
    for pair in args.pairs.split(","):
        f = pair.split(":")
        n, st = int(f[0]), int(f[1])
        pipe.streams = st
        pipe.fork_chunks = len(f) > 2 and f[2] == "f"        # "192:2:f": also fork ControlNet / UNet encoder inside every chunk
        pipe.fork_max_scenes = (1 << 30) if (len(f) > 2 and f[2] == "F") else 0      # "48:1:F": ONE chunk with the two branches of a step side by side
        kw = dict(prompt=None, image=cat("bev_map", n), camera_param=None, height=224, width=400, num_inference_steps=args.ddim_steps,
                  guidance_scale=1.0, latents=cat("latents", n), prompt_embeds=cat("prompt_embeds", n),
                  negative_prompt_embeds=cat("negative_prompt_embeds", n), output_type="latent")
        out = pipe(**kw).images
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            out = pipe(**kw).images
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        assert torch.isfinite(out).all()
        # scene 0 must not depend on how the batch was chunked
        if ref is None:
            ref = out[0].float().clone()
        dev_rel = ((out[0].float() - ref).norm() / ref.norm()).item()
        print(json.dumps({"scenes": n, "streams": st, "fork_chunks": pipe.fork_chunks, "fork_one_chunk": pipe.fork_max_scenes > 0, "scenes_per_s": round(n / min(ts), 4), "seconds_per_call": [round(t, 3) for t in ts],
                          "scene0_vs_first_config_rel": round(dev_rel, 6)}), flush=True)
        pipe._plans.clear()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
