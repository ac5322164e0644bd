#!/usr/bin/env python
"""Which generic tile (gemm_conv.hip) should a small grid get?  Takes the real step program of an N-scene plan, and for every distinct GEMM / conv shape the library
routes to the generic kernel times the op under forced tile sizes (options GEMM_BM / GEMM_BN) and split-K factors: 40 back-to-back launches between two HIP events.
Prints, per shape, the default route's time next to the best configurations, and the step-level sum.  Usage: python tools/tile_sweep.py [--scenes 1] [--json out.json]"""
import argparse, dataclasses, itertools, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from magicdrive_amd import _lib as L, ops as O, synthetic  # noqa: E402
from magicdrive_amd.networks import spec  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=1); ap.add_argument("--reps", type=int, default=40); ap.add_argument("--json", default="")
ap.add_argument("--default-only", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
pipe, unet, cn = bench.build_pipeline(spec.SD15_CONFIG, dev, "ddim", torch.bfloat16)
sc = [synthetic.make_scene_batch(1, seed=1234 + i, max_len=None, zero_map=True) for i in range(a.scenes)]
cat = lambda k: torch.cat([s[k] for s in sc]).to(dev)
pipe(prompt=None, image=cat("bev_map"), camera_param=None, height=224, width=400, num_inference_steps=2, guidance_scale=1.0, latents=cat("latents"),
     prompt_embeds=cat("prompt_embeds"), negative_prompt_embeds=cat("negative_prompt_embeds"), output_type="latent")
torch.cuda.synchronize()
plan = next(iter(pipe._plans.values()))
st = torch.cuda.current_stream().cuda_stream
def timeit(op):
    code, desc, dt = O.lower_with_dtype(op)
    for _ in range(2): L.call_op(code, desc, st, dt)
    k = (L.lib().mdx_last_kernel() or b"").decode()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): L.call_op(code, desc, st, dt)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / a.reps, k
shapes = {}
for op in plan.step_ops:
    if isinstance(op, O.Conv) and not op.direct:
        B, H, W, Ci = op.X.shape; Co = op.Wt.shape[0]; Ho, Wo = op.Y.shape[1:3]
        key = ("conv", B * Ho * Wo, Co, Ci * op.Wt.shape[1] * op.Wt.shape[2], f"{H}x{W} {Ci}->{Co} s{op.stride[0]}", op.epilogue)
    elif isinstance(op, O.Gemm) and op.A.dim() == 2 and op.Vt is None and op.ln_eps == 0.0:
        key = ("gemm", op.A.shape[0], op.W.shape[0], op.A.shape[1], "R" if op.R is not None else "", op.epilogue)
    else:
        continue
    shapes.setdefault(key, []).append(op)
rows = []
tot_def = tot_best = 0.0
for key, ops in sorted(shapes.items(), key=lambda kv: (kv[0][0], -kv[0][1], kv[0][2], kv[0][3])):
    op = ops[0]
    t0, k0 = timeit(op)
    if not (k0.startswith("gemm_conv_kernel") or k0.startswith("splitk_reduce")):
        continue
    res = []
    if not a.default_only:
        for bm, bn, sk in itertools.product((64, 128), (64, 128), (0, 1, 2, 3, 4, 6, 8, 12, 16)):
            if sk > 1 and key[3] / sk < 256: continue
            if key[5] == 1 and bn == 64: continue
            try:
                with L.options(GEMM_BM=bm, GEMM_BN=bn):
                    t, k = timeit(dataclasses.replace(op, splitk=sk))
            except Exception:
                continue
            res.append((round(t, 1), bm, bn, sk))
        res.sort()
    best = res[0] if res else (t0, 0, 0, 0)
    tot_def += t0 * len(ops); tot_best += min(best[0], t0) * len(ops)
    fl = 2.0 * key[1] * key[2] * key[3]
    print(f"{key[0]} M={key[1]:6d} N={key[2]:5d} K={key[3]:6d} {str(key[4]):22s} epi{key[5]} x{len(ops):3d}  default {t0:6.1f} us ({fl / t0 / 1e6:5.0f} TF/s) {k0[17:36]:20s} best: " +
          "  ".join(f"{t:.1f}us bm{bm} bn{bn} sk{sk}" for t, bm, bn, sk in res[:4]), flush=True)
    rows.append({"key": list(key), "count": len(ops), "default_us": t0, "default_kernel": k0, "sweep": res})
print(f"sum over the step program: default {tot_def / 1e3:.3f} ms, best-per-shape {tot_best / 1e3:.3f} ms")
if a.json:
    with open(a.json, "w") as f:
        json.dump({"scenes": a.scenes, "rows": rows, "default_ms": tot_def / 1e3, "best_ms": tot_best / 1e3}, f)
