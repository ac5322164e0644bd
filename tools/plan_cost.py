#!/usr/bin/env python
"""What a NEW plan geometry costs at the reference's operating point (one scene per call, a new padded box count almost every batch): wall time of the
pieces of the first pipe() call of a geometry — SamplerPlan construction (engine walk, pool allocations, packed-weight lookups), program lowering,
hipGraph capture + instantiation — next to a cached call.  Usage: python tools/plan_cost.py [--scenes 1] [--steps 20]"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from magicdrive_amd import denoiser as DN, ops as O, synthetic  # noqa: E402
from magicdrive_amd.networks import spec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=1); ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
pipe, unet, cn = bench.build_pipeline(spec.SD15_CONFIG, dev, "ddim", torch.bfloat16)
sync = torch.cuda.synchronize


def kwargs(L):
    sc = synthetic.make_scene_batch(a.scenes, seed=77, max_len=L)
    boxes = {k: v.to(dev) for k, v in sc["bboxes_3d_data"].items()}
    return dict(prompt=None, image=sc["bev_map"].to(dev), camera_param=sc["camera_param"].to(dev), height=224, width=400, num_inference_steps=a.steps, guidance_scale=2.0,
                latents=sc["latents"].to(dev), prompt_embeds=sc["prompt_embeds"].to(dev), negative_prompt_embeds=sc["negative_prompt_embeds"].to(dev), output_type="latent",
                bev_controlnet_kwargs={"bboxes_3d_data": boxes})


pipe(**kwargs(4)); sync()                                           # weights packed, kernels loaded
rows = []
for L in (5, 6, 7, 9, 12):
    kw = kwargs(L)
    t0 = time.perf_counter(); pipe(**kw); sync(); t_first = time.perf_counter() - t0
    t0 = time.perf_counter(); pipe(**kw); sync(); t_cached = time.perf_counter() - t0
    rows.append(dict(L=L, first_call_s=round(t_first, 4), cached_call_s=round(t_cached, 4), new_geometry_cost_s=round(t_first - t_cached, 4)))
    print(json.dumps(rows[-1]), flush=True)
# the pieces, outside the pipeline
t0 = time.perf_counter()
plan = DN.SamplerPlan(pipe._plan_config(), unet.packed(), cn.packed(), dev, a.scenes, True, 13, (28, 50), num_steps=a.steps, guidance_scale=2.0, fork=True)
sync(); t_plan = time.perf_counter() - t0
t0 = time.perf_counter(); plan.compile(); t_lower = time.perf_counter() - t0
t0 = time.perf_counter()
for pr in (plan.step_cn, plan.step_enc, plan.step_tail):
    pr.capture()
sync(); t_cap = time.perf_counter() - t0
print(json.dumps(dict(plan_construction_s=round(t_plan, 4), lowering_s=round(t_lower, 4), graph_capture_s=round(t_cap, 4), step_ops=len(plan.step_ops), prologue_ops=len(plan.prologue_ops))))
# eviction: what PlanCache.put pays when the LRU plan is released
sync()
t0 = time.perf_counter(); torch.cuda.synchronize(dev); t_sync = time.perf_counter() - t0
t0 = time.perf_counter()
for prog in (plan.prologue, plan.step, plan.step_cn, plan.step_enc, plan.step_tail):
    if prog is not None:
        prog.destroy()
t_destroy = time.perf_counter() - t0
t0 = time.perf_counter()
plan.prologue_ops = plan.step_ops = []; plan.bld = None; plan.cond = plan.temb_cn = plan.temb_un = plan.kv_cn = plan.kv_un = None
import gc; gc.collect(); sync()
t_free = time.perf_counter() - t0
print(json.dumps(dict(release_sync_s=round(t_sync, 4), graph_destroy_s=round(t_destroy, 4), buffers_free_s=round(t_free, 4))))
# the reference's validation flow: one scene per call, a different padded box count (almost) every call — 20 calls over 20 DISTINCT box counts (more than the
# plan cache holds: every call builds a plan and evicts one) against 20 calls at one box count
import random
random.seed(0)
Ls = random.sample(range(1, 118), 20)
kws = {L: kwargs(L) for L in Ls}
pipe(**kws[Ls[0]]); sync()
t0 = time.perf_counter()
for _ in range(20):
    pipe(**kws[Ls[0]])
sync(); t_fixed = time.perf_counter() - t0
t0 = time.perf_counter()
for L in Ls:
    pipe(**kws[L])
sync(); t_var = time.perf_counter() - t0
t0 = time.perf_counter()
for L in Ls:                                                        # second sweep over the same 20 box counts (the cache holds PLAN_CACHE of them)
    pipe(**kws[L])
sync(); t_var2 = time.perf_counter() - t0
from magicdrive_amd import _lib as L_
print(json.dumps(dict(scenes=a.scenes, ddim_steps=a.steps, plan_cache=int(L_.get_option("PLAN_CACHE")), fixed_L_20_calls_s=round(t_fixed, 3), varying_L_20_calls_s=round(t_var, 3),
                      varying_L_second_sweep_s=round(t_var2, 3), varying_over_fixed=round(t_var / t_fixed, 4), second_over_fixed=round(t_var2 / t_fixed, 4))))
