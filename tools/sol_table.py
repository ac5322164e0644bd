#!/usr/bin/env python
"""Speed-of-light table of the step program: for every launch max(FLOPs / MFMA peak, algorithmic bytes / HBM peak) next to the measured
time (an --ops-json written by bench.py), summed per kernel.  The step is a mix of MFMA-bound launches (convs, attention, K >= 640
GEMMs) and HBM-bound ones (norms, the K = 320 projections, residual adds): this is the per-launch roofline the bench line's single
`roofline` object cannot show.  CPU only: the plan is built on the host at a small batch and its activation bytes are scaled.
Usage: python tools/sol_table.py profiles/r02e_ops_b128.json [--scenes 128] [--out profiles/r03_sol_table.json]
       --practical: the MEASURED roofs instead of the data-sheet ones — the 16x16x32 MFMA stream on random operands at the power budget
       (2.05 PFLOP/s, profiles/r04_ubench_mfma_shape.log) and the achievable HBM rate (6.3 TB/s, MI355X_MICROARCH.md)."""
import argparse
import json
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import denoiser as DN, flops as FL, ops as O  # noqa: E402
from magicdrive_amd.engine import PackedNet  # noqa: E402
from magicdrive_amd.networks import spec  # noqa: E402

MFMA_PEAK = 2500e12        # dense bf16, MI355X_MICROARCH.md "Matrix cores"
HBM_PEAK = 8e12            # spec; ~6.3e12 achievable (same guide, "HBM")


def op_bytes_split(op):
    """(activation bytes, weight bytes): activations scale with the batch, weights do not."""
    nb = lambda t: 0 if t is None else t.numel() * t.element_size()
    if isinstance(op, O.Gemm):
        # batched / flattened V^T GEMMs carry the weight in A (shared) and the activations in W
        if op.C.dim() == 3:
            return nb(op.W) + nb(op.C) + nb(op.R), nb(op.A)
        return nb(op.A) + nb(op.C) + nb(op.R), nb(op.W)
    if isinstance(op, O.Conv):
        return nb(op.X) + nb(op.Y) + nb(op.R), nb(op.Wt)
    return FL.op_bytes(op), 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ops_json")
    ap.add_argument("--scenes", type=int, default=128)
    ap.add_argument("--build-scenes", type=int, default=2)
    ap.add_argument("--out", default="")
    ap.add_argument("--practical", action="store_true")
    a = ap.parse_args()
    global MFMA_PEAK, HBM_PEAK
    if a.practical:
        MFMA_PEAK, HBM_PEAK = 2051e12, 6.3e12
    cfg = spec.SD15_CONFIG
    dev = torch.device("cpu")
    usd = spec.random_state_dict(spec.unet_param_shapes(cfg), 0)
    csd = spec.random_state_dict(spec.controlnet_param_shapes(cfg), 1)
    plan = DN.SamplerPlan(cfg, PackedNet(usd, dev), PackedNet(csd, dev), dev, a.build_scenes, False, 0, (28, 50), num_steps=50)
    scale = a.scenes / a.build_scenes
    meas = json.load(open(a.ops_json))
    assert len(meas) == len(plan.step_ops), (len(meas), len(plan.step_ops))
    rows = []
    agg = defaultdict(lambda: dict(ms=0.0, sol_ms=0.0, mfma_ms=0.0, hbm_ms=0.0, n=0, flops=0.0, bytes=0.0))
    for op, m in zip(plan.step_ops, meas):
        fl = FL.op_flops(op) * scale
        ab, wb = op_bytes_split(op)
        by = ab * scale + wb
        t_m, t_h = fl / MFMA_PEAK * 1e3, by / HBM_PEAK * 1e3
        r = dict(name=m["name"], kernel=m["kernel"], ms=m["ms"], gflop=fl / 1e9, mbytes=by / 1e6, mfma_ms=t_m, hbm_ms=t_h, sol_ms=max(t_m, t_h),
                 bound="mfma" if t_m >= t_h else "hbm")
        rows.append(r)
        g = agg[m["kernel"]]
        g["ms"] += m["ms"]; g["sol_ms"] += r["sol_ms"]; g["mfma_ms"] += t_m; g["hbm_ms"] += t_h; g["n"] += 1; g["flops"] += fl; g["bytes"] += by
    tot = sum(r["ms"] for r in rows); sol = sum(r["sol_ms"] for r in rows)
    print(f"step {tot:.1f} ms measured, sum of per-launch speed-of-light bounds {sol:.1f} ms ({sol / tot:.3f}); "
          f"MFMA-only {sum(r['mfma_ms'] for r in rows):.1f} ms, HBM-only {sum(r['hbm_ms'] for r in rows):.1f} ms")
    print(f"{'kernel':44s} {'n':>4s} {'ms':>8s} {'sol':>8s} {'frac':>6s} {'gap ms':>7s} {'TF/s':>7s} {'GB/s':>7s}")
    for k, g in sorted(agg.items(), key=lambda kv: -(kv[1]["ms"] - kv[1]["sol_ms"])):
        print(f"{k:44s} {g['n']:4d} {g['ms']:8.2f} {g['sol_ms']:8.2f} {g['sol_ms'] / g['ms']:6.3f} {g['ms'] - g['sol_ms']:7.2f} "
              f"{g['flops'] / g['ms'] / 1e9:7.0f} {g['bytes'] / g['ms'] / 1e6:7.0f}")
    if a.out:
        json.dump(dict(scenes=a.scenes, step_ms=tot, sol_ms=sol, mfma_peak=MFMA_PEAK, hbm_peak=HBM_PEAK,
                       per_kernel={k: {kk: round(v, 3) for kk, v in g.items()} for k, g in agg.items()}, ops=rows), open(a.out, "w"), indent=0)


if __name__ == "__main__":
    main()
