#!/usr/bin/env python
"""Regenerate the FLOP model of SURVEY.md §8d.

  * as-executed FLOPs of the reference algorithm: torch's FlopCounterMode over one oracle pass (GEMM+conv) plus
    attention cores from the shapes (4·B·Tq·Tk·C per call; the counter does not see softmax(QK^T)V built from matmuls
    on every backend, so we count it from the oracle's own matmuls: FlopCounterMode does count `@`);
  * algorithmic FLOPs of the de-duplicated HIP program: magicdrive_amd.flops over the engine's op list.
Usage: python tools/count_flops.py [L_boxes=32]
"""
import os
import sys

import torch
from torch.utils.flop_counter import FlopCounterMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from magicdrive_amd import denoiser as DN, flops, synthetic  # noqa: E402
from magicdrive_amd.engine import PackedNet  # noqa: E402
from magicdrive_amd.networks import spec  # noqa: E402
from oracle import denoiser as D  # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    cfg = spec.SD15_CONFIG
    usd = {k: torch.zeros(s) for k, s in spec.unet_param_shapes(cfg).items()}
    csd = {k: torch.zeros(s) for k, s in spec.controlnet_param_shapes(cfg).items()}
    sc = synthetic.make_scene_batch(1, max_len=L or None)
    lat = torch.stack([sc["latents"]] * 6, 1)
    with torch.no_grad(), FlopCounterMode(display=False) as fc:
        d, m, ctx = D.controlnet_forward(csd, cfg, lat, torch.tensor([981]), sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"])
        cn_f = fc.get_total_flops()
        D.unet_forward(usd, cfg, lat.reshape(-1, 4, 28, 50), 981, ctx, d, m)
        tot_f = fc.get_total_flops()
    print(f"reference algorithm as executed by the oracle (L={L}, c=1, b=1): controlnet {cn_f/1e12:.4f} TF, unet {(tot_f-cn_f)/1e12:.4f} TF, total {tot_f/1e12:.4f} TF per step")
    print("  (the oracle already encodes the map only once per scene; the reference repeats it x6: +0.0193 TF)")
    dev = torch.device("cpu")
    sp = DN.SamplerPlan(cfg, PackedNet(usd, dev), PackedNet(csd, dev), dev, 1, False, L, (28, 50), num_steps=50)
    fs = flops.program_flops(sp.step_ops); fp = flops.program_flops(sp.prologue_ops)
    print("HIP program, per scene-step:", {k: round(v / 1e12, 4) for k, v in fs.items()})
    print("HIP program, prologue      :", {k: round(v / 1e12, 4) for k, v in fp.items()})
    print(f"F_scene (50 steps, c=1) = {(50*fs['total']+fp['total'])/1e12:.2f} TF ; with CFG (c=2) = {(100*fs['total']+2*fp['total'])/1e12:.2f} TF")
    print(f"launches per step: {len(sp.step_ops)}  prologue: {len(sp.prologue_ops)}")
    # the VAE decoder (diffusers AutoencoderKL.decode, vae.py:152-281: row a14) — once per scene, outside the denoising loop
    from magicdrive_amd import vae as V
    vcfg = spec.VAE_SD15_CONFIG
    vsd = {k: torch.zeros(s_) for k, s_ in spec.vae_decoder_param_shapes(vcfg).items()}
    vp = V.VaeDecodePlan(vcfg, PackedNet(vsd, dev), dev, 6, (28, 50))
    fv = flops.program_flops(vp.ops)
    print("VAE decode, per 6-view scene:", {k: round(v / 1e12, 4) for k, v in fv.items()}, f"launches: {len(vp.ops)}")


if __name__ == "__main__":
    main()
