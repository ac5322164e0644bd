#!/usr/bin/env python
"""Per-op HIP-event profile of the VAE decode program (vae.VaeDecodePlan: diffusers AutoencoderKL.decode, vae.py:152-281) at the bench's shape:
one 6-view scene, 28x50 latents -> 224x400 images.  Usage: python tools/vaeone.py [--scenes 1] [--reps 5]"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, flops as FL, ops as O  # noqa: E402
from magicdrive_amd.networks import spec  # noqa: E402
from magicdrive_amd.networks.autoencoder_kl import AutoencoderKL  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=1); ap.add_argument("--reps", type=int, default=5); ap.add_argument("--json", default="")
a = ap.parse_args()
dev = torch.device("cuda")
vae = AutoencoderKL.from_config(spec.VAE_SD15_CONFIG, 7).to(dev)
z = torch.randn(6 * a.scenes, 4, 28, 50, device=dev)
vae.decode(z); torch.cuda.synchronize()
plan = next(iter(vae._plans.values()))
st = torch.cuda.current_stream().cuda_stream
low = [O.lower_with_dtype(op) for op in plan.ops]
n = len(low); samples = [[] for _ in range(n)]; kn = [""] * n
for _ in range(a.reps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i, (code, desc, dt) in enumerate(low):
        L.call_op(code, desc, st, dt); kn[i] = (L.lib().mdx_last_kernel() or b"").decode(); ev[i + 1].record()
    torch.cuda.synchronize()
    for i in range(n): samples[i].append(ev[i].elapsed_time(ev[i + 1]))
med = [sorted(v)[len(v) // 2] for v in samples]
rows = []
tot = 0.0
for op, ms, k in zip(plan.ops, med, kn):
    fl = FL.op_flops(op); by = FL.op_bytes(op); tot += ms
    rows.append(dict(name=getattr(op, "name", ""), kernel=k, ms=round(ms, 4), gflop=round(fl / 1e9, 2), tflops=round(fl / ms / 1e9, 1) if fl else None, gbps=round(by / ms / 1e6, 0)))
by_k = {}
for r in rows:
    d = by_k.setdefault(r["kernel"], [0.0, 0, 0.0]); d[0] += r["ms"]; d[1] += 1; d[2] += r["gflop"]
print(f"total {tot:.3f} ms for {a.scenes} scene(s); {sum(r['gflop'] for r in rows) / 1e3:.3f} TF -> {sum(r['gflop'] for r in rows) / tot:.0f} TFLOP/s")
for k, (ms, cnt, gf) in sorted(by_k.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:50s} {ms:8.3f} ms {cnt:3d} launches {gf / ms if gf else 0:8.0f} TF/s")
for r in sorted(rows, key=lambda r: -r["ms"])[:14]:
    print("   ", r)
if a.json:
    json.dump(rows, open(a.json, "w"))
