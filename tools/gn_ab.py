#!/usr/bin/env python
"""A/B of the GroupNorm read order (GN_REVERSE) in a producer -> GroupNorm -> consumer chain, as in a resnet: a conv writes the tensor,
GroupNorm (statistics + apply) normalises it, the next conv reads it.  Times the GroupNorm op alone (HIP events) inside the chain.
Usage: python tools/gn_ab.py [--views 768]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O  # noqa: E402
BF = torch.bfloat16
ap = argparse.ArgumentParser(); ap.add_argument("--views", type=int, default=768); ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda"); B = a.views
ws = torch.empty(64 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
st = torch.cuda.current_stream().cuda_stream
for (h, w, C) in [(28, 50, 320), (14, 25, 640), (28, 50, 640), (7, 13, 1280)]:
    x0 = r(B, h, w, C); wt = r(C, 3, 3, C); x = torch.empty(B, h, w, C, dtype=BF, device=dev); y = torch.empty_like(x); z = torch.empty_like(x)
    conv1 = O.Conv(x0, wt, x, bias=torch.randn(C, device=dev), ws=ws)
    gn = O.GroupNorm(x.view(B, h * w, C), y.view(B, h * w, C), torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5, True, ws=ws)
    conv2 = O.Conv(y, wt, z, bias=torch.randn(C, device=dev), ws=ws)
    for rev in (1, 0, 1, 0):
        with L.options(GN_REVERSE=rev):
            tg = tc = 0.0
            for i in range(a.reps + 1):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                O.run_ops([conv1]); e[0].record(); O.run_ops([gn]); e[1].record(); O.run_ops([conv2]); e[2].record()
                torch.cuda.synchronize()
                if i:
                    tg += e[0].elapsed_time(e[1]); tc += e[1].elapsed_time(e[2])
            print(f"{h}x{w} C={C} GN_REVERSE={rev}: groupnorm {tg / a.reps * 1e3:8.1f} us, following conv {tc / a.reps * 1e3:8.1f} us", flush=True)
