#!/usr/bin/env python
"""attention3.hip vs attention2.hip at level-0 shape, kv length swept: the slope is the cost of a (128 queries x 64 kv) workgroup tile, the
intercept the per-workgroup fixed cost.  Usage: [MDX_LIB_PATH=side.so] python tools/a3_sweep.py [--views 576] [--tks 256,704,1408] [--xview]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O, packing as PK  # noqa: E402
BF = torch.bfloat16
ap = argparse.ArgumentParser(); ap.add_argument("--views", type=int, default=576); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tks", default="256,704,1408"); ap.add_argument("--attn3", default="0,1"); ap.add_argument("--xview", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda"); B = a.views; T, C, d = 1400, 320, 40
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
st = torch.cuda.current_stream().cuda_stream
q = r(B, T, C)
for tk in [int(x) for x in a.tks.split(",")]:
    k = r(B, tk, C); vt = torch.zeros(B, C, PK.round_up(tk, 8), dtype=BF, device=dev); vt[:, :, :tk] = r(B, C, tk)
    o = torch.empty(B, T, C, dtype=BF, device=dev)
    kw = dict(kvmap=torch.tensor([(i // 6) * 6 + ((i % 6 + s) % 6) for i in range(B) for s in (5, 1)], dtype=torch.int32, device=dev), nsrc=2) if a.xview else {}
    for a3 in [int(x) for x in a.attn3.split(",")]:
        with L.options(ATTN3=a3, ATTN2_RES=0):
            code, desc = O.Attn(q, k, vt, o, heads=8, Tk=tk, scale=d ** -0.5, q_prescaled=True, **kw).lower()
            for _ in range(2): L.call_op(code, desc, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps): L.call_op(code, desc, st)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            ntile = (tk + 63) // 64 * (2 if a.xview else 1)
            print(f"Tk={tk:5d} tiles={ntile:3d} attn3={a3} {us:9.1f} us  {us / ntile:7.2f} us/tile  {(L.lib().mdx_last_kernel() or b'').decode()}", flush=True)
