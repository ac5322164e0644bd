#!/usr/bin/env python
"""Time the level-0 / level-1 attention shapes (MDX_ATTN_DBG ablations).  Usage: [MDX_ATTN_DBG=..] python tools/attnone.py [--views 384]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O, packing as PK  # noqa: E402
BF = torch.bfloat16
ap = argparse.ArgumentParser(); ap.add_argument("--views", type=int, default=384); ap.add_argument("--reps", type=int, default=5); ap.add_argument("--layout", default="token", help="token: [B,T,heads*d] (product layout); head: heads folded into the batch, K rows contiguous, V^T rows 128-B aligned; head160: as head but K rows at a 160-B stride")
a = ap.parse_args()
dev = torch.device("cuda"); B = a.views
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
st = torch.cuda.current_stream().cuda_stream
for (T, C, xv) in [(1400, 320, False), (1400, 320, True), (350, 640, False), (350, 640, True)]:
    d = C // 8
    if a.layout == "token":
        qk = r(B, T, 2 * C); vt = torch.zeros(B, C, PK.round_up(T, 8), dtype=BF, device=dev); vt[:, :, :T] = r(B, C, T)
        o = torch.empty(B, T, C, dtype=BF, device=dev)
        kw = dict(kvmap=torch.tensor([(i // 6) * 6 + ((i % 6 + s) % 6) for i in range(B) for s in (5, 1)], dtype=torch.int32, device=dev), nsrc=2) if xv else {}
        code, desc = O.Attn(qk[:, :, :C], qk[:, :, C:], vt, o, heads=8, Tk=T, scale=d ** -0.5, q_prescaled=True, **kw).lower()   # the product's form: to_q pre-scaled
    else:
        BH = B * 8
        qq = r(BH, T, d); kk = r(BH, T, 2 * d if a.layout == "head160" else d)[:, :, :d]
        vt = torch.zeros(BH, d, PK.round_up(T, 64), dtype=BF, device=dev); vt[:, :, :T] = r(BH, d, T)
        o = torch.empty(BH, T, d, dtype=BF, device=dev)
        kw = dict(kvmap=torch.tensor([((i // 8) // 6 * 6 + (((i // 8) % 6 + s) % 6)) * 8 + i % 8 for i in range(BH) for s in (5, 1)], dtype=torch.int32, device=dev), nsrc=2) if xv else {}
        code, desc = O.Attn(qq, kk, vt, o, heads=1, Tk=T, scale=d ** -0.5, **kw).lower()
    for _ in range(2): L.call_op(code, desc, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): L.call_op(code, desc, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.reps * 1e3
    fl = (8.0 if xv else 4.0) * B * T * T * C
    print(f"T={T} d={d} {'xview' if xv else 'self '} {us:9.1f} us {fl/us/1e6:7.1f} TF/s  {(L.lib().mdx_last_kernel() or b'').decode()}", flush=True)
# text-context attention of level 0 (S = 1 + 77 tokens, K / V^T from the prologue)
if a.layout == "token":
    T, C, S = 1400, 320, 78
    q = r(B, T, C); kc = r(B, S, C); vt = torch.zeros(B, C, PK.round_up(S, 8), dtype=BF, device=dev); vt[:, :, :S] = r(B, C, S)
    o = torch.empty(B, T, C, dtype=BF, device=dev)
    code, desc = O.Attn(q, kc, vt, o, heads=8, Tk=S, scale=40 ** -0.5, q_prescaled=True).lower()
    for _ in range(2): L.call_op(code, desc, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): L.call_op(code, desc, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.reps * 1e3
    print(f"T={T} S={S} d=40 ctx   {us:9.1f} us {4.0 * B * T * S * C / us / 1e6:7.1f} TF/s  {(L.lib().mdx_last_kernel() or b'').decode()}", flush=True)
