#!/usr/bin/env python
"""Per-tile phase times of the XL GEMM / conv kernel (gemm_xl.hip, MDX_XL_TIMING=1: s_memtime stamps of wave 0 of every workgroup):
bookkeeping, first-slab latency, main loop, C staging, store issue — averaged over the workgroups of one launch, for a few shapes.
Usage: MDX_XL_TIMING=1 python tools/xl_timing.py [--views 384]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MDX_XL_TIMING", "1")
from magicdrive_amd import _lib as L, ops as O  # noqa: E402
BF = torch.bfloat16
ap = argparse.ArgumentParser(); ap.add_argument("--views", type=int, default=384)
a = ap.parse_args()
dev = torch.device("cuda"); B = a.views
ws = torch.zeros(64 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
st = torch.cuda.current_stream().cuda_stream
def conv(h, w, cin, cout, res=True):
    x = r(B, h, w, cin); wt = r(cout, 3, 3, cin); y = torch.empty(B, h, w, cout, dtype=BF, device=dev)
    return O.Conv(x, wt, y, bias=torch.randn(cout, device=dev), R=r(B, h, w, cout) if res else None, ws=ws)
def gemm(M, N, K, epi=0, res=False):
    A = r(M, K); W = r(N, K); No = N // 2 if epi == 1 else N
    return O.Gemm(A, W, torch.empty(M, No, dtype=BF, device=dev), bias=torch.randn(N, device=dev), R=r(M, No) if res else None, epilogue=epi, ws=ws)
cases = [("conv 14x25 1280->1280 +res", conv(14, 25, 1280, 1280)), ("conv 28x50 640->640 +res", conv(28, 50, 640, 640)),
         ("gemm qk L1 (N=1280,K=640)", gemm(B * 350, 1280, 640)), ("gemm out L1 +res (N=640,K=640)", gemm(B * 350, 640, 640, res=True)),
         ("gemm ff.out L1 +res (N=640,K=2560)", gemm(B * 350, 640, 2560, res=True)), ("gemm geglu L1 (N=5120,K=640)", gemm(B * 350, 5120, 640, epi=1))]
for name, op in cases:
    code, desc = op.lower()
    for _ in range(2): L.call_op(code, desc, st)
    ws.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.call_op(code, desc, st); e1.record(); torch.cuda.synchronize()
    kern = (L.lib().mdx_last_kernel() or b"").decode()
    t = ws.view(torch.int64)[: 8 * 200000].view(-1, 8)
    t = t[(t[:, 0] != 0) & (t[:, 5] != 0)].double()
    n = t.shape[0]
    t = t[:, [0, 1, 2, 3, 6, 7, 4, 5]]                               # program order: entry, setup, slab 0, loop end, barrier 1, staged, barrier 2, stores issued
    d = (t[:, 1:] - t[:, :-1]).mean(0) / 100.0                       # s_memtime ticks (shader clock); printed in units of 100 ticks
    tot = ((t[:, 7] - t[:, 0]).mean() / 100.0).item()
    span = ((t[:, 7].max() - t[:, 0].min()) / 100.0).item()
    us = e0.elapsed_time(e1) * 1e3 / max(1.0, n / 256.0) / tot          # microseconds per printed unit, from the launch time and the rounds of tiles
    # are the CUs synchronised, and do the burst-bound phases (first slab, store issue) shrink once rounds of tiles have drifted apart?
    order = torch.argsort(t[:, 0])
    ts_ = t[order]
    nq = 4
    for qi in range(nq):
        seg = ts_[qi * n // nq:(qi + 1) * n // nq]
        dd = (seg[:, 1:] - seg[:, :-1]).mean(0) / 100.0
        print(f"    tiles by start time, quarter {qi}: first slab {dd[1]*us:5.2f}  main {dd[2]*us:6.2f}  stage {dd[4]*us:5.2f}  store {dd[6]*us:5.2f} us; "
              f"start-time spread inside the quarter's first 256 tiles {(seg[:256, 0].max() - seg[:256, 0].min()).item() / 100.0 * us:7.1f} us")
    print(f"{name:36s} {kern:30s} {n:6d} tiles, launch {e0.elapsed_time(e1)*1e3:8.1f} us (stamps span {span:8.1f}) | per tile {tot*us:6.1f} us = setup {d[0]*us:4.1f} + first slab {d[1]*us:4.1f} + main loop {d[2]*us:6.1f} + barrier {d[3]*us:4.1f} + stage {d[4]*us:4.1f} + barrier {d[5]*us:4.1f} + residual/store {d[6]*us:4.1f}")
