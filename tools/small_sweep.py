#!/usr/bin/env python
"""Generic-tile choices at the latency operating point (1 scene = 6 views): every representative GEMM / conv shape under forced tile sizes, split-K factors and
stage counts; 50 back-to-back launches between two HIP events.  Usage: python tools/small_sweep.py [--views 6]"""
import argparse, os, sys, itertools
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O
ap = argparse.ArgumentParser(); ap.add_argument("--views", type=int, default=6); ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
BF = torch.bfloat16; dev = torch.device("cuda")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
ws = torch.empty(64 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
V = a.views
def gemm(M, N, K, sk=0):
    return O.Gemm(r(M, K), r(N, K), torch.empty(M, N, dtype=BF, device=dev), bias=torch.randn(N, device=dev), R=r(M, N), splitk=sk, ws=ws)
def conv(H, W, Ci, Co, sk=0):
    return O.Conv(r(V, H, W, Ci), r(Co, 3, 3, Ci), torch.empty(V, H, W, Co, dtype=BF, device=dev), bias=torch.randn(Co, device=dev), R=r(V, H, W, Co), splitk=sk, ws=ws)
shapes = [("gemm L1 640x640", lambda sk: gemm(V * 350, 640, 640, sk), 2 * V * 350 * 640 * 640),
          ("gemm L1 ff.out 640x2560", lambda sk: gemm(V * 350, 640, 2560, sk), 2 * V * 350 * 640 * 2560),
          ("gemm L2 1280x1280", lambda sk: gemm(V * 91, 1280, 1280, sk), 2 * V * 91 * 1280 * 1280),
          ("gemm L0 320x1280", lambda sk: gemm(V * 1400, 320, 1280, sk), 2 * V * 1400 * 320 * 1280),
          ("conv L0 320->320", lambda sk: conv(28, 50, 320, 320, sk), 2 * V * 1400 * 320 * 2880),
          ("conv L0 640->320", lambda sk: conv(28, 50, 640, 320, sk), 2 * V * 1400 * 320 * 5760),
          ("conv L1 640->640", lambda sk: conv(14, 25, 640, 640, sk), 2 * V * 350 * 640 * 5760),
          ("conv L2 1280->1280", lambda sk: conv(7, 13, 1280, 1280, sk), 2 * V * 91 * 1280 * 11520),
          ("conv L3 1280->1280", lambda sk: conv(4, 7, 1280, 1280, sk), 2 * V * 28 * 1280 * 11520)]
st = torch.cuda.current_stream().cuda_stream
def timeit(op):
    code, desc = op.lower()
    for _ in range(3): L.call_op(code, desc, st)
    k = (L.lib().mdx_last_kernel() or b"").decode()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): L.call_op(code, desc, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / a.reps, k
for name, mk, fl in shapes:
    print(f"== {name}  ({fl / 1e9:.1f} GFLOP)")
    t, k = timeit(mk(0)); print(f"   default                                  {t:7.1f} us  {fl / t / 1e6:7.1f} TF/s  {k}")
    res = []
    for bm, bn, sk, ns, xl, c3 in itertools.product((64, 128), (64, 128), (1, 2, 4, 8, 0), (1, 3), (0,), (0, 1)):
        if c3 and (not name.startswith("conv") or sk != 1 or bm != 128 or bn != 128 or ns != 1): continue
        if name.startswith("gemm") and sk in (2, 8): continue
        with L.options(GEMM_BM=bm, GEMM_BN=bn, GEMM_STAGES=ns, GEMM_XL=xl, CONV3=c3, GEMM_WS=0):
            try:
                t, k = timeit(mk(sk))
            except Exception as e:
                continue
        res.append((t, f"bm{bm} bn{bn} splitk{sk} ns{ns} c3={c3}", k))
    for t, tag, k in sorted(res)[:6]:
        print(f"   {tag:40s} {t:7.1f} us  {fl / t / 1e6:7.1f} TF/s  {k}")
    with L.options(GEMM_XL=2):
        t, k = timeit(mk(0)); print(f"   XL forced                                {t:7.1f} us  {fl / t / 1e6:7.1f} TF/s  {k}")
