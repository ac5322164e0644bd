#!/usr/bin/env python
"""Kernel micro-benchmark: the representative GEMM / conv / attention / norm shapes of one denoising step
(SD-1.5 size, `--views` = c*b*6 views), each timed with HIP events on the launch stream.  Seconds, not minutes —
the A/B tool for kernel work.  Usage: python tools/kbench.py [--views 24] [--only gemm,conv,attn,norm] [--reps 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O, packing as PK  # noqa: E402

BF = torch.bfloat16


def timeit(op, reps):
    st = torch.cuda.current_stream().cuda_stream
    code, desc = op.lower()
    for _ in range(3):
        L.call_op(code, desc, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.call_op(code, desc, st)
    e1.record()
    torch.cuda.synchronize()
    global LAST_KERNEL
    LAST_KERNEL = (L.lib().mdx_last_kernel() or b"").decode()
    return e0.elapsed_time(e1) / reps * 1e3    # us


LAST_KERNEL = ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=24)
    ap.add_argument("--only", type=str, default="gemm,conv,attn,norm")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--prescaled", action="store_true", help="attention: Q pre-scaled by scale * log2(e) (q_prescaled, as the engine packs to_q)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    B = a.views
    ws = torch.empty(64 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
    lv = [(28, 50, 320), (14, 25, 640), (7, 13, 1280), (4, 7, 1280)]
    tot = {}
    rows = []

    def rec(kind, name, us, gf):
        rows.append((kind, name, us, gf, LAST_KERNEL))
        t = tot.setdefault(kind, [0.0, 0.0]); t[0] += us; t[1] += gf

    if "gemm" in a.only:
        for (h, w, C) in lv[:3]:
            M = B * h * w
            x = r(M, C); res = r(M, C); bias = torch.randn(C, device=dev)
            for name, N, K, epi in [("out(C,C)+res", C, C, 0), ("qk(2C,C)", 2 * C, C, 0), ("ff.out(C,4C)+res", C, 4 * C, 0), ("geglu(8C,C)", 8 * C, C, 1)]:
                A = r(M, K); W = r(N, K)
                No = N // 2 if epi else N
                Cc = torch.empty(M, No, dtype=BF, device=dev)
                op = O.Gemm(A, W, Cc, bias=torch.randn(N, device=dev), R=res if No == C and "res" in name else None, epilogue=epi, ws=ws)
                us = timeit(op, a.reps)
                rec("gemm", f"M={M} {name}", us, 2.0 * M * N * K / 1e9)
            # V^T batched
            T = h * w
            X3 = r(B, T, C); Wv = r(C, C); Vt = torch.zeros(B, C, PK.round_up(T, 8), dtype=BF, device=dev)
            us = timeit(O.Gemm(Wv, X3, Vt[:, :, :T]), a.reps)
            rec("gemm", f"M={M} vT batched", us, 2.0 * M * C * C / 1e9)
    if "conv" in a.only:
        for (h, w, Cin, Cout) in [(28, 50, 320, 320), (28, 50, 640, 320), (28, 50, 960, 320), (14, 25, 640, 640), (14, 25, 1280, 640), (14, 25, 1920, 640),
                                   (7, 13, 1280, 1280), (7, 13, 2560, 1280), (4, 7, 1280, 1280), (4, 7, 2560, 1280), (28, 50, 640, 640), (14, 25, 1280, 1280)]:
            x = r(B, h, w, Cin); wt = r(Cout, 3, 3, Cin); y = torch.empty(B, h, w, Cout, dtype=BF, device=dev)
            us = timeit(O.Conv(x, wt, y, bias=torch.randn(Cout, device=dev), R=r(B, h, w, Cout), ws=ws), a.reps)
            rec("conv", f"{h}x{w} {Cin}->{Cout}", us, 2.0 * B * h * w * Cout * 9 * Cin / 1e9)
    if "attn" in a.only:
        for (h, w, C) in lv:
            T = h * w; d = C // 8
            qk = r(B, T, 2 * C); vt = torch.zeros(B, C, PK.round_up(T, 8), dtype=BF, device=dev); vt[:, :, :T] = r(B, C, T)
            o = torch.empty(B, T, C, dtype=BF, device=dev)
            pre = a.prescaled
            if pre:
                qk[:, :, :C] = (qk[:, :, :C].float() * (d ** -0.5 * 1.4426950408889634)).to(BF)
            us = timeit(O.Attn(qk[:, :, :C], qk[:, :, C:], vt, o, heads=8, Tk=T, scale=d ** -0.5, q_prescaled=pre), a.reps)
            rec("attn", f"self T={T} d={d}", us, 4.0 * B * T * T * C / 1e9)
            kvmap = torch.tensor([(i // 6) * 6 + ((i % 6 + s) % 6) for i in range(B) for s in (5, 1)], dtype=torch.int32, device=dev)
            us = timeit(O.Attn(qk[:, :, :C], qk[:, :, C:], vt, o, heads=8, Tk=T, scale=d ** -0.5, kvmap=kvmap, nsrc=2, q_prescaled=pre), a.reps)
            rec("attn", f"xview T={T} d={d}", us, 8.0 * B * T * T * C / 1e9)
            S = 78
            kc = r(B, S, C); vtc = torch.zeros(B, C, 80, dtype=BF, device=dev); vtc[:, :, :S] = r(B, C, S)
            q = r(B, T, C)
            if pre:
                q = (q.float() * (d ** -0.5 * 1.4426950408889634)).to(BF)
            us = timeit(O.Attn(q, kc, vtc, o, heads=8, Tk=S, scale=d ** -0.5, q_prescaled=pre), a.reps)
            rec("attn", f"ctx T={T} S={S} d={d}", us, 4.0 * B * T * S * C / 1e9)
    if "norm" in a.only:
        for (h, w, C) in [(28, 50, 320), (28, 50, 640), (14, 25, 640), (14, 25, 1920), (7, 13, 1280), (4, 7, 2560)]:
            x = r(B, h * w, C); y = torch.empty_like(x)
            us = timeit(O.GroupNorm(x, y, torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5, True, ws=ws), a.reps)
            rec("norm", f"GN {h}x{w} C={C}", us, 0)
        for (h, w, C) in lv[:3]:
            x = r(B * h * w, C); y = torch.empty_like(x)
            us = timeit(O.LayerNorm(x, y, torch.ones(C, device=dev), torch.zeros(C, device=dev)), a.reps)
            rec("norm", f"LN M={B*h*w} C={C}", us, 0)
    for kind, name, us, gf, kern in rows:
        print(f"{kind:5s} {name:34s} {us:9.1f} us  {gf/us*1e3 if gf else 0:8.1f} TF/s  {kern}")
    for k, (us, gf) in tot.items():
        print(f"== {k}: {us:.0f} us total, {gf/us*1e3 if gf else 0:.1f} TF/s aggregate")


if __name__ == "__main__":
    main()
