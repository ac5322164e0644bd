#!/usr/bin/env python
"""Time a few representative shapes of the XL GEMM / conv main loop (gemm_xl.hip) — the A/B tool for its schedule knobs.
Usage: [MDX_XL_DBG=..] python tools/xlone.py [--views 384] [--reps 10] [--only c160,c256,...]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O  # noqa: E402

BF = torch.bfloat16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=384)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--bn", type=str, default="", help="comma list of XL tile widths to force one after the other (160,256,320); default: the library's choice")
    ap.add_argument("--raster", type=str, default="", help="semicolon list of forced panel shapes 'gm,gn' of the XCD-blocked tile order (XL_GM / XL_GN), "
                                                         "timed after the cost model's choice, e.g. '16,2;32,1;8,4'")
    a = ap.parse_args()
    dev = torch.device("cuda")
    B = a.views
    ws = torch.empty(64 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
    cases = {}

    def conv(name, h, w, cin, cout):
        def mk():
            x = r(B, h, w, cin); wt = r(cout, 3, 3, cin); y = torch.empty(B, h, w, cout, dtype=BF, device=dev)
            return O.Conv(x, wt, y, bias=torch.randn(cout, device=dev), R=r(B, h, w, cout), ws=ws), 2.0 * B * h * w * cout * 9 * cin
        cases[name] = mk

    def gemm(name, M, N, K, epi=0, res=False):
        def mk():
            A = r(M, K); W = r(N, K); No = N // 2 if epi == 1 else N
            C = torch.empty(M, No, dtype=BF, device=dev)
            return O.Gemm(A, W, C, bias=torch.randn(N, device=dev), R=r(M, No) if res else None, epilogue=epi, ws=ws), 2.0 * M * N * K
        cases[name] = mk

    def gemm_vt(name, T, C):            # the batch-flattened V^T projection of levels 1 / 2 (GCParams.col_split epilogue)
        def mk():
            from magicdrive_amd import packing as PK
            X = r(B, T, C); Wv = r(C, C); Vt = torch.zeros(B, C, PK.round_up(T, 8), dtype=BF, device=dev)
            return O.Gemm(Wv, X, Vt[:, :, :T]), 2.0 * B * T * C * C
        cases[name] = mk

    def conv_out(name, h, w, cin):      # the UNet's conv_out: Cout = 4, fp32 eps (direct K-parallel kernels)
        def mk():
            x = r(B, h, w, cin); wt = r(4, 3, 3, cin); y = torch.empty(B, h, w, 4, dtype=torch.float32, device=dev)
            return O.Conv(x, wt, y, bias=torch.randn(4, device=dev), direct=True), 2.0 * B * h * w * 4 * 9 * cin
        cases[name] = mk

    gemm_vt("vt_L1", 350, 640)
    gemm_vt("vt_L2", 91, 1280)
    conv_out("convout_28x50_320", 28, 50, 320)
    conv("c_28x50_640_320", 28, 50, 640, 320)
    conv("c_28x50_960_320", 28, 50, 960, 320)
    conv("c_14x25_640_640", 14, 25, 640, 640)
    conv("c_14x25_1920_640", 14, 25, 1920, 640)
    conv("c_7x13_2560_1280", 7, 13, 2560, 1280)
    conv("c160_28x50_640_640", 28, 50, 640, 640)
    conv("c160_28x50_320_320", 28, 50, 320, 320)
    conv("c256_14x25_1280_1280", 14, 25, 1280, 1280)
    conv("c256_7x13_1280_1280", 7, 13, 1280, 1280)
    gemm("g256_geglu_L1", B * 350, 5120, 640, epi=1)
    gemm("g256_ffout_L1", B * 350, 640, 2560, res=True)
    gemm("g160_ffout_L0", B * 1400, 320, 1280, res=True)
    gemm("g256_qk_L1", B * 350, 1280, 640)
    gemm("g256_cc_L1", B * 350, 640, 640, res=True)
    gemm("g256_cc_L2", B * 91, 1280, 1280, res=True)
    gemm("g256_geglu_L2", B * 91, 10240, 1280, epi=1)
    gemm("g256_ffout_L2", B * 91, 1280, 5120, res=True)
    gemm("g256_qk_L2", B * 91, 2560, 1280)
    gemm("g_geglu_L0", B * 1400, 2560, 320, epi=1)
    gemm("g_out_L0", B * 1400, 320, 320, res=True)
    only = [s for s in a.only.split(",") if s]
    st = torch.cuda.current_stream().cuda_stream
    bns = [int(x) for x in a.bn.split(",") if x] or [0]
    for name, mk in cases.items():
        if only and not any(name.startswith(o) for o in only):
            continue
        op, fl = mk()
        code, desc = op.lower()
        rasters = [(0, 0)] + [tuple(int(v) for v in r_.split(",")) for r_ in a.raster.split(";") if r_]
        for bn, (gm, gn) in [(b_, r_) for b_ in bns for r_ in rasters]:
            with L.options(**dict(({"GEMM_XL": 2, "XL_BN": bn} if bn else {}), XL_GM=gm, XL_GN=gn)):
                try:
                    for _ in range(2):
                        L.call_op(code, desc, st)
                except L.MdxError as e:
                    print(f"{name:26s} bn={bn}: {e}")
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    L.call_op(code, desc, st)
                e1.record()
                torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            print(f"{name:26s} {('bn=%d' % bn) if bn else '':7s} {('gm,gn=%d,%d' % (gm, gn)) if gm or gn else '':12s} {us:9.1f} us {fl / us / 1e6:8.1f} TF/s  {(L.lib().mdx_last_kernel() or b'').decode()}", flush=True)
        del op
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
