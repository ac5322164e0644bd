#!/usr/bin/env python
"""W-direct persistent GEMM (csrc/gemm_xd.hip) against the LDS-both persistent kernel (gemm_xlp_kernel) and an fp32 reference:
same descriptor, option XD = 1 / 0, outputs compared bit for bit, then both timed.
Usage: python tools/xdone.py [--views 576] [--reps 10] [--only qk_L1,...] [--check-only]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O, packing as PK  # noqa: E402

BF = torch.bfloat16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=576)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--f16", action="store_true")
    ap.add_argument("--timing", action="store_true", help="-DXD_TIMING side builds: print the per-workgroup cycle split (main loops / conversions) the kernel left in ws")
    ap.add_argument("--stress", type=int, default=0, help="run the XD = 1 launch this many more times, each compared with the XD = 0 output")
    a = ap.parse_args()
    dev = torch.device("cuda")
    dt = torch.float16 if a.f16 else BF
    B = a.views
    ws = torch.empty(64 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(dt)
    cases = []

    def gemm(name, M, N, K, epi=0, res=False, bias=True, ldc_extra=0):
        cases.append((name, M, N, K, epi, res, bias, ldc_extra))

    gemm("qk_L1", B * 350, 1280, 640)
    gemm("cc_L1", B * 350, 640, 640, res=True)
    gemm("geglu_L1", B * 350, 5120, 640, epi=1)
    gemm("ffout_L1", B * 350, 640, 2560, res=True)
    gemm("qk_L2", B * 91, 2560, 1280)
    gemm("cc_L2", B * 91, 1280, 1280, res=True)
    gemm("geglu_L2", B * 91, 10240, 1280, epi=1)
    gemm("ffout_L2", B * 91, 1280, 5120, res=True)
    gemm("projin_L1", B * 350, 640, 640)
    gemm("tail_nobias", B * 350 - 37, 1296, 768, bias=False)
    gemm("tail_slice", B * 350 - 5, 640, 640, res=True, ldc_extra=64)
    only = [s for s in a.only.split(",") if s]
    st = torch.cuda.current_stream().cuda_stream
    bad = 0
    for name, M, N, K, epi, res, bias, ldx in cases:
        if only and not any(name.startswith(o) for o in only):
            continue
        A = r(M, K); W = r(N, K) * 0.1
        if epi == 1:
            W, bv = PK.pack_geglu(W.float(), torch.randn(N, device=dev, generator=g), dt)
        else:
            bv = torch.randn(N, device=dev, generator=g)
        No = N // 2 if epi == 1 else N
        Cbuf = torch.zeros(M, No + ldx, dtype=dt, device=dev)
        C = Cbuf[:, :No]
        R = r(M, No) if res else None
        Wq = PK.pack_wq(W)
        outs = {}
        times = {}
        for xd in (1, 0):
            op = O.Gemm(A, W, C, bias=bv if bias else None, R=R, epilogue=epi, ws=ws, Wq=Wq)
            code, desc = op.lower()
            with L.options(XD=xd):
                Cbuf.fill_(7.0)
                L.call_op(code, desc, st, dtype=L.DTYPE_F16 if a.f16 else L.DTYPE_BF16)
                torch.cuda.synchronize()
                kern = (L.lib().mdx_last_kernel() or b"").decode()
                outs[xd] = Cbuf.clone()
                if not a.check_only:
                    for _ in range(2):
                        L.call_op(code, desc, st, dtype=L.DTYPE_F16 if a.f16 else L.DTYPE_BF16)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.reps):
                        L.call_op(code, desc, st, dtype=L.DTYPE_F16 if a.f16 else L.DTYPE_BF16)
                    e1.record()
                    torch.cuda.synchronize()
                    times[xd] = (e0.elapsed_time(e1) / a.reps * 1e3, kern)
                else:
                    times[xd] = (0.0, kern)
        if a.timing:
            op = O.Gemm(A, W, C, bias=bv if bias else None, R=R, epilogue=epi, ws=ws, Wq=Wq)
            code, desc = op.lower()
            ws.zero_()
            with L.options(XD=1):
                L.call_op(code, desc, st, dtype=L.DTYPE_F16 if a.f16 else L.DTYPE_BF16)
            torch.cuda.synchronize()
            tt = ws.view(torch.int64)[:256 * 8].reshape(256, 8).double()
            tiles = tt[:, 2].clamp_min(1)
            print(f"   timing (s_memtime ticks, mean over workgroups): per tile main loop {(tt[:, 0] / tiles).mean():.0f}, conversion {(tt[:, 1] / tiles).mean():.0f}; "
                  f"units 0-4 (incl. the tile's set-up) {(tt[:, 4] / tiles).mean():.0f}, units 5-8 {(tt[:, 5] / tiles).mean():.0f}; tiles per workgroup {tt[:, 2].mean():.1f}; whole kernel {tt[:, 3].mean():.0f} (max {tt[:, 3].max():.0f}); units per tile {K // 64}")
        if a.stress:
            op = O.Gemm(A, W, C, bias=bv if bias else None, R=R, epilogue=epi, ws=ws, Wq=Wq)
            code, desc = op.lower()
            nbad = 0
            with L.options(XD=1):
                for it in range(a.stress):
                    Cbuf.fill_(7.0)
                    L.call_op(code, desc, st, dtype=L.DTYPE_F16 if a.f16 else L.DTYPE_BF16)
                    torch.cuda.synchronize()
                    if not torch.equal(Cbuf, outs[0]):
                        nbad += 1
                        if nbad == 1:
                            outs[1] = Cbuf.clone()
            print(f"   stress: {nbad} of {a.stress} launches differ from XD=0")
        # reference on a row sample (fp32)
        idx = torch.randint(0, M, (2048,), device=dev, generator=g)
        idx[:8] = torch.arange(M - 8, M, device=dev)
        ref = A[idx].float() @ W.float().t()
        if bias:
            ref = ref + bv[None, :]
        if epi == 1:
            rr_ = ref.reshape(-1, N // 64, 2, 32)
            ref = (rr_[:, :, 0] * torch.nn.functional.gelu(rr_[:, :, 1])).reshape(-1, No)
        ref = ref.to(dt).float()
        if res:
            ref = ref + R[idx].float()
        got = outs[1][idx, :No].float()
        err = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()
        same = torch.equal(outs[1], outs[0])
        ndiff = (outs[1] != outs[0]).sum().item()
        pad_ok = bool((outs[1][:, No:] == 7.0).all().item()) if ldx else True
        if ndiff:
            d = (outs[1][:, :No] != outs[0][:, :No]).nonzero()
            rows, cols = d[:, 0], d[:, 1]
            tiles = torch.stack([rows // 256, cols // (128 if epi == 1 else 256)], 1).unique(dim=0)
            print(f"   mismatching tiles ({tiles.shape[0]}): {tiles[:24].tolist()}")
            print(f"   row blocks (row % 256 // 16): {torch.bincount((rows % 256) // 16, minlength=16).tolist()}")
            print(f"   row in block (row % 16):      {torch.bincount(rows % 16, minlength=16).tolist()}")
            print(f"   wave (col % 256 // 64):       {torch.bincount((cols % 256) // 64, minlength=4).tolist()}")
            print(f"   col in wave // 8:             {torch.bincount((cols % 64) // 8, minlength=8).tolist()}")
            r0, c0 = rows[0].item(), cols[0].item()
            print(f"   sevens among the mismatches: {(outs[1][:, :No][rows, cols] == 7.0).sum().item()} of {rows.numel()}")
            print(f"   first: row {r0} col {c0}: xd {outs[1][r0, c0 - c0 % 8:c0 - c0 % 8 + 8].tolist()} xlp {outs[0][r0, c0 - c0 % 8:c0 - c0 % 8 + 8].tolist()}")
        ok = err < 2e-2 and pad_ok and times[1][1].startswith("gemm_xd")
        bad += 0 if ok else 1
        fl = 2.0 * M * N * K
        t1, t0 = times[1][0], times[0][0]
        print(f"{name:14s} M={M:7d} N={N:5d} K={K:5d}  rel err vs fp32 {err:.2e}  identical to XD=0: {same} ({ndiff} differ)  pad {pad_ok}  "
              + (f"xd {t1:8.1f} us {fl / t1 / 1e6:7.1f} TF/s | xlp {t0:8.1f} us {fl / t0 / 1e6:7.1f} TF/s | {t0 / t1:5.3f}x  " if not a.check_only else "")
              + f"[{times[1][1]} | {times[0][1]}] {'ok' if ok else 'FAIL'}", flush=True)
        del A, W, Cbuf, R, Wq, outs
        torch.cuda.empty_cache()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
