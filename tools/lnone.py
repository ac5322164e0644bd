#!/usr/bin/env python
"""Time the level-0 LayerNorm -> projection pairs with the norm inside the weight-stationary GEMM (MdxGemmDesc.ln_eps) against the same
projection on pre-normalised rows and against LayerNorm + projection — the A/B tool for gemm_ws.hip's fused-LayerNorm variants
(side builds with -DMDX_WS_LN_ABLATE=1|2|3 through MDX_LIB_PATH separate the sums, the extra fragment read and the epilogue).
Usage: python tools/lnone.py [--views 768] [--reps 5]"""
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
    ap.add_argument("--views", type=int, default=768)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda")
    T, C = 1400, 320
    M = a.views * T
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
    x = r(M, C); scratch = torch.empty(M, C, dtype=BF, device=dev); y = torch.empty(M, C, dtype=BF, device=dev)
    g = torch.ones(C, device=dev); bt = torch.zeros(C, device=dev)

    def mk(N, epi=0, vt=False, ln=False):
        W = r(N, C); b = torch.randn(N, device=dev); cs = W.float().sum(1)
        No = N // 2 if epi == 1 else (2 * C if vt else N)
        Cm = torch.empty(M, No, dtype=BF, device=dev)
        kw = dict(bias=b, epilogue=epi)
        if vt:
            kw.update(Vt=torch.empty(a.views, C, T, dtype=BF, device=dev), vt_from=2 * C, vt_T=T)
        if ln:
            kw.update(ln_eps=1e-5, ln_csum=cs, ln_scratch=scratch)
        return O.Gemm(x, W, Cm, **kw)

    cases = [("qkv", dict(N=3 * C, vt=True)), ("to_q", dict(N=C)), ("geglu", dict(N=8 * C, epi=1))]
    ln_op = O.LayerNorm(x, y, g, bt, 1e-5)

    def timeit(ops):
        O.run_ops(ops); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            O.run_ops(ops)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps * 1e3

    t_ln = timeit([ln_op])
    print(f"layernorm alone                      {t_ln:8.1f} us")
    for name, kw in cases:
        plain = timeit([mk(**kw)])
        fused = timeit([mk(ln=True, **kw)])
        kern = (L.lib().mdx_last_kernel() or b"").decode()
        print(f"{name:8s} plain {plain:8.1f} us   ln+plain {t_ln + plain:8.1f} us   fused {fused:8.1f} us  ({kern})")


if __name__ == "__main__":
    main()
