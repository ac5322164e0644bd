#!/usr/bin/env python
"""Pipeline ceiling of the GEMM main loops: the same launch with real operands, with every A row aliased to row 0
(lda = 0: A is L1/L2 resident) and with W aliased too — separates what the memory system costs from what the
LDS / barrier / MFMA structure costs.  Run under MDX_GEMM_PP=0|2."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O
BF = torch.bfloat16
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(65536, 256, 4096), (13056, 1280, 4096), (65280, 1280, 1280)]:
    A = (torch.randn(M, K, device=dev) * 0.5).to(BF); W = (torch.randn(N, K, device=dev) * 0.5).to(BF)
    C = torch.empty(M, N, dtype=BF, device=dev)
    for mode in ("real", "lda=0", "lda=ldw=0"):
        code, d = O.Gemm(A, W, C, bias=torch.zeros(N, device=dev)).lower()
        if mode != "real": d.lda = 0
        if mode == "lda=ldw=0": d.ldw = 0
        for _ in range(2): L.call_op(code, d, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): L.call_op(code, d, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        print(f"M={M} N={N} K={K} {mode:10s} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TF/s")
