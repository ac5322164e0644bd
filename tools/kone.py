#!/usr/bin/env python
"""Run ONE representative kernel a few times (for rocprofv3 --pmc passes).  Usage: kone.py conv|gemm|attn|gn"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O, packing as PK
BF = torch.bfloat16
dev = torch.device("cuda")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
kind = sys.argv[1] if len(sys.argv) > 1 else "conv"
B = int(os.environ.get('KONE_VIEWS', '24'))
ws = torch.empty(64 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
if kind == "conv":
    x = r(B, 28, 50, 640); wt = r(640, 3, 3, 640); y = torch.empty(B, 28, 50, 640, dtype=BF, device=dev)
    op = O.Conv(x, wt, y, bias=torch.randn(640, device=dev), R=r(B, 28, 50, 640), ws=ws)
elif kind == "conv320":
    x = r(B, 28, 50, 320); wt = r(320, 3, 3, 320); y = torch.empty(B, 28, 50, 320, dtype=BF, device=dev)
    op = O.Conv(x, wt, y, bias=torch.randn(320, device=dev), R=r(B, 28, 50, 320), ws=ws)
elif kind == "conv1280":
    x = r(B, 14, 25, 1280); wt = r(1280, 3, 3, 1280); y = torch.empty(B, 14, 25, 1280, dtype=BF, device=dev)
    op = O.Conv(x, wt, y, bias=torch.randn(1280, device=dev), R=r(B, 14, 25, 1280), ws=ws)
elif kind == "gemm":
    M = B * 1400; A = r(M, 320); W = r(2560, 320); C = torch.empty(M, 1280, dtype=BF, device=dev)
    op = O.Gemm(A, W, C, bias=torch.randn(2560, device=dev), epilogue=1, ws=ws)
elif kind == "gemmcc":
    M = B * 1400; A = r(M, 320); W = r(320, 320); C = torch.empty(M, 320, dtype=BF, device=dev)
    op = O.Gemm(A, W, C, bias=torch.randn(320, device=dev), R=r(M, 320), ws=ws)
elif kind == "attn":
    T, C = 1400, 320
    qk = r(B, T, 2 * C); vt = r(B, C, T); o = torch.empty(B, T, C, dtype=BF, device=dev)
    op = O.Attn(qk[:, :, :C], qk[:, :, C:], vt, o, heads=8, Tk=T, scale=40 ** -0.5)
else:
    x = r(B, 1400, 320); y = torch.empty_like(x)
    op = O.GroupNorm(x, y, torch.ones(320, device=dev), torch.zeros(320, device=dev), 32, 1e-5, True)
code, desc = op.lower()
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    L.call_op(code, desc, st)
torch.cuda.synchronize()
