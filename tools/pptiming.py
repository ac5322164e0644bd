#!/usr/bin/env python
"""Debug: where a ping-pong GEMM tile spends its slab phases (MDX_GEMM_TIMING=1 MDX_GEMM_PP=2).  s_memtime ticks."""
import os, sys
os.environ["MDX_GEMM_TIMING"] = "1"; os.environ["MDX_GEMM_PP"] = "2"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O
BF = torch.bfloat16
dev = torch.device("cuda")
M, N, K = 13056, 1280, 4096
A = (torch.randn(M, K, device=dev) * 0.5).to(BF); W = (torch.randn(N, K, device=dev) * 0.5).to(BF)
C = torch.empty(M, N, dtype=BF, device=dev)
ws = torch.zeros(16 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for mode in ("real", "lda=ldw=0"):
    code, d = O.Gemm(A, W, C, bias=torch.zeros(N, device=dev), ws=ws).lower()
    if mode != "real": d.lda = 0; d.ldw = 0
    for _ in range(2): L.call_op(code, d, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.call_op(code, d, st); e1.record(); torch.cuda.synchronize()
    nb = ((M + 255) // 256) * (N // 256)
    t = ws.view(torch.int64)[: nb * 8].view(nb, 2, 4).cpu().double()
    nt = K // 64
    print(f"{mode}: {e0.elapsed_time(e1)*1e3:.1f} us; per slab, mean over {nb} blocks (ticks): ")
    for g in (0, 1):
        m = t[:, g].mean(0) / nt
        print(f"  group {g}: compute {m[0]:.1f}  barrier-after-compute {m[1]:.1f}  load-phase {m[2]:.1f}  barrier-after-load {m[3]:.1f}  sum {m.sum():.1f}")
