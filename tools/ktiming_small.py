#!/usr/bin/env python
"""Per-block s_memtime stamps of the generic tile at the 1-scene shapes (MDX_GEMM_TIMING=1): where a 10 us launch spends its time."""
import os, sys
os.environ["MDX_GEMM_TIMING"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O
BF = torch.bfloat16
dev = torch.device("cuda")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
st = torch.cuda.current_stream().cuda_stream
q = lambda x: [round(float(v)) for v in torch.quantile(x, torch.tensor([0.0, 0.5, 1.0], dtype=torch.double))]
for (M, N, K, conv) in ((2100, 640, 640, None), (546, 1280, 1280, None), (8400, 320, 1280, None), (8400, 320, 2880, (6, 28, 50, 320))):
    ws = torch.zeros(64 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    if conv:
        B, H, W_, Ci = conv
        op = O.Conv(r(B, H, W_, Ci), r(N, 3, 3, Ci), torch.empty(B, H, W_, N, dtype=BF, device=dev), bias=torch.randn(N, device=dev), R=r(B, H, W_, N), splitk=1, ws=ws)
    else:
        op = O.Gemm(r(M, K), r(N, K), torch.empty(M, N, dtype=BF, device=dev), bias=torch.randn(N, device=dev), R=r(M, N), splitk=1, ws=ws)
    code, desc = op.lower()
    for _ in range(3): L.call_op(code, desc, st)
    torch.cuda.synchronize()
    k = (L.lib().mdx_last_kernel() or b"").decode()
    if not k.startswith("gemm_conv_kernel<"):
        print(f"{k} M={M} N={N} K={K}: not the generic tile"); continue
    bm = int(k.split("<")[1].split(",")[0]); bn = int(k.split(",")[1])
    nb = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    t = ws.view(torch.int64)[: nb * 5].view(nb, 5).cpu().double()
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    print(f"{k} M={M} N={N} K={K}: {len(t)} blocks (ticks of 10 ns): start {q(t[:,0]-t0)} prologue {q(t[:,1]-t[:,0])} main {q(t[:,2]-t[:,1])} epilogue {q(t[:,3]-t[:,2])} end {q(t[:,3]-t0)}")
