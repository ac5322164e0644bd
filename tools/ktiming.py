#!/usr/bin/env python
"""Debug: per-block s_memtime stamps of the register-staged GEMM (MDX_GEMM_TIMING=1)."""
import os, sys
os.environ["MDX_GEMM_TIMING"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O
BF = torch.bfloat16
dev = torch.device("cuda")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
M, N, K = 33600, 320, int(sys.argv[1]) if len(sys.argv) > 1 else 320
A = r(M, K); W = r(N, K); C = torch.empty(M, N, dtype=BF, device=dev); R = r(M, N)
ws = torch.zeros(64 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
op = O.Gemm(A, W, C, bias=torch.randn(N, device=dev), R=R, ws=ws)
code, desc = op.lower()
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.call_op(code, desc, st)
torch.cuda.synchronize()
nb = ((M + 127) // 128) * ((N + 127) // 128)
t = ws.view(torch.int64)[: nb * 5].view(nb, 5).cpu().double()
t0 = t[:, 0].min()
start = (t[:, 0] - t0); pro = t[:, 1] - t[:, 0]; main = t[:, 2] - t[:, 1]; epi = t[:, 3] - t[:, 2]; end = t[:, 3] - t0
q = lambda x: [round(float(v)) for v in torch.quantile(x, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.double))]
print(f"blocks {nb}  (s_memtime ticks; 100 MHz => 10 ns per tick)")
print("start offs ", q(start)); print("prologue   ", q(pro)); print("main loop  ", q(main)); print("epilogue   ", q(epi)); print("end offs   ", q(end))
srt = torch.sort(start).values
print("start times of blocks #0,#255,#511,#512,#600,#788:", [round(float(srt[i])) for i in (0, 255, 511, 512, 600, nb - 1)])
