"""Shared by tests/test_routes_gpu.py and tests/route_worker.py: GEGLU GEMM case vs a torch reference (returns the kernel it was routed to)."""
import torch
import torch.nn.functional as F

from magicdrive_amd import _lib as L, ops as O, packing as PK

BF = torch.bfloat16


def geglu_check(M, F_, K):
    from test_routes_gpu import rnd, close, ws_buf, run_one
    dev = torch.device("cuda")
    A = rnd(M, K, seed=1); W = rnd(2 * F_, K, scale=K ** -0.5, seed=2, dtype=torch.float32); b = rnd(2 * F_, seed=3, dtype=torch.float32)
    Wp, bp = PK.pack_geglu(W.cpu(), b.cpu())
    C = torch.zeros(M, F_, dtype=BF, device=dev)
    k = run_one(O.Gemm(A, Wp.to(dev), C, bias=bp.to(dev), epilogue=L.EPI_GEGLU, ws=ws_buf()))
    h, g = (A.float() @ W.to(BF).float().T + b).chunk(2, dim=-1)
    close(C, h * F.gelu(g), name=f"geglu {M}x{F_}x{K} ({k})")
    return k
