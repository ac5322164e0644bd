"""Stand-alone case behind test_routes_gpu.py::test_xl320_edge_tile_reads_no_residual_past_its_buffer (run in a child process: the failure
mode is a GPU memory fault, which aborts the process).

A 320-wide XL tile stores two 128-row halves; an edge tile with fewer than 129 rows has a second half that starts past M.  Its residual
prefetch must clamp to row M - 1.  The residual R is placed so that it CLOSES a fresh device allocation (the bytes after it are not mapped):
round 5 found the denoiser faulting at 24 scenes (cn.d2.r0.conv2: M = 144 x 91 = 51 tiles + 48 rows, R last in the plan's pool)."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from magicdrive_amd import _lib as L, ops as O, packing as PK  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
rnd = lambda *s, scale=1.0, dtype=BF: (torch.randn(*s, generator=g) * scale).to(dtype).to(dev)


def tail_of_fresh_segment(numel):
    """bf16 tensor of `numel` elements whose last byte is the last byte of a device segment the caching allocator has just mapped"""
    torch.cuda.empty_cache()
    seg = 64 << 20                                           # >= 10 MiB: the allocator maps exactly the (2 MiB-rounded) request
    buf = torch.empty(seg, dtype=torch.uint8, device=dev)
    return buf, buf[seg - 2 * numel:].view(BF)


def main():
    B, H, W, C = 144, 7, 13, 1280                             # the shape that faulted; the 48-row edge tile is tile 52 of 52
    x = rnd(B, H, W, C)
    w = rnd(C, C, 3, 3, scale=(9 * C) ** -0.5, dtype=torch.float32); b = rnd(C, dtype=torch.float32)
    wp = PK.pack_conv_weight(w.cpu(), BF).to(dev)
    y = torch.full((B, H, W, C), float("nan"), dtype=BF, device=dev)
    ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)
    keep, Rflat = tail_of_fresh_segment(B * H * W * C)
    R = Rflat.view(B, H, W, C); R.copy_(rnd(B, H, W, C))
    with L.options(GEMM_XL=2, XL_BN=320):
        O.run_ops([O.Conv(x, wp, y, bias=b, R=R, stride=(1, 1), pad=(1, 1), ws=ws)])
        k = (L.lib().mdx_last_kernel() or b"").decode()
    torch.cuda.synchronize()
    assert k == "gemm_xl_kernel<256x320,conv>", k
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(BF).float(), b, padding=1).permute(0, 2, 3, 1) + R.float()
    err = (y.float() - ref).abs().max().item()
    assert err < 0.06, err
    # the same edge geometry through the GEMM loader (M = 256 + 48)
    M, N, K = 304, 640, 128
    A = rnd(M, K); Wg = rnd(N, K, scale=K ** -0.5)
    keep2, R2 = tail_of_fresh_segment(M * N); R2 = R2.view(M, N); R2.copy_(rnd(M, N))
    Cg = torch.full((M, N), float("nan"), dtype=BF, device=dev)
    with L.options(GEMM_XL=2, XL_BN=320):
        O.run_ops([O.Gemm(A, Wg, Cg, R=R2, ws=ws)])
        k = (L.lib().mdx_last_kernel() or b"").decode()
    torch.cuda.synchronize()
    assert k == "gemm_xl_kernel<256x320,gemm>", k
    err2 = (Cg.float() - (A.float() @ Wg.float().T + R2.float())).abs().max().item()
    assert err2 < 0.06, err2
    print(f"ok conv_err={err:.4f} gemm_err={err2:.4f}")


if __name__ == "__main__":
    main()
