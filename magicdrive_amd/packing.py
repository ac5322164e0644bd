"""Host-side weight / index packing for the libmdx kernels (runs once at load time).

Reference state-dict tensors (diffusers / MagicDrive layout) -> the layouts the kernels read:
  * conv filters  [Cout, Cin, kh, kw]  -> [Cout, kh, kw, Cin] bf16  (K-contiguous for the implicit GEMM)
  * GEGLU proj    [2F, K] (value rows | gate rows, attention.py:259-280) -> 32-row interleave so one
    MFMA wave tile holds a feature's value and gate in the same register slot
  * nearest-neighbour source indices for Upsample2D (resnet.py:154-163), torch's float32 rule
"""
from __future__ import annotations

import numpy as np
import torch


def pack_conv_weight(w: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    assert w.dim() == 4
    return w.detach().permute(0, 2, 3, 1).contiguous().to(dtype)


def pack_linear_weight(w: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    assert w.dim() == 2
    return w.detach().contiguous().to(dtype)


def pack_geglu(w: torch.Tensor, b: torch.Tensor, dtype=torch.bfloat16):
    """[2F, K] with rows [value(F) | gate(F)] -> rows [v0..31, g0..31, v32..63, g32..63, ...]."""
    two_f, k = w.shape
    f = two_f // 2
    assert two_f == 2 * f and f % 32 == 0, f"GEGLU inner dim {f} must be a multiple of 32"
    wv = w[:f].reshape(f // 32, 32, k)
    wg = w[f:].reshape(f // 32, 32, k)
    wp = torch.stack([wv, wg], dim=1).reshape(two_f, k)
    bv = b[:f].reshape(f // 32, 32)
    bg = b[f:].reshape(f // 32, 32)
    bp = torch.stack([bv, bg], dim=1).reshape(two_f)
    return wp.contiguous().to(dtype), bp.contiguous().to(torch.float32)


def pack_wq(w: torch.Tensor) -> torch.Tensor:
    """Row-major linear weight [N, K] (16-bit, K % 32 == 0) -> MFMA-fragment order for the W-direct GEMM (include/mdx.h: MdxGemmDesc.Wq,
    csrc/gemm_xd.hip):  Wq[n // 16][k // 32][((k % 32) // 8) * 16 + n % 16][k % 8], N padded with zero rows to a multiple of 256.
    A pure permutation: one contiguous KiB per 16-column x 32-deep block."""
    n, k = w.shape
    assert k % 32 == 0, f"pack_wq: K={k} must be a multiple of 32"
    npad = round_up(n, 256)
    wp = w if npad == n else torch.cat([w, torch.zeros(npad - n, k, dtype=w.dtype, device=w.device)], 0)
    return wp.reshape(npad // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(npad, k)


def nearest_index(n_in: int, n_out: int) -> torch.Tensor:
    """Source index of F.interpolate(mode='nearest', size=n_out): min(floor(dst * (in/out)), in-1) in fp32."""
    scale = np.float32(n_in) / np.float32(n_out)
    dst = np.arange(n_out, dtype=np.float32)
    src = np.minimum(np.floor(dst * scale).astype(np.int64), n_in - 1)
    return torch.from_numpy(src.astype(np.int32))


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m
