"""Algorithmic FLOP / byte accounting of an op program (2 FLOPs per MAC), per kernel family.

This is the de-duplicated work the HIP path executes (SURVEY.md §8d "F_step"): cross-view projections once
per view, context K/V and the map encoder in the prologue.  tools/count_flops.py cross-checks the totals
against the reference's as-executed FLOPs counted on the oracle.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, Iterable

from . import ops as O


def kernel_family(op) -> str:
    if isinstance(op, O.Conv):
        return "conv_direct" if op.direct else "gemm_conv_kernel<conv>"
    if isinstance(op, O.Gemm):
        return "gemm_conv_kernel<gemm>"
    if isinstance(op, O.Attn):
        return "attn_kernel"
    if isinstance(op, O.GroupNorm):
        return "groupnorm_kernel"
    if isinstance(op, O.LayerNorm):
        return "layernorm_kernel"
    return "elementwise"


def op_flops(op) -> float:
    if isinstance(op, O.Gemm):
        C = op.C
        batch = C.shape[0] if C.dim() == 3 else 1
        M = C.shape[-2]
        N = op.W.shape[-2]
        K = op.W.shape[-1]
        return 2.0 * batch * M * N * K
    if isinstance(op, O.Conv):
        B, Ho, Wo, Cout = op.Y.shape
        _, kh, kw, Cin = op.Wt.shape
        return 2.0 * B * Ho * Wo * Cout * kh * kw * Cin
    if isinstance(op, O.Attn):
        B, Tq, C = op.Q.shape
        return 4.0 * B * Tq * op.Tk * C * op.nsrc          # QK^T + PV
    return 0.0


def op_bytes(op) -> float:
    """Algorithmic HBM bytes (each operand once)."""
    def nb(t):
        return 0 if t is None else t.numel() * t.element_size()
    if isinstance(op, O.Gemm):
        return nb(op.A) + nb(op.W) + nb(op.C) + nb(op.R)
    if isinstance(op, O.Conv):
        return nb(op.X) + nb(op.Wt) + nb(op.Y) + nb(op.R)
    if isinstance(op, O.Attn):
        return nb(op.Q) + op.nsrc * (nb(op.K) + op.K.shape[0] * op.K.shape[2] * op.Tk * 2) // max(op.nsrc, 1) + nb(op.O)
    if isinstance(op, (O.GroupNorm, O.LayerNorm)):
        return nb(op.X) + nb(op.Y)
    if isinstance(op, (O.Ew, O.Upsample, O.Layout)):
        return nb(op.X) + nb(op.Y) * (2 if getattr(op, "kind", 0) == 1 else 1)
    return 0.0


def program_flops(ops: Iterable) -> Dict[str, float]:
    out: Dict[str, float] = defaultdict(float)
    for op in ops:
        out[kernel_family(op)] += op_flops(op)
    out["total"] = sum(v for k, v in out.items() if k != "total")
    return dict(out)
