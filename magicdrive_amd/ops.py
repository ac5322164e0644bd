"""Op IR of the MI355X denoiser: one Python record per kernel launch, holding torch tensor
*views* (device memory owned by the engine).  `lower()` turns a record into the C descriptor of
include/mdx.h; `magicdrive_amd._lib.Program` packs the descriptors into the array libmdx executes.

The records carry no arithmetic: the only implementation of each op is the HIP kernel behind its
C entry point.  (tests/plan_interp.py holds an independent torch restatement of the records'
semantics, used only to check the op graph on CPU.)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch

from . import _lib as L

BF16 = torch.bfloat16
F16 = torch.float16
H16 = (BF16, F16)          # the two 16-bit storage types: every 16-bit operand of ONE op must use the same one (dtype_code)
F32 = torch.float32


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(cond: bool, msg: str):
    if not cond:
        raise ValueError(msg)


def _rows2d(t: torch.Tensor):
    """[M, C] view with unit inner stride -> (M, C, ld)."""
    _chk(t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), f"need [M,C] view with unit inner stride, got {tuple(t.shape)} {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


def _nhwc(t: torch.Tensor):
    """[B,H,W,C] channels-last view, pixels dense -> (B,H,W,C,ld)."""
    _chk(t.dim() == 4 and t.stride(3) == 1, f"need [B,H,W,C] view, got {tuple(t.shape)} {t.stride()}")
    B, H, W, Cc = t.shape
    ld = t.stride(2)
    _chk((W == 1 or True) and (H == 1 or t.stride(1) == W * ld) and (B == 1 or t.stride(0) == H * W * ld),
         f"pixels must be dense: shape {tuple(t.shape)} stride {t.stride()}")
    return B, H, W, Cc, ld


@dataclass
class Gemm:
    """C = epi(A @ W^T + bias + temb[row]) + R.  A [M,K] or [Bt,M,K]; W [N,K] or [Bt,N,K]; C [M,No] or [Bt,M,No]."""
    A: torch.Tensor
    W: torch.Tensor
    C: torch.Tensor
    bias: Optional[torch.Tensor] = None      # fp32 [N]
    R: Optional[torch.Tensor] = None         # same shape/dtype as C
    temb: Optional[torch.Tensor] = None      # fp32 table
    sel: Optional[torch.Tensor] = None       # int32 [1]
    epilogue: int = L.EPI_NONE
    temb_sel_stride: int = 0
    temb_b_stride: int = 0
    rows_per_b: int = 1
    splitk: int = 0
    ws: Optional[torch.Tensor] = None
    name: str = ""
    # fused q/k/v projection: raw columns >= vt_from go, transposed, to Vt [Bv, N - vt_from, >= vt_T] (token m -> view m // vt_T,
    # position m % vt_T); C then has vt_from columns (include/mdx.h: MdxGemmDesc.Vt)
    Vt: Optional[torch.Tensor] = None
    vt_from: int = 0
    vt_T: int = 0
    # LayerNorm of the A rows fused in (include/mdx.h: MdxGemmDesc.ln_eps): A = raw rows, W / bias carry gamma / beta (engine.PackedNet.ln_lin),
    # ln_csum fp32 [N] = row sums of W, ln_scratch [M, K] = where routes that cannot fuse put the normalised rows
    ln_eps: float = 0.0
    ln_csum: Optional[torch.Tensor] = None
    ln_scratch: Optional[torch.Tensor] = None
    # row statistics (include/mdx.h: MdxGemmDesc.rowstat_out / ln_stats): fp32 [parts, M, 2] = (sum, sum of squares) of the stored C rows per column part,
    # written by the producer of a LayerNorm's input and read by the GEMM that carries that LayerNorm (ln_eps > 0)
    rowstat: Optional[torch.Tensor] = None
    ln_stats: Optional[torch.Tensor] = None
    # W once more in MFMA-fragment order (packing.pack_wq; include/mdx.h: MdxGemmDesc.Wq): the W-direct persistent kernel reads this copy
    Wq: Optional[torch.Tensor] = None
    opcode = L.OP_GEMM

    def lower(self):
        A, W, Cm = self.A, self.W, self.C
        batch = 1
        sA = sW = sC = sR = 0
        if Cm.dim() == 3:
            batch = Cm.shape[0]
            sC = Cm.stride(0)
            if A.dim() == 3:
                sA = A.stride(0); A = A[0]
            if W.dim() == 3:
                sW = W.stride(0); W = W[0]
            C2 = Cm[0]
        else:
            C2 = Cm
        M, K, lda = _rows2d(A)
        N, K2, ldw = _rows2d(W)
        Mc, No, ldc = _rows2d(C2)
        _chk(K == K2 and M == Mc, f"gemm {self.name}: shape mismatch A{tuple(A.shape)} W{tuple(W.shape)} C{tuple(C2.shape)}")
        if self.Vt is not None:
            _chk(self.epilogue == L.EPI_NONE and batch == 1 and No == self.vt_from and 0 < self.vt_from < N and self.vt_T > 0,
                 f"gemm {self.name}: fused V^T output needs a plain epilogue and C with vt_from columns")
            _chk(self.Vt.dim() == 3 and self.Vt.dtype == A.dtype and self.Vt.stride(2) == 1 and self.Vt.shape[1] == N - self.vt_from
                 and self.Vt.shape[0] * self.vt_T == M and self.Vt.shape[2] >= self.vt_T, f"gemm {self.name}: Vt shape {tuple(self.Vt.shape)}")
        else:
            _chk(No == (N // 2 if self.epilogue == L.EPI_GEGLU else N), f"gemm {self.name}: C has {No} cols for N={N}")
        _chk(A.dtype in H16 and W.dtype == A.dtype and Cm.dtype in (A.dtype, F32), f"gemm {self.name}: dtypes")
        d = L.MdxGemmDesc()
        d.A, d.W, d.C = _p(A), _p(W), _p(C2)
        ldr = 0
        if self.R is not None:
            R2 = self.R
            if R2.dim() == 3:
                sR = R2.stride(0); R2 = R2[0]
            _chk(R2.dtype == Cm.dtype and tuple(R2.shape) == tuple(C2.shape), f"gemm {self.name}: residual shape/dtype")
            ldr = R2.stride(0)
            d.R = _p(R2)
        if self.bias is not None:
            _chk(self.bias.dtype == F32 and self.bias.numel() == N, f"gemm {self.name}: bias")
            d.bias = _p(self.bias)
        if self.temb is not None:
            _chk(self.temb.dtype == F32, "temb must be fp32")
            d.temb = _p(self.temb)
        if self.sel is not None:
            _chk(self.sel.dtype == torch.int32, "sel must be int32")
            d.sel_ptr = _p(self.sel)
        if self.ws is not None:
            d.ws = _p(self.ws); d.ws_bytes = self.ws.numel() * self.ws.element_size()
        d.M, d.N, d.K = M, N, K
        d.lda, d.ldw, d.ldc, d.ldr = lda, ldw, ldc, ldr
        d.batch, d.sA, d.sW, d.sC, d.sR = batch, sA, sW, sC, sR
        d.temb_sel_stride, d.temb_b_stride, d.rows_per_b = self.temb_sel_stride, self.temb_b_stride, self.rows_per_b
        d.epilogue, d.splitk, d.c_is_f32 = self.epilogue, self.splitk, int(Cm.dtype == F32)
        if self.Vt is not None:
            d.Vt = _p(self.Vt)
            d.vt_from, d.vt_T, d.vt_ld, d.vt_stride = self.vt_from, self.vt_T, self.Vt.stride(1), self.Vt.stride(0)
        if self.ln_eps > 0:
            _chk(batch == 1 and self.ln_csum is not None and self.ln_csum.dtype == F32 and self.ln_csum.numel() == N, f"gemm {self.name}: ln_csum")
            _chk(self.ln_scratch is None or (self.ln_scratch.dtype == A.dtype and self.ln_scratch.is_contiguous() and self.ln_scratch.numel() >= M * lda),
                 f"gemm {self.name}: ln_scratch must be a contiguous [M, lda] buffer of the operand type")
            d.ln_eps, d.ln_csum, d.ln_scratch = float(self.ln_eps), _p(self.ln_csum), _p(self.ln_scratch)
            if self.ln_stats is not None:
                st = self.ln_stats
                _chk(st.dtype == F32 and st.dim() == 3 and st.is_contiguous() and st.shape[1] == M and st.shape[2] == 2, f"gemm {self.name}: ln_stats must be fp32 [parts, M, 2]")
                d.ln_stats, d.ln_stats_parts = _p(st), st.shape[0]
        else:
            _chk(self.ln_stats is None, f"gemm {self.name}: ln_stats without ln_eps")
        if self.rowstat is not None:
            rs = self.rowstat
            _chk(rs.dtype == F32 and rs.dim() == 3 and rs.is_contiguous() and rs.shape[1] == M and rs.shape[2] == 2 and rs.shape[0] >= 1,
                 f"gemm {self.name}: rowstat must be fp32 [parts, M, 2]")
            _chk(batch == 1 and self.epilogue == L.EPI_NONE and self.Vt is None and Cm.dtype in H16, f"gemm {self.name}: rowstat needs a plain 2-D GEMM with 16-bit C")
            d.rowstat_out, d.rowstat_parts = _p(rs), rs.shape[0]
        if self.Wq is not None:
            _chk(batch == 1 and self.Wq.dtype == W.dtype and self.Wq.is_contiguous() and self.Wq.numel() == (N + 255) // 256 * 256 * K and K % 32 == 0,
                 f"gemm {self.name}: Wq must be packing.pack_wq(W)")
            d.Wq = _p(self.Wq)
        return self.opcode, d


@dataclass
class Conv:
    """Channels-last conv: X [B,Hi,Wi,Cin], Wt [Cout,kh,kw,Cin] packed bf16, Y [B,Ho,Wo,Cout]."""
    X: torch.Tensor
    Wt: torch.Tensor
    Y: torch.Tensor
    bias: Optional[torch.Tensor] = None
    R: Optional[torch.Tensor] = None
    temb: Optional[torch.Tensor] = None
    sel: Optional[torch.Tensor] = None
    stride: tuple = (1, 1)
    pad: tuple = (1, 1)
    pad_end: Optional[tuple] = None          # zero padding at the bottom / right when it differs from `pad` (Downsample2D with padding=0:
                                             # F.pad(x, (0, 1, 0, 1)), resnet.py:215-217) — the kernels zero-fill every tap outside the image,
                                             # so only the output size depends on it
    epilogue: int = L.EPI_NONE
    temb_sel_stride: int = 0
    temb_b_stride: int = 0
    direct: bool = False                     # vector-ALU path (tiny Cin / Cout, fp32 I/O allowed)
    splitk: int = 0
    ws: Optional[torch.Tensor] = None
    name: str = ""

    @property
    def opcode(self):
        return L.OP_CONV_DIRECT if self.direct else L.OP_CONV

    def lower(self):
        B, Hi, Wi, Cin, ldx = _nhwc(self.X)
        B2, Ho, Wo, Cout, ldy = _nhwc(self.Y)
        Co2, kh, kw, Ci2 = self.Wt.shape
        _chk(self.Wt.is_contiguous() and self.Wt.dtype in H16, f"conv {self.name}: weights must be packed bf16 / fp16")
        _chk(B == B2 and Cout == Co2 and Cin == Ci2, f"conv {self.name}: shape mismatch")
        sh, sw = self.stride
        ph, pw = self.pad
        phe, pwe = self.pad_end if self.pad_end is not None else self.pad
        _chk(Ho == (Hi + ph + phe - kh) // sh + 1 and Wo == (Wi + pw + pwe - kw) // sw + 1, f"conv {self.name}: output size")
        d = L.MdxConvDirectDesc() if self.direct else L.MdxConvDesc()
        d.X, d.Wt, d.Y = _p(self.X), _p(self.Wt), _p(self.Y)
        if self.R is not None:
            rB, rH, rW, rC, ldr = _nhwc(self.R)
            _chk((rB, rH, rW, rC) == (B, Ho, Wo, Cout) and self.R.dtype == self.Y.dtype, f"conv {self.name}: residual")
            d.R, d.ldr = _p(self.R), ldr
        if self.bias is not None:
            _chk(self.bias.dtype == F32 and self.bias.numel() == Cout, f"conv {self.name}: bias")
            d.bias = _p(self.bias)
        if self.temb is not None:
            d.temb = _p(self.temb)
        if self.sel is not None:
            d.sel_ptr = _p(self.sel)
        d.B, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = B, Hi, Wi, Cin, Ho, Wo, Cout
        d.kh, d.kw, d.sh, d.sw, d.ph, d.pw = kh, kw, sh, sw, ph, pw
        d.ldx, d.ldy = ldx, ldy
        d.temb_sel_stride, d.temb_b_stride = self.temb_sel_stride, self.temb_b_stride
        d.epilogue = self.epilogue
        if self.direct:
            d.x_is_f32 = int(self.X.dtype == F32)
            d.y_is_f32 = int(self.Y.dtype == F32)
        else:
            _chk(self.X.dtype == self.Wt.dtype and self.Y.dtype == self.Wt.dtype, f"conv {self.name}: the MFMA path takes one 16-bit type")
            d.splitk = self.splitk
            if self.ws is not None:
                d.ws = _p(self.ws); d.ws_bytes = self.ws.numel() * self.ws.element_size()
        return self.opcode, d


@dataclass
class Attn:
    """O = sum_s softmax(Q K_s^T * scale) V_s.  Q [B,Tq,C]; K [Bkv,Tk,C]; Vt [Bkv,C,ldv] (V transposed); O [B,Tq,C]."""
    Q: torch.Tensor
    K: torch.Tensor
    Vt: torch.Tensor
    O: torch.Tensor
    heads: int
    Tk: int
    scale: float
    kvmap: Optional[torch.Tensor] = None     # int32 [B*nsrc]
    nsrc: int = 1
    name: str = ""
    joint: bool = False                      # one softmax over the concatenated sources (neighboring_attn_type concat / self) instead of a sum of per-source attentions
    q_prescaled: bool = False                # Q already carries scale * log2(e) (folded into to_q at pack time): probabilities are exp2(Q K^T - max)
    opcode = L.OP_ATTN

    def lower(self):
        Q, K, Vt, O = self.Q, self.K, self.Vt, self.O
        B, Tq, Cc = Q.shape
        _chk(Q.stride(2) == 1 and K.stride(2) == 1 and O.stride(2) == 1 and Vt.stride(2) == 1, f"attn {self.name}: inner strides")
        _chk(Cc % self.heads == 0 and K.shape[2] == Cc and Vt.shape[1] == Cc and O.shape == Q.shape, f"attn {self.name}: shapes")
        _chk(K.shape[1] == self.Tk and Vt.shape[2] >= self.Tk and Vt.shape[0] == K.shape[0], f"attn {self.name}: kv shapes")
        d = L.MdxAttnDesc()
        d.Q, d.K, d.Vt, d.O = _p(Q), _p(K), _p(Vt), _p(O)
        if self.kvmap is not None:
            _chk(self.kvmap.dtype == torch.int32 and self.kvmap.numel() == B * self.nsrc, f"attn {self.name}: kvmap")
            d.kvmap = _p(self.kvmap)
        else:
            _chk(self.nsrc == 1 and K.shape[0] == B, f"attn {self.name}: identity kv map needs Bkv == B")
        d.B, d.H, d.Tq, d.Tk, d.d, d.nsrc = B, self.heads, Tq, self.Tk, Cc // self.heads, self.nsrc
        d.ldq, d.sQ = Q.stride(1), Q.stride(0)
        d.ldk, d.sK = K.stride(1), K.stride(0)
        d.ldv, d.sV = Vt.stride(1), Vt.stride(0)
        d.ldo, d.sO = O.stride(1), O.stride(0)
        d.scale = float(self.scale)
        d.joint = int(self.joint)
        d.q_prescaled = int(self.q_prescaled)
        _chk(self.nsrc >= 1 and (self.nsrc <= 8 if self.joint else self.nsrc <= 2), f"attn {self.name}: nsrc={self.nsrc} (joint={self.joint})")
        return self.opcode, d


@dataclass
class GroupNorm:
    X: torch.Tensor      # [B, HW, C] view
    Y: torch.Tensor
    gamma: torch.Tensor  # fp32 [C]
    beta: torch.Tensor
    groups: int
    eps: float
    silu: bool = False
    ws: Optional[torch.Tensor] = None
    name: str = ""
    opcode = L.OP_GROUPNORM

    def lower(self):
        B, HW, Cc = self.X.shape
        _chk(self.X.stride(2) == 1 and self.Y.stride(2) == 1 and self.Y.shape == self.X.shape, f"gn {self.name}: layout")
        _chk(B == 1 or (self.X.stride(0) == HW * self.X.stride(1) and self.Y.stride(0) == HW * self.Y.stride(1)), f"gn {self.name}: batch stride")
        _chk(self.gamma.dtype == F32 and self.beta.dtype == F32 and self.gamma.numel() == Cc, f"gn {self.name}: affine")
        d = L.MdxGroupNormDesc()
        d.X, d.Y, d.gamma, d.beta = _p(self.X), _p(self.Y), _p(self.gamma), _p(self.beta)
        d.B, d.HW, d.C, d.G, d.ldx, d.ldy = B, HW, Cc, self.groups, self.X.stride(1), self.Y.stride(1)
        d.eps, d.silu = float(self.eps), int(self.silu)
        if self.ws is not None:
            d.ws = _p(self.ws); d.ws_bytes = self.ws.numel() * self.ws.element_size()
        return self.opcode, d


@dataclass
class LayerNorm:
    X: torch.Tensor      # [M, C]
    Y: torch.Tensor
    gamma: torch.Tensor
    beta: torch.Tensor
    eps: float = 1e-5
    name: str = ""
    opcode = L.OP_LAYERNORM

    def lower(self):
        M, Cc, ldx = _rows2d(self.X)
        M2, C2, ldy = _rows2d(self.Y)
        _chk(M == M2 and Cc == C2, f"ln {self.name}: shapes")
        d = L.MdxLayerNormDesc()
        d.X, d.Y, d.gamma, d.beta = _p(self.X), _p(self.Y), _p(self.gamma), _p(self.beta)
        d.M, d.C, d.ldx, d.ldy, d.eps = M, Cc, ldx, ldy, float(self.eps)
        return self.opcode, d


@dataclass
class Softmax:
    """Y[r, :T] = softmax(scale * X[r, :T]) (fp32 scores -> bf16 probabilities); Y's columns T.. are zero-filled."""
    X: torch.Tensor      # fp32 [rows, >=T] (row stride = X.stride(0))
    Y: torch.Tensor      # bf16 [rows, ldy >= T] contiguous rows
    T: int
    scale: float = 1.0
    name: str = ""
    opcode = L.OP_SOFTMAX

    def lower(self):
        _chk(self.X.dtype == F32 and self.Y.dtype in H16 and self.X.dim() == 2 and self.Y.dim() == 2, "softmax: dtypes / rank")
        _chk(self.X.stride(1) == 1 and self.Y.stride(1) == 1 and self.X.shape[0] == self.Y.shape[0], "softmax: layout")
        _chk(0 < self.T <= self.X.shape[1] and self.T <= self.Y.shape[1], "softmax: T")
        d = L.MdxSoftmaxDesc()
        d.X, d.Y, d.rows, d.T, d.ldx, d.ldy, d.scale = _p(self.X), _p(self.Y), self.X.shape[0], self.T, self.X.stride(0), self.Y.stride(0), float(self.scale)
        # the kernel zero-fills Y[:, T:ldy]: Y must own its whole row pitch
        _chk(self.Y.shape[1] == self.Y.stride(0), "softmax: Y rows must be dense (the pad columns are written)")
        return self.opcode, d


@dataclass
class Ew:
    """kind in {ADD (Y += X), COPY, SILU, SCALE} on [M,C] views."""
    kind: int
    X: torch.Tensor
    Y: torch.Tensor
    alpha: float = 1.0
    name: str = ""
    opcode = L.OP_EW

    def lower(self):
        M, Cc, ldx = _rows2d(self.X)
        M2, C2, ldy = _rows2d(self.Y)
        _chk(M == M2 and Cc == C2, f"ew {self.name}: shapes {tuple(self.X.shape)} vs {tuple(self.Y.shape)}")
        d = L.MdxEwDesc()
        d.X, d.Y, d.kind, d.M, d.C, d.ldx, d.ldy = _p(self.X), _p(self.Y), self.kind, M, Cc, ldx, ldy
        d.x_is_f32, d.y_is_f32, d.alpha = int(self.X.dtype == F32), int(self.Y.dtype == F32), float(self.alpha)
        return self.opcode, d


@dataclass
class Upsample:
    """Nearest resize X [B,Hi,Wi,C] -> Y [B,Ho,Wo,C]; ymap/xmap int32 source indices (host-computed, torch's rule)."""
    X: torch.Tensor
    Y: torch.Tensor
    ymap: torch.Tensor
    xmap: torch.Tensor
    name: str = ""
    opcode = L.OP_EW

    def lower(self):
        B, Hi, Wi, Cc, ldx = _nhwc(self.X)
        B2, Ho, Wo, C2, ldy = _nhwc(self.Y)
        _chk(B == B2 and Cc == C2 and self.ymap.numel() == Ho and self.xmap.numel() == Wo, f"upsample {self.name}")
        d = L.MdxEwDesc()
        d.X, d.Y, d.ymap, d.xmap = _p(self.X), _p(self.Y), _p(self.ymap), _p(self.xmap)
        d.kind, d.C, d.ldx, d.ldy = L.EW_UPSAMPLE, Cc, ldx, ldy
        d.B, d.Hi, d.Wi, d.Ho, d.Wo = B, Hi, Wi, Ho, Wo
        d.x_is_f32, d.y_is_f32 = int(self.X.dtype == F32), int(self.Y.dtype == F32)
        return self.opcode, d


@dataclass
class Layout:
    """to_nhwc: X contiguous [B,C,H,W] -> Y [B,H,W,C] view ; else the reverse (X nhwc view, Y contiguous nchw)."""
    X: torch.Tensor
    Y: torch.Tensor
    to_nhwc: bool
    name: str = ""
    opcode = L.OP_EW

    def lower(self):
        d = L.MdxEwDesc()
        if self.to_nhwc:
            _chk(self.X.is_contiguous() and self.X.dim() == 4, "layout: X must be contiguous NCHW")
            B, Cc, H, W = self.X.shape
            B2, H2, W2, C2, ldy = _nhwc(self.Y)
            _chk((B, H, W, Cc) == (B2, H2, W2, C2), "layout: shapes")
            d.kind, d.ldy = L.EW_NCHW_TO_NHWC, ldy
        else:
            _chk(self.Y.is_contiguous() and self.Y.dim() == 4, "layout: Y must be contiguous NCHW")
            B, Cc, H, W = self.Y.shape
            B2, H2, W2, C2, ldx = _nhwc(self.X)
            _chk((B, H, W, Cc) == (B2, H2, W2, C2), "layout: shapes")
            d.kind, d.ldx = L.EW_NHWC_TO_NCHW, ldx
        d.X, d.Y, d.B, d.C, d.Hi, d.Wi = _p(self.X), _p(self.Y), B, Cc, H, W
        d.x_is_f32, d.y_is_f32 = int(self.X.dtype == F32), int(self.Y.dtype == F32)
        return self.opcode, d


@dataclass
class Fourier:
    X: torch.Tensor                      # fp32 [n, P, 3] contiguous
    Y: torch.Tensor                      # bf16 [n, P*(3+6F)] view
    F: int
    mask: Optional[torch.Tensor] = None  # uint8 [n]
    null_feat: Optional[torch.Tensor] = None  # fp32 [P*(3+6F)]
    name: str = ""
    opcode = L.OP_FOURIER

    def lower(self):
        n, Pn, three = self.X.shape
        _chk(three == 3 and self.X.is_contiguous() and self.X.dtype == F32, "fourier: X fp32 [n,P,3]")
        n2, width, ldy = _rows2d(self.Y)
        _chk(n2 == n and width == Pn * (3 + 6 * self.F) and self.Y.dtype in H16, "fourier: Y")
        d = L.MdxFourierDesc()
        d.X, d.Y, d.mask, d.null_feat = _p(self.X), _p(self.Y), _p(self.mask), _p(self.null_feat)
        if self.mask is not None:
            _chk(self.mask.dtype == torch.uint8 and self.mask.numel() == n, "fourier: mask")
        if self.null_feat is not None:
            _chk(self.null_feat.dtype == F32 and self.null_feat.numel() == width, "fourier: null")
        d.n, d.P, d.F, d.ldy = n, Pn, self.F, ldy
        return self.opcode, d


@dataclass
class Gather:
    T: torch.Tensor                      # bf16 [rows, C]
    Y: torch.Tensor                      # bf16 [n, C] view
    idx: torch.Tensor                    # int64 [n]
    mask: Optional[torch.Tensor] = None  # uint8 [n]
    null_row: Optional[torch.Tensor] = None  # bf16 [C]
    name: str = ""
    opcode = L.OP_GATHER

    def lower(self):
        rows, Cc, ldt = _rows2d(self.T)
        n, C2, ldy = _rows2d(self.Y)
        _chk(Cc == C2 and self.idx.dtype == torch.int64 and self.idx.numel() == n, "gather: shapes")
        d = L.MdxGatherDesc()
        d.T, d.Y, d.idx, d.mask, d.null_row = _p(self.T), _p(self.Y), _p(self.idx), _p(self.mask), _p(self.null_row)
        d.n, d.C, d.ldt, d.ldy, d.n_rows = n, Cc, ldt, ldy, rows
        return self.opcode, d


@dataclass
class TimeEmb:
    t: torch.Tensor                      # fp32 [n]
    Y: torch.Tensor                      # fp32 [n, dim]
    flip_sin_to_cos: bool = True
    freq_shift: float = 0.0
    max_period: float = 10000.0
    name: str = ""
    opcode = L.OP_TIMEEMB

    def lower(self):
        n, dim, ldy = _rows2d(self.Y)
        _chk(self.t.dtype == F32 and self.t.numel() == n and self.Y.dtype == F32, "timeemb: dtypes")
        d = L.MdxTimeEmbDesc()
        d.t, d.Y, d.n, d.dim, d.flip_sin_to_cos, d.ldy = _p(self.t), _p(self.Y), n, dim, int(self.flip_sin_to_cos), ldy
        d.freq_shift, d.max_period = float(self.freq_shift), float(self.max_period)
        return self.opcode, d


@dataclass
class DdimStep:
    x: torch.Tensor                      # fp32 [n] latents (any layout), updated in place
    eps: torch.Tensor                    # fp32 [c*n], same layout ([uncond | cond] when cfg)
    coef: torch.Tensor                   # fp32 [steps, 4]
    step: torch.Tensor                   # int32 [1]
    x_in: Optional[torch.Tensor] = None  # fp32 flat [c*n]  or  bf16 [c*pixels, ld] channels-last with ld >= C (padded)
    cfg: bool = False
    guidance: float = 1.0
    xin_c: int = 0                        # channels per pixel of x (for the bf16 padded x_in)
    name: str = ""
    # given views (MdxDdimDesc.gv_*): mask uint8 [views], cond / noise fp32 [n] in x's layout, mode 1 | 2, last step index
    gv_mask: Optional[torch.Tensor] = None
    gv_cond: Optional[torch.Tensor] = None
    gv_noise: Optional[torch.Tensor] = None
    gv_mode: int = 0
    gv_last_step: int = 0
    opcode = L.OP_DDIM

    def lower(self):
        n = self.x.numel()
        _chk(self.x.is_contiguous() and self.eps.is_contiguous() and self.eps.numel() == n * (2 if self.cfg else 1), "ddim: sizes")
        _chk(self.x.dtype == F32 and self.eps.dtype == F32 and self.coef.dtype == F32 and self.step.dtype == torch.int32, "ddim: dtypes")
        d = L.MdxDdimDesc()
        d.x, d.eps, d.coef, d.step_ptr, d.x_in = _p(self.x), _p(self.eps), _p(self.coef), _p(self.step), _p(self.x_in)
        if self.x_in is not None:
            if self.x_in.dtype == F32:
                _chk(self.x_in.is_contiguous() and self.x_in.numel() == self.eps.numel(), "ddim: x_in")
            else:
                _chk(self.x_in.dtype in H16 and self.x_in.dim() == 2 and self.x_in.is_contiguous() and self.xin_c > 0, "ddim: bf16 x_in must be [pixels, ld]")
                _chk(self.x_in.shape[0] * self.xin_c == self.eps.numel() and self.x_in.shape[1] >= self.xin_c, "ddim: bf16 x_in shape")
                d.xin_c, d.xin_ld = self.xin_c, self.x_in.shape[1]
        d.n, d.cfg, d.guidance = n, int(self.cfg), float(self.guidance)
        if self.gv_mode:
            _chk(self.gv_mode in (1, 2) and self.gv_mask is not None and self.gv_mask.dtype == torch.uint8 and self.gv_mask.is_contiguous()
                 and n % self.gv_mask.numel() == 0, "ddim: given-view mask")
            _chk(self.gv_noise is not None and self.gv_noise.dtype == F32 and self.gv_noise.is_contiguous() and self.gv_noise.numel() == n, "ddim: gv_noise")
            _chk(self.gv_mode == 2 or (self.gv_cond is not None and self.gv_cond.dtype == F32 and self.gv_cond.is_contiguous()
                                       and self.gv_cond.numel() == n), "ddim: gv_cond")
            d.gv_mask, d.gv_noise, d.gv_cond = _p(self.gv_mask), _p(self.gv_noise), _p(self.gv_cond)
            d.gv_mode, d.gv_view_elems, d.gv_last_step = self.gv_mode, n // self.gv_mask.numel(), self.gv_last_step
        return self.opcode, d


@dataclass
class UniPCStep:
    """Fused CFG + UniPC (order <= 2) predictor-corrector step; state buffers x_last / m1 / m2 are fp32 [n]."""
    x: torch.Tensor
    eps: torch.Tensor
    coef: torch.Tensor                   # fp32 [steps, 12]
    step: torch.Tensor
    x_last: torch.Tensor
    m1: torch.Tensor
    m2: torch.Tensor
    x_in: Optional[torch.Tensor] = None
    cfg: bool = False
    guidance: float = 1.0
    xin_c: int = 0
    name: str = ""
    # given views (MdxUniPCDesc.gv_*): as in DdimStep
    gv_mask: Optional[torch.Tensor] = None
    gv_cond: Optional[torch.Tensor] = None
    gv_noise: Optional[torch.Tensor] = None
    gv_mode: int = 0
    gv_last_step: int = 0
    opcode = L.OP_UNIPC

    def lower(self):
        n = self.x.numel()
        _chk(self.eps.numel() == n * (2 if self.cfg else 1) and all(t.numel() == n and t.dtype == F32 and t.is_contiguous() for t in (self.x, self.x_last, self.m1, self.m2)), "unipc: sizes")
        _chk(self.coef.dtype == F32 and self.coef.shape[1] == 12 and self.step.dtype == torch.int32, "unipc: coef/step")
        d = L.MdxUniPCDesc()
        d.x, d.eps, d.coef, d.step_ptr, d.x_in = _p(self.x), _p(self.eps), _p(self.coef), _p(self.step), _p(self.x_in)
        d.x_last, d.m1, d.m2 = _p(self.x_last), _p(self.m1), _p(self.m2)
        if self.x_in is not None:
            if self.x_in.dtype == F32:
                _chk(self.x_in.numel() == self.eps.numel(), "unipc: x_in")
            else:
                _chk(self.x_in.dtype in H16 and self.x_in.dim() == 2 and self.xin_c > 0 and self.x_in.shape[0] * self.xin_c == self.eps.numel(), "unipc: bf16 x_in")
                d.xin_c, d.xin_ld = self.xin_c, self.x_in.shape[1]
        d.n, d.cfg, d.guidance = n, int(self.cfg), float(self.guidance)
        if self.gv_mode:
            _chk(self.gv_mode in (1, 2) and self.gv_mask is not None and self.gv_mask.dtype == torch.uint8 and self.gv_mask.is_contiguous()
                 and n % self.gv_mask.numel() == 0, "unipc: given-view mask")
            _chk(self.gv_noise is not None and self.gv_noise.dtype == F32 and self.gv_noise.is_contiguous() and self.gv_noise.numel() == n, "unipc: gv_noise")
            _chk(self.gv_mode == 2 or (self.gv_cond is not None and self.gv_cond.dtype == F32 and self.gv_cond.is_contiguous()
                                       and self.gv_cond.numel() == n), "unipc: gv_cond")
            d.gv_mask, d.gv_noise, d.gv_cond = _p(self.gv_mask), _p(self.gv_noise), _p(self.gv_cond)
            d.gv_mode, d.gv_view_elems, d.gv_last_step = self.gv_mode, n // self.gv_mask.numel(), self.gv_last_step
        return self.opcode, d


def dtype_code(op) -> int:
    """MdxOp.dtype of an IR op: fp16 when its 16-bit tensors are torch.float16, else bf16 (ops without 16-bit tensors run either build)."""
    for v in vars(op).values():
        if isinstance(v, torch.Tensor):
            if v.dtype == F16:
                return L.DTYPE_F16
            if v.dtype == BF16:
                return L.DTYPE_BF16
    return L.DTYPE_BF16


def lower_with_dtype(op):
    code, desc = op.lower()
    return code, desc, dtype_code(op)


def build_program(ops) -> L.Program:
    return L.Program([lower_with_dtype(op) for op in ops])


def run_ops(ops, stream: Optional[int] = None) -> None:
    """Eagerly launch a list of IR ops on `stream` (default: torch's current stream)."""
    if stream is None:
        dev = None
        for op in ops:                       # launch on the device that owns the operands, not on whatever device is current
            for v in vars(op).values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    dev = v.device
                    break
            if dev is not None:
                break
        if dev is None:
            raise RuntimeError("magicdrive_amd: ops run on the MI355X only (no CPU fallback): none of the operands is a CUDA / HIP tensor")
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            for op in ops:
                L.call_op(*op.lower(), stream, dtype_code(op))
        return
    for op in ops:
        L.call_op(*op.lower(), stream, dtype_code(op))
