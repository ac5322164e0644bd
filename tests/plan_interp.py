"""TEST INFRASTRUCTURE: an independent torch (CPU, fp32 math) interpreter of the op IR in
magicdrive_amd/ops.py.  It lets the *op graph* the engine builds (topology, weight packing, buffer
aliasing, hoisting) be checked against the oracle on a machine without a GPU.  It is not a fallback:
nothing in magicdrive_amd imports it, and the product only ever executes the HIP kernels.
Outputs are written through the same (bf16 / fp32) tensor views the kernels would write, so storage
rounding and buffer reuse behave as on the device.
"""
import math

import torch
import torch.nn.functional as F

from magicdrive_amd import _lib as L
from magicdrive_amd import ops as O


def _sel(op):
    return int(op.sel.item()) if getattr(op, "sel", None) is not None else 0


def _temb_rows(op, n_rows_b, width, col0=0):
    """fp32 [n_rows_b, width] rows of the temb table addressed like the kernels do."""
    flat = op.temb.reshape(-1) if op.temb.is_contiguous() else None
    base = op.temb
    # op.temb is a column-offset view of the table; rebuild addressing from storage
    st = torch.as_strided(base, (n_rows_b, width), (op.temb_b_stride, 1), base.storage_offset() + _sel(op) * op.temb_sel_stride)
    return st.float()


def run_gemm(op: O.Gemm):
    A, W, C = op.A.float(), op.W.float(), op.C
    if op.ln_eps > 0:                         # fused LayerNorm, exactly as include/mdx.h states it: rstd (A W'^T - mean csum) + bias
        if op.ln_stats is not None:           # statistics handed over by the producer of A (MdxGemmDesc.ln_stats): sum / sum of squares per column part
            s = op.ln_stats.float().sum(0)    # [M, 2]
            K = A.shape[-1]
            mean = (s[:, 0] / K)[:, None]
            rstd = ((s[:, 1] / K - mean[:, 0] ** 2).clamp_min(0) + op.ln_eps).rsqrt()[:, None]
        else:
            mean = A.mean(-1, keepdim=True)
            rstd = (A.var(-1, unbiased=False, keepdim=True) + op.ln_eps).rsqrt()
        raw = rstd * (A @ W.transpose(-1, -2) - mean * op.ln_csum.float())
    else:
        raw = A @ W.transpose(-1, -2)
    N = W.shape[-2]
    if op.bias is not None:
        raw = raw + op.bias.float()
    if op.temb is not None:
        M = raw.shape[-2]
        nb = (M + op.rows_per_b - 1) // op.rows_per_b
        t = _temb_rows(op, nb, N)
        raw = raw + t.repeat_interleave(op.rows_per_b, 0)[:M]
    if op.epilogue == L.EPI_GEGLU:
        r = raw.reshape(*raw.shape[:-1], N // 64, 2, 32)
        raw = (r[..., 0, :] * F.gelu(r[..., 1, :])).reshape(*raw.shape[:-1], N // 2)
    elif op.epilogue == L.EPI_SILU:
        raw = F.silu(raw)
    if op.Vt is not None:                     # fused q/k/v: columns >= vt_from go transposed to Vt[view][channel][token]
        Bv, Cv = op.Vt.shape[0], op.Vt.shape[1]
        v = raw[:, op.vt_from:].reshape(Bv, op.vt_T, Cv).transpose(1, 2)
        op.Vt[:, :, :op.vt_T].copy_(v.to(op.Vt.dtype))
        raw = raw[:, :op.vt_from]
    if op.R is not None:
        raw = raw + op.R.float()
    C.copy_(raw.to(C.dtype))
    if op.rowstat is not None:                # MdxGemmDesc.rowstat_out: sums of the STORED values; any split into parts is allowed, unused parts are zero
        cs = C.float()
        op.rowstat.zero_()
        op.rowstat[0, :, 0] = cs.sum(-1)
        op.rowstat[0, :, 1] = (cs * cs).sum(-1)


def run_conv(op: O.Conv):
    x = op.X.float().permute(0, 3, 1, 2)
    w = op.Wt.float().permute(0, 3, 1, 2)
    if op.pad_end is not None and tuple(op.pad_end) != tuple(op.pad):       # asymmetric zero padding (Downsample2D with padding = 0)
        x = F.pad(x, (op.pad[1], op.pad_end[1], op.pad[0], op.pad_end[0]))
        y = F.conv2d(x, w, None if op.bias is None else op.bias.float(), stride=op.stride, padding=0)
    else:
        y = F.conv2d(x, w, None if op.bias is None else op.bias.float(), stride=op.stride, padding=op.pad)
    if op.temb is not None:
        B, Cout = y.shape[0], y.shape[1]
        y = y + _temb_rows(op, B, Cout)[:, :, None, None]
    if op.epilogue == L.EPI_SILU:
        y = F.silu(y)
    y = y.permute(0, 2, 3, 1)
    if op.R is not None:
        y = y + op.R.float()
    op.Y.copy_(y.to(op.Y.dtype))


def run_attn(op: O.Attn):
    Q, K, Vt = op.Q.float(), op.K.float(), op.Vt.float()[:, :, :op.Tk]
    B, Tq, Cc = Q.shape
    H = op.heads
    d = Cc // H
    out = torch.zeros(B, Tq, Cc)
    kvmap = op.kvmap.tolist() if op.kvmap is not None else list(range(B))
    qh = Q.view(B, Tq, H, d).transpose(1, 2)
    khs, vhs = [], []
    for s in range(op.nsrc):
        idx = torch.tensor([kvmap[b * op.nsrc + s] for b in range(B)])
        khs.append(K[idx].view(B, op.Tk, H, d).transpose(1, 2))
        vhs.append(Vt[idx].transpose(1, 2).reshape(B, op.Tk, H, d).transpose(1, 2))
    if getattr(op, "joint", False):            # one softmax over the concatenated sources
        khs, vhs = [torch.cat(khs, 2)], [torch.cat(vhs, 2)]
    # q_prescaled: the to_q weights carry scale * log2(e), scores are base-2 exponents: exp2(s) = exp(s ln 2)
    sc = 0.6931471805599453 if getattr(op, "q_prescaled", False) else op.scale
    for kh, vh in zip(khs, vhs):
        att = torch.softmax(qh @ kh.transpose(-1, -2) * sc, -1)
        out += (att @ vh).transpose(1, 2).reshape(B, Tq, Cc)
    op.O.copy_(out.to(op.O.dtype))


def run_groupnorm(op: O.GroupNorm):
    y = F.group_norm(op.X.float().transpose(1, 2), op.groups, op.gamma.float(), op.beta.float(), op.eps)
    if op.silu:
        y = F.silu(y)
    op.Y.copy_(y.transpose(1, 2).to(op.Y.dtype))


def run_layernorm(op: O.LayerNorm):
    op.Y.copy_(F.layer_norm(op.X.float(), (op.X.shape[-1],), op.gamma.float(), op.beta.float(), op.eps).to(op.Y.dtype))


def run_ew(op: O.Ew):
    x = op.X.float()
    if op.kind == L.EW_ADD:
        r = op.Y.float() + x
    elif op.kind == L.EW_COPY:
        r = x
    elif op.kind == L.EW_SILU:
        r = F.silu(x)
    elif op.kind == L.EW_SCALE:
        r = x * op.alpha
    else:
        raise ValueError(op.kind)
    op.Y.copy_(r.to(op.Y.dtype))


def run_upsample(op: O.Upsample):
    x = op.X
    op.Y.copy_(x[:, op.ymap.long()][:, :, op.xmap.long()])


def run_layout(op: O.Layout):
    if op.to_nhwc:
        op.Y.copy_(op.X.permute(0, 2, 3, 1).to(op.Y.dtype))
    else:
        op.Y.copy_(op.X.permute(0, 3, 1, 2).to(op.Y.dtype))


def run_fourier(op: O.Fourier):
    x = op.X.float()
    parts = [x]
    for k in range(op.F):
        parts += [torch.sin(x * (2.0 ** k)), torch.cos(x * (2.0 ** k))]
    e = torch.cat(parts, -1).reshape(x.shape[0], -1)
    if op.mask is not None:
        m = op.mask.float()[:, None]
        null = op.null_feat.float()[None] if op.null_feat is not None else torch.zeros(1, e.shape[1])
        e = e * m + null * (1 - m)
    op.Y.copy_(e.to(op.Y.dtype))


def run_gather(op: O.Gather):
    idx = op.idx.clone()
    idx[idx < 0] += op.T.shape[0]
    rows = op.T[idx.clamp(0, op.T.shape[0] - 1)]
    if op.mask is not None:
        m = op.mask.bool()[:, None]
        null = op.null_row[None] if op.null_row is not None else torch.zeros_like(rows[:1])
        rows = torch.where(m, rows, null.expand_as(rows))
    op.Y.copy_(rows)


def run_timeemb(op: O.TimeEmb):
    dim = op.Y.shape[1]
    half = dim // 2
    e = -math.log(op.max_period) * torch.arange(half, dtype=torch.float32) / (half - op.freq_shift)
    a = op.t.float()[:, None] * torch.exp(e)[None]
    emb = torch.cat([torch.cos(a), torch.sin(a)], -1) if op.flip_sin_to_cos else torch.cat([torch.sin(a), torch.cos(a)], -1)
    op.Y.copy_(emb)


def run_ddim(op: O.DdimStep):
    n = op.x.numel()
    c = op.coef[int(op.step.item())]
    e = op.eps[:n] + op.guidance * (op.eps[n:] - op.eps[:n]) if op.cfg else op.eps
    given = None
    if op.gv_mode:
        given = op.gv_mask.bool().repeat_interleave(n // op.gv_mask.numel())
        if op.gv_mode == 2:
            e = torch.where(given, op.gv_noise, e)
    x0 = (op.x - c[1] * e) / c[0]
    xn = c[2] * x0 + c[3] * e
    if op.gv_mode == 1 and int(op.step.item()) < op.gv_last_step:
        xn = torch.where(given, c[2] * op.gv_cond + c[3] * op.gv_noise, xn)
    op.x.copy_(xn)
    if op.x_in is not None:
        if op.x_in.dtype == torch.float32:
            op.x_in[:n].copy_(xn)
            if op.cfg:
                op.x_in[n:].copy_(xn)
        else:
            npx = n // op.xin_c
            v = xn.view(npx, op.xin_c).to(op.x_in.dtype)
            op.x_in[:npx, :op.xin_c].copy_(v)
            if op.cfg:
                op.x_in[npx:, :op.xin_c].copy_(v)
    op.step += 1


def run_unipc(op: O.UniPCStep):
    n = op.x.numel()
    c = op.coef[int(op.step.item())]
    e = op.eps[:n] + op.guidance * (op.eps[n:] - op.eps[:n]) if op.cfg else op.eps
    given = None
    if op.gv_mode:
        given = op.gv_mask.bool().repeat_interleave(n // op.gv_mask.numel())
        if op.gv_mode == 2:
            e = torch.where(given, op.gv_noise, e)
    x = op.x.clone(); m1 = op.m1.clone(); m2 = op.m2.clone()
    mt = c[0] * x + c[1] * e
    xc = c[3] * op.x_last + c[4] * m1 + c[5] * m2 + c[6] * mt if c[2] != 0 else x
    xn = c[7] * xc + c[8] * mt + c[9] * m1
    if op.gv_mode == 1 and int(op.step.item()) < op.gv_last_step:
        xn = torch.where(given, c[10] * op.gv_cond + c[11] * op.gv_noise, xn)
    op.x.copy_(xn); op.x_last.copy_(xc); op.m2.copy_(m1); op.m1.copy_(mt)
    if op.x_in is not None:
        if op.x_in.dtype == torch.float32:
            op.x_in[:n].copy_(xn)
            if op.cfg:
                op.x_in[n:].copy_(xn)
        else:
            npx = n // op.xin_c
            v = xn.view(npx, op.xin_c).to(op.x_in.dtype)
            op.x_in[:npx, :op.xin_c].copy_(v)
            if op.cfg:
                op.x_in[npx:, :op.xin_c].copy_(v)
    op.step += 1


def run_softmax(op: O.Softmax):
    y = torch.softmax(op.X[:, :op.T].float() * op.scale, dim=-1)
    op.Y.zero_()
    op.Y[:, :op.T].copy_(y.to(op.Y.dtype))


DISPATCH = {O.Gemm: run_gemm, O.Conv: run_conv, O.Attn: run_attn, O.GroupNorm: run_groupnorm, O.LayerNorm: run_layernorm,
            O.Ew: run_ew, O.Upsample: run_upsample, O.Layout: run_layout, O.Fourier: run_fourier, O.Gather: run_gather,
            O.TimeEmb: run_timeemb, O.DdimStep: run_ddim, O.UniPCStep: run_unipc, O.Softmax: run_softmax}


def run(ops, lower_check: bool = True):
    with torch.no_grad():
        for op in ops:
            if lower_check:
                op.lower()            # descriptor construction must succeed for every op (shape/stride/dtype checks)
            DISPATCH[type(op)](op)
