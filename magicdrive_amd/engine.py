"""Host-side engine: turns (config, reference-layout state dicts, batch shape) into the flat op
programs libmdx executes — a step-invariant *prologue* (conditioning encoders, context K/V, timestep
tables) and the per-step *denoise* program (BEV-ControlNet -> multi-view UNet -> CFG + DDIM).

What is hoisted out of the 50-step loop (the reference recomputes all of it every step):
  * camera Fourier+cam2token, box Fourier+MLP, the [cam | text | box] context assembly
      (magicdrive/networks/unet_addon_rawbox.py:743-793),
  * the BEV-map conv encoder — once per map, not 6 x per step (unet_addon_rawbox.py:842-850),
  * every attn2.to_k / to_v projection of the context (attention_processor.py:520-525),
  * timestep sinusoid + time MLP + every resnet's time_emb_proj, for ALL steps at once, as one GEMM
      (unet_2d_condition_multiview.py:386-411; resnet.py:612-618).
What is de-duplicated inside a step: the cross-view attention's q/k/v/out projections run once per view
instead of once per (view, neighbour) pair (magicdrive/networks/blocks.py:112-121).

Every arithmetic op is a libmdx kernel (see ops.py); torch is used for device memory and for marshalling
user inputs (dtype/layout of the arguments, uncond/cond concatenation).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import ops as O
from . import packing as PK
from .networks.spec import heads_at

BF16, F32 = torch.bfloat16, torch.float32


def q_prescale(head_dim: int) -> float:
    """softmax scale * log2(e) (attention_processor.py: scale = head_dim ** -0.5): folded into the to_q weights when they are packed, so
    that Q K^T is the base-2 exponent and the attention kernel can subtract the running maximum inside the QK MFMA."""
    return float(head_dim) ** -0.5 * 1.4426950408889634


class PackedNet:
    """Device-resident, kernel-layout weights of one network, packed lazily from a reference state dict."""

    def __init__(self, sd: Dict[str, torch.Tensor], device, dtype=BF16):
        assert dtype in (torch.bfloat16, torch.float16), "the kernels take bf16 or fp16 operands (fp32 accumulate)"
        self.sd = sd
        self.device = device
        self.dtype = dtype                 # 16-bit storage type of the packed weights = the type the activations are computed in
        self.cache: Dict[Tuple, torch.Tensor] = {}

    def has(self, key: str) -> bool:
        return key in self.sd

    def _get(self, tag, keys, fn):
        ck = (tag,) + tuple(keys)
        if ck not in self.cache:
            self.cache[ck] = fn(*[self.sd[k].detach().float() for k in keys]).to(self.device)
        return self.cache[ck]

    def lin(self, key, scale: float = 1.0):            # [N,K] bf16 (1x1 convs too)
        return self._get(("lin", scale), [key], lambda w: (w.reshape(w.shape[0], -1) * scale).contiguous().to(self.dtype))

    def conv(self, key):                               # [Cout,kh,kw,Cin] bf16
        return self._get("conv", [key], lambda w: PK.pack_conv_weight(w, self.dtype))

    def conv_cin_padded(self, key, cin_pad: int):      # [Cout,kh,kw,cin_pad] bf16, extra input channels zero
        def f(w):
            wp = torch.zeros(w.shape[0], cin_pad, w.shape[2], w.shape[3])
            wp[:, :w.shape[1]] = w
            return PK.pack_conv_weight(wp, self.dtype)
        return self._get(("convpad", cin_pad), [key], f)

    def vec(self, key, scale: float = 1.0):            # fp32 vector
        return self._get(("vec", scale), [key], lambda v: (v.reshape(-1) * scale).contiguous().to(F32))

    def vec_bf16(self, key):
        return self._get("vecbf", [key], lambda v: v.reshape(-1).contiguous().to(self.dtype))

    def cat_lin(self, keys: Sequence[str], scales: Sequence[float] = ()):            # rows concatenated (each part optionally scaled, in fp32)
        sc = tuple(scales) if scales else (1.0,) * len(keys)
        return self._get(("catlin",) + sc, keys,
                         lambda *ws: torch.cat([w.reshape(w.shape[0], -1) * f for w, f in zip(ws, sc)], 0).contiguous().to(self.dtype))

    def ln_lin(self, keys: Sequence[str], ln_pre: str, scales: Sequence[float] = ()):
        """Linear(s) that consume LayerNorm `ln_pre`'s output, with the norm's affine part folded in (MdxGemmDesc.ln_eps):
            W' = [s_i W_i] diag(gamma)   (16-bit),   b' = [s_i W_i] beta   (fp32; diffusers' to_q / to_k / to_v carry no bias of their own),
            csum[n] = sum_k W'[n][k]     (of the ROUNDED W': the kernel subtracts mean * csum from sum_k x_k W'[n][k]).
        Returns (W', b', csum)."""
        sc = tuple(scales) if scales else (1.0,) * len(keys)
        allk = list(keys) + [ln_pre + "weight", ln_pre + "bias"]

        def cat(*ts):
            ws, g, b = ts[:-2], ts[-2].reshape(-1), ts[-1].reshape(-1)
            return torch.cat([w.reshape(w.shape[0], -1) * f for w, f in zip(ws, sc)], 0), g, b

        w = self._get(("lnw",) + sc, allk, lambda *ts: (cat(*ts)[0] * cat(*ts)[1][None, :]).contiguous().to(self.dtype))
        b = self._get(("lnb",) + sc, allk, lambda *ts: (cat(*ts)[0] @ cat(*ts)[2]).contiguous().to(F32))
        cs = self._get(("lncs",) + sc, allk, lambda *ts: (cat(*ts)[0] * cat(*ts)[1][None, :]).to(self.dtype).float().sum(1).contiguous())
        return w, b, cs

    def cat_vec(self, keys: Sequence[str]):
        return self._get("catvec", keys, lambda *vs: torch.cat([v.reshape(-1) for v in vs]).contiguous().to(F32))

    def geglu(self, wkey, bkey):
        w = self._get("gegluw", [wkey, bkey], lambda w, b: PK.pack_geglu(w, b, self.dtype)[0])
        b = self._get("geglub", [wkey, bkey], lambda w, b: PK.pack_geglu(w, b, self.dtype)[1])
        return w, b

    def ln_geglu(self, wkey, bkey, ln_pre: str):
        """GEGLU projection that consumes LayerNorm `ln_pre` with the affine part folded in and the rows in packed [32 value | 32 gate] order
        (PackedNet.ln_lin + PackedNet.geglu): W' = pack(W diag(gamma)), b' = pack(b + W beta), csum = row sums of the rounded W'.  Only taken
        when the row statistics come from the producer (MdxGemmDesc.ln_stats)."""
        keys = [wkey, bkey, ln_pre + "weight", ln_pre + "bias"]
        fold = lambda w, b, g, be: PK.pack_geglu(w * g.reshape(-1)[None, :], b.reshape(-1) + w @ be.reshape(-1), self.dtype)
        w = self._get("lngegluw", keys, lambda w, b, g, be: fold(w, b, g, be)[0])
        bb = self._get("lngeglub", keys, lambda w, b, g, be: fold(w, b, g, be)[1])
        cs = self._get("lngeglucs", keys, lambda w, b, g, be: fold(w, b, g, be)[0].float().sum(1).contiguous())
        return w, bb, cs

    def folded_affine(self, w2key, b2key, w1key, b1key, b1_scale: float = 1.0):
        """y = W2 (W1 x + s b1) + b2  ->  (W2 W1) x + (W2 s b1 + b2), folded in fp32."""
        keys = [w2key, b2key, w1key, b1key]
        w = self._get(("foldw", b1_scale), keys, lambda w2, b2, w1, b1: (w2 @ w1).contiguous().to(self.dtype))
        b = self._get(("foldb", b1_scale), keys, lambda w2, b2, w1, b1: (w2 @ (b1 * b1_scale) + b2).contiguous().to(F32))
        return w, b

    def gated_affine(self, w1key, b1key, alpha_key: Optional[str], b1_scale: float = 1.0):
        """y = tanh(alpha) * (W1 x + s b1) (GatedConnector, blocks.py:24-32) -> (diag(tanh alpha) W1) x + tanh(alpha) s b1, folded in fp32;
        alpha_key None = the identity connector (zero_module_type "none"): W1 x + s b1."""
        keys = [w1key, b1key] + ([alpha_key] if alpha_key else [])
        gate = lambda a: torch.tanh(a[0].reshape(-1)) if a else None
        w = self._get(("gatew", b1_scale), keys, lambda w1, b1, *a: (w1 if not a else gate(a)[:, None] * w1).contiguous().to(self.dtype))
        b = self._get(("gateb", b1_scale), keys, lambda w1, b1, *a: (b1 * b1_scale if not a else gate(a) * b1 * b1_scale).contiguous().to(F32))
        return w, b

    def table(self, key):                              # 2-D bf16 table (class tokens)
        return self._get("table", [key], lambda t: t.contiguous().to(self.dtype))


class Pool:
    """Exact-size free-list of device buffers; the program's ops alias freed buffers, which is safe because
    a program executes in order on one stream."""

    # Debugging aid, off in the product (tools/batch_sweep.py --guard sets it): every buffer CLOSES a device segment of its own (the caching
    # allocator maps a request of >= 10 MiB as one segment of exactly the 2 MiB-rounded size), so a kernel that reads or writes past the end
    # of a buffer runs into unmapped addresses and faults instead of quietly touching its neighbour.  Round 5's edge-tile over-read
    # (gemm_xl.hip fetch_residual) sat in the kernel for two rounds because something was always mapped behind the residual.
    guard = False
    GUARD_SEGMENT = 12 << 20

    def __init__(self, device, dtype=BF16):
        self.device = device
        self.dtype = dtype                 # default element type of a buffer: the plan's 16-bit activation type
        self.free_list: Dict[Tuple, List[torch.Tensor]] = {}
        self.total_bytes = 0

    @staticmethod
    def alloc(n: int, dtype, device) -> torch.Tensor:
        """`n` elements of device memory; under Pool.guard the buffer ENDS at the end of a device segment of its own (the VAE plans, which keep
        every buffer alive instead of pooling them, allocate through this too)."""
        if not Pool.guard:
            return torch.empty(n, dtype=dtype, device=device)
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        seg = max(Pool.GUARD_SEGMENT, -(-nbytes // (2 << 20)) * (2 << 20))
        raw = torch.empty(seg, dtype=torch.uint8, device=device)
        off = (seg - nbytes) & ~15         # 16-byte aligned start (vector loads, LDS-DMA): at most 15 bytes of slack behind the buffer
        return raw[off:off + nbytes].view(dtype)

    def _alloc(self, n: int, dtype) -> torch.Tensor:
        return Pool.alloc(n, dtype, self.device)

    def get(self, shape, dtype=None) -> torch.Tensor:
        dtype = dtype or self.dtype
        n = 1
        for s in shape:
            n *= int(s)
        key = (n, dtype)
        fl = self.free_list.get(key)
        if fl:
            return fl.pop().view(*shape)
        self.total_bytes += n * torch.empty(0, dtype=dtype).element_size()
        return self._alloc(n, dtype).view(*shape)

    def put(self, t: Optional[torch.Tensor]):
        if t is None:
            return
        base = t.reshape(-1) if t.is_contiguous() else None
        assert base is not None, "only whole contiguous buffers go back to the pool"
        self.free_list.setdefault((base.numel(), base.dtype), []).append(base)


def level_sizes(h: int, w: int, n_levels: int) -> List[Tuple[int, int]]:
    """Spatial size per UNet level: Downsample2D = conv3x3 stride 2 pad 1 (resnet.py:198-222)."""
    out = [(h, w)]
    for _ in range(n_levels - 1):
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        out.append((h, w))
    return out


def skip_geometry(cfg, h: int, w: int) -> List[Tuple[int, int, int]]:
    """(H, W, C) of the UNet's skip tensors in encoder order (conv_in output, every down-block layer, every downsampler):
    unet_2d_condition.py:880-905 `down_block_res_samples`."""
    boc = cfg["block_out_channels"]
    sizes = level_sizes(h, w, len(boc))
    geo = [(sizes[0][0], sizes[0][1], boc[0])]
    for i in range(len(boc)):
        geo += [(sizes[i][0], sizes[i][1], boc[i])] * cfg["layers_per_block"]
        if i + 1 < len(boc):
            geo.append((sizes[i + 1][0], sizes[i + 1][1], boc[i]))
    return geo


class Act:
    """A channels-last activation: tokens [B*H*W, C] bf16 plus its geometry."""

    __slots__ = ("t", "B", "H", "W", "C")

    def __init__(self, t, B, H, W, C):
        self.t, self.B, self.H, self.W, self.C = t, B, H, W, C

    @property
    def tok(self):            # [M, C]
        return self.t.view(self.B * self.H * self.W, self.C)

    @property
    def bhwc(self):
        return self.t.view(self.B, self.H, self.W, self.C)

    @property
    def btc(self):
        return self.t.view(self.B, self.H * self.W, self.C)

    def channels(self, c0: int, c1: int) -> "ActSlice":
        """The channel range [c0, c1) of this activation as a strided view (row stride = self.C): what a producer writes when its output
        is the first part of a decoder concat (Builder.decoder)."""
        return ActSlice(self, c0, c1)


class ActSlice:
    """Channels [c0, c1) of a wider channels-last buffer: same accessors as Act, strided rows.  Owns no storage (Builder.free ignores it)."""

    __slots__ = ("parent", "c0", "B", "H", "W", "C")

    def __init__(self, parent: Act, c0: int, c1: int):
        self.parent, self.c0, self.B, self.H, self.W, self.C = parent, c0, parent.B, parent.H, parent.W, c1 - c0

    @property
    def tok(self):
        return self.parent.tok[:, self.c0:self.c0 + self.C]

    @property
    def bhwc(self):
        return self.parent.bhwc[..., self.c0:self.c0 + self.C]

    @property
    def btc(self):
        return self.parent.btc[..., self.c0:self.c0 + self.C]


class Builder:
    """Emits IR ops for the two networks at a fixed batch geometry."""

    def __init__(self, cfg, device, n_views: int, n_cam: int, ws_mb: int = 64, dtype=BF16):
        self.cfg = cfg
        self.device = device
        self.B = n_views                  # c * b * n_cam views through the nets
        self.n_cam = n_cam
        self.dtype = dtype                # bf16 or fp16: activations and weights of every op this builder emits
        self.pool = Pool(device, dtype)
        self.ops: List[object] = []
        self.ws = torch.empty(ws_mb * 1024 * 1024 // 4, dtype=F32, device=device)
        self.groups = cfg["norm_num_groups"]
        self.eps = cfg["norm_eps"]
        pair = cfg["neighboring_view_pair"]
        pair = {int(k): [int(x) for x in v] for k, v in pair.items()}
        # cross-view attention form (BasicMultiviewTransformerBlock._construct_attn_input, blocks.py:106-142):
        #   add    (default): one attention per (view, neighbour), outputs summed -> 2 sources, separate softmax, to_out bias twice
        #   concat : the neighbours' tokens concatenated into one kv sequence     -> 2 sources, ONE softmax, bias once
        #   self   : all cameras of the scene as one sequence (queries of every view see every view) -> n_cam sources, ONE softmax, bias once
        self.nattn = cfg.get("neighboring_attn_type", "add")
        if self.nattn not in ("add", "concat", "self"):
            raise NotImplementedError(f"Unknown type: {self.nattn}")          # the reference's error (blocks.py:139-141)
        kv = []
        for i in range(n_views):
            base = (i // n_cam) * n_cam
            srcs = list(range(n_cam)) if self.nattn == "self" else pair[i % n_cam]
            for nb in srcs:
                kv.append(base + nb)
        if self.nattn == "add":
            assert all(len(v) == 2 for v in pair.values()), "the summed cross-view form handles exactly 2 neighbours per view"
        else:
            assert len({len(v) for v in pair.values()}) == 1 and (self.nattn == "self" or len(pair[0]) <= 8), "concat / self: equal source counts, <= 8"
        self.xv_nsrc = n_cam if self.nattn == "self" else len(pair[0])
        assert self.xv_nsrc <= 8, "joint cross-view attention handles <= 8 sources"
        self.kvmap = torch.tensor(kv, dtype=torch.int32, device=device)

    # ---- small helpers ---------------------------------------------------------------
    def emit(self, op):
        self.ops.append(op)
        return op

    def new(self, B, H, W, C) -> Act:
        return Act(self.pool.get((B * H * W, C)), B, H, W, C)

    def free(self, a):
        if isinstance(a, ActSlice):           # a view into a concat buffer: the buffer is freed by whoever allocated it
            return
        self.pool.put(a.t if isinstance(a, Act) else a)

    def groupnorm(self, net, pre, x: Act, eps, silu, name) -> Act:
        y = self.new(x.B, x.H, x.W, x.C)
        self.emit(O.GroupNorm(x.btc, y.btc, net.vec(pre + "weight"), net.vec(pre + "bias"), self.groups, eps, silu, ws=self.ws, name=name))
        return y

    def layernorm(self, net, pre, x: torch.Tensor, name) -> torch.Tensor:
        y = self.pool.get(tuple(x.shape))
        self.emit(O.LayerNorm(x, y, net.vec(pre + "weight"), net.vec(pre + "bias"), 1e-5, name=name))
        return y

    @staticmethod
    def wq_of(W: torch.Tensor, M: int, epilogue: int) -> Optional[torch.Tensor]:
        """The fragment-ordered copy of a linear weight (packing.pack_wq) for the W-direct persistent GEMM (csrc/gemm_xd.hip), made once per
        weight tensor and kept on it; None when the route is switched off (option XD, the default: measured on par with the LDS-both persistent kernel), for shapes that kernel does not take (K % 128, K < 640) or launches too small for the persistent tile
        walk (fewer than two 256 x 256 tiles per CU: the library would not route them there)."""
        if W.dim() != 2 or not W.is_contiguous() or epilogue not in (L.EPI_NONE, L.EPI_GEGLU) or not L.get_option("XD"):
            return None
        N, K = W.shape
        if K % 128 or K < 640 or N % 16 or ((M + 255) // 256) * ((N + 255) // 256) < 512:
            return None
        wq = getattr(W, "_mdx_wq", None)
        if wq is None:
            wq = PK.pack_wq(W)
            W._mdx_wq = wq
        return wq

    def gemm(self, A, W, N_out, bias=None, R=None, epilogue=L.EPI_NONE, name="", out=None, **kw) -> torch.Tensor:
        C = out if out is not None else self.pool.get((A.shape[0], N_out))
        if "Wq" not in kw and not any(kw.get(k) is not None for k in ("Vt", "temb", "rowstat")) and not kw.get("ln_eps"):
            kw["Wq"] = self.wq_of(W, A.shape[0], epilogue)
        self.emit(O.Gemm(A, W, C, bias=bias, R=R, epilogue=epilogue, ws=self.ws, name=name, **kw))
        return C

    # ---- resnet -------------------------------------------------------------------------
    def resnet(self, net, pre, x: Act, temb: "TembTable", name, out=None) -> Act:
        """ResnetBlock2D.forward (resnet.py:590-640).  `out`: where the block's output goes (an ActSlice of the next decoder concat)."""
        cout = net.sd[pre + "conv1.weight"].shape[0]
        a = self.groupnorm(net, pre + "norm1.", x, self.eps, True, name + ".norm1")
        h = self.new(x.B, x.H, x.W, cout)
        off = temb.offset[pre + "time_emb_proj.weight"]
        self.emit(O.Conv(a.bhwc, net.conv(pre + "conv1.weight"), h.bhwc, bias=net.vec(pre + "conv1.bias"),
                         temb=temb.table[:, off:], sel=temb.sel, temb_sel_stride=temb.sel_stride, temb_b_stride=temb.b_stride,
                         ws=self.ws, name=name + ".conv1"))
        self.free(a)
        b = self.groupnorm(net, pre + "norm2.", h, self.eps, True, name + ".norm2")
        self.free(h)
        if net.has(pre + "conv_shortcut.weight"):
            sc = self.new(x.B, x.H, x.W, cout)
            self.gemm(x.tok, net.lin(pre + "conv_shortcut.weight"), cout, bias=net.vec(pre + "conv_shortcut.bias"), out=sc.tok, name=name + ".shortcut")
        else:
            sc = x
        if out is None:
            out = self.new(x.B, x.H, x.W, cout)
        assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, cout)
        self.emit(O.Conv(b.bhwc, net.conv(pre + "conv2.weight"), out.bhwc, bias=net.vec(pre + "conv2.bias"), R=sc.bhwc, ws=self.ws, name=name + ".conv2"))
        self.free(b)
        if sc is not x:
            self.free(sc)
        return out

    # ---- transformer ------------------------------------------------------------------------
    @staticmethod
    def fuses_qkv(B, T, C) -> bool:
        """Level 0 at real batch sizes: q, k and V^T come from ONE weight-stationary launch pair (gemm_ws.hip), which can also take the
        LayerNorm in front of it (transformer_block)."""
        return C == 320 and T % 8 == 0 and B * T >= 8192

    def self_like_attention(self, net, pre, n: torch.Tensor, B, T, C, heads, cross_view: bool, name, ln_pre: Optional[str] = None,
                            ln_scratch: Optional[torch.Tensor] = None, ln_stats: Optional[torch.Tensor] = None) -> torch.Tensor:
        """q,k fused projection + V^T projection + fused attention over the same token set (attn1) or over the
        two neighbour views (attn4).  n: normalised tokens [B*T, C] — or, with ln_pre (only when fuses_qkv), the RAW tokens: LayerNorm
        `ln_pre` is then applied inside the projection (ops.Gemm.ln_eps)."""
        ldv = PK.round_up(T, 8)
        vt = self.pool.get((B, C, ldv))
        qs = q_prescale(C // heads)         # softmax scale * log2(e), folded into to_q (MdxAttnDesc.q_prescaled)
        qkv_keys = [pre + "to_q.weight", pre + "to_k.weight", pre + "to_v.weight"]
        if self.fuses_qkv(B, T, C):
            # level 0: ONE weight-stationary launch pair reads the tokens for q, k and v; the V columns are stored transposed
            # (gemm_ws.hip) — replaces the batched V^T GEMM (245 TFLOP/s, 2 % of the step)
            qk = self.pool.get((B * T, 2 * C))
            if ln_pre is not None:
                w, b, cs = net.ln_lin(qkv_keys, ln_pre, (qs, 1.0, 1.0))
                self.emit(O.Gemm(n, w, qk, bias=b, Vt=vt, vt_from=2 * C, vt_T=T, ln_eps=1e-5, ln_csum=cs, ln_scratch=ln_scratch, ln_stats=ln_stats,
                                 ws=self.ws, name=name + ".ln+qkv"))
            else:
                self.emit(O.Gemm(n, net.cat_lin(qkv_keys, (qs, 1.0, 1.0)), qk, Vt=vt, vt_from=2 * C, vt_T=T, ws=self.ws, name=name + ".qkv"))
        else:
            assert ln_pre is None
            qk = self.gemm(n, net.cat_lin([pre + "to_q.weight", pre + "to_k.weight"], (qs, 1.0)), 2 * C, name=name + ".qk")
            self.emit(O.Gemm(net.lin(pre + "to_v.weight"), n.view(B, T, C), vt[:, :, :T], name=name + ".vT"))
        ao = self.pool.get((B * T, C))
        qk3 = qk.view(B, T, 2 * C)
        self.emit(O.Attn(qk3[:, :, :C], qk3[:, :, C:], vt, ao.view(B, T, C), heads=heads, Tk=T, scale=(C // heads) ** -0.5, q_prescaled=True,
                         kvmap=self.kvmap if cross_view else None, nsrc=self.xv_nsrc if cross_view else 1,
                         joint=cross_view and self.nattn != "add", name=name + ".attn"))
        self.pool.put(qk)
        self.pool.put(vt)
        return ao

    ROWSTAT_PARTS = 3          # column parts of a producer's row statistics (gemm_ws.hip: one per 128-column tile of N = 320; other routes fill part 0)
    # norm3 -> ff.net.0 folded into the GEGLU epilogue (PackedNet.ln_geglu, gemm_ws_kernel<geglu,lns>): built, tested, measured and NOT the default —
    # the two extra fused multiply-adds per value and gate on 2560 raw columns cost the (VALU-bound) GEGLU epilogue +1.9 ms per step over its 7
    # level-0 launches, against 1.3 ms for the 7 LayerNorm passes they replace (576 views, profiles/r06_ln_stats_ab.log).
    FOLD_NORM3 = False

    def rowstat(self, M: int) -> torch.Tensor:
        """fp32 [parts, M, 2] buffer for the (sum, sum of squares) a projection's store phase leaves for the LayerNorm behind it."""
        return self.pool.get((self.ROWSTAT_PARTS, M, 2), F32)

    def transformer_block(self, net, pre, h: torch.Tensor, B, T, C, heads, ctx_kv, name, h_stats: Optional[torch.Tensor] = None,
                          out_stats: bool = False):
        """BasicTransformerBlock / BasicMultiviewTransformerBlock.forward (magicdrive/networks/blocks.py:144-238).
        `h_stats`: row statistics of h from its producer (proj_in or the previous block), given exactly when the LayerNorms of this block are
        folded (fuse_ln); out_stats: also return the row statistics of the block's output (a further block's norm1 reads it): (h, stats)."""
        # LayerNorm -> projection pairs whose GEMM has K = 320 and takes the weight-stationary route normalise inside the GEMM (gemm_ws.hip reads
        # the whole row anyway): norm1 -> q/k/v, norm2 -> to_q, norm4 -> cross-view q/k/v, and (round 6) norm3 -> GEGLU.  The mean / rstd of a row
        # come from the (sum, sum of squares) that the PRODUCER of the tensor wrote from its store phase (MdxGemmDesc.rowstat_out -> ln_stats:
        # proj_in, attn1.to_out + residual, attn2.to_out + residual, connector(attn4.to_out) + residual are all C x C projections of the same
        # kernel family) instead of being recomputed from the streamed rows in every N-tile's workgroup (round 3: 8x for a fused q/k/v launch pair,
        # +150 us per pair; for the 20 N-tiles of the GEGLU more than the LayerNorm pass it would have saved).  The scratch buffer is only written by
        # routes that cannot fuse (small M, forced routes).
        fuse_ln = self.fuses_qkv(B, T, C)
        assert (h_stats is not None) == fuse_ln
        M = B * T
        # 1. self-attention
        if fuse_ln:
            n1 = self.pool.get(tuple(h.shape))
            ao = self.self_like_attention(net, pre + "attn1.", h, B, T, C, heads, False, name + ".attn1", ln_pre=pre + "norm1.", ln_scratch=n1, ln_stats=h_stats)
            self.pool.put(h_stats)
        else:
            n1 = self.layernorm(net, pre + "norm1.", h, name + ".norm1")
            ao = self.self_like_attention(net, pre + "attn1.", n1, B, T, C, heads, False, name + ".attn1")
        self.pool.put(n1)
        s1 = self.rowstat(M) if fuse_ln else None
        h1 = self.gemm(ao, net.lin(pre + "attn1.to_out.0.weight"), C, bias=net.vec(pre + "attn1.to_out.0.bias"), R=h, rowstat=s1, name=name + ".attn1.out")
        self.pool.put(ao); self.pool.put(h)
        # 2. context cross-attention with prologue-computed K / V^T
        if fuse_ln:
            n2 = self.pool.get(tuple(h1.shape))
            w, b, cs = net.ln_lin([pre + "attn2.to_q.weight"], pre + "norm2.", (q_prescale(C // heads),))
            q2 = self.gemm(h1, w, C, bias=b, ln_eps=1e-5, ln_csum=cs, ln_scratch=n2, ln_stats=s1, name=name + ".attn2.ln+q")
            self.pool.put(s1)
        else:
            n2 = self.layernorm(net, pre + "norm2.", h1, name + ".norm2")
            q2 = self.gemm(n2, net.lin(pre + "attn2.to_q.weight", q_prescale(C // heads)), C, name=name + ".attn2.q")
        self.pool.put(n2)
        Kc, Vtc, S = ctx_kv[pre + "attn2."]
        ao2 = self.pool.get((B * T, C))
        self.emit(O.Attn(q2.view(B, T, C), Kc, Vtc, ao2.view(B, T, C), heads=heads, Tk=S, scale=(C // heads) ** -0.5, q_prescaled=True, name=name + ".attn2"))
        self.pool.put(q2)
        has4 = net.has(pre + "attn4.to_q.weight")
        fold3 = fuse_ln and self.FOLD_NORM3
        s2 = self.rowstat(M) if (fuse_ln and (has4 or fold3)) else None       # read by norm4 (multiview blocks) or, folded, by norm3 (ControlNet blocks)
        h2 = self.gemm(ao2, net.lin(pre + "attn2.to_out.0.weight"), C, bias=net.vec(pre + "attn2.to_out.0.bias"), R=h1, rowstat=s2, name=name + ".attn2.out")
        self.pool.put(ao2); self.pool.put(h1)
        # 2b. cross-view attention: out = W_o (o_left + o_right) + 2 b_o ; connector ; residual (blocks.py:190-222)
        s3 = s2
        if has4:
            if fuse_ln:
                n4 = self.pool.get(tuple(h2.shape))
                ao4 = self.self_like_attention(net, pre + "attn4.", h2, B, T, C, heads, True, name + ".attn4", ln_pre=pre + "norm4.", ln_scratch=n4, ln_stats=s2)
                self.pool.put(s2)
            else:
                n4 = self.layernorm(net, pre + "norm4.", h2, name + ".norm4")
                ao4 = self.self_like_attention(net, pre + "attn4.", n4, B, T, C, heads, True, name + ".attn4")
            self.pool.put(n4)
            # connector(to_out(o_l + o_r) + 2 b_o) is one affine map: fold it at pack time,
            #   W = W_c W_o ,  b = W_c (2 b_o) + b_c     (one GEMM instead of two per block; fp32 fold, bf16 weights)
            bo_scale = 2.0 if self.nattn == "add" else 1.0                        # concat / self: ONE attention per view, its out-bias once
            if net.has(pre + "connector.weight"):                                 # zero_module_type zero_linear (blocks.py:81-83)
                wf, bf_ = net.folded_affine(pre + "connector.weight", pre + "connector.bias", pre + "attn4.to_out.0.weight", pre + "attn4.to_out.0.bias", bo_scale)
            else:                                                                 # gated: tanh(alpha) per channel (blocks.py:24-32, 84-85); none: identity (:86-88)
                wf, bf_ = net.gated_affine(pre + "attn4.to_out.0.weight", pre + "attn4.to_out.0.bias",
                                           pre + "connector.alpha" if net.has(pre + "connector.alpha") else None, bo_scale)
            s3 = self.rowstat(M) if fold3 else None
            h3 = self.gemm(ao4, wf, C, bias=bf_, R=h2, rowstat=s3, name=name + ".attn4.out+connector")
            self.pool.put(ao4); self.pool.put(h2)
        else:
            h3 = h2
        # 3. GEGLU feed-forward
        if fold3:
            n3 = self.pool.get(tuple(h3.shape))
            wg, bg, csg = net.ln_geglu(pre + "ff.net.0.proj.weight", pre + "ff.net.0.proj.bias", pre + "norm3.")
            g = self.gemm(h3, wg, 4 * C, bias=bg, epilogue=L.EPI_GEGLU, ln_eps=1e-5, ln_csum=csg, ln_scratch=n3, ln_stats=s3, name=name + ".ff.ln+geglu")
            self.pool.put(s3)
        else:
            n3 = self.layernorm(net, pre + "norm3.", h3, name + ".norm3")
            wg, bg = net.geglu(pre + "ff.net.0.proj.weight", pre + "ff.net.0.proj.bias")
            g = self.gemm(n3, wg, 4 * C, bias=bg, epilogue=L.EPI_GEGLU, name=name + ".ff.geglu")
        self.pool.put(n3)
        s4 = self.rowstat(M) if (fuse_ln and out_stats) else None
        h4 = self.gemm(g, net.lin(pre + "ff.net.2.weight"), C, bias=net.vec(pre + "ff.net.2.bias"), R=h3, rowstat=s4, name=name + ".ff.out")
        self.pool.put(g); self.pool.put(h3)
        return (h4, s4) if out_stats else h4

    def transformer2d(self, net, pre, x: Act, heads, ctx_kv, name, out=None) -> Act:
        """Transformer2DModel.forward (transformer_2d.py:276-315): GN(eps 1e-6) -> 1x1 -> block -> 1x1 -> + input.  `out` as in resnet."""
        B, T, C = x.B, x.H * x.W, x.C
        gn = self.groupnorm(net, pre + "norm.", x, 1e-6, False, name + ".norm")
        fuse_ln = self.fuses_qkv(B, T, C)
        hs = self.rowstat(B * T) if fuse_ln else None                    # row statistics of proj_in's output for the first block's norm1
        h = self.gemm(gn.tok, net.lin(pre + "proj_in.weight"), C, bias=net.vec(pre + "proj_in.bias"), rowstat=hs, name=name + ".proj_in")
        self.free(gn)
        i = 0
        while net.has(f"{pre}transformer_blocks.{i}.norm1.weight"):
            more = net.has(f"{pre}transformer_blocks.{i + 1}.norm1.weight")       # a further block's norm1 reads this block's ff.out + residual
            r = self.transformer_block(net, f"{pre}transformer_blocks.{i}.", h, B, T, C, heads, ctx_kv, f"{name}.tb{i}", h_stats=hs, out_stats=more)
            h, hs = r if more else (r, None)
            i += 1
        if out is None:
            out = self.new(x.B, x.H, x.W, C)
        assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, C)
        self.gemm(h, net.lin(pre + "proj_out.weight"), C, bias=net.vec(pre + "proj_out.bias"), R=x.tok, out=out.tok, name=name + ".proj_out")
        self.pool.put(h)
        return out

    # ---- encoder (shared by ControlNet and UNet) ----------------------------------------------
    def encoder(self, net, x0: Act, temb, ctx_kv, tag, mid_out=None) -> Tuple[Act, List[Act]]:
        """down blocks + mid block; returns (mid output, skip list incl. conv_in output).  `mid_out`: where the mid block's output goes (the x half
        of the decoder's first concat buffer, decoder_concats)."""
        cfg = self.cfg
        skips = [x0]
        x = x0
        nblk = len(cfg["block_out_channels"])
        for i in range(nblk):
            has_attn = cfg["down_block_types"][i].startswith("CrossAttn")
            for j in range(cfg["layers_per_block"]):
                y = self.resnet(net, f"down_blocks.{i}.resnets.{j}.", x, temb, f"{tag}.d{i}.r{j}")
                if has_attn:
                    z = self.transformer2d(net, f"down_blocks.{i}.attentions.{j}.", y, heads_at(cfg, i), ctx_kv, f"{tag}.d{i}.a{j}")
                    self.free(y)
                    y = z
                skips.append(y)
                x = y
            dk = f"down_blocks.{i}.downsamplers.0.conv."
            if net.has(dk + "weight"):
                Ho, Wo = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
                y = self.new(x.B, Ho, Wo, x.C)
                self.emit(O.Conv(x.bhwc, net.conv(dk + "weight"), y.bhwc, bias=net.vec(dk + "bias"), stride=(2, 2), pad=(1, 1), ws=self.ws, name=f"{tag}.d{i}.down"))
                skips.append(y)
                x = y
        m = self.resnet(net, "mid_block.resnets.0.", x, temb, f"{tag}.mid.r0")
        m2 = self.transformer2d(net, "mid_block.attentions.0.", m, heads_at(cfg, nblk - 1), ctx_kv, f"{tag}.mid.a0")
        self.free(m)
        m3 = self.resnet(net, "mid_block.resnets.1.", m2, temb, f"{tag}.mid.r1", out=mid_out)
        self.free(m2)
        return m3, skips

    def decoder_concats(self, net, B: int, x_c: int, skip_geo: Sequence[Tuple[int, int, int]]) -> List[Act]:
        """The concat buffers [x | skip] of every up-block layer, allocated up front in the order the decoder consumes them.  `skip_geo`: (H, W, C)
        of the skips in encoder order (the decoder pops from the end); x_c: channels of the mid output.  A caller that owns them lets the PRODUCER of
        each half write it in place — the mid block / previous layer the x half (as decoder() always does), the ControlNet zero-convs the skip half
        (SamplerPlan: `skip + residual` lands in the concat, unet_2d_condition_multiview.py:464-488 + unet_2d_blocks.py:1948-1951) — so that the
        step program holds no concat copy at all (round 5: 13 ew_vec8 copies per step)."""
        cfg = self.cfg
        geo = list(skip_geo)
        cats = []
        for i in range(len(cfg["block_out_channels"])):
            for j in range(cfg["layers_per_block"] + 1):
                H, W, sc = geo.pop()
                cats.append(self.new(B, H, W, x_c + sc))
                x_c = net.sd[f"up_blocks.{i}.resnets.{j}.conv1.weight"].shape[0]
        assert not geo
        return cats

    def decoder(self, net, x: Act, skips: List[Act], temb, ctx_kv, tag, cats: Optional[List[Act]] = None) -> Act:
        """up blocks (unet_2d_blocks.py:1886-2111) with explicit upsample sizes
        (unet_2d_condition_multiview.py:491-516), then conv_norm_out + SiLU (:519-521).
        `cats` (decoder_concats): the concat buffers, with every skip half already in place and x = cats[0].channels(0, x.C); `skips` is then only
        read for its geometry."""
        cfg = self.cfg
        nblk = len(cfg["block_out_channels"])
        rev_heads = [heads_at(cfg, k) for k in reversed(range(nblk))]
        # The concat [x | skip] of every up-block layer (unet_2d_blocks.py:1948-1951): the layer that PRODUCES x writes it straight into
        # the first channels of the next concat buffer (strided output of its last conv / GEMM), so only the skip half is copied — the
        # x half was a read + write of the whole tensor per layer (round 3: 12 of the step's 24 ew_vec8 launches).  `cat` is allocated
        # before the producer runs; x's only other consumers (the upsampler, conv_norm_out) get a standalone buffer.
        n_layers = cfg["layers_per_block"] + 1
        placed = cats is not None
        cats = list(cats) if placed else None
        skips = list(skips)
        cat = None                                 # concat buffer whose first x.C channels already hold x
        if placed:
            cat = cats.pop(0)
            assert isinstance(x, ActSlice) and x.parent is cat and x.c0 == 0, "with pre-placed concats the mid output must live in cats[0]"
        for i in range(nblk):
            has_attn = cfg["up_block_types"][i].startswith("CrossAttn")
            for j in range(n_layers):
                s = skips.pop()
                if cat is None:
                    cat = self.new(x.B, x.H, x.W, x.C + s.C)
                    self.emit(O.Ew(L.EW_COPY, x.tok, cat.tok[:, :x.C], name=f"{tag}.u{i}.cat{j}a"))
                    self.free(x)
                assert cat.C == x.C + s.C and (cat.H, cat.W) == (s.H, s.W)
                if not placed:
                    self.emit(O.Ew(L.EW_COPY, s.tok, cat.tok[:, x.C:], name=f"{tag}.u{i}.cat{j}b"))
                    self.free(s)
                # where this layer's output goes: into the next layer's concat when that one follows at the same resolution
                cout = net.sd[f"up_blocks.{i}.resnets.{j}.conv1.weight"].shape[0]
                nxt = None
                if j + 1 < n_layers:
                    nxt = cats.pop(0) if placed else self.new(x.B, x.H, x.W, cout + skips[-1].C)
                    assert (nxt.B, nxt.H, nxt.W, nxt.C) == (x.B, x.H, x.W, cout + skips[-1].C)
                dst = nxt.channels(0, cout) if nxt is not None else None
                y = self.resnet(net, f"up_blocks.{i}.resnets.{j}.", cat, temb, f"{tag}.u{i}.r{j}", out=None if has_attn else dst)
                self.free(cat)
                if has_attn:
                    z = self.transformer2d(net, f"up_blocks.{i}.attentions.{j}.", y, rev_heads[i], ctx_kv, f"{tag}.u{i}.a{j}", out=dst)
                    self.free(y)
                    y = z
                x, cat = y, nxt
            uk = f"up_blocks.{i}.upsamplers.0.conv."
            if net.has(uk + "weight"):
                Ho, Wo = skips[-1].H, skips[-1].W          # next skip's size (forced interpolation size)
                up = self.new(x.B, Ho, Wo, x.C)
                self.emit(O.Upsample(x.bhwc, up.bhwc, PK.nearest_index(x.H, Ho).to(self.device), PK.nearest_index(x.W, Wo).to(self.device), name=f"{tag}.u{i}.nearest"))
                self.free(x)
                cat = cats.pop(0) if placed else self.new(x.B, Ho, Wo, x.C + skips[-1].C)      # the upsampler conv writes the x half of the next block's first concat
                assert (cat.H, cat.W, cat.C) == (Ho, Wo, x.C + skips[-1].C)
                y = cat.channels(0, x.C)
                self.emit(O.Conv(up.bhwc, net.conv(uk + "weight"), y.bhwc, bias=net.vec(uk + "bias"), ws=self.ws, name=f"{tag}.u{i}.upconv"))
                self.free(up)
                x = y
        assert not placed or not cats
        y = self.groupnorm(net, "conv_norm_out.", x, self.eps, True, f"{tag}.norm_out")
        self.free(x)
        return y


class TembTable:
    """fp32 [rows, sum(Cout)] table of time_emb_proj(silu(time_mlp(sinusoid(t)))) for every resnet of one network.
    rows = DDIM steps (pipeline: row picked by the device-side step counter) or batch entries (module API)."""

    def __init__(self, net: PackedNet, rows: int, device, per_sample: bool):
        keys = [k for k in net.sd.keys() if k.endswith("time_emb_proj.weight")]
        self.keys = keys
        self.offset: Dict[str, int] = {}
        off = 0
        for k in keys:
            self.offset[k] = off
            off += net.sd[k].shape[0]
        self.width = off
        self.rows = rows
        self.table = torch.zeros(rows, off, dtype=F32, device=device)
        self.sel = torch.zeros(1, dtype=torch.int32, device=device)
        self.sel_stride = 0 if per_sample else off
        self.b_stride = off if per_sample else 0
        self.t = torch.zeros(rows, dtype=F32, device=device)          # timesteps, filled by the caller

    def emit_fill(self, bld: Builder, net: PackedNet, cfg):
        """Ops that (re)compute the whole table from self.t."""
        c0 = cfg["block_out_channels"][0]
        dev = bld.device
        sin = torch.empty(self.rows, c0, dtype=F32, device=dev)
        bld.emit(O.TimeEmb(self.t, sin, flip_sin_to_cos=cfg["flip_sin_to_cos"], freq_shift=cfg["freq_shift"], name="temb.sin"))
        h1 = torch.empty(self.rows, 4 * c0, dtype=F32, device=dev)
        h2 = torch.empty(self.rows, 4 * c0, dtype=F32, device=dev)
        act = torch.empty(self.rows, 4 * c0, dtype=net.dtype, device=dev)
        as4 = lambda t: t.view(self.rows, 1, 1, t.shape[1])
        w1 = net.lin("time_embedding.linear_1.weight"); w2 = net.lin("time_embedding.linear_2.weight")
        bld.emit(O.Conv(as4(sin), w1.view(w1.shape[0], 1, 1, w1.shape[1]), as4(h1), bias=net.vec("time_embedding.linear_1.bias"),
                        pad=(0, 0), epilogue=L.EPI_SILU, direct=True, name="temb.linear_1+silu"))
        bld.emit(O.Conv(as4(h1), w2.view(w2.shape[0], 1, 1, w2.shape[1]), as4(h2), bias=net.vec("time_embedding.linear_2.bias"),
                        pad=(0, 0), direct=True, name="temb.linear_2"))
        bld.emit(O.Ew(L.EW_SILU, h2, act, name="temb.silu"))
        wcat = net.cat_lin(self.keys)
        bcat = net.cat_vec([k[:-len("weight")] + "bias" for k in self.keys])
        bld.emit(O.Gemm(act, wcat, self.table, bias=bcat, name="temb.table"))
        self._keep = (sin, h1, h2, act)


def build_context_kv(bld: Builder, net: PackedNet, ctx: torch.Tensor, B: int, S: int) -> Dict[str, Tuple[torch.Tensor, torch.Tensor, int]]:
    """attn2 K and V^T of every transformer block of `net`, from ctx [B, S, D] (step-invariant)."""
    out = {}
    D = ctx.shape[-1]
    ldv = PK.round_up(S, 8)
    for k in net.sd.keys():
        if not k.endswith("attn2.to_k.weight"):
            continue
        pre = k[:-len("to_k.weight")]
        C = net.sd[k].shape[0]
        Kc = torch.empty(B, S, C, dtype=net.dtype, device=bld.device)
        Vt = torch.zeros(B, C, ldv, dtype=net.dtype, device=bld.device)
        bld.emit(O.Gemm(ctx.view(B * S, D), net.lin(k), Kc.view(B * S, C), ws=bld.ws, name=pre + "K"))
        bld.emit(O.Gemm(net.lin(pre + "to_v.weight"), ctx, Vt[:, :, :S], name=pre + "vT"))
        out[pre] = (Kc, Vt, S)
    return out
