"""Plans: the op programs of the hot path at a fixed batch geometry.

  SamplerPlan   — what StableDiffusionBEVControlNetPipeline.__call__ runs: one prologue program and one
                  per-step program (ControlNet -> UNet -> CFG + DDIM), replayed num_inference_steps times
                  (eager, or as a captured hipGraph).
  ControlNetPlan / UNetPlan — the module-level forwards behind BEVControlNetModel.forward /
                  UNet2DConditionModelMultiview.forward (reference signatures).

View ordering everywhere: view index = (cfg_half * b + scene) * n_cam + cam, "uncond first, cond second"
(pipeline_bev_controlnet.py:330-343, 352-354).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import ops as O
from . import packing as PK
from .networks import spec
from .engine import Act, Builder, PackedNet, Pool, TembTable, build_context_kv, level_sizes, skip_geometry

BF16, F32 = torch.bfloat16, torch.float32
CIN_PAD = 8      # latent channels (4) zero-padded so conv_in meets the MFMA path's Cin % 8 == 0


def _rows_as_pixels(t2d: torch.Tensor) -> torch.Tensor:
    """[M, C] view with row stride ld -> [M, 1, 1, C] 'image' of M one-pixel samples (for linear layers run
    through the direct-conv kernel, which may write into a column/row slice of a wider buffer)."""
    M, C = t2d.shape
    ld = t2d.stride(0)
    assert t2d.stride(1) == 1
    return torch.as_strided(t2d, (M, 1, 1, C), (ld, ld, ld, 1), t2d.storage_offset())


def device_stream(device=None) -> int:
    """Raw hipStream_t of torch's current stream ON `device` (not on whatever device happens to be current)."""
    if not torch.cuda.is_available():
        return 0
    return torch.cuda.current_stream(device).cuda_stream


def _stream(device=None) -> int:
    return device_stream(device)


class PlanCache:
    """Small LRU of plans keyed by batch geometry.  A plan owns a 64 MB workspace, its activation pool, K/V and temb tables and a
    captured hipGraph; the reference's validation flow changes the padded box count (and with it the plan key) almost every batch,
    so an unbounded dict would accumulate ~100 plans per batch size.  Evicted plans release their graph and device buffers.
    Size: the library option PLAN_CACHE (csrc/options.h, default 6: a multi-stream call holds one plan per scene chunk)."""

    def __init__(self, maxsize: Optional[int] = None):
        from collections import OrderedDict
        from . import _lib
        self.maxsize = maxsize if maxsize is not None else max(1, int(_lib.get_option("PLAN_CACHE")))
        self._d = OrderedDict()

    def get(self, key):
        p = self._d.get(key)
        if p is not None:
            self._d.move_to_end(key)
        return p

    def put(self, key, plan):
        self._d[key] = plan
        self._d.move_to_end(key)
        while len(self._d) > self.maxsize:
            _, old = self._d.popitem(last=False)
            rel = getattr(old, "release", None)
            if rel is not None:
                rel()

    def clear(self):
        for old in self._d.values():
            rel = getattr(old, "release", None)
            if rel is not None:
                rel()
        self._d.clear()

    def values(self):
        return self._d.values()

    def __len__(self):
        return len(self._d)

    def __contains__(self, key):
        return key in self._d


class ConditioningBuffers:
    """Input-side device buffers of the conditioning prologue + the ops that turn them into the context tokens
    ehs_with_cam [B, 1+77+L, D] and the per-view map feature (BEVControlNetModel.forward :743-793, 842-850)."""

    def __init__(self, bld: Builder, cn: PackedNet, cfg, n_scene: int, n_cam: int, L_box: int, latent_hw, n_text: int = 77):
        dev = bld.device
        H16 = bld.dtype                     # the plan's 16-bit activation type (bf16 or fp16)
        self.dtype = H16
        cc = cfg["controlnet"]; bb = cc["bbox"]
        self.minmax_normalize = bool(bb.get("minmax_normalize", False))
        self.n_scene, self.n_cam, self.L = n_scene, n_cam, L_box
        B = n_scene * n_cam
        D = cfg["cross_attention_dim"]
        self.S = 1 + n_text + L_box
        self.n_text = n_text
        S = self.S
        self.ctx = torch.zeros(B, S, D, dtype=H16, device=dev)
        # camera: [B, 7, 3] fp32 (columns of the (3,7) matrix) -> Fourier 189 -> cam2token -> ctx[:, 0]
        ncol = cc["uncond_cam_in_dim"][1]
        F_cam = cc["cam_embedder_num_freqs"]
        self.cam_in = torch.zeros(B, ncol, 3, dtype=F32, device=dev)
        cam_emb = torch.empty(B, ncol * (3 + 6 * F_cam), dtype=H16, device=dev)
        bld.emit(O.Fourier(self.cam_in, cam_emb, F_cam, name="cam.fourier"))
        wc = cn.lin("cam2token.weight")
        bld.emit(O.Conv(cam_emb.view(B, 1, 1, -1), wc.view(wc.shape[0], 1, 1, wc.shape[1]), _rows_as_pixels(self.ctx[:, 0, :]),
                        bias=cn.vec("cam2token.bias"), pad=(0, 0), direct=True, name="cam2token"))
        # boxes
        if L_box > 0:
            P, Fq = bb["n_corners"], bb["embedder_num_freq"]
            fdim = P * (3 + 6 * Fq)
            pd = bb["proj_dims"]; ctd = bb["class_token_dim"]
            n = B * L_box
            self.box_in = torch.zeros(n, P, 3, dtype=F32, device=dev)
            self.box_mask = torch.zeros(n, dtype=torch.uint8, device=dev)
            self.box_cls = torch.zeros(n, dtype=torch.int64, device=dev)
            pos = torch.empty(n, fdim, dtype=H16, device=dev)
            bld.emit(O.Fourier(self.box_in, pos, Fq, mask=self.box_mask, null_feat=cn.vec("bbox_embedder.null_pos_feature"), name="box.fourier"))
            e1 = torch.empty(n, pd[0] + ctd, dtype=H16, device=dev)       # [silu(bbox_proj) | class token]
            wp = cn.lin("bbox_embedder.bbox_proj.weight")
            bld.emit(O.Conv(pos.view(n, 1, 1, fdim), wp.view(pd[0], 1, 1, fdim), _rows_as_pixels(e1[:, :pd[0]]),
                            bias=cn.vec("bbox_embedder.bbox_proj.bias"), pad=(0, 0), epilogue=L.EPI_SILU, direct=True, name="box.bbox_proj"))
            bld.emit(O.Gather(cn.table("bbox_embedder._class_tokens"), e1[:, pd[0]:], self.box_cls, mask=self.box_mask,
                              null_row=cn.vec_bf16("bbox_embedder.null_class_feature"), name="box.class_token"))
            e2 = torch.empty(n, pd[1], dtype=H16, device=dev)
            bld.emit(O.Gemm(e1, cn.lin("bbox_embedder.second_linear.0.weight"), e2, bias=cn.vec("bbox_embedder.second_linear.0.bias"), epilogue=L.EPI_SILU, name="box.mlp0"))
            e3 = torch.empty(n, pd[2], dtype=H16, device=dev)
            bld.emit(O.Gemm(e2, cn.lin("bbox_embedder.second_linear.2.weight"), e3, bias=cn.vec("bbox_embedder.second_linear.2.bias"), epilogue=L.EPI_SILU, name="box.mlp2"))
            # last layer writes straight into the box rows of every view's context (batched over views)
            bld.emit(O.Gemm(e3.view(B, L_box, pd[2]), cn.lin("bbox_embedder.second_linear.4.weight"), self.ctx[:, 1 + n_text:, :],
                            bias=cn.vec("bbox_embedder.second_linear.4.bias"), name="box.mlp4"))
            self._keep = (pos, e1, e2, e3)
        self._keep_cam = cam_emb
        # BEV map encoder: once per scene (the reference convolves 6 identical copies every step)
        h, w = latent_hw
        msz = cc["map_size"]
        ch = cc["conditioning_embedding_out_channels"]
        self.map_in = torch.zeros(n_scene, msz[0], msz[1], msz[2], dtype=F32, device=dev)
        x = torch.empty(n_scene, msz[1], msz[2], msz[0], dtype=H16, device=dev)
        bld.emit(O.Layout(self.map_in, x, True, name="map.nhwc"))
        pre = "controlnet_cond_embedding."
        layers = [(pre + "conv_in.", (1, 1), (1, 1), True)]
        nblk = 0
        while cn.has(f"{pre}blocks.{nblk}.weight"):
            nblk += 1
        plus = spec.map_embedder_plus_size(cfg)
        for i in range(nblk):
            if plus is None:                                # BEVControlNetConditioningEmbedding, map_embedder.py:36-56
                if i < nblk - 2:
                    layers.append((f"{pre}blocks.{i}.", (1, 1), (1, 1), True) if i % 2 == 0 else (f"{pre}blocks.{i}.", (2, 2), (2, 1), True))
                elif i == nblk - 2:
                    layers.append((f"{pre}blocks.{i}.", (1, 1), (2, 1), True))
                else:
                    layers.append((f"{pre}blocks.{i}.", (2, 1), (2, 1), True))
            else:                                           # ...Plus, map_embedder.py:94-117: pad 1 everywhere; strides 1,1 | 1,2 | 1,(2,1)
                stride = (1, 1) if i % 2 == 0 else ((2, 1) if i == nblk - 1 else ((1, 1) if i == 1 else (2, 2)))
                layers.append((f"{pre}blocks.{i}.", stride, (1, 1), True))
        layers.append((pre + "conv_out.", (1, 1), (1, 1), False))
        keep = [x]
        for key, stride, pad, act in layers:
            if plus is not None and key == pre + "conv_out.":
                x = self._adaptive_avg_pool_silu(bld, x, plus, keep)      # blocks[-1] = AdaptiveAvgPool2d, then SiLU (map_embedder.py:117, :71-73)
            wt = cn.conv(key + "weight")
            Hi, Wi = x.shape[1], x.shape[2]
            Ho = (Hi + 2 * pad[0] - 3) // stride[0] + 1
            Wo = (Wi + 2 * pad[1] - 3) // stride[1] + 1
            y = torch.empty(n_scene, Ho, Wo, wt.shape[0], dtype=H16, device=dev)
            bld.emit(O.Conv(x, wt, y, bias=cn.vec(key + "bias"), stride=stride, pad=pad, epilogue=L.EPI_SILU if act else L.EPI_NONE,
                            direct=(wt.shape[3] % 8 != 0 or wt.shape[0] % 4 != 0), ws=bld.ws, name="map." + key))
            keep.append(y)
            x = y
        if (x.shape[1], x.shape[2]) != (h, w):
            raise ValueError(f"map encoder output {tuple(x.shape[1:3])} != latent size {(h, w)}: select BEVControlNetConditioningEmbeddingPlus with "
                             f"conditioning_embedding_size={[h, w]} (configs/exp/272x736.yaml:15-22; spec.with_plus_map_embedder)")
        C0 = x.shape[3]
        self.map_rep = torch.empty(B, h, w, C0, dtype=H16, device=dev)
        for s in range(n_scene):
            for c in range(n_cam):
                bld.emit(O.Ew(L.EW_COPY, x[s].view(h * w, C0), self.map_rep[s * n_cam + c].view(h * w, C0), name="map.repeat"))
        self._keep_map = keep

    @staticmethod
    def _adaptive_avg_pool_silu(bld, x, out_hw, keep):
        """SiLU(AdaptiveAvgPool2d(out_hw)(x)) for channels-last x [n, Hi, Wi, C] on the existing kernels (runs once per scene, in the
        prologue): the pooling is a fixed linear map over pixels, so it is ONE batched GEMM  out[n] = P @ x[n]  with the
        [Ho*Wo, Hi*Wi] matrix of window weights (torch's windows: start = floor(o*I/O), end = ceil((o+1)*I/O)) and the SiLU epilogue;
        the GEMM wants its second operand K-contiguous, i.e. x as [C, Hi*Wi] = NCHW, hence the layout op.  Weights 1/count are bf16:
        exact for counts 1, 2, 4 (every window of the 432x768 case), 2^-9 relative otherwise."""
        n, Hi, Wi, C = x.shape
        Ho, Wo = out_hw
        dev = x.device
        H16 = x.dtype

        def windows(I, Oo):
            m = torch.zeros(Oo, I, dtype=torch.float64)
            for o in range(Oo):
                a, b = (o * I) // Oo, -((-(o + 1) * I) // Oo)
                m[o, a:b] = 1.0 / (b - a)
            return m
        P = torch.kron(windows(Hi, Ho), windows(Wi, Wo)).to(H16).to(dev)              # [(oy,ox), (iy,ix)]
        xn = torch.empty(n, C, Hi, Wi, dtype=H16, device=dev)
        bld.emit(O.Layout(x, xn, False, name="map.pool.nchw"))
        y = torch.empty(n, Ho, Wo, C, dtype=H16, device=dev)
        bld.emit(O.Gemm(P, xn.view(n, C, Hi * Wi), y.view(n, Ho * Wo, C), epilogue=L.EPI_SILU, ws=bld.ws, name="map.pool"))
        keep += [P, xn, y]
        return y

    # ---- input marshalling (torch: dtype / layout of user tensors only) ----
    def load(self, camera_param: torch.Tensor, text: torch.Tensor, bev_map: torch.Tensor, boxes: Optional[Dict[str, torch.Tensor]]):
        """camera_param (n_scene, n_cam, 3, 7); text (n_scene, n_text, D); bev_map (n_scene, C, H, W);
        boxes {bboxes (n_scene, n_cam | 1, L, 8, 3), classes, masks} or None."""
        ns, nc = self.n_scene, self.n_cam
        assert camera_param.shape[:2] == (ns, nc), f"camera_param {tuple(camera_param.shape)} vs scenes {ns} cams {nc}"
        self.cam_in.copy_(camera_param.to(self.cam_in.device, F32).permute(0, 1, 3, 2).reshape(ns * nc, -1, 3))
        assert text.shape[0] == ns and text.shape[1] == self.n_text
        self.ctx.view(ns, nc, self.S, -1)[:, :, 1:1 + self.n_text, :] = text.to(self.ctx.device, self.ctx.dtype).unsqueeze(1)
        self.map_in.copy_(bev_map.to(self.map_in.device, F32))
        if self.L > 0:
            assert boxes is not None
            bb, cl, mk = boxes["bboxes"], boxes["classes"], boxes["masks"]
            if bb.shape[1] != nc:                               # view-shared boxes (unet_addon_rawbox.py:785-787)
                assert bb.shape[1] == 1
                bb = bb.expand(ns, nc, *bb.shape[2:]); cl = cl.expand(ns, nc, -1); mk = mk.expand(ns, nc, -1)
            assert bb.shape[2] == self.L, f"boxes padded to {bb.shape[2]} but plan built for L={self.L}"
            bb = bb.to(self.box_in.device, F32)
            if self.minmax_normalize:          # normalizer('all-xyz'): (xyz - XYZ_MIN) / XYZ_RANGE (bbox_embedder.py:10-25, :175-176)
                bb = (bb - bb.new_tensor([-200.0, -300.0, -20.0])) / bb.new_tensor([350.0, 650.0, 80.0])
            self.box_in.copy_(bb.reshape(-1, *bb.shape[3:]))
            self.box_cls.copy_(cl.to(self.box_cls.device, torch.int64).reshape(-1))
            self.box_mask.copy_(mk.to(self.box_mask.device).reshape(-1).to(torch.uint8))


def _emit_controlnet(bld: Builder, cn: PackedNet, x_in: torch.Tensor, cond: ConditioningBuffers, temb: TembTable, ctx_kv, h, w):
    """conv_in + map feature, then the encoder copy (unet_addon_rawbox.py:846-880).
    x_in: bf16 [B,h,w,CIN_PAD] — latent channels zero-padded to 8 so conv_in runs on the MFMA implicit-GEMM path."""
    c0 = bld.cfg["block_out_channels"][0]
    x0 = bld.new(bld.B, h, w, c0)
    bld.emit(O.Conv(x_in, cn.conv_cin_padded("conv_in.weight", x_in.shape[3]), x0.bhwc, bias=cn.vec("conv_in.bias"), R=cond.map_rep, ws=bld.ws, name="cn.conv_in+map"))
    return bld.encoder(cn, x0, temb, ctx_kv, "cn")


def _zero_conv_keys(n_skips: int):
    return [f"controlnet_down_blocks.{k}." for k in range(n_skips)]


class SamplerPlan:
    """Everything the DDIM loop needs for `b` scenes (x `n_cam` views, x2 with CFG) on one GPU."""

    def __init__(self, cfg, unet: PackedNet, cn: PackedNet, device, b: int, do_cfg: bool, L_box: int, latent_hw=(28, 50),
                 num_steps: int = 50, guidance_scale: float = 2.0, conditioning_scale: float = 1.0, n_text: int = 77,
                 scheduler_kind: str = "ddim", given_view_mode: int = 0, fork: bool = False):
        self.cfg, self.device = cfg, device
        assert scheduler_kind in ("ddim", "unipc")
        assert given_view_mode in (0, 1, 2)
        self.scheduler_kind = scheduler_kind
        self.given_view_mode = given_view_mode
        n_cam = len(cfg["neighboring_view_pair"])
        self.b, self.n_cam, self.c = b, n_cam, (2 if do_cfg else 1)
        self.do_cfg = do_cfg
        self.num_steps = num_steps
        h, w = latent_hw
        self.h, self.w = h, w
        B = self.c * b * n_cam
        self.B = B
        Cl = cfg["in_channels"]
        assert unet.dtype == cn.dtype, "UNet and ControlNet must be packed in the same 16-bit type"
        self.dtype = unet.dtype
        bld = Builder(cfg, device, B, n_cam, dtype=self.dtype)
        self.bld = bld
        # state
        self.x = torch.zeros(b * n_cam, h, w, Cl, dtype=F32, device=device)            # latents, NHWC
        self.x_in = torch.zeros(B, h, w, CIN_PAD, dtype=self.dtype, device=device)      # model input ([uncond|cond] copies), channels padded
        self.eps = torch.zeros(B, h, w, cfg["out_channels"], dtype=F32, device=device)
        self.coef = torch.zeros(num_steps, 4 if scheduler_kind == "ddim" else 12, dtype=F32, device=device)
        if scheduler_kind == "unipc":       # multistep history of the fused UniPC update (scheduling_unipc_multistep.py:518-600)
            self.x_last = torch.zeros_like(self.x); self.m1 = torch.zeros_like(self.x); self.m2 = torch.zeros_like(self.x)
        self.step_ctr = torch.zeros(1, dtype=torch.int32, device=device)
        if given_view_mode:       # pipeline_bev_controlnet_given_view.py: known clean latents of some views + every view's initial noise
            self.gv_mask = torch.zeros(b * n_cam, dtype=torch.uint8, device=device)
            self.gv_cond = torch.zeros_like(self.x)
            self.gv_noise = torch.zeros_like(self.x)
        # ---------------- prologue ----------------
        self.cond = ConditioningBuffers(bld, cn, cfg, self.c * b, n_cam, L_box, latent_hw, n_text)
        self.temb_cn = TembTable(cn, num_steps, device, per_sample=False)
        self.temb_un = TembTable(unet, num_steps, device, per_sample=False)
        self.temb_cn.sel = self.step_ctr
        self.temb_un.sel = self.step_ctr
        self.temb_cn.emit_fill(bld, cn, cfg)
        self.temb_un.emit_fill(bld, unet, cfg)
        self.kv_cn = build_context_kv(bld, cn, self.cond.ctx, B, self.cond.S)
        self.kv_un = build_context_kv(bld, unet, self.cond.ctx, B, self.cond.S)
        self.prologue_ops = bld.ops
        bld.ops = []
        # ---------------- one denoising step ----------------
        # The ControlNet and the UNet encoder of a step are INDEPENDENT until the zero-convs add the one into the other's skips (both read x_in;
        # unet_addon_rawbox.py:882-910 / unet_2d_condition_multiview.py:464-488).  fork: the two are emitted with SEPARATE buffer pools and workspaces
        # (the pool's aliasing of freed buffers is only safe for ops that execute in order), so that the pipeline may replay them side by side on
        # two streams and join before the zero-convs (round 5: the small-batch operating point, where a launch fills a fraction of the chip and
        # every kernel boundary is a dependency bubble).  The linear program (self.step) stays valid: same ops, same buffers.
        cn_mid, cn_skips = _emit_controlnet(bld, cn, self.x_in, self.cond, self.temb_cn, self.kv_cn, h, w)
        n_cn = len(bld.ops)
        if fork:
            pool_cn = bld.pool
            bld.pool = Pool(device, self.dtype)
            bld.ws = torch.empty_like(bld.ws)
        c0 = cfg["block_out_channels"][0]
        # The decoder's concat buffers [x | skip] exist before the UNet encoder runs: the mid block writes its output into the first one's x half,
        # and each zero-conv below writes `skip + ControlNet residual` straight into its concat's skip half — no concat copy is left in the step
        # (round 5: 13 per step).  The reference adds the residuals out of place (unet_2d_condition_multiview.py:464-488) and
        # concatenates in every up-block layer (unet_2d_blocks.py:1948-1951): same values, one pass instead of three.
        cats = bld.decoder_concats(unet, B, cfg["block_out_channels"][-1], skip_geometry(cfg, h, w))
        u0 = bld.new(B, h, w, c0)
        bld.emit(O.Conv(self.x_in, unet.conv_cin_padded("conv_in.weight", CIN_PAD), u0.bhwc, bias=unet.vec("conv_in.bias"), ws=bld.ws, name="unet.conv_in"))
        u_mid, u_skips = bld.encoder(unet, u0, self.temb_un, self.kv_un, "unet", mid_out=cats[0].channels(0, cfg["block_out_channels"][-1]))
        self.fork_at = (n_cn, len(bld.ops)) if fork else None   # step_ops[:a] ControlNet | [a:b] UNet conv_in + encoder | [b:] zero-convs, decoder, scheduler
        if fork:                                                # behind the join the tail may reuse what either branch has freed
            for k_, v_ in pool_cn.free_list.items():
                bld.pool.free_list.setdefault(k_, []).extend(v_)
            bld.pool.total_bytes += pool_cn.total_bytes
        # zero-convs (unet_addon_rawbox.py:882-910 + unet_2d_condition_multiview.py:464-488): after the UNet encoder + mid consumed the un-added
        # tensors, exactly like the reference's out-of-place `sample + residual`; skip k (encoder order) is popped by decoder layer n - 1 - k
        assert len(cn_skips) == len(u_skips) == len(cats)
        for k, (cs, us) in enumerate(zip(cn_skips, u_skips)):
            key = f"controlnet_down_blocks.{k}."
            cat = cats[len(cats) - 1 - k]
            assert (cat.H, cat.W) == (us.H, us.W)
            dst = cat.channels(cat.C - us.C, cat.C)
            bld.emit(O.Gemm(cs.tok, cn.lin(key + "weight", conditioning_scale), dst.tok, bias=cn.vec(key + "bias", conditioning_scale), R=us.tok, name=f"zero_conv.{k}"))
            bld.free(cs)
            bld.free(us)
        bld.emit(O.Gemm(cn_mid.tok, cn.lin("controlnet_mid_block.weight", conditioning_scale), u_mid.tok,
                        bias=cn.vec("controlnet_mid_block.bias", conditioning_scale), R=u_mid.tok, name="zero_conv.mid"))
        bld.free(cn_mid)
        y = bld.decoder(unet, u_mid, u_skips, self.temb_un, self.kv_un, "unet", cats=cats)
        bld.emit(O.Conv(y.bhwc, unet.conv("conv_out.weight"), self.eps, bias=unet.vec("conv_out.bias"), direct=True, name="unet.conv_out"))
        bld.free(y)
        if scheduler_kind == "ddim":
            gv = {}
            if given_view_mode:
                gv = dict(gv_mask=self.gv_mask, gv_cond=self.gv_cond.view(-1), gv_noise=self.gv_noise.view(-1), gv_mode=given_view_mode,
                          gv_last_step=num_steps - 1)
            bld.emit(O.DdimStep(self.x.view(-1), self.eps.view(-1), self.coef, self.step_ctr, x_in=self.x_in.view(-1, CIN_PAD), cfg=do_cfg,
                                guidance=guidance_scale, xin_c=Cl, name="cfg+ddim", **gv))
        else:
            gv = {}
            if given_view_mode:
                gv = dict(gv_mask=self.gv_mask, gv_cond=self.gv_cond.view(-1), gv_noise=self.gv_noise.view(-1), gv_mode=given_view_mode,
                          gv_last_step=num_steps - 1)
            bld.emit(O.UniPCStep(self.x.view(-1), self.eps.view(-1), self.coef, self.step_ctr, self.x_last.view(-1), self.m1.view(-1),
                                 self.m2.view(-1), x_in=self.x_in.view(-1, CIN_PAD), cfg=do_cfg, guidance=guidance_scale, xin_c=Cl, name="cfg+unipc", **gv))
        self.step_ops = bld.ops
        bld.ops = []
        self.prologue: Optional[L.Program] = None
        self.step: Optional[L.Program] = None
        self.step_cn = self.step_enc = self.step_tail = None    # fork: the three parts of the step as programs of their own
        self._ev = None                                         # fork: (ready, done) events of launch_step

    def compile(self):
        self.prologue = O.build_program(self.prologue_ops)
        if self.fork_at is None:
            self.step = O.build_program(self.step_ops)
        else:
            # a forked plan replays its three parts; the linear program (and its hipGraph) is only built if launch_step ever falls back to it
            a, b_ = self.fork_at
            self.step_cn = O.build_program(self.step_ops[:a])
            self.step_enc = O.build_program(self.step_ops[a:b_])
            self.step_tail = O.build_program(self.step_ops[b_:])

    def launch_step(self, main, side, use_graph: bool = True):
        """One denoising step.  Without fork (or without a side stream): the linear program on `main`.  With fork: the ControlNet on `side`, the UNet
        encoder on `main`, joined before the zero-convs — `main` / `side` are torch streams; everything else the caller does stays on `main`."""
        go = (lambda prog, st: prog.launch(st.cuda_stream)) if use_graph else (lambda prog, st: prog.run(st.cuda_stream))
        if self.fork_at is None or side is None:
            if self.step is None:
                self.step = O.build_program(self.step_ops)
            go(self.step, main)
            return
        if self._ev is None:                                    # two events per plan, re-recorded every step (not two new ones per step)
            self._ev = (torch.cuda.Event(), torch.cuda.Event())
        ready, done = self._ev
        ready.record(main)                                      # x_in of this step: written by the previous step's scheduler kernel on main
        side.wait_event(ready)
        go(self.step_cn, side)
        done.record(side)
        go(self.step_enc, main)
        main.wait_event(done)
        go(self.step_tail, main)

    def release(self):
        """Drop the captured graph and every device buffer this plan owns (called on LRU eviction).  The plan's last replay may still
        be in flight (an output_type='latent' caller that has not synchronised): destroying a hipGraphExec / freeing its buffers under
        a running graph is not safe, so the device is drained first (evictions are rare: once per new batch geometry)."""
        if self.device is not None and torch.device(self.device).type == "cuda":
            torch.cuda.synchronize(self.device)
        for prog in (self.prologue, self.step, self.step_cn, self.step_enc, self.step_tail):
            if prog is not None:
                prog.destroy()
        self.prologue = self.step = self.step_cn = self.step_enc = self.step_tail = None
        self.prologue_ops = self.step_ops = []
        self.bld = None
        self.cond = self.temb_cn = self.temb_un = self.kv_cn = self.kv_un = None

    # ---- per-call inputs ----
    def load_inputs(self, latents: torch.Tensor, camera_param, text, bev_map, boxes, timesteps: torch.Tensor, coef: torch.Tensor,
                    given_mask: Optional[torch.Tensor] = None, given_latents: Optional[torch.Tensor] = None):
        """latents (b, n_cam, C, h, w) any float dtype; camera/text/map/boxes already hold the [uncond | cond] halves.
        given_mask (b, n_cam) bool + given_latents (b, n_cam, C, h, w): the known views of a given-view plan."""
        b, nc = self.b, self.n_cam
        assert latents.shape[:2] == (b, nc)
        xl = latents.to(self.device, F32).reshape(b * nc, *latents.shape[2:]).permute(0, 2, 3, 1).contiguous()
        if self.given_view_mode:
            assert given_mask is not None and given_latents is not None and tuple(given_mask.shape) == (b, nc)
            gm = given_mask.to(self.device).reshape(-1).bool()
            self.gv_mask.copy_(gm.to(torch.uint8))
            self.gv_noise.copy_(xl)                                                  # original_noise (:263)
            self.gv_cond.copy_(given_latents.to(self.device, F32).reshape(b * nc, *latents.shape[2:]).permute(0, 2, 3, 1))
            c0 = coef[0].to(self.device, F32)                                        # add_noise at the first timestep (:265-275, :284-291)
            if self.scheduler_kind == "ddim":
                a0, s0 = c0[0], c0[1]                                                # DDIM row: sqrt(acp_t), sqrt(1 - acp_t), ...
            else:
                a0, s0 = 1.0 / c0[0], -c0[1] / c0[0]                                 # UniPC row: a = 1 / alpha_t, b = -sigma_t / alpha_t
            xl = torch.where(gm.view(-1, 1, 1, 1), a0 * self.gv_cond + s0 * self.gv_noise, xl)
        else:
            assert given_mask is None, "this plan was built without given views"
        self.x.copy_(xl)
        self.x_in.view(self.c, b * nc, self.h, self.w, CIN_PAD)[..., :xl.shape[-1]].copy_(self.x.unsqueeze(0).expand(self.c, *self.x.shape))
        self.cond.load(camera_param, text, bev_map, boxes)
        t = timesteps.to(self.device, F32)
        assert t.numel() == self.num_steps
        self.temb_cn.t.copy_(t); self.temb_un.t.copy_(t)
        self.coef.copy_(coef.to(self.device, F32))
        self.step_ctr.zero_()
        if self.scheduler_kind == "unipc":
            self.x_last.zero_(); self.m1.zero_(); self.m2.zero_()

    def run(self, use_graph: bool = True) -> torch.Tensor:
        """prologue + num_steps denoising steps on the current stream; returns latents (b, n_cam, C, h, w) fp32."""
        if self.prologue is None:
            self.compile()
        if self.step is None:
            self.step = O.build_program(self.step_ops)
        st = _stream(self.device)
        self.prologue.run(st)
        for _ in range(self.num_steps):
            if use_graph:
                self.step.launch(st)
            else:
                self.step.run(st)
        return self.latents()

    def latents(self) -> torch.Tensor:
        return self.x.view(self.b, self.n_cam, self.h, self.w, -1).permute(0, 1, 4, 2, 3).contiguous()

    def latents_on(self, stream) -> torch.Tensor:
        """latents() as a copy made on `stream` (a torch stream that already waits for this plan's last replay)."""
        with torch.cuda.stream(stream):
            return self.latents()


class ControlNetPlan:
    """BEVControlNetModel.forward at a fixed geometry: returns 12 down residuals + mid (NCHW) + ehs_with_cam."""

    def __init__(self, cfg, cn: PackedNet, device, n_scene: int, L_box: int, latent_hw, conditioning_scale: float = 1.0, n_text: int = 77):
        n_cam = len(cfg["neighboring_view_pair"])
        B = n_scene * n_cam
        h, w = latent_hw
        self.cfg, self.device, self.B, self.n_scene, self.n_cam, self.h, self.w = cfg, device, B, n_scene, n_cam, h, w
        self.dtype = cn.dtype
        bld = Builder(cfg, device, B, n_cam, dtype=self.dtype)
        self.bld = bld
        self.sample_nchw = torch.zeros(B, cfg["in_channels"], h, w, dtype=F32, device=device)
        self.x_in = torch.zeros(B, h, w, CIN_PAD, dtype=self.dtype, device=device)
        bld.emit(O.Layout(self.sample_nchw, self.x_in[..., :cfg["in_channels"]], True, name="cn.sample.nhwc"))
        self.cond = ConditioningBuffers(bld, cn, cfg, n_scene, n_cam, L_box, latent_hw, n_text)
        self.temb = TembTable(cn, B, device, per_sample=True)        # one timestep per view row
        self.temb.emit_fill(bld, cn, cfg)
        self.kv = build_context_kv(bld, cn, self.cond.ctx, B, self.cond.S)
        mid, skips = _emit_controlnet(bld, cn, self.x_in, self.cond, self.temb, self.kv, h, w)
        self.down_out: List[torch.Tensor] = []
        for k, s in enumerate(skips):
            key = f"controlnet_down_blocks.{k}."
            r = bld.new(s.B, s.H, s.W, s.C)
            bld.emit(O.Gemm(s.tok, cn.lin(key + "weight", conditioning_scale), r.tok, bias=cn.vec(key + "bias", conditioning_scale), name=f"zero_conv.{k}"))
            o = torch.zeros(s.B, s.C, s.H, s.W, dtype=self.dtype, device=device)
            bld.emit(O.Layout(r.bhwc, o, False, name=f"res{k}.nchw"))
            self.down_out.append(o)
        r = bld.new(mid.B, mid.H, mid.W, mid.C)
        bld.emit(O.Gemm(mid.tok, cn.lin("controlnet_mid_block.weight", conditioning_scale), r.tok, bias=cn.vec("controlnet_mid_block.bias", conditioning_scale), name="zero_conv.mid"))
        self.mid_out = torch.zeros(mid.B, mid.C, mid.H, mid.W, dtype=self.dtype, device=device)
        bld.emit(O.Layout(r.bhwc, self.mid_out, False, name="mid.nchw"))
        self.ops = bld.ops
        self.program: Optional[L.Program] = None

    def run(self, sample, timestep, camera_param, boxes, text, bev_map):
        if self.program is None:
            self.program = O.build_program(self.ops)
        self.sample_nchw.copy_(sample.to(self.device, F32).reshape(self.B, *sample.shape[-3:]))
        t = torch.as_tensor(timestep).to(self.device, F32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(self.n_scene)
        if t.numel() == self.n_scene:
            t = t.repeat_interleave(self.n_cam)                      # unet_addon_rawbox.py:840-841
        self.temb.t.copy_(t)
        self.cond.load(camera_param, text, bev_map, boxes)
        with torch.cuda.device(self.device):
            self.program.run(_stream(self.device))
        return self.down_out, self.mid_out, self.cond.ctx


class UNetPlan:
    """UNet2DConditionModelMultiview.forward at a fixed geometry (sample (B,4,h,w), ehs (B,S,D), residuals)."""

    def __init__(self, cfg, unet: PackedNet, device, B: int, S: int, latent_hw, with_residuals: bool = True):
        n_cam = len(cfg["neighboring_view_pair"])
        assert B % n_cam == 0, "batch must hold whole scenes: (b n) views (blocks.py:196-197)"
        h, w = latent_hw
        self.cfg, self.device, self.B, self.S, self.h, self.w = cfg, device, B, S, h, w
        self.dtype = unet.dtype
        bld = Builder(cfg, device, B, n_cam, dtype=self.dtype)
        self.bld = bld
        D = cfg["cross_attention_dim"]
        self.sample_nchw = torch.zeros(B, cfg["in_channels"], h, w, dtype=F32, device=device)
        self.x_in = torch.zeros(B, h, w, CIN_PAD, dtype=self.dtype, device=device)
        self.ctx = torch.zeros(B, S, D, dtype=self.dtype, device=device)
        bld.emit(O.Layout(self.sample_nchw, self.x_in[..., :cfg["in_channels"]], True, name="unet.sample.nhwc"))
        self.temb = TembTable(unet, B, device, per_sample=True)
        self.temb.emit_fill(bld, unet, cfg)
        self.kv = build_context_kv(bld, unet, self.ctx, B, S)
        c0 = cfg["block_out_channels"][0]
        u0 = bld.new(B, h, w, c0)
        bld.emit(O.Conv(self.x_in, unet.conv_cin_padded("conv_in.weight", CIN_PAD), u0.bhwc, bias=unet.vec("conv_in.bias"), ws=bld.ws, name="unet.conv_in"))
        mid, skips = bld.encoder(unet, u0, self.temb, self.kv, "unet")
        self.res_in: List[torch.Tensor] = []
        self.mid_in = None
        if with_residuals:
            for k, s in enumerate(skips):
                rin = torch.zeros(s.B, s.C, s.H, s.W, dtype=self.dtype, device=device)
                rn = bld.new(s.B, s.H, s.W, s.C)
                bld.emit(O.Layout(rin, rn.bhwc, True, name=f"res{k}.nhwc"))
                bld.emit(O.Ew(L.EW_ADD, rn.tok, s.tok, name=f"skip{k}+=res"))
                bld.free(rn)
                self.res_in.append(rin)
            self.mid_in = torch.zeros(mid.B, mid.C, mid.H, mid.W, dtype=self.dtype, device=device)
            rn = bld.new(mid.B, mid.H, mid.W, mid.C)
            bld.emit(O.Layout(self.mid_in, rn.bhwc, True, name="midres.nhwc"))
            bld.emit(O.Ew(L.EW_ADD, rn.tok, mid.tok, name="mid+=res"))
            bld.free(rn)
        y = bld.decoder(unet, mid, skips, self.temb, self.kv, "unet")
        self.eps_nhwc = torch.zeros(B, h, w, cfg["out_channels"], dtype=F32, device=device)
        bld.emit(O.Conv(y.bhwc, unet.conv("conv_out.weight"), self.eps_nhwc, bias=unet.vec("conv_out.bias"), direct=True, name="unet.conv_out"))
        self.out_nchw = torch.zeros(B, cfg["out_channels"], h, w, dtype=F32, device=device)
        bld.emit(O.Layout(self.eps_nhwc, self.out_nchw, False, name="eps.nchw"))
        self.ops = bld.ops
        self.program: Optional[L.Program] = None

    def run(self, sample, timestep, ehs, down_res=None, mid_res=None):
        if self.program is None:
            self.program = O.build_program(self.ops)
        self.sample_nchw.copy_(sample.to(self.device, F32))
        t = torch.as_tensor(timestep).to(self.device, F32).reshape(-1)
        self.temb.t.copy_(t.expand(self.B) if t.numel() == 1 else t)
        self.ctx.copy_(ehs.to(self.device, self.dtype))
        if self.res_in:
            assert down_res is not None and len(down_res) == len(self.res_in)
            for dst, src in zip(self.res_in, down_res):
                dst.copy_(src.to(self.device, self.dtype))
            self.mid_in.copy_(mid_res.to(self.device, self.dtype))
        with torch.cuda.device(self.device):
            self.program.run(_stream(self.device))
        return self.out_nchw
