"""VAE decode — and, round 4, VAE encode (what the given-view demo calls on its known views) — as op programs on
the same HIP kernels as the sampler.

AutoencoderKL.decode = post_quant_conv (1x1) -> Decoder (dif:models/vae.py:152-275): conv_in, mid block (resnet, single-head
512-channel attention, resnet), four UpDecoderBlock2D (3 resnets + nearest x2 + conv), GroupNorm + SiLU + conv_out.
Everything is GroupNorm(+SiLU) / 3x3 conv / 1x1 GEMM / nearest upsample — kernels the UNet path already has — except the mid-block
attention: one head of dim C = 512 is outside the fused attention kernel's head-dim range, so it is spelled out on the GEMM kernel:
S = Q K^T (fp32 out), P = softmax(S / sqrt(C)) (mdx_softmax_rows), O = P V with V^T from the "W_v as A" GEMM; to_v's bias is folded
through the softmax (rows of P sum to 1) into to_out's bias.  One plan decodes `n_img` latents of a fixed size (the pipeline runs
it once per scene: 6 views), channels-last bf16 activations like the rest of the library.
"""
from __future__ import annotations

from typing import List

import torch

from . import _lib as L
from . import ops as O
from . import packing as PK
from .engine import PackedNet, Pool

BF16, F32 = torch.bfloat16, torch.float32
ZPAD = 8      # latent channels padded 4 -> 8 so conv_in runs on the MFMA path


def _vae_attention(plan, net, x, a, G, eps, buf, emit):
    return VaeDecodePlan._attention(plan, net, x, a, G, eps, buf, emit)


class VaeEncodePlan:
    """AutoencoderKL.encode up to the moments (dif:models/autoencoder_kl.py:127-171: `quant_conv(encoder(x))`): conv_in, four
    DownEncoderBlock2D (2 resnets + a stride-2 3x3 conv on the input zero-padded by one row / column at the bottom / right,
    resnet.py:215-217), the mid block (resnet, single-head attention, resnet), GroupNorm + SiLU + conv_out (2 x latent channels) and the
    1x1 quant_conv — the caller of `pipe.vae.encode(pixel_values).latent_dist.mean` in demo/run_cond_on_view.py:79-86.  Same kernels and
    the same mid-block attention as the decoder; the image's 3 channels are zero-padded to 8 so conv_in runs on the MFMA path; the last two
    layers (Cout = 8) run on the direct kernels with fp32 outputs, so the moments never pass through a 16-bit store."""

    def __init__(self, vcfg, net: PackedNet, device, n_img: int, image_hw):
        self.vcfg, self.device, self.n = vcfg, device, n_img
        H, W = image_hw
        G, eps = vcfg["norm_num_groups"], 1e-6
        zc = vcfg["latent_channels"]
        cin = vcfg.get("in_channels", 3)
        self.ops: List[object] = []
        self.keep: List[torch.Tensor] = []
        self.ws = torch.empty(64 * 1024 * 1024 // 4, dtype=F32, device=device)
        emit = self.ops.append
        H16 = net.dtype

        def buf(*shape, dtype=H16, zero=False):
            n_ = 1
            for d_ in shape:
                n_ *= int(d_)
            t = Pool.alloc(n_, dtype, device).view(*shape)          # engine.Pool.guard: every buffer closes its own device segment
            if zero:
                t.zero_()
            self.keep.append(t)
            return t

        def gn(x, pre, silu):
            y = buf(*x.shape)
            n_, h_, w_, C = x.shape
            emit(O.GroupNorm(x.view(n_, h_ * w_, C), y.view(n_, h_ * w_, C), net.vec(pre + "weight"), net.vec(pre + "bias"), G, eps, silu, ws=self.ws, name="vae." + pre))
            return y

        def conv3(x, key, R=None):
            wt = net.conv(key + "weight")
            y = buf(x.shape[0], x.shape[1], x.shape[2], wt.shape[0])
            emit(O.Conv(x, wt, y, bias=net.vec(key + "bias"), R=R, ws=self.ws, name="vae." + key))
            return y

        def resnet(x, pre):
            a = gn(x, pre + "norm1.", True)
            hcv = conv3(a, pre + "conv1.")
            b = gn(hcv, pre + "norm2.", True)
            sc = x
            if net.has(pre + "conv_shortcut.weight"):
                cout = net.sd[pre + "conv_shortcut.weight"].shape[0]
                sc = buf(x.shape[0], x.shape[1], x.shape[2], cout)
                emit(O.Gemm(x.view(-1, x.shape[3]), net.lin(pre + "conv_shortcut.weight"), sc.view(-1, cout), bias=net.vec(pre + "conv_shortcut.bias"),
                            ws=self.ws, name="vae." + pre + "shortcut"))
            return conv3(b, pre + "conv2.", R=sc)

        # ---- input: image fp32 NCHW in [-1, 1] -> 16-bit NHWC, channels zero-padded 3 -> ZPAD
        self.x_in = torch.zeros(n_img, cin, H, W, dtype=F32, device=device)
        xp = buf(n_img, H, W, ZPAD, zero=True)
        emit(O.Layout(self.x_in, xp[..., :cin], True, name="vae.x.nhwc"))
        c0 = net.sd["encoder.conv_in.weight"].shape[0]
        x = buf(n_img, H, W, c0)
        emit(O.Conv(xp, net.conv_cin_padded("encoder.conv_in.weight", ZPAD), x, bias=net.vec("encoder.conv_in.bias"), ws=self.ws, name="vae.enc.conv_in"))
        nlev = len(vcfg["block_out_channels"])
        for i in range(nlev):
            for j in range(vcfg["layers_per_block"]):
                x = resnet(x, f"encoder.down_blocks.{i}.resnets.{j}.")
            if i != nlev - 1:
                key = f"encoder.down_blocks.{i}.downsamplers.0.conv."
                Ho, Wo = (x.shape[1] + 1 - 3) // 2 + 1, (x.shape[2] + 1 - 3) // 2 + 1
                y = buf(n_img, Ho, Wo, x.shape[3])
                emit(O.Conv(x, net.conv(key + "weight"), y, bias=net.vec(key + "bias"), stride=(2, 2), pad=(0, 0), pad_end=(1, 1), ws=self.ws, name="vae." + key))
                x = y
        x = resnet(x, "encoder.mid_block.resnets.0.")
        x = _vae_attention(self, net, x, "encoder.mid_block.attentions.0.", G, eps, buf, emit)
        x = resnet(x, "encoder.mid_block.resnets.1.")
        x = gn(x, "encoder.conv_norm_out.", True)
        m1 = buf(n_img, x.shape[1], x.shape[2], 2 * zc, dtype=F32)
        emit(O.Conv(x, net.conv("encoder.conv_out.weight"), m1, bias=net.vec("encoder.conv_out.bias"), direct=True, name="vae.enc.conv_out"))
        self.moments_nhwc = buf(n_img, x.shape[1], x.shape[2], 2 * zc, dtype=F32)
        emit(O.Conv(m1, net.conv("quant_conv.weight"), self.moments_nhwc, bias=net.vec("quant_conv.bias"), stride=(1, 1), pad=(0, 0), direct=True, name="vae.quant_conv"))
        self.program = None

    def compile(self):
        self.program = O.build_program(self.ops)

    def run(self, x: torch.Tensor) -> torch.Tensor:
        """x (n_img, 3, H, W) in [-1, 1] -> moments (n_img, 2 * latent_channels, H / 8, W / 8) fp32 = [mean | logvar]."""
        if self.program is None:
            self.compile()
        with torch.cuda.device(self.device):
            self.x_in.copy_(x.to(self.device, F32))
            self.program.run(torch.cuda.current_stream(self.device).cuda_stream)
            return self.moments_nhwc.permute(0, 3, 1, 2).contiguous()

    def release(self):
        if self.program is not None:
            self.program.destroy()
        self.program = None
        self.ops, self.keep = [], []


class VaeDecodePlan:
    def __init__(self, vcfg, net: PackedNet, device, n_img: int, latent_hw):
        self.vcfg, self.device, self.n = vcfg, device, n_img
        h, w = latent_hw
        G, eps = vcfg["norm_num_groups"], 1e-6
        zc = vcfg["latent_channels"]
        self.ops: List[object] = []
        self.keep: List[torch.Tensor] = []
        self.ws = torch.empty(64 * 1024 * 1024 // 4, dtype=F32, device=device)
        emit = self.ops.append

        H16 = net.dtype                     # bf16 or fp16: the type the decoder's weights were packed in

        def buf(*shape, dtype=H16, zero=False):
            n_ = 1
            for d_ in shape:
                n_ *= int(d_)
            t = Pool.alloc(n_, dtype, device).view(*shape)          # engine.Pool.guard: every buffer closes its own device segment
            if zero:
                t.zero_()
            self.keep.append(t)
            return t

        def gn(x, pre, silu):                  # x [n, H, W, C]
            y = buf(*x.shape)
            n_, H, W, C = x.shape
            emit(O.GroupNorm(x.view(n_, H * W, C), y.view(n_, H * W, C), net.vec(pre + "weight"), net.vec(pre + "bias"), G, eps, silu, ws=self.ws, name="vae." + pre))
            return y

        def conv3(x, key, R=None, name=""):
            wt = net.conv(key + "weight")
            y = buf(x.shape[0], x.shape[1], x.shape[2], wt.shape[0])
            emit(O.Conv(x, wt, y, bias=net.vec(key + "bias"), R=R, ws=self.ws, name="vae." + key))
            return y

        def resnet(x, pre):
            """ResnetBlock2D without temb (dif:models/resnet.py:590-640)."""
            a = gn(x, pre + "norm1.", True)
            hcv = conv3(a, pre + "conv1.")
            b = gn(hcv, pre + "norm2.", True)
            sc = x
            if net.has(pre + "conv_shortcut.weight"):
                cout = net.sd[pre + "conv_shortcut.weight"].shape[0]
                sc = buf(x.shape[0], x.shape[1], x.shape[2], cout)
                emit(O.Gemm(x.view(-1, x.shape[3]), net.lin(pre + "conv_shortcut.weight"), sc.view(-1, cout), bias=net.vec(pre + "conv_shortcut.bias"),
                            ws=self.ws, name="vae." + pre + "shortcut"))
            return conv3(b, pre + "conv2.", R=sc)

        # ---- input: z (already / scaling_factor) fp32 NCHW -> post_quant_conv -> bf16 NHWC padded to ZPAD channels
        self.z_in = torch.zeros(n_img, zc, h, w, dtype=F32, device=device)
        z_nhwc = buf(n_img, h, w, zc, dtype=F32)
        emit(O.Layout(self.z_in, z_nhwc, True, name="vae.z.nhwc"))
        zq = buf(n_img, h, w, ZPAD, zero=True)
        emit(O.Conv(z_nhwc, net.conv("post_quant_conv.weight"), zq[..., :zc], bias=net.vec("post_quant_conv.bias"), stride=(1, 1), pad=(0, 0),
                    direct=True, name="vae.post_quant_conv"))
        top = net.sd["decoder.conv_in.weight"].shape[0]
        x = buf(n_img, h, w, top)
        emit(O.Conv(zq, net.conv_cin_padded("decoder.conv_in.weight", ZPAD), x, bias=net.vec("decoder.conv_in.bias"), ws=self.ws, name="vae.conv_in"))
        # ---- mid block
        x = resnet(x, "decoder.mid_block.resnets.0.")
        x = self._attention(net, x, "decoder.mid_block.attentions.0.", G, eps, buf, emit)
        x = resnet(x, "decoder.mid_block.resnets.1.")
        # ---- up blocks
        nlev = len(vcfg["block_out_channels"])
        for i in range(nlev):
            for j in range(vcfg["layers_per_block"] + 1):
                x = resnet(x, f"decoder.up_blocks.{i}.resnets.{j}.")
            if i != nlev - 1:
                H2, W2 = 2 * x.shape[1], 2 * x.shape[2]
                up = buf(n_img, H2, W2, x.shape[3])
                emit(O.Upsample(x, up, PK.nearest_index(x.shape[1], H2).to(device), PK.nearest_index(x.shape[2], W2).to(device), name=f"vae.up{i}.nearest"))
                x = conv3(up, f"decoder.up_blocks.{i}.upsamplers.0.conv.")
        x = gn(x, "decoder.conv_norm_out.", True)
        oc = vcfg["out_channels"]
        self.out_nhwc = buf(n_img, x.shape[1], x.shape[2], oc, dtype=F32)
        emit(O.Conv(x, net.conv("decoder.conv_out.weight"), self.out_nhwc, bias=net.vec("decoder.conv_out.bias"), direct=True, name="vae.conv_out"))
        self.program = None

    def _attention(self, net, x, a, G, eps, buf, emit):
        """Attention(heads=1, dim_head=C, group_norm, bias, residual_connection) with the vanilla processor
        (attention_processor.py:495-558; unet_2d_blocks.py:433-445)."""
        n, H, W, C = x.shape
        T = H * W
        Tp = PK.round_up(T, 8)
        t = buf(n, H, W, C)
        emit(O.GroupNorm(x.view(n, T, C), t.view(n, T, C), net.vec(a + "group_norm.weight"), net.vec(a + "group_norm.bias"), G, eps, False, ws=self.ws, name="vae.attn.gn"))
        qk = buf(n * T, 2 * C)
        emit(O.Gemm(t.view(n * T, C), net.cat_lin([a + "to_q.weight", a + "to_k.weight"]), qk, bias=net.cat_vec([a + "to_q.bias", a + "to_k.bias"]),
                    ws=self.ws, name="vae.attn.qk"))
        vt = buf(n, C, Tp, zero=True)                                            # V^T (bias folded into the output projection)
        emit(O.Gemm(net.lin(a + "to_v.weight"), t.view(n, T, C), vt[:, :, :T], name="vae.attn.vT"))
        qk3 = qk.view(n, T, 2 * C)
        S = buf(n, T, Tp, dtype=F32)
        emit(O.Gemm(qk3[:, :, :C], qk3[:, :, C:], S[:, :, :T], name="vae.attn.scores"))       # [n][T][T] fp32
        P = buf(n * T, Tp)
        emit(O.Softmax(S.view(n * T, Tp), P, T, scale=float(C) ** -0.5, name="vae.attn.softmax"))
        o = buf(n, T, C)
        emit(O.Gemm(P.view(n, T, Tp), vt, o, name="vae.attn.pv"))
        # to_out(o + b_v) + b_o = W_o o + (W_o b_v + b_o); residual add of the block input
        wo = net.lin(a + "to_out.0.weight")
        bo = net._get("vae_attn_bias", [a + "to_out.0.weight", a + "to_out.0.bias", a + "to_v.bias"], lambda w, b, bv: (w @ bv + b).contiguous().to(F32))
        y = buf(n, H, W, C)
        emit(O.Gemm(o.view(n * T, C), wo, y.view(n * T, C), bias=bo, R=x.view(n * T, C), ws=self.ws, name="vae.attn.out"))
        return y

    def compile(self):
        self.program = O.build_program(self.ops)

    def run(self, z: torch.Tensor) -> torch.Tensor:
        """z (n_img, 4, h, w), already divided by the scaling factor -> images (n_img, 3, 8h, 8w) fp32, un-clamped."""
        if self.program is None:
            self.compile()
        with torch.cuda.device(self.device):
            self.z_in.copy_(z.to(self.device, F32))
            self.program.run(torch.cuda.current_stream(self.device).cuda_stream)
            return self.out_nhwc.permute(0, 3, 1, 2).contiguous()

    def release(self):
        if self.program is not None:
            self.program.destroy()
        self.program = None
        self.ops, self.keep = [], []
