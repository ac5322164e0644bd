"""Architecture spec: config -> parameter names and shapes of the two networks on the hot path,
in the reference's checkpoint layout (`<ckpt>/{unet,controlnet}/diffusion_pytorch_model.*`,
SURVEY.md §5 "Checkpoint / resume"), plus seeded random initialisation for benchmarks/tests
(no pretrained weights exist offline).

Key layout follows the module trees of
  magicdrive/networks/unet_2d_condition_multiview.py:117-234 (UNet2DConditionModelMultiview)
  magicdrive/networks/unet_addon_rawbox.py:46-286 (BEVControlNetModel)
  third_party/diffusers/src/diffusers/models/unet_2d_condition.py:144-505, unet_2d_blocks.py
"""
from __future__ import annotations

import copy
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

SD15_CONFIG = dict(
    in_channels=4, out_channels=4, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, norm_num_groups=32, norm_eps=1e-5,
    cross_attention_dim=768, attention_head_dim=8,
    # multiview additions (configs/model/SDv1.5mv_rawbox.yaml:16-22, configs/dataset/Nuscenes.yaml:27-33)
    neighboring_view_pair={0: [5, 1], 1: [0, 2], 2: [1, 3], 3: [2, 4], 4: [3, 5], 5: [4, 0]},
    neighboring_attn_type="add", zero_module_type="zero_linear",
    # BEV-ControlNet additions (configs/model/SDv1.5mv_rawbox.yaml:26-56)
    controlnet=dict(
        camera_in_dim=189, camera_out_dim=768, map_size=(8, 200, 200),
        conditioning_embedding_out_channels=(16, 32, 96, 256), uncond_cam_in_dim=(3, 7),
        cam_embedder_num_freqs=4,
        bbox=dict(n_classes=10, class_token_dim=768, embedder_num_freq=4, proj_dims=(768, 512, 512, 768), n_corners=8),
    ),
)

# A small architecture with the same topology (4 levels, 2 layers/block, cross-view blocks) for parity tests.
TINY_CONFIG = copy.deepcopy(SD15_CONFIG)
TINY_CONFIG.update(block_out_channels=(32, 64, 64, 64), cross_attention_dim=64, attention_head_dim=2)
TINY_CONFIG["controlnet"].update(camera_out_dim=64, conditioning_embedding_out_channels=(8, 16, 16, 32))
TINY_CONFIG["controlnet"]["bbox"].update(class_token_dim=64, proj_dims=(64, 48, 48, 64))


# AutoencoderKL of SD-1.5 (vae/config.json of runwayml/stable-diffusion-v1-5; dif:models/autoencoder_kl.py:64-112): decoder side only
VAE_SD15_CONFIG = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                       norm_num_groups=32, scaling_factor=0.18215)
VAE_TINY_CONFIG = dict(latent_channels=4, out_channels=3, block_out_channels=(32, 32, 64, 64), layers_per_block=2,
                       norm_num_groups=8, scaling_factor=0.18215)


def vae_decoder_param_shapes(vcfg) -> "OrderedDict[str, Tuple[int, ...]]":
    """post_quant_conv + Decoder of AutoencoderKL (dif:models/autoencoder_kl.py:111, dif:models/vae.py:152-226): conv_in, mid block
    (resnet, single-head attention, resnet), one UpDecoderBlock2D per level (layers_per_block + 1 resnets, nearest x2 + conv except
    the last), GroupNorm + SiLU + conv_out.  Keys and shapes of the reference state dict (encoder / quant_conv are not on the path)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = vcfg["block_out_channels"]; L = vcfg["layers_per_block"]; zc = vcfg["latent_channels"]
    sh["post_quant_conv.weight"] = (zc, zc, 1, 1); sh["post_quant_conv.bias"] = (zc,)
    top = boc[-1]
    sh["decoder.conv_in.weight"] = (top, zc, 3, 3); sh["decoder.conv_in.bias"] = (top,)

    def resnet(pre, cin, cout):
        sh[pre + "norm1.weight"] = (cin,); sh[pre + "norm1.bias"] = (cin,)
        sh[pre + "conv1.weight"] = (cout, cin, 3, 3); sh[pre + "conv1.bias"] = (cout,)
        sh[pre + "norm2.weight"] = (cout,); sh[pre + "norm2.bias"] = (cout,)
        sh[pre + "conv2.weight"] = (cout, cout, 3, 3); sh[pre + "conv2.bias"] = (cout,)
        if cin != cout:
            sh[pre + "conv_shortcut.weight"] = (cout, cin, 1, 1); sh[pre + "conv_shortcut.bias"] = (cout,)
    rev = list(reversed(boc))
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(L + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}.", prev if j == 0 else c, c)
        if i != len(rev) - 1:
            sh[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3); sh[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
        prev = c
    a = "decoder.mid_block.attentions.0."
    sh[a + "group_norm.weight"] = (top,); sh[a + "group_norm.bias"] = (top,)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        sh[a + nm + ".weight"] = (top, top); sh[a + nm + ".bias"] = (top,)
    resnet("decoder.mid_block.resnets.0.", top, top)
    resnet("decoder.mid_block.resnets.1.", top, top)
    sh["decoder.conv_norm_out.weight"] = (boc[0],); sh["decoder.conv_norm_out.bias"] = (boc[0],)
    sh["decoder.conv_out.weight"] = (vcfg["out_channels"], boc[0], 3, 3); sh["decoder.conv_out.bias"] = (vcfg["out_channels"],)
    return sh


def vae_encoder_param_shapes(vcfg) -> "OrderedDict[str, Tuple[int, ...]]":
    """Encoder + quant_conv of AutoencoderKL (dif:models/vae.py:38-150, dif:models/autoencoder_kl.py:95-110): conv_in, one DownEncoderBlock2D
    per level (layers_per_block resnets, a stride-2 conv with (0, 1, 0, 1) zero padding except after the last), mid block (resnet,
    single-head attention, resnet), GroupNorm + SiLU + conv_out to 2 x latent_channels, quant_conv 1x1.  What the given-view demo calls
    through `pipe.vae.encode(...).latent_dist.mean` (demo/run_cond_on_view.py:79-86)."""
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = vcfg["block_out_channels"]; L = vcfg["layers_per_block"]; zc = vcfg["latent_channels"]
    cin = vcfg.get("in_channels", 3)
    sh["encoder.conv_in.weight"] = (boc[0], cin, 3, 3); sh["encoder.conv_in.bias"] = (boc[0],)

    def resnet(pre, ci, co):
        sh[pre + "norm1.weight"] = (ci,); sh[pre + "norm1.bias"] = (ci,)
        sh[pre + "conv1.weight"] = (co, ci, 3, 3); sh[pre + "conv1.bias"] = (co,)
        sh[pre + "norm2.weight"] = (co,); sh[pre + "norm2.bias"] = (co,)
        sh[pre + "conv2.weight"] = (co, co, 3, 3); sh[pre + "conv2.bias"] = (co,)
        if ci != co:
            sh[pre + "conv_shortcut.weight"] = (co, ci, 1, 1); sh[pre + "conv_shortcut.bias"] = (co,)
    prev = boc[0]
    for i, c in enumerate(boc):
        for j in range(L):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}.", prev if j == 0 else c, c)
        if i != len(boc) - 1:
            sh[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3); sh[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
        prev = c
    top = boc[-1]
    a = "encoder.mid_block.attentions.0."
    sh[a + "group_norm.weight"] = (top,); sh[a + "group_norm.bias"] = (top,)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        sh[a + nm + ".weight"] = (top, top); sh[a + nm + ".bias"] = (top,)
    resnet("encoder.mid_block.resnets.0.", top, top)
    resnet("encoder.mid_block.resnets.1.", top, top)
    sh["encoder.conv_norm_out.weight"] = (top,); sh["encoder.conv_norm_out.bias"] = (top,)
    sh["encoder.conv_out.weight"] = (2 * zc, top, 3, 3); sh["encoder.conv_out.bias"] = (2 * zc,)
    sh["quant_conv.weight"] = (2 * zc, 2 * zc, 1, 1); sh["quant_conv.bias"] = (2 * zc,)
    return sh


PLUS_MAP_EMBEDDER = "magicdrive.networks.map_embedder.BEVControlNetConditioningEmbeddingPlus"   # configs/exp/272x736.yaml:18


def map_embedder_plus_size(cfg):
    """None for the default BEVControlNetConditioningEmbedding; (h, w) of the AdaptiveAvgPool2d target when the config selects
    BEVControlNetConditioningEmbeddingPlus (map_embedder_cls / map_embedder_param, unet_addon_rawbox.py:172-184)."""
    cn = cfg["controlnet"]
    cls = cn.get("map_embedder_cls")
    if not cls:
        return None
    if not str(cls).endswith("BEVControlNetConditioningEmbeddingPlus"):
        raise NotImplementedError(f"map_embedder_cls={cls!r}: only the two embedders of magicdrive/networks/map_embedder.py are built")
    return tuple(cn["map_embedder_param"]["conditioning_embedding_size"])


def with_plus_map_embedder(cfg, size):
    """A copy of cfg that selects the ...Plus map encoder for latent size `size` (what configs/exp/272x736.yaml:15-22 does)."""
    c = copy.deepcopy(cfg)
    cn = c["controlnet"]
    cn["map_embedder_cls"] = PLUS_MAP_EMBEDDER
    cn["map_embedder_param"] = dict(conditioning_embedding_size=tuple(size), conditioning_size=tuple(cn["map_size"]),
                                    block_out_channels=tuple(cn["conditioning_embedding_out_channels"]))
    return c


def heads_at(cfg, level: int) -> int:
    h = cfg["attention_head_dim"]       # NUMBER of heads (diffusers 0.17 naming quirk, unet_2d_blocks.py:842-844)
    return h[level] if isinstance(h, (tuple, list)) else h


def _resnet(sh, pre, cin, cout, temb):
    sh[pre + "norm1.weight"] = (cin,); sh[pre + "norm1.bias"] = (cin,)
    sh[pre + "conv1.weight"] = (cout, cin, 3, 3); sh[pre + "conv1.bias"] = (cout,)
    sh[pre + "time_emb_proj.weight"] = (cout, temb); sh[pre + "time_emb_proj.bias"] = (cout,)
    sh[pre + "norm2.weight"] = (cout,); sh[pre + "norm2.bias"] = (cout,)
    sh[pre + "conv2.weight"] = (cout, cout, 3, 3); sh[pre + "conv2.bias"] = (cout,)
    if cin != cout:
        sh[pre + "conv_shortcut.weight"] = (cout, cin, 1, 1); sh[pre + "conv_shortcut.bias"] = (cout,)


def _attn(sh, pre, c, ctx):
    sh[pre + "to_q.weight"] = (c, c); sh[pre + "to_k.weight"] = (c, ctx); sh[pre + "to_v.weight"] = (c, ctx)
    sh[pre + "to_out.0.weight"] = (c, c); sh[pre + "to_out.0.bias"] = (c,)


def _transformer(sh, pre, c, cross, multiview, zero_module_type="zero_linear"):
    sh[pre + "norm.weight"] = (c,); sh[pre + "norm.bias"] = (c,)
    sh[pre + "proj_in.weight"] = (c, c, 1, 1); sh[pre + "proj_in.bias"] = (c,)
    b = pre + "transformer_blocks.0."
    sh[b + "norm1.weight"] = (c,); sh[b + "norm1.bias"] = (c,)
    _attn(sh, b + "attn1.", c, c)
    sh[b + "norm2.weight"] = (c,); sh[b + "norm2.bias"] = (c,)
    _attn(sh, b + "attn2.", c, cross)
    sh[b + "norm3.weight"] = (c,); sh[b + "norm3.bias"] = (c,)
    sh[b + "ff.net.0.proj.weight"] = (8 * c, c); sh[b + "ff.net.0.proj.bias"] = (8 * c,)
    sh[b + "ff.net.2.weight"] = (c, 4 * c); sh[b + "ff.net.2.bias"] = (c,)
    if multiview:
        sh[b + "norm4.weight"] = (c,); sh[b + "norm4.bias"] = (c,)
        _attn(sh, b + "attn4.", c, c)
        # BasicMultiviewTransformerBlock.__init__ (blocks.py:81-90): zero_linear -> nn.Linear, gated -> GatedConnector (one alpha per channel,
        # blocks.py:24-32), none -> identity (no tensors)
        if zero_module_type == "zero_linear":
            sh[b + "connector.weight"] = (c, c); sh[b + "connector.bias"] = (c,)
        elif zero_module_type == "gated":
            sh[b + "connector.alpha"] = (c,)
        elif zero_module_type != "none":
            raise TypeError(f"Unknown zero module type: {zero_module_type}")          # the reference's error (blocks.py:89-90)
    sh[pre + "proj_out.weight"] = (c, c, 1, 1); sh[pre + "proj_out.bias"] = (c,)


def _encoder(sh, cfg, multiview: bool):
    """conv_in, time_embedding, down_blocks, mid_block — shared by the UNet and the ControlNet copy."""
    boc = cfg["block_out_channels"]; L = cfg["layers_per_block"]; cross = cfg["cross_attention_dim"]
    temb = boc[0] * 4
    sh["conv_in.weight"] = (boc[0], cfg["in_channels"], 3, 3); sh["conv_in.bias"] = (boc[0],)
    sh["time_embedding.linear_1.weight"] = (temb, boc[0]); sh["time_embedding.linear_1.bias"] = (temb,)
    sh["time_embedding.linear_2.weight"] = (temb, temb); sh["time_embedding.linear_2.bias"] = (temb,)
    out = boc[0]
    for i, typ in enumerate(cfg["down_block_types"]):
        cin, out = out, boc[i]
        for j in range(L):
            _resnet(sh, f"down_blocks.{i}.resnets.{j}.", cin if j == 0 else out, out, temb)
            if typ.startswith("CrossAttn"):
                _transformer(sh, f"down_blocks.{i}.attentions.{j}.", out, cross, multiview, cfg.get("zero_module_type", "zero_linear"))
        if i != len(boc) - 1:
            sh[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (out, out, 3, 3)
            sh[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (out,)
    c = boc[-1]
    _resnet(sh, "mid_block.resnets.0.", c, c, temb)
    _transformer(sh, "mid_block.attentions.0.", c, cross, multiview, cfg.get("zero_module_type", "zero_linear"))
    _resnet(sh, "mid_block.resnets.1.", c, c, temb)


def unet_param_shapes(cfg) -> "OrderedDict[str, Tuple[int, ...]]":
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    _encoder(sh, cfg, multiview=True)
    boc = cfg["block_out_channels"]; L = cfg["layers_per_block"]; cross = cfg["cross_attention_dim"]
    temb = boc[0] * 4
    rev = list(reversed(boc))
    out = rev[0]
    for i, typ in enumerate(cfg["up_block_types"]):        # unet_2d_condition.py:425-470
        prev, out = out, rev[i]
        inp = rev[min(i + 1, len(boc) - 1)]
        for j in range(L + 1):
            skip = inp if j == L else out
            rin = prev if j == 0 else out
            _resnet(sh, f"up_blocks.{i}.resnets.{j}.", rin + skip, out, temb)
            if typ.startswith("CrossAttn"):
                _transformer(sh, f"up_blocks.{i}.attentions.{j}.", out, cross, True, cfg.get("zero_module_type", "zero_linear"))
        if i != len(boc) - 1:
            sh[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (out, out, 3, 3)
            sh[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (out,)
    sh["conv_norm_out.weight"] = (boc[0],); sh["conv_norm_out.bias"] = (boc[0],)
    sh["conv_out.weight"] = (cfg["out_channels"], boc[0], 3, 3); sh["conv_out.bias"] = (cfg["out_channels"],)
    return sh


def controlnet_param_shapes(cfg) -> "OrderedDict[str, Tuple[int, ...]]":
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    cn = cfg["controlnet"]; bb = cn["bbox"]
    boc = cfg["block_out_channels"]; L = cfg["layers_per_block"]
    sh["cam2token.weight"] = (cn["camera_out_dim"], cn["camera_in_dim"]); sh["cam2token.bias"] = (cn["camera_out_dim"],)
    sh["uncond_cam.weight"] = (1, cn["uncond_cam_in_dim"][0] * cn["uncond_cam_in_dim"][1])
    _encoder(sh, cfg, multiview=False)
    ch = cn["conditioning_embedding_out_channels"]         # map_embedder.py:28-64
    pre = "controlnet_cond_embedding."
    sh[pre + "conv_in.weight"] = (ch[0], cn["map_size"][0], 3, 3); sh[pre + "conv_in.bias"] = (ch[0],)
    k = 0
    for i in range(len(ch) - 2):
        sh[f"{pre}blocks.{k}.weight"] = (ch[i], ch[i], 3, 3); sh[f"{pre}blocks.{k}.bias"] = (ch[i],); k += 1
        sh[f"{pre}blocks.{k}.weight"] = (ch[i + 1], ch[i], 3, 3); sh[f"{pre}blocks.{k}.bias"] = (ch[i + 1],); k += 1
    sh[f"{pre}blocks.{k}.weight"] = (ch[-2], ch[-2], 3, 3); sh[f"{pre}blocks.{k}.bias"] = (ch[-2],); k += 1
    sh[f"{pre}blocks.{k}.weight"] = (ch[-1], ch[-2], 3, 3); sh[f"{pre}blocks.{k}.bias"] = (ch[-1],)
    sh[pre + "conv_out.weight"] = (boc[0], ch[-1], 3, 3); sh[pre + "conv_out.bias"] = (boc[0],)
    # bbox embedder (bbox_embedder.py:60-101)
    fdim = bb["n_corners"] * (3 + 6 * bb["embedder_num_freq"])
    pd = bb["proj_dims"]
    p = "bbox_embedder."
    sh[p + "null_class_feature"] = (bb["class_token_dim"],)
    sh[p + "null_pos_feature"] = (fdim,)
    sh[p + "_class_tokens"] = (bb["n_classes"], bb["class_token_dim"])
    sh[p + "bbox_proj.weight"] = (pd[0], fdim); sh[p + "bbox_proj.bias"] = (pd[0],)
    sh[p + "second_linear.0.weight"] = (pd[1], pd[0] + bb["class_token_dim"]); sh[p + "second_linear.0.bias"] = (pd[1],)
    sh[p + "second_linear.2.weight"] = (pd[2], pd[1]); sh[p + "second_linear.2.bias"] = (pd[2],)
    sh[p + "second_linear.4.weight"] = (pd[3], pd[2]); sh[p + "second_linear.4.bias"] = (pd[3],)
    # zero convs (unet_addon_rawbox.py:221-272)
    k = 0
    sh[f"controlnet_down_blocks.{k}.weight"] = (boc[0], boc[0], 1, 1); sh[f"controlnet_down_blocks.{k}.bias"] = (boc[0],); k += 1
    for i in range(len(boc)):
        for _ in range(L + (0 if i == len(boc) - 1 else 1)):
            sh[f"controlnet_down_blocks.{k}.weight"] = (boc[i], boc[i], 1, 1); sh[f"controlnet_down_blocks.{k}.bias"] = (boc[i],); k += 1
    sh["controlnet_mid_block.weight"] = (boc[-1], boc[-1], 1, 1); sh["controlnet_mid_block.bias"] = (boc[-1],)
    return sh


# tensors the reference zero-initialises: random weights must re-randomise them or
# the MagicDrive-specific paths are dead in any parity test
ZERO_INIT_MARKERS = ("controlnet_down_blocks.", "controlnet_mid_block.", "controlnet_cond_embedding.conv_out.",
                     "bbox_embedder.null_class_feature", "bbox_embedder.null_pos_feature", ".connector.")
# of those, the cross-view connector gets a full-scale init so the neighbour attention matters as much
# as the other attentions in a random-weight parity test (the rest: N(0, 0.02^2))
_FULL_SCALE_ZERO_INIT = (".connector.weight", "controlnet_cond_embedding.conv_out.weight")


# init gains (measured with tools/path_sensitivity.py: bf16-weight error ~1 % of eps, every conditioning path alive)
_G = dict(all=3.0 ** 0.5, qk=1.0, map=1.8, conn=1.0)


def random_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Seeded init, one CPU generator per tensor (order independent, reproducible across machines):
    conv/linear weights U(+-1/sqrt(fan_in)); norm gains 1 + 0.1 N; biases 0.05 N; zero-init families N(0, 0.02^2);
    class tokens / uncond camera N(0,1) like the reference's randn / nn.Embedding init."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        if any(m in name for m in ZERO_INIT_MARKERS) and not any(name.endswith(m) for m in _FULL_SCALE_ZERO_INIT) and not name.endswith(".connector.alpha"):
            t = torch.randn(shape, generator=g) * 0.02
        elif name.endswith(".connector.alpha"):
            # GatedConnector: tanh(alpha) scales the cross-view branch; the reference's zero init would switch the branch off in a parity test
            t = (torch.rand(shape, generator=g) * 2 - 1) * 1.5
        elif name.endswith("_class_tokens") or name.startswith("uncond_cam"):
            t = torch.randn(shape, generator=g)
        elif ".norm" in name or "conv_norm_out." in name or name.endswith("norm.weight") or name.endswith("norm.bias"):
            t = (1.0 + 0.1 * torch.randn(shape, generator=g)) if name.endswith("weight") else 0.05 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            bound = _G['all'] * fan_in ** -0.5
            # Gains that keep every conditioning path of a RANDOM-weight model well above bf16 noise
            # (a trained model has them; PyTorch's default init does not — see DESIGN.md "fixtures"):
            if name.endswith(("to_q.weight", "to_k.weight")):
                bound *= _G['qk']                   # peaky (non-uniform) attention maps
            elif name.startswith("controlnet_cond_embedding."):
                bound *= _G['map']                  # 8 conv+SiLU layers otherwise attenuate the BEV map to nothing
            # raw intrinsics (~1.3e3) and box coordinates (~50 m) enter these two layers unscaled
            # (SURVEY.md §8a a11); a trained layer maps them to O(1) tokens, so the random init must too,
            # otherwise the context tokens saturate every softmax and the net is chaotic in its inputs.
            if name.endswith(".connector.weight"):
                bound *= _G['conn']
            if name == "cam2token.weight":
                bound *= 7e-3
            elif name == "bbox_embedder.bbox_proj.weight":
                bound *= 5e-2
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        sd[name] = t.to(dtype)
    return sd
