"""TEST INFRASTRUCTURE (authoring container only): build the REAL reference modules from
/root/reference with a given config and load a reference-layout state dict into them."""
from __future__ import annotations

import torch

from . import refshim


def build_reference(cfg, unet_sd, cn_sd, img_size=(224, 400)):
    ns = refshim.load()
    base = ns.UNet2DConditionModel(
        sample_size=64, in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
        block_out_channels=cfg["block_out_channels"], layers_per_block=cfg["layers_per_block"],
        cross_attention_dim=cfg["cross_attention_dim"], attention_head_dim=cfg["attention_head_dim"],
        norm_num_groups=cfg["norm_num_groups"])
    unet = ns.unet_mv.UNet2DConditionModelMultiview.from_unet_2d_condition(
        base, neighboring_view_pair=cfg["neighboring_view_pair"], neighboring_attn_type=cfg.get("neighboring_attn_type", "add"),
        zero_module_type=cfg.get("zero_module_type", "zero_linear"), img_size=list(img_size))
    cn = cfg["controlnet"]; bb = cn["bbox"]
    extra = {}
    if cn.get("map_embedder_cls"):          # configs/exp/272x736.yaml:15-22
        mp = cn["map_embedder_param"]
        extra = dict(map_embedder_cls=cn["map_embedder_cls"],
                     map_embedder_param=dict(conditioning_embedding_size=list(mp["conditioning_embedding_size"]),
                                             conditioning_size=list(mp["conditioning_size"]), block_out_channels=list(mp["block_out_channels"])))
    cnet = ns.controlnet.BEVControlNetModel.from_unet(
        base, **extra, camera_in_dim=cn["camera_in_dim"], camera_out_dim=cn["camera_out_dim"], map_size=list(cn["map_size"]),
        conditioning_embedding_out_channels=cn["conditioning_embedding_out_channels"],
        uncond_cam_in_dim=cn["uncond_cam_in_dim"], use_uncond_map=None, drop_cond_ratio=0.25, drop_cam_num=6,
        drop_cam_with_box=False,
        cam_embedder_param=dict(input_dims=3, num_freqs=cn["cam_embedder_num_freqs"], include_input=True, log_sampling=True),
        bbox_embedder_cls="magicdrive.networks.bbox_embedder.ContinuousBBoxWithTextEmbedding",
        bbox_embedder_param=dict(n_classes=bb["n_classes"], class_token_dim=bb["class_token_dim"], trainable_class_token=False,
                                 use_text_encoder_init=False, embedder_num_freq=bb["embedder_num_freq"],
                                 proj_dims=list(bb["proj_dims"]), mode=bb.get("mode", "all-xyz"),
                                 minmax_normalize=bool(bb.get("minmax_normalize", False))))
    unet.load_state_dict(unet_sd, strict=True)
    cnet.load_state_dict(cn_sd, strict=True)
    return ns, unet.eval(), cnet.eval()


def build_reference_pipeline(cfg, unet_sd, cn_sd, scheduler="ddim", given_view=False, img_size=None):
    """The reference StableDiffusionBEVControlNetPipeline with a 1-layer random CLIP (only .dtype is read when
    prompt_embeds are given, pipeline_controlnet.py:371) and a generator-free DDIM subclass."""
    ns, unet, cnet = build_reference(cfg, unet_sd, cn_sd, **({} if img_size is None else {"img_size": img_size}))
    import transformers

    class DDIMNoGen(ns.DDIMScheduler):
        def step(self, model_output, timestep, sample, eta: float = 0.0, **kw):
            return super().step(model_output, timestep, sample, eta=eta)

    sch_cfg = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000)
    if scheduler == "ddim":
        sch = DDIMNoGen(clip_sample=False, set_alpha_to_one=False, steps_offset=1, **sch_cfg)
    else:
        sch = ns.UniPCMultistepScheduler(**sch_cfg)
    tcfg = transformers.CLIPTextConfig(hidden_size=cfg["cross_attention_dim"], intermediate_size=64, num_hidden_layers=1,
                                       num_attention_heads=2, vocab_size=100, max_position_embeddings=77)
    te = transformers.CLIPTextModel(tcfg).eval()
    vae = ns.diffusers.AutoencoderKL(in_channels=3, out_channels=3, block_out_channels=(32, 32, 32, 32),
                                     down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
                                     latent_channels=4, norm_num_groups=8)
    pipe_cls = ns.pipeline.StableDiffusionBEVControlNetPipeline
    if given_view:
        import importlib
        pipe_cls = importlib.import_module("magicdrive.pipeline.pipeline_bev_controlnet_given_view").StableDiffusionBEVControlNetGivenViewPipeline
    pipe = pipe_cls(vae=vae, text_encoder=te, unet=unet, controlnet=cnet, scheduler=sch, tokenizer=None)
    pipe.set_progress_bar_config(disable=True)
    return ns, pipe
