"""One tiny invocation of the hot path on cuda:0, checked against the CPU oracle (driver smoke test)."""
import torch


def run_smoke(verbose: bool = True) -> float:
    from . import _lib, schedulers, synthetic
    from .networks import spec
    from .networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    from .networks.unet_addon_rawbox import BEVControlNetModel
    from .pipeline.pipeline_bev_controlnet import StableDiffusionBEVControlNetPipeline
    from oracle import denoiser as D          # the checker, not the thing run

    _lib.lib()
    cfg = spec.TINY_CONFIG
    dev = torch.device("cuda:0")
    unet = UNet2DConditionModelMultiview.from_config(cfg, seed=0)
    cn = BEVControlNetModel.from_config(cfg, seed=1)
    pipe = StableDiffusionBEVControlNetPipeline(unet=unet, controlnet=cn, scheduler=schedulers.DDIMScheduler()).to(dev)
    sc = synthetic.make_scene_batch(1, ctx_dim=cfg["cross_attention_dim"], max_len=4, latent_hw=(28, 50))
    steps, gs = 3, 2.0
    out = pipe(prompt=None, image=sc["bev_map"], camera_param=sc["camera_param"], height=224, width=400,
               num_inference_steps=steps, guidance_scale=gs, latents=sc["latents"], prompt_embeds=sc["prompt_embeds"],
               negative_prompt_embeds=sc["negative_prompt_embeds"], output_type="latent",
               bev_controlnet_kwargs={"bboxes_3d_data": sc["bboxes_3d_data"]}).images
    torch.cuda.synchronize()
    rnd = lambda sd: {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    with torch.no_grad():
        ref = D.sample_loop(rnd(unet.state_dict()), rnd(cn.state_dict()), cfg, sc["latents"], sc["prompt_embeds"], sc["negative_prompt_embeds"],
                            sc["bev_map"], sc["camera_param"], sc["bboxes_3d_data"], num_steps=steps, guidance_scale=gs)
    err = ((out.float().cpu() - ref).norm() / ref.norm()).item()
    if verbose:
        print(f"[smoke] tiny 6-view {steps}-step CFG sampler on {torch.cuda.get_device_name(0)}: rel L2 vs oracle = {err:.4f}")
    assert torch.isfinite(out).all() and err < 5e-2, f"smoke parity failed: rel L2 {err}"
    return err
