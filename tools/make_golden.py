#!/usr/bin/env python
"""Generate tests/golden/*.pt by running the REAL reference (imported from /root/reference through
oracle/refshim.py) on seeded inputs.  Runs only in the authoring container; the fixtures travel.

Weights are NOT stored: they are regenerated anywhere from seeds by
magicdrive_amd.networks.spec.random_state_dict (CPU generator, order independent); each fixture records a
checksum of the state dicts it was produced with so a drifted init is detected, not silently compared.

Fixtures
  tiny_forward.pt   BEVControlNetModel.forward + UNet2DConditionModelMultiview.forward (reference modules),
                    2 scenes x 6 views with DISTINCT noise per view, 5 boxes/view, two different timesteps.
  tiny_pipeline.pt  StableDiffusionBEVControlNetPipeline.__call__ (reference pipeline, generator-free DDIM subclass,
                    SURVEY.md §0.2), 5 steps, guidance 2.0, output_type="latent"; and the camera_param=None path.
  tiny_pipeline_unipc.pt  the same call with the scheduler tools/test.py really installs (UniPCMultistepScheduler,
                    magicdrive/misc/test_utils.py:129), 6 steps (warm-up, order-2 and lower-order-final steps all occur).

  tiny_forward_hires.pt  BASELINE.json configs[3] shape: 432x768 images = 54x96 latents with the
                    BEVControlNetConditioningEmbeddingPlus map encoder (configs/exp/272x736.yaml:15-22 with size [54, 96]);
                    reference BEVControlNetModel.forward + UNet2DConditionModelMultiview.forward, 1 scene x 6 views, 3 boxes/view.

  tiny_pipeline_given_view.pt  StableDiffusionBEVControlNetGivenViewPipeline.__call__ (reference), 5 DDIM steps, guidance 2.0, views 0 and 3
                    of scene 0 and view 5 of scene 1 given; both `conditional_latents_change_every_input` modes.

  tiny_vae_decode.pt  diffusers AutoencoderKL.decode (the reference's `vae`, pipeline_bev_controlnet.py:100-112) of a tiny decoder config
                    (spec.VAE_TINY_CONFIG, seeded weights) on 2 latents of 7x13.

`python tools/make_golden.py unipc` / `... hires` / `... given` / `... vae` / `... nattn` / `... zmod` regenerate only that fixture.

Round 4:
  tiny_pipeline_given_view_unipc.pt `... givenunipc`  the given-view pipeline with UniPC — the scheduler demo/run_cond_on_view.py's
                    build_pipe installs — 6 steps, both re-noising modes.
  tiny_forward_cxyz.pt `... cxyz`   bbox_embedder mode='cxyz' (the reference class default: 4 points per box), module forwards.
  sd15_loop_given_view.pt `... sd15given`  SD-1.5 size, REAL reference given-view pipeline, camera + 32 boxes + map, CFG 2.0, 10 DDIM
                    steps, views 0 and 3 given: the headline-size loop whose views genuinely differ.
  tiny_vae_encode.pt `... vaeenc`   diffusers AutoencoderKL.encode (mean / logvar / one sample) of the tiny VAE on two 56x104 images.
  `... cpuref` writes profiles/r04_cpu_reference_vs_port.json (reference vs CPU port seconds per denoise step, same threads).

SD-1.5-SIZE fixtures (round 3; the REAL reference at spec.SD15_CONFIG, fp32 arithmetic on the bf16-rounded seeded weights — the
weights the HIP model holds — so that the fixture measures arithmetic, not weight rounding; 8 host threads, minutes each):
  sd15_loop50.pt    `... sd15`      BASELINE configs[1]: reference pipeline __call__ (pipeline_bev_controlnet.py:349-451), 1 scene,
                    text-only (camera_param=None, zero map, no boxes), 50 DDIM steps; final latents + the latents after every 10th step (fp16).
  sd15_loop_cfg.pt  `... sd15cfg`   BASELINE configs[2]: camera + 32 padded boxes per view + BEV map, guidance 2.0, 10 DDIM steps,
                    latents after every 2nd step (fp16).
  sd15_forward_hires.pt `... sd15hires`  BASELINE configs[3] shape at REAL width: 54x96 latents, ...Plus map encoder, one
                    BEVControlNetModel.forward + UNet forward of 1 scene x 6 views, 3 boxes per view (eps fp16).
  tiny_forward_272x736.pt `... res272`   configs/exp/272x736.yaml:15-22: 34x92 latents, ...Plus [34, 92] (tiny width).
  tiny_forward_424x800.pt `... res424`   configs/exp/424x800abox0.1_nockpt.yaml:15-17: 53x100 latents, map_size [8, 400, 400] through the
                    plain BEVControlNetConditioningEmbedding (tiny width).
  tiny_forward_nattn.pt  the reference UNet forward with neighboring_attn_type = concat and = self (same tiny weights and inputs).
  tiny_forward_zmod.pt   the reference UNet forward with zero_module_type = gated (GatedConnector) and = none (identity connector).
"""
import copy
import time
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import scene, state_dicts  # noqa: E402
from magicdrive_amd.networks import spec  # noqa: E402
from oracle import ref_models  # noqa: E402


def checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def unipc_fixture(out_dir, cfg, usd, csd, meta, hw=(28, 50)):
    ns, pipe = ref_models.build_reference_pipeline(cfg, usd, csd, scheduler="unipc")
    sc = scene(cfg, 2, 5, hw)
    with torch.no_grad():
        out = pipe(prompt=None, image=sc["bev_map"], camera_param=sc["camera_param"], height=224, width=400, num_inference_steps=6,
                   guidance_scale=2.0, latents=sc["latents"].clone(), prompt_embeds=sc["prompt_embeds"],
                   negative_prompt_embeds=sc["negative_prompt_embeds"], output_type="latent",
                   bev_controlnet_kwargs={"bboxes_3d_data": sc["bboxes_3d_data"]}).images
    torch.save({"meta": meta, "steps": 6, "guidance": 2.0, "latents_cfg": out.clone()}, os.path.join(out_dir, "tiny_pipeline_unipc.pt"))
    print("tiny_pipeline_unipc: |x|", out.abs().mean().item())


def given_view_inputs(hw=(28, 50)):
    """Seeded clean latents of the known views: scene 0 views 0 and 3, scene 1 view 5."""
    g = torch.Generator().manual_seed(77)
    cl = [[None] * 6 for _ in range(2)]
    for (i, j) in ((0, 0), (0, 3), (1, 5)):
        cl[i][j] = torch.randn(4, *hw, generator=g) * 0.8
    return cl


def given_view_fixture(out_dir, cfg, usd, csd, meta, hw=(28, 50), scheduler="ddim"):
    """scheduler="unipc": what demo/run_cond_on_view.py really runs (build_pipe installs UniPC, misc/test_utils.py:129), 6 steps so that
    the warm-up, order-2 and lower-order-final updates all occur beside the per-step re-noising."""
    ns, pipe = ref_models.build_reference_pipeline(cfg, usd, csd, given_view=True, scheduler=scheduler)
    sc = scene(cfg, 2, 5, hw)
    steps = 5 if scheduler == "ddim" else 6
    outs = {}
    with torch.no_grad():
        for every in (True, False):
            outs[every] = pipe(prompt=None, image=sc["bev_map"], camera_param=sc["camera_param"], height=224, width=400,
                               conditional_latents=given_view_inputs(hw), conditional_latents_change_every_input=every,
                               num_inference_steps=steps, guidance_scale=2.0, latents=sc["latents"].clone(), prompt_embeds=sc["prompt_embeds"],
                               negative_prompt_embeds=sc["negative_prompt_embeds"], output_type="latent",
                               bev_controlnet_kwargs={"bboxes_3d_data": sc["bboxes_3d_data"]}).images
    name = "tiny_pipeline_given_view.pt" if scheduler == "ddim" else "tiny_pipeline_given_view_unipc.pt"
    torch.save({"meta": meta, "steps": steps, "guidance": 2.0, "scheduler": scheduler, "latents_every": outs[True].clone(),
                "latents_once": outs[False].clone()}, os.path.join(out_dir, name))
    print(name, "|x|", outs[True].abs().mean().item(), outs[False].abs().mean().item())


def cxyz_fixture(out_dir, cfg0, usd, meta, hw=(28, 50)):
    """bbox_embedder mode='cxyz' — the reference CLASS default (bbox_embedder.py:41, :52-54): 4 points per box instead of 8 corners, so
    bbox_proj is Linear(4 * 27, .) and null_pos_feature has 108 entries.  Reference BEVControlNetModel.forward + UNet forward, 1 scene
    x 6 views, 5 boxes per view (the first four corners of the synthetic boxes stand in for the four points), minmax_normalize on (the
    class default too) with metre-scale coordinates."""
    cfg = copy.deepcopy(cfg0)
    cfg["controlnet"]["bbox"].update(mode="cxyz", n_corners=4, minmax_normalize=True)
    csd = spec.random_state_dict(spec.controlnet_param_shapes(cfg), 1)
    meta = dict(meta); meta["cn_checksum"] = checksum(csd)
    ns, unet, cnet = ref_models.build_reference(cfg, usd, csd)
    sc = scene(cfg, 1, 5, hw)
    boxes = dict(sc["bboxes_3d_data"]); boxes["bboxes"] = boxes["bboxes"][..., :4, :].contiguous() * 20.0      # metres
    lat = torch.randn(1, 6, 4, *hw, generator=torch.Generator().manual_seed(19))
    t = torch.tensor([620])
    with torch.no_grad():
        d, m, ctx = cnet(lat, t, sc["camera_param"], boxes, sc["prompt_embeds"], sc["bev_map"], return_dict=False)
        e = unet(lat.reshape(-1, 4, *hw), t.repeat_interleave(6), encoder_hidden_states=ctx,
                 down_block_additional_residuals=d, mid_block_additional_residual=m).sample
    torch.save({"meta": meta, "lat_seed": 19, "timesteps": t, "ctx": ctx.half(), "mid": m.clone(), "eps": e.half(),
                "down_absmean": torch.tensor([x.abs().mean() for x in d])}, os.path.join(out_dir, "tiny_forward_cxyz.pt"))
    print("tiny_forward_cxyz: eps std", e.std().item(), "ctx box rows |x|", ctx[:, 78:].abs().mean().item())


def sd15_given_view_inputs(hw=(28, 50)):
    """Known views of sd15_loop_given_view.pt: views 0 and 3 of the one scene (seed 78)."""
    g = torch.Generator().manual_seed(78)
    cl = [[None] * 6]
    for j in (0, 3):
        cl[0][j] = torch.randn(4, *hw, generator=g) * 0.8
    return cl


def sd15_given_view_fixture(out_dir):
    """A headline-size loop whose six views GENUINELY differ: the REAL reference given-view pipeline
    (pipeline_bev_controlnet_given_view.py:263-296) at spec.SD15_CONFIG, camera + 32 boxes + map, guidance 2.0, 10 DDIM steps, views 0
    and 3 given and re-noised every step; latents after every 2nd step.  The reference __call__ stacks ONE latent over the cameras
    (pipeline_bev_controlnet.py:326), so only the given views make the per-view states differ by O(1) instead of by the conditioning's ~1 %."""
    import time
    from helpers import bf16_round
    torch.set_num_threads(8)
    cfg = spec.SD15_CONFIG
    usd, csd = state_dicts(cfg)
    usd, csd = bf16_round(usd), bf16_round(csd)
    meta = {"unet_checksum": checksum(usd), "cn_checksum": checksum(csd), "torch": str(torch.__version__),
            "weights": "spec.random_state_dict seeds (0, 1), bf16-rounded; reference arithmetic fp32"}
    ns, pipe = ref_models.build_reference_pipeline(cfg, usd, csd, given_view=True)
    sc = scene(cfg, 1, 32, (28, 50))
    trace = {}
    t0 = time.time()
    with torch.no_grad():
        out = pipe(prompt=None, image=sc["bev_map"], camera_param=sc["camera_param"], height=224, width=400,
                   conditional_latents=sd15_given_view_inputs(), conditional_latents_change_every_input=True,
                   num_inference_steps=10, guidance_scale=2.0, latents=sc["latents"].clone(), prompt_embeds=sc["prompt_embeds"],
                   negative_prompt_embeds=sc["negative_prompt_embeds"], output_type="latent", callback=_trace_cb(trace, 2), callback_steps=1,
                   bev_controlnet_kwargs={"bboxes_3d_data": sc["bboxes_3d_data"]}).images
    dt = time.time() - t0
    meta["reference_seconds"] = dt; meta["reference_threads"] = torch.get_num_threads()
    torch.save({"meta": meta, "steps": 10, "guidance": 2.0, "given": [0, 3], "latents": out.half().clone(), "trace": trace,
                "absmean": out.abs().mean().item()}, os.path.join(out_dir, "sd15_loop_given_view.pt"))
    v = out[0].float()
    print("sd15_loop_given_view |x|", out.abs().mean().item(), f"reference: {dt:.1f} s; view-to-view rel diff vs view 1:",
          [round(((v[j] - v[1]).norm() / v[1].norm()).item(), 3) for j in range(6)])


def cpu_reference_vs_port(out_path):
    """Time the REAL reference denoise step (BEVControlNetModel.forward + UNet forward through diffusers, fp32) next to the CPU port
    (oracle/denoiser.py) on the same scene, weights and thread count — the ratio bench.py quotes beside its `cpu_baseline` (kind "port")
    on boxes where /root/reference does not exist.  Written to profiles/ (no literals in bench.py)."""
    import json
    import time
    from helpers import bf16_round
    from oracle import denoiser as D
    torch.set_num_threads(8)
    cfg = spec.SD15_CONFIG
    usd, csd = state_dicts(cfg)
    usd, csd = bf16_round(usd), bf16_round(csd)
    ns, unet, cnet = ref_models.build_reference(cfg, usd, csd)
    sc = scene(cfg, 1, None, (28, 50), zero_map=True)
    lat = torch.stack([sc["latents"]] * 6, 1)
    cam = D.uncond_cam_param(csd, 1, 6)

    def ref_step(t):
        tt = torch.tensor([t])
        d, m, ctx = cnet(lat, tt, cam, None, sc["prompt_embeds"], sc["bev_map"], return_dict=False)
        return unet(lat.reshape(-1, 4, 28, 50), tt.repeat_interleave(6), encoder_hidden_states=ctx,
                    down_block_additional_residuals=d, mid_block_additional_residual=m).sample

    def port_step(t):
        d, m, ctx = D.controlnet_forward(csd, cfg, lat, torch.tensor([t]), cam, None, sc["prompt_embeds"], sc["bev_map"])
        return D.unet_forward(usd, cfg, lat.reshape(-1, 4, 28, 50), t, ctx, d, m)

    res = {}
    with torch.no_grad():
        for name, fn in (("reference", ref_step), ("port", port_step)):
            fn(981)
            t0 = time.time()
            outs = [fn(961 - 20 * i) for i in range(3)]
            res[name] = (time.time() - t0) / 3
            res[name + "_out"] = outs[-1]
    rel = ((res["port_out"] - res["reference_out"]).norm() / res["reference_out"].norm()).item()
    out = {"what": "one 6-view text-only denoise step (BEV-ControlNet + multi-view UNet, SD-1.5 size, fp32) on the host cores: the REAL reference "
                   "(modules imported from /root/reference through oracle/refshim.py) vs the CPU port oracle/denoiser.py; 1 warm-up + 3 timed steps each",
           "threads": torch.get_num_threads(), "reference_s_per_step": round(res["reference"], 3), "port_s_per_step": round(res["port"], 3),
           "reference_over_port": round(res["port"] / res["reference"], 3), "port_vs_reference_rel_l2": rel,
           "torch": str(torch.__version__), "where": "authoring container (no GPU); written by tools/make_golden.py cpuref"}
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


def vae_fixture(out_dir):
    from oracle import refshim
    ns = refshim.load()
    vcfg = spec.VAE_TINY_CONFIG
    sd = spec.random_state_dict(spec.vae_decoder_param_shapes(vcfg), 5)
    vae = ns.diffusers.AutoencoderKL(in_channels=3, out_channels=vcfg["out_channels"], block_out_channels=vcfg["block_out_channels"],
                                     down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
                                     latent_channels=vcfg["latent_channels"], norm_num_groups=vcfg["norm_num_groups"],
                                     layers_per_block=vcfg["layers_per_block"]).eval()
    missing, unexpected = vae.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("encoder.", "quant_conv.")) for k in missing), (missing[:3], unexpected[:3])
    z = torch.randn(2, 4, 7, 13, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        img = vae.decode(z).sample
    torch.save({"z_seed": 3, "weights_seed": 5, "checksum": checksum(sd), "image": img.half()}, os.path.join(out_dir, "tiny_vae_decode.pt"))
    print("tiny_vae_decode:", tuple(img.shape), "|x|", img.abs().mean().item())


def vae_encode_fixture(out_dir):
    """diffusers AutoencoderKL.encode (dif:models/autoencoder_kl.py:127-171: quant_conv(encoder(x)) -> DiagonalGaussianDistribution) of the tiny
    VAE config with seeded weights (decoder seed 5 as in tiny_vae_decode.pt, encoder + quant_conv seed 1005 = what
    magicdrive_amd AutoencoderKL.from_config(VAE_TINY_CONFIG, 5, with_encoder=True) builds) on two 56x104 images in [-1, 1]: the call
    demo/run_cond_on_view.py:79-86 makes on its known views."""
    from oracle import refshim
    ns = refshim.load()
    vcfg = spec.VAE_TINY_CONFIG
    sd = spec.random_state_dict(spec.vae_decoder_param_shapes(vcfg), 5)
    sd.update(spec.random_state_dict(spec.vae_encoder_param_shapes(vcfg), 1005))
    vae = ns.diffusers.AutoencoderKL(in_channels=3, out_channels=vcfg["out_channels"], block_out_channels=vcfg["block_out_channels"],
                                     down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
                                     latent_channels=vcfg["latent_channels"], norm_num_groups=vcfg["norm_num_groups"],
                                     layers_per_block=vcfg["layers_per_block"]).eval()
    missing, unexpected = vae.load_state_dict(sd, strict=True)
    x = torch.rand(2, 3, 56, 104, generator=torch.Generator().manual_seed(9)) * 2 - 1
    with torch.no_grad():
        dist = vae.encode(x).latent_dist
    torch.save({"x_seed": 9, "weights_seed": 5, "checksum": checksum(sd), "mean": dist.mean.clone(), "logvar": dist.logvar.clone(),
                "sample_seed0": dist.sample(torch.Generator().manual_seed(0)).clone()}, os.path.join(out_dir, "tiny_vae_encode.pt"))
    print("tiny_vae_encode: mean", tuple(dist.mean.shape), "|mean|", dist.mean.abs().mean().item(), "|logvar|", dist.logvar.abs().mean().item())


def hires_fixture(out_dir, cfg0, usd, csd, meta, hw=(54, 96)):
    cfg = spec.with_plus_map_embedder(cfg0, hw)
    ns, unet, cnet = ref_models.build_reference(cfg, usd, csd, img_size=(hw[0] * 8, hw[1] * 8))
    sc = scene(cfg, 1, 3, hw)
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, 6, 4, *hw, generator=g)
    t = torch.tensor([741])
    with torch.no_grad():
        d, m, ctx = cnet(lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"], return_dict=False)
        e = unet(lat.reshape(-1, 4, *hw), t.repeat_interleave(6), encoder_hidden_states=ctx,
                 down_block_additional_residuals=d, mid_block_additional_residual=m).sample
    torch.save({"meta": meta, "lat_seed": 11, "timesteps": t, "hw": hw, "mid": m.clone(), "eps": e.half(),
                "down_absmean": torch.tensor([x.abs().mean() for x in d]), "down_first": d[0][:, :, ::9, ::12].clone()},
               os.path.join(out_dir, "tiny_forward_hires.pt"))
    print("tiny_forward_hires: eps std", e.std().item(), "down0 |x|", d[0].abs().mean().item())


def nattn_fixture(out_dir, cfg0, usd, csd, meta, hw=(28, 50)):
    """neighboring_attn_type "concat" and "self" (blocks.py:106-142, 206-217): the UNet forward of the REAL reference on the same tiny
    weights (the ControlNet has no cross-view attention: its residuals come from the default-mode golden's inputs)."""
    out = {"meta": meta, "lat_seed": 13, "timesteps": torch.tensor([500])}
    for mode in ("concat", "self"):
        cfg = dict(cfg0); cfg["neighboring_attn_type"] = mode
        ns, unet, cnet = ref_models.build_reference(cfg, usd, csd)
        sc = scene(cfg, 1, 3, hw)
        lat = torch.randn(1, 6, 4, *hw, generator=torch.Generator().manual_seed(13))
        t = out["timesteps"]
        with torch.no_grad():
            d, m, ctx = cnet(lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"], return_dict=False)
            e = unet(lat.reshape(-1, 4, *hw), t.repeat_interleave(6), encoder_hidden_states=ctx,
                     down_block_additional_residuals=d, mid_block_additional_residual=m).sample
        out["eps_" + mode] = e.half()
        print("tiny_forward_nattn", mode, "eps std", e.std().item())
    torch.save(out, os.path.join(out_dir, "tiny_forward_nattn.pt"))


def zmod_fixture(out_dir, cfg0, csd, meta, hw=(28, 50)):
    """zero_module_type "gated" (GatedConnector: tanh(alpha) * x, blocks.py:24-32, 84-85) and "none" (identity connector, :86-88): the UNet
    forward of the REAL reference; every tensor the three variants share has the same seeded values (spec.random_state_dict seeds per name)."""
    out = {"meta": meta, "lat_seed": 17, "timesteps": torch.tensor([400])}
    for mode in ("gated", "none"):
        cfg = dict(cfg0); cfg["zero_module_type"] = mode
        usd = spec.random_state_dict(spec.unet_param_shapes(cfg), 0)
        ns, unet, cnet = ref_models.build_reference(cfg, usd, csd)
        sc = scene(cfg, 1, 3, hw)
        lat = torch.randn(1, 6, 4, *hw, generator=torch.Generator().manual_seed(17))
        t = out["timesteps"]
        with torch.no_grad():
            d, m, ctx = cnet(lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"], return_dict=False)
            e = unet(lat.reshape(-1, 4, *hw), t.repeat_interleave(6), encoder_hidden_states=ctx,
                     down_block_additional_residuals=d, mid_block_additional_residual=m).sample
        out["eps_" + mode] = e.half()
        out["unet_checksum_" + mode] = checksum(usd)
        print("tiny_forward_zmod", mode, "eps std", e.std().item())
    torch.save(out, os.path.join(out_dir, "tiny_forward_zmod.pt"))


def _trace_cb(store, every):
    def cb(i, t, latents):
        if (i + 1) % every == 0:
            store[i + 1] = latents.detach().half().clone()
    return cb


def sd15_loop_fixture(out_dir, full_cond):
    """The headline configuration against the REAL reference."""
    import time
    from helpers import bf16_round
    torch.set_num_threads(8)
    cfg = spec.SD15_CONFIG
    usd, csd = state_dicts(cfg)
    usd, csd = bf16_round(usd), bf16_round(csd)
    meta = {"unet_checksum": checksum(usd), "cn_checksum": checksum(csd), "torch": str(torch.__version__),
            "weights": "spec.random_state_dict seeds (0, 1), bf16-rounded; reference arithmetic fp32"}
    ns, pipe = ref_models.build_reference_pipeline(cfg, usd, csd)
    trace = {}
    t0 = time.time()
    with torch.no_grad():
        if not full_cond:
            steps, every, name = 50, 10, "sd15_loop50.pt"
            sc = scene(cfg, 1, None, (28, 50), zero_map=True)
            out = pipe(prompt=None, image=sc["bev_map"], camera_param=None, height=224, width=400, num_inference_steps=steps, guidance_scale=2.0,
                       latents=sc["latents"].clone(), prompt_embeds=sc["prompt_embeds"], negative_prompt_embeds=sc["negative_prompt_embeds"],
                       output_type="latent", callback=_trace_cb(trace, every), callback_steps=1,
                       bev_controlnet_kwargs={"bboxes_3d_data": None}).images
        else:
            steps, every, name = 10, 2, "sd15_loop_cfg.pt"
            sc = scene(cfg, 1, 32, (28, 50))
            out = pipe(prompt=None, image=sc["bev_map"], camera_param=sc["camera_param"], height=224, width=400, num_inference_steps=steps,
                       guidance_scale=2.0, latents=sc["latents"].clone(), prompt_embeds=sc["prompt_embeds"],
                       negative_prompt_embeds=sc["negative_prompt_embeds"], output_type="latent", callback=_trace_cb(trace, every), callback_steps=1,
                       bev_controlnet_kwargs={"bboxes_3d_data": sc["bboxes_3d_data"]}).images
    dt = time.time() - t0
    meta["reference_seconds"] = dt; meta["reference_threads"] = torch.get_num_threads()
    torch.save({"meta": meta, "steps": steps, "guidance": 2.0, "latents": out.half().clone(), "trace": trace,
                "absmean": out.abs().mean().item()}, os.path.join(out_dir, name))
    print(name, "|x|", out.abs().mean().item(), f"reference: {dt:.1f} s for {steps} steps = {dt / steps:.2f} s/step on {torch.get_num_threads()} threads",
          os.path.getsize(os.path.join(out_dir, name)) // 1024, "KiB")


def sd15_hires_fixture(out_dir, hw=(54, 96)):
    from helpers import bf16_round
    torch.set_num_threads(8)
    cfg = spec.with_plus_map_embedder(spec.SD15_CONFIG, hw)
    usd, csd = state_dicts(cfg)
    usd, csd = bf16_round(usd), bf16_round(csd)
    meta = {"unet_checksum": checksum(usd), "cn_checksum": checksum(csd), "torch": str(torch.__version__),
            "weights": "spec.random_state_dict seeds (0, 1), bf16-rounded; reference arithmetic fp32"}
    ns, unet, cnet = ref_models.build_reference(cfg, usd, csd, img_size=(hw[0] * 8, hw[1] * 8))
    sc = scene(cfg, 1, 3, hw)
    lat = torch.randn(1, 6, 4, *hw, generator=torch.Generator().manual_seed(11))
    t = torch.tensor([741])
    with torch.no_grad():
        d, m, ctx = cnet(lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"], return_dict=False)
        e = unet(lat.reshape(-1, 4, *hw), t.repeat_interleave(6), encoder_hidden_states=ctx,
                 down_block_additional_residuals=d, mid_block_additional_residual=m).sample
    torch.save({"meta": meta, "lat_seed": 11, "timesteps": t, "hw": hw, "mid_sub": m.half()[:, ::4].clone(), "eps": e.half(),
                "down_absmean": torch.tensor([x.abs().mean() for x in d])}, os.path.join(out_dir, "sd15_forward_hires.pt"))   # mid: every 4th channel
    print("sd15_forward_hires: eps std", e.std().item(), "mid |x|", m.abs().mean().item())


def sd15_hires_loop_fixture(out_dir, hw=(54, 96), steps=6):
    """BASELINE configs[3] as a LOOP at real width: the REAL reference pipeline's __call__ (pipeline_bev_controlnet.py:349-451) at 432x768
    (54x96 latents, ...Plus map encoder), camera + 3 boxes + BEV map, CFG 2.0, `steps` DDIM steps -> sd15_loop_hires.pt.  (The 224x400
    loops pin the sampler / CFG machinery; this one pins it at the geometry whose level-0 sequences are 5184 tokens.)"""
    from helpers import bf16_round
    torch.set_num_threads(8)
    cfg = spec.with_plus_map_embedder(spec.SD15_CONFIG, hw)
    usd, csd = state_dicts(cfg)
    usd, csd = bf16_round(usd), bf16_round(csd)
    meta = {"unet_checksum": checksum(usd), "cn_checksum": checksum(csd), "torch": str(torch.__version__),
            "weights": "spec.random_state_dict seeds (0, 1), bf16-rounded; reference arithmetic fp32"}
    ns, pipe = ref_models.build_reference_pipeline(cfg, usd, csd, img_size=(hw[0] * 8, hw[1] * 8))
    sc = scene(cfg, 1, 3, hw)
    trace = {}
    t0 = time.time()
    with torch.no_grad():
        out = pipe(prompt=None, image=sc["bev_map"], camera_param=sc["camera_param"], height=hw[0] * 8, width=hw[1] * 8, num_inference_steps=steps,
                   guidance_scale=2.0, latents=sc["latents"].clone(), prompt_embeds=sc["prompt_embeds"],
                   negative_prompt_embeds=sc["negative_prompt_embeds"], output_type="latent", callback=_trace_cb(trace, 2), callback_steps=1,
                   bev_controlnet_kwargs={"bboxes_3d_data": sc["bboxes_3d_data"]}).images
    dt = time.time() - t0
    meta["reference_seconds"] = dt; meta["reference_threads"] = torch.get_num_threads()
    name = "sd15_loop_hires.pt"
    torch.save({"meta": meta, "steps": steps, "guidance": 2.0, "hw": hw, "latents": out.half().clone(), "trace": trace,
                "absmean": out.abs().mean().item()}, os.path.join(out_dir, name))
    print(name, "|x|", out.abs().mean().item(), f"reference: {dt:.1f} s for {steps} steps = {dt / steps:.2f} s/step on {torch.get_num_threads()} threads",
          os.path.getsize(os.path.join(out_dir, name)) // 1024, "KiB")


def resolution_fixture(out_dir, cfg0, usd, meta, which):
    """The reference's two other shipped resolutions at tiny width (module forwards of the REAL reference)."""
    if which == "272x736":
        hw = (34, 92)
        cfg = spec.with_plus_map_embedder(cfg0, hw)                # configs/exp/272x736.yaml:15-22
    else:
        hw = (53, 100)
        cfg = copy.deepcopy(cfg0); cfg["controlnet"]["map_size"] = (8, 400, 400)   # configs/exp/424x800abox0.1_nockpt.yaml:15-17
    csd = spec.random_state_dict(spec.controlnet_param_shapes(cfg), 1)
    meta = dict(meta); meta["cn_checksum"] = checksum(csd)
    ns, unet, cnet = ref_models.build_reference(cfg, usd, csd, img_size=(hw[0] * 8, hw[1] * 8))
    sc = scene(cfg, 1, 3, hw, map_size=cfg["controlnet"]["map_size"][1])
    lat = torch.randn(1, 6, 4, *hw, generator=torch.Generator().manual_seed(17))
    t = torch.tensor([333])
    with torch.no_grad():
        d, m, ctx = cnet(lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"], return_dict=False)
        e = unet(lat.reshape(-1, 4, *hw), t.repeat_interleave(6), encoder_hidden_states=ctx,
                 down_block_additional_residuals=d, mid_block_additional_residual=m).sample
    name = f"tiny_forward_{which}.pt"
    torch.save({"meta": meta, "lat_seed": 17, "timesteps": t, "hw": hw, "mid": m.clone(), "eps": e.half(),
                "down_absmean": torch.tensor([x.abs().mean() for x in d])}, os.path.join(out_dir, name))
    print(name, "eps std", e.std().item(), "down0 |x|", d[0].abs().mean().item(), "map", tuple(sc["bev_map"].shape))


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    if sys.argv[1:] == ["sd15"]:
        return sd15_loop_fixture(out_dir, False)
    if sys.argv[1:] == ["sd15cfg"]:
        return sd15_loop_fixture(out_dir, True)
    if sys.argv[1:] == ["sd15hires"]:
        return sd15_hires_fixture(out_dir)
    if sys.argv[1:] == ["sd15hiresloop"]:
        return sd15_hires_loop_fixture(out_dir)
    if sys.argv[1:] == ["sd15given"]:
        return sd15_given_view_fixture(out_dir)
    if sys.argv[1:] == ["cpuref"]:
        return cpu_reference_vs_port(os.path.join(ROOT, "profiles", "r04_cpu_reference_vs_port.json"))
    cfg = spec.TINY_CONFIG
    usd, csd = state_dicts(cfg)
    meta = {"unet_checksum": checksum(usd), "cn_checksum": checksum(csd), "torch": str(torch.__version__)}
    if sys.argv[1:] == ["unipc"]:
        return unipc_fixture(out_dir, cfg, usd, csd, meta)
    if sys.argv[1:] == ["hires"]:
        return hires_fixture(out_dir, cfg, usd, csd, meta)
    if sys.argv[1:] == ["given"]:
        return given_view_fixture(out_dir, cfg, usd, csd, meta)
    if sys.argv[1:] == ["givenunipc"]:
        return given_view_fixture(out_dir, cfg, usd, csd, meta, scheduler="unipc")
    if sys.argv[1:] == ["cxyz"]:
        return cxyz_fixture(out_dir, cfg, usd, meta)
    if sys.argv[1:] == ["vae"]:
        return vae_fixture(out_dir)
    if sys.argv[1:] == ["vaeenc"]:
        return vae_encode_fixture(out_dir)
    if sys.argv[1:] == ["nattn"]:
        return nattn_fixture(out_dir, cfg, usd, csd, meta)
    if sys.argv[1:] == ["zmod"]:
        return zmod_fixture(out_dir, cfg, csd, meta)
    if sys.argv[1:] in (["res272"], ["res424"]):
        return resolution_fixture(out_dir, cfg, usd, meta, "272x736" if sys.argv[1] == "res272" else "424x800")

    # ---- module-level forwards
    ns, unet, cnet = ref_models.build_reference(cfg, usd, csd)
    nb, Lb, hw = 2, 5, (28, 50)
    sc = scene(cfg, nb, Lb, hw)
    g = torch.Generator().manual_seed(7)
    lat = torch.randn(nb, 6, 4, *hw, generator=g)
    t = torch.tensor([981, 501])
    with torch.no_grad():
        d, m, ctx = cnet(lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"], return_dict=False)
        e = unet(lat.reshape(-1, 4, *hw), t.repeat_interleave(6), encoder_hidden_states=ctx,
                 down_block_additional_residuals=d, mid_block_additional_residual=m).sample
    torch.save({"meta": meta, "lat_seed": 7, "timesteps": t, "ctx": ctx.half(), "mid": m.clone(), "eps": e.clone(),
                "down_absmean": torch.tensor([x.abs().mean() for x in d]), "down_first": d[0][:, :, ::7, ::10].clone(),
                "down_last": d[-1].clone()}, os.path.join(out_dir, "tiny_forward.pt"))
    print("tiny_forward: eps std", e.std().item())

    # ---- the reference pipeline __call__
    ns, pipe = ref_models.build_reference_pipeline(cfg, usd, csd)
    sc = scene(cfg, 2, 5, hw)
    with torch.no_grad():
        out = pipe(prompt=None, image=sc["bev_map"], camera_param=sc["camera_param"], height=224, width=400, num_inference_steps=5,
                   guidance_scale=2.0, latents=sc["latents"].clone(), prompt_embeds=sc["prompt_embeds"],
                   negative_prompt_embeds=sc["negative_prompt_embeds"], output_type="latent",
                   bev_controlnet_kwargs={"bboxes_3d_data": sc["bboxes_3d_data"]}).images
        out_nocam = pipe(prompt=None, image=torch.zeros_like(sc["bev_map"]), camera_param=None, height=224, width=400, num_inference_steps=5,
                         guidance_scale=2.0, latents=sc["latents"].clone(), prompt_embeds=sc["prompt_embeds"],
                         negative_prompt_embeds=sc["negative_prompt_embeds"], output_type="latent",
                         bev_controlnet_kwargs={"bboxes_3d_data": None}).images
    torch.save({"meta": meta, "steps": 5, "guidance": 2.0, "latents_cfg": out.clone(), "latents_textonly": out_nocam.clone()},
               os.path.join(out_dir, "tiny_pipeline.pt"))
    print("tiny_pipeline: |x|", out.abs().mean().item(), out_nocam.abs().mean().item())
    unipc_fixture(out_dir, cfg, usd, csd, meta, hw)
    hires_fixture(out_dir, cfg, usd, csd, meta)
    given_view_fixture(out_dir, cfg, usd, csd, meta)
    vae_fixture(out_dir)
    for f in os.listdir(out_dir):
        print(f, os.path.getsize(os.path.join(out_dir, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
