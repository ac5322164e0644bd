"""CPU: pin the oracle (oracle/denoiser.py).

1. Known-answer tests the reference vendors (third_party/diffusers/tests/...): the module inputs/weights are
   re-created with plain torch.nn layers constructed in the same order under torch.manual_seed(0), so the
   hard-coded expected slices apply without importing any reference code.
2. Golden vectors produced by the real reference (tools/make_golden.py -> tests/golden/*.pt).
3. When /root/reference is present (authoring container): live comparison against the reference modules.
"""
import os

import pytest
import torch
import torch.nn as nn

from helpers import bf16_round, given_view_inputs, rel_l2, scene, state_dicts
from magicdrive_amd.networks import spec
from oracle import denoiser as D
from oracle import refshim

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ---------------------------------------------------------------- diffusers KATs
def test_kat_sinusoid_embeddings_hardcoded():
    # tests/models/test_layers_utils.py:89-115
    t = torch.arange(128)
    t1 = D.timestep_embedding(t, 64, flip_sin_to_cos=False, freq_shift=1)
    t2 = D.timestep_embedding(t, 64, flip_sin_to_cos=True, freq_shift=0)
    assert torch.allclose(t1[23:26, 47:50].flatten(), torch.tensor([0.9646, 0.9804, 0.9892, 0.9615, 0.9787, 0.9882, 0.9582, 0.9769, 0.9872]), 1e-3)
    assert torch.allclose(t2[23:26, 47:50].flatten(), torch.tensor([0.3019, 0.2280, 0.1716, 0.3146, 0.2377, 0.1790, 0.3272, 0.2474, 0.1864]), 1e-3)


def _sd_of(mods):
    sd = {}
    for pre, m in mods.items():
        for k, v in m.state_dict().items():
            sd[f"{pre}.{k}"] = v
    return sd


def test_kat_resnet_default():
    # tests/models/test_layers_utils.py:223-236: ResnetBlock2D(in_channels=32, temb_channels=128), groups 32, eps 1e-6
    torch.manual_seed(0)
    sample = torch.randn(1, 32, 64, 64)
    temb = torch.randn(1, 128)
    mods = {"norm1": nn.GroupNorm(32, 32), "conv1": nn.Conv2d(32, 32, 3, padding=1), "time_emb_proj": nn.Linear(128, 32),
            "norm2": nn.GroupNorm(32, 32), "conv2": nn.Conv2d(32, 32, 3, padding=1)}       # ctor order resnet.py:535-561
    with torch.no_grad():
        out = D.resnet_block(_sd_of(mods), "", sample, temb, 32, 1e-6)
    exp = torch.tensor([-1.9010, -0.2974, -0.8245, -1.3533, 0.8742, -0.9645, -2.0584, 1.3387, -0.4746])
    assert torch.allclose(out[0, -1, -3:, -3:].flatten(), exp, atol=1e-3)


def _attention_mods(c, ctx):
    return {"to_q": nn.Linear(c, c, bias=False), "to_k": nn.Linear(ctx, c, bias=False), "to_v": nn.Linear(ctx, c, bias=False), "to_out.0": nn.Linear(c, c)}


def _transformer_sd(c, cross):
    mods = {"norm": nn.GroupNorm(32, c, eps=1e-6), "proj_in": nn.Conv2d(c, c, 1)}
    b = "transformer_blocks.0."
    mods[b + "norm1"] = nn.LayerNorm(c)
    for k, v in _attention_mods(c, c).items():
        mods[b + "attn1." + k] = v
    if cross is not None:
        mods[b + "norm2"] = nn.LayerNorm(c)
        for k, v in _attention_mods(c, cross).items():
            mods[b + "attn2." + k] = v
    mods[b + "norm3"] = nn.LayerNorm(c)
    mods[b + "ff.net.0.proj"] = nn.Linear(c, 8 * c)
    mods[b + "ff.net.2"] = nn.Linear(4 * c, c)
    mods["proj_out"] = nn.Conv2d(c, c, 1)
    return _sd_of(mods)


def test_kat_spatial_transformer_default():
    # tests/models/test_layers_utils.py:315-337: Transformer2DModel(32 ch, 1 head x 32, no cross attention)
    torch.manual_seed(0)
    sample = torch.randn(1, 32, 64, 64)
    sd = _transformer_sd(32, None)
    with torch.no_grad():
        out = D.transformer_2d(sd, "", sample, None, 1, 32, None)
    exp = torch.tensor([-1.9455, -0.0066, -1.3933, -1.5878, 0.5325, -0.6486, -1.8648, 0.7515, -0.9689])
    assert torch.allclose(out[0, -1, -3:, -3:].flatten(), exp, atol=1e-3)


def test_kat_spatial_transformer_cross_attention():
    # tests/models/test_layers_utils.py:339-361: 64 ch, 2 heads x 32, cross_attention_dim 64, context (1,4,64)
    torch.manual_seed(0)
    sample = torch.randn(1, 64, 64, 64)
    sd = _transformer_sd(64, 64)
    with torch.no_grad():
        ctx = torch.randn(1, 4, 64)
        out = D.transformer_2d(sd, "", sample, ctx, 2, 32, None)
    exp = torch.tensor([0.0143, -0.6909, -2.1547, -1.8893, 1.4097, 0.1359, -0.2521, -1.3359, 0.2598])
    assert torch.allclose(out[0, -1, -3:, -3:].flatten(), exp, atol=1e-3)


def test_kat_ddim():
    # tests/schedulers/test_scheduler_ddim.py:50-54 (timesteps), :102-112 (variance), :114-121 (10-step loop)
    kw = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear")
    s = D.DDIM(clip_sample=True, set_alpha_to_one=True, steps_offset=1, **kw)
    assert s.set_timesteps(5).tolist() == [801, 601, 401, 201, 1]
    s = D.DDIM(clip_sample=True, set_alpha_to_one=True, steps_offset=0, **kw)
    for (t, p), v in {(0, 0): 0.0, (420, 400): 0.14771, (980, 960): 0.32460, (487, 486): 0.00979, (999, 998): 0.02}.items():
        assert abs(float(s.variance(t, p)) - v) < 1e-5
    n = 4 * 3 * 8 * 8
    sample = (torch.arange(n).reshape(3, 8, 8, 4) / n).permute(3, 0, 1, 2)      # test_schedulers.py:222-234
    for t in s.set_timesteps(10):
        sample = s.step(sample * t / (t + 1), int(t), sample)
    assert abs(sample.abs().sum().item() - 172.0067) < 1e-2 and abs(sample.abs().mean().item() - 0.223967) < 1e-3


def test_kat_unipc():
    # third_party/diffusers/tests/schedulers/test_scheduler_unipc.py:19-30 (config: linear betas, order 2, bh1), :205-209
    # test_full_loop_no_noise -> mean |x| = 0.2521; dummy model = sample * t / (t + 1), test_schedulers.py:222-241
    s = D.UniPC(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", solver_order=2, solver_type="bh1")
    n = 4 * 3 * 8 * 8
    sample = (torch.arange(n).reshape(3, 8, 8, 4) / n).permute(3, 0, 1, 2)
    for t in s.set_timesteps(10):
        sample = s.step(sample * t / (t + 1), int(t), sample)
    assert abs(sample.abs().mean().item() - 0.2521) < 1e-3
    # tools/test.py's sampler: 20 steps over SD-1.5's betas (configs/runner/default.yaml:54-57)
    s = D.UniPC()
    ts = s.set_timesteps(20)
    assert ts[0] == 999 and ts[-1] == 50 and len(ts) == 20


def test_sd15_sampler_timesteps():
    # 50 steps, steps_offset 1 -> 981, 961, ..., 1; final alpha_prev = alphas_cumprod[0]
    s = D.DDIM()
    ts = s.set_timesteps(50)
    assert ts[0] == 981 and ts[-1] == 1 and (ts[:-1] - ts[1:] == 20).all()
    a_t, a_p = s.coefficients(1)
    assert a_p == s.alphas_cumprod[0]


# ---------------------------------------------------------------- golden vectors from the real reference
def _checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


@pytest.fixture(scope="module")
def tiny():
    cfg = spec.TINY_CONFIG
    usd, csd = state_dicts(cfg)
    return cfg, usd, csd


def test_golden_forward(tiny):
    cfg, usd, csd = tiny
    G = torch.load(os.path.join(GOLD, "tiny_forward.pt"))
    assert abs(_checksum(usd) - G["meta"]["unet_checksum"]) < 1e-6 * G["meta"]["unet_checksum"], "seeded init drifted: regenerate goldens"
    assert abs(_checksum(csd) - G["meta"]["cn_checksum"]) < 1e-6 * G["meta"]["cn_checksum"]
    sc = scene(cfg, 2, 5)
    lat = torch.randn(2, 6, 4, 28, 50, generator=torch.Generator().manual_seed(G["lat_seed"]))
    t = G["timesteps"]
    with torch.no_grad():
        d, m, ctx = D.controlnet_forward(csd, cfg, lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"])
        e = D.unet_forward(usd, cfg, lat.reshape(-1, 4, 28, 50), t.repeat_interleave(6), ctx, d, m)
    assert rel_l2(ctx, G["ctx"].float()) < 1e-3            # stored in fp16
    assert rel_l2(m, G["mid"]) < 1e-4 and rel_l2(d[-1], G["down_last"]) < 1e-4
    assert rel_l2(d[0][:, :, ::7, ::10], G["down_first"]) < 1e-4
    assert torch.allclose(torch.tensor([x.abs().mean() for x in d]), G["down_absmean"], rtol=1e-4)
    assert rel_l2(e, G["eps"]) < 1e-4, rel_l2(e, G["eps"])


def test_golden_pipeline(tiny):
    cfg, usd, csd = tiny
    G = torch.load(os.path.join(GOLD, "tiny_pipeline.pt"))
    sc = scene(cfg, 2, 5)
    with torch.no_grad():
        out = D.sample_loop(usd, csd, cfg, sc["latents"], sc["prompt_embeds"], sc["negative_prompt_embeds"], sc["bev_map"],
                            sc["camera_param"], sc["bboxes_3d_data"], num_steps=G["steps"], guidance_scale=G["guidance"])
        out2 = D.sample_loop(usd, csd, cfg, sc["latents"], sc["prompt_embeds"], sc["negative_prompt_embeds"], torch.zeros_like(sc["bev_map"]),
                             None, None, num_steps=G["steps"], guidance_scale=G["guidance"])
    assert rel_l2(out, G["latents_cfg"]) < 2e-4, rel_l2(out, G["latents_cfg"])
    assert rel_l2(out2, G["latents_textonly"]) < 2e-4, rel_l2(out2, G["latents_textonly"])


def test_golden_pipeline_unipc(tiny):
    """The oracle loop with the restated UniPC vs the REAL reference pipeline running diffusers' UniPCMultistepScheduler."""
    cfg, usd, csd = tiny
    G = torch.load(os.path.join(GOLD, "tiny_pipeline_unipc.pt"))
    assert abs(_checksum(usd) - G["meta"]["unet_checksum"]) < 1e-3 * G["meta"]["unet_checksum"]
    sc = scene(cfg, 2, 5)
    with torch.no_grad():
        out = D.sample_loop(usd, csd, cfg, sc["latents"], sc["prompt_embeds"], sc["negative_prompt_embeds"], sc["bev_map"],
                            sc["camera_param"], sc["bboxes_3d_data"], num_steps=G["steps"], guidance_scale=G["guidance"], scheduler=D.UniPC())
    assert rel_l2(out, G["latents_cfg"]) < 2e-4, rel_l2(out, G["latents_cfg"])


@pytest.mark.parametrize("every", [True, False])
def test_golden_pipeline_given_view(tiny, every):
    """The restated given-view loop vs the REAL StableDiffusionBEVControlNetGivenViewPipeline (both re-noising modes)."""
    cfg, usd, csd = tiny
    G = torch.load(os.path.join(GOLD, "tiny_pipeline_given_view.pt"))
    sc = scene(cfg, 2, 5)
    with torch.no_grad():
        out = D.sample_loop(usd, csd, cfg, sc["latents"], sc["prompt_embeds"], sc["negative_prompt_embeds"], sc["bev_map"],
                            sc["camera_param"], sc["bboxes_3d_data"], num_steps=G["steps"], guidance_scale=G["guidance"],
                            conditional_latents=given_view_inputs(), conditional_latents_change_every_input=every)
    gold = G["latents_every" if every else "latents_once"]
    assert rel_l2(out, gold) < 2e-4, rel_l2(out, gold)


def test_golden_vae_decode():
    """the restated AutoencoderKL.decode vs diffusers' (tests/golden/tiny_vae_decode.pt)."""
    G = torch.load(os.path.join(GOLD, "tiny_vae_decode.pt"))
    vcfg = spec.VAE_TINY_CONFIG
    sd = spec.random_state_dict(spec.vae_decoder_param_shapes(vcfg), G["weights_seed"])
    assert abs(_checksum(sd) - G["checksum"]) < 1e-6 * G["checksum"]
    z = torch.randn(2, 4, 7, 13, generator=torch.Generator().manual_seed(G["z_seed"]))
    with torch.no_grad():
        img = D.vae_decode(sd, vcfg, z)
    assert img.shape == (2, 3, 56, 104) and rel_l2(img, G["image"].float()) < 1e-3      # golden stored as fp16


def test_golden_forward_hires_plus_map_encoder(tiny):
    """BASELINE.json configs[3] shape (432x768 -> 54x96 latents, BEVControlNetConditioningEmbeddingPlus): the oracle vs the real
    reference modules (tests/golden/tiny_forward_hires.pt, tools/make_golden.py hires)."""
    cfg0, usd, csd = tiny
    G = torch.load(os.path.join(GOLD, "tiny_forward_hires.pt"))
    hw = tuple(G["hw"])
    cfg = spec.with_plus_map_embedder(cfg0, hw)
    sc = scene(cfg, 1, 3, hw)
    lat = torch.randn(1, 6, 4, *hw, generator=torch.Generator().manual_seed(G["lat_seed"]))
    t = G["timesteps"]
    with torch.no_grad():
        d, m, ctx = D.controlnet_forward(csd, cfg, lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"])
        e = D.unet_forward(usd, cfg, lat.reshape(-1, 4, *hw), t.repeat_interleave(6), ctx, d, m)
    assert rel_l2(m, G["mid"]) < 1e-4 and torch.allclose(torch.tensor([x.abs().mean() for x in d]), G["down_absmean"], rtol=1e-4)
    assert rel_l2(d[0][:, :, ::9, ::12], G["down_first"]) < 1e-4
    assert rel_l2(e, G["eps"].float()) < 2e-3          # golden eps stored as fp16


@pytest.mark.parametrize("mode", ["concat", "self"])
def test_golden_forward_neighboring_attn_modes(tiny, mode):
    """neighboring_attn_type concat / self (blocks.py:106-142, 206-217): the oracle vs the real reference UNet (tools/make_golden.py nattn);
    the modes differ from each other and from the default by far more than the comparison tolerance."""
    cfg0, usd, csd = tiny
    G = torch.load(os.path.join(GOLD, "tiny_forward_nattn.pt"))
    cfg = dict(cfg0); cfg["neighboring_attn_type"] = mode
    sc = scene(cfg, 1, 3)
    lat = torch.randn(1, 6, 4, 28, 50, generator=torch.Generator().manual_seed(G["lat_seed"]))
    t = G["timesteps"]
    with torch.no_grad():
        d, m, ctx = D.controlnet_forward(csd, cfg, lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"])
        e = D.unet_forward(usd, cfg, lat.reshape(-1, 4, 28, 50), t.repeat_interleave(6), ctx, d, m)
        e_add = D.unet_forward(usd, cfg0, lat.reshape(-1, 4, 28, 50), t.repeat_interleave(6), ctx, d, m)
    assert rel_l2(e, G["eps_" + mode].float()) < 2e-3          # golden eps stored as fp16
    other = "self" if mode == "concat" else "concat"
    assert rel_l2(e, G["eps_" + other].float()) > 1e-2 and rel_l2(e, e_add) > 1e-2


@pytest.mark.parametrize("mode", ["gated", "none"])
def test_golden_forward_zero_module_types(tiny, mode):
    """zero_module_type gated (GatedConnector: tanh(alpha) * x, blocks.py:24-32, 84-85) / none (identity, :86-88): the oracle vs the real
    reference UNet (tools/make_golden.py zmod); the variants differ from each other by far more than the comparison tolerance."""
    cfg0, _, csd = tiny
    G = torch.load(os.path.join(GOLD, "tiny_forward_zmod.pt"))
    cfg = dict(cfg0); cfg["zero_module_type"] = mode
    usd = spec.random_state_dict(spec.unet_param_shapes(cfg), 0)
    assert (mode == "gated") == any(k.endswith("connector.alpha") for k in usd) and not any(k.endswith("connector.weight") for k in usd)
    sc = scene(cfg, 1, 3)
    lat = torch.randn(1, 6, 4, 28, 50, generator=torch.Generator().manual_seed(G["lat_seed"]))
    t = G["timesteps"]
    with torch.no_grad():
        d, m, ctx = D.controlnet_forward(csd, cfg, lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"])
        e = D.unet_forward(usd, cfg, lat.reshape(-1, 4, 28, 50), t.repeat_interleave(6), ctx, d, m)
    assert rel_l2(e, G["eps_" + mode].float()) < 2e-3          # golden eps stored as fp16
    assert rel_l2(e, G["eps_" + ("none" if mode == "gated" else "gated")].float()) > 1e-2
    with pytest.raises(TypeError):                              # the reference's error for an unknown type (blocks.py:89-90)
        spec.unet_param_shapes(dict(cfg0, zero_module_type="bogus"))


# ---------------------------------------------------------------- live reference (authoring container only)
needs_ref = pytest.mark.skipif(not refshim.available(), reason="/root/reference not present")


@needs_ref
def test_param_shapes_match_reference_modules():
    from oracle import ref_models
    for cfg in (spec.TINY_CONFIG,):
        usd, csd = state_dicts(cfg)
        ns, unet, cnet = ref_models.build_reference(cfg, usd, csd)       # strict load_state_dict inside
        assert {k: tuple(v.shape) for k, v in unet.state_dict().items()} == dict(spec.unet_param_shapes(cfg))
        assert {k: tuple(v.shape) for k, v in cnet.state_dict().items()} == dict(spec.controlnet_param_shapes(cfg))


@needs_ref
@pytest.mark.parametrize("n,order,stype", [(20, 2, "bh2"), (7, 2, "bh1"), (5, 1, "bh2"), (9, 3, "bh2")])
def test_unipc_restatement_matches_live_reference(n, order, stype):
    """The restated UniPC against the reference environment's own UniPCMultistepScheduler (what
    magicdrive/misc/test_utils.py:129 installs), driven by a fixed pseudo-model."""
    from oracle import refshim
    r = refshim.load().UniPCMultistepScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", solver_order=order, solver_type=stype)
    o = D.UniPC(solver_order=order, solver_type=stype)
    r.set_timesteps(n)
    assert torch.equal(o.set_timesteps(n), r.timesteps.cpu().long())
    g = torch.Generator().manual_seed(3)
    xr = xo = torch.randn(2, 4, 6, 5, generator=g)
    for t in r.timesteps:
        er = torch.tanh(xr * 0.7) + 0.1 * torch.cos(xr * float(t) / 300)
        eo = torch.tanh(xo * 0.7) + 0.1 * torch.cos(xo * float(t) / 300)
        xr = r.step(er, t, xr).prev_sample
        xo = o.step(eo, int(t), xo)
    assert (xr - xo).abs().max() < 5e-5 * max(1.0, float(xr.abs().max()))


@needs_ref
def test_reference_refuses_stock_ddim_like_survey_says():
    """pipeline_bev_controlnet.py:94-97: a scheduler whose step() takes `generator` raises RuntimeError."""
    from oracle import ref_models
    cfg = spec.TINY_CONFIG
    usd, csd = state_dicts(cfg)
    ns, pipe = ref_models.build_reference_pipeline(cfg, usd, csd)
    pipe.scheduler = ns.DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    sc = scene(cfg, 1, 2)
    with pytest.raises(RuntimeError):
        pipe(prompt=None, image=sc["bev_map"], camera_param=sc["camera_param"], height=224, width=400, num_inference_steps=2,
             guidance_scale=2.0, latents=sc["latents"], prompt_embeds=sc["prompt_embeds"], negative_prompt_embeds=sc["negative_prompt_embeds"],
             output_type="latent", bev_controlnet_kwargs={"bboxes_3d_data": sc["bboxes_3d_data"]})


def test_oracle_matches_reference_at_sd15_geometry():
    """Pins the oracle at the REAL shapes (head dims 40 / 80 / 160, 320..1280 channels, 2560-channel concat resnets): the first 10 of
    the 50 DDIM steps of BASELINE configs[1] on the same bf16-rounded seeded weights vs the latents the real reference pipeline produced
    (tests/golden/sd15_loop50.pt: trace after step 10).  ~1 minute on 8 host threads: the one long test of the CPU suite."""
    G = torch.load(os.path.join(GOLD, "sd15_loop50.pt"), weights_only=False)
    cfg = spec.SD15_CONFIG
    usd, csd = state_dicts(cfg)
    usd, csd = bf16_round(usd), bf16_round(csd)
    chk = lambda sd: float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(chk(usd) - G["meta"]["unet_checksum"]) < 1e-6 * G["meta"]["unet_checksum"], "seeded init drifted from the fixture's"
    sc = scene(cfg, 1, None, (28, 50), zero_map=True)
    sch = D.DDIM(); ts = sch.set_timesteps(G["steps"])
    n_cam = 6
    lat = torch.stack([sc["latents"]] * n_cam, 1)
    cam = D.uncond_cam_param(csd, 1, n_cam)
    x = lat.reshape(-1, *lat.shape[2:])
    with torch.no_grad():
        for t in ts[:10].tolist():
            d, m, ctx = D.controlnet_forward(csd, cfg, x.reshape(1, n_cam, *x.shape[1:]), torch.tensor([t]), cam, None, sc["prompt_embeds"], sc["bev_map"])
            x = sch.step(D.unet_forward(usd, cfg, x, t, ctx, d, m), t, x)
    ref = G["trace"][10].float().reshape(x.shape)
    e = rel_l2(x, ref)
    print(f"[oracle vs real reference, SD-1.5 size, 10 DDIM steps] rel L2 {e:.2e}")
    assert e < 2e-3, e            # fp16 storage of the fixture alone is ~3e-4
