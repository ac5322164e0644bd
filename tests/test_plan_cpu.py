"""CPU: the op programs the engine builds (topology, weight packing, hoisting, buffer aliasing) are executed
by the independent torch interpreter tests/plan_interp.py and compared with the oracle and the goldens.
This checks everything about the HIP path except the kernels' own arithmetic (that is tests/test_kernels_gpu.py)."""
import os

import pytest
import torch

import plan_interp
from helpers import given_view_inputs, bf16_round, cfg_inputs, rel_l2, scene, state_dicts
from magicdrive_amd import denoiser as DN, flops, schedulers
from magicdrive_amd.engine import PackedNet
from magicdrive_amd.networks import spec
from oracle import denoiser as D

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CPU = torch.device("cpu")


def one_scene(sc, i):
    """Scene i of a synthetic scene batch.  The sampler-plan tests below interpret ONE of the two scenes of a reference fixture: scenes are
    independent (cross-view attention couples only the cameras of one scene), so scene i of a 1-scene plan must reproduce row i of the
    2-scene golden — at half the CPU time (the 2-scene batching itself runs in tests/test_e2e_gpu.py against the same fixtures)."""
    out = {}
    for k, v in sc.items():
        if isinstance(v, dict):
            out[k] = {kk: vv[i:i + 1] for kk, vv in v.items()}
        elif isinstance(v, torch.Tensor):
            out[k] = v[i:i + 1]
        else:
            out[k] = v
    return out


@pytest.fixture(scope="module")
def tiny():
    cfg = spec.TINY_CONFIG
    usd, csd = state_dicts(cfg)
    return cfg, usd, csd, PackedNet(usd, CPU), PackedNet(csd, CPU)


def test_module_plans_match_golden_forward(tiny):
    cfg, usd, csd, un, cn = tiny
    G = torch.load(os.path.join(GOLD, "tiny_forward.pt"))
    sc = scene(cfg, 2, 5)
    lat = torch.randn(2, 6, 4, 28, 50, generator=torch.Generator().manual_seed(G["lat_seed"]))
    t = G["timesteps"]
    cp = DN.ControlNetPlan(cfg, cn, CPU, 2, 5, (28, 50))
    cp.sample_nchw.copy_(lat.reshape(-1, 4, 28, 50)); cp.temb.t.copy_(t.float().repeat_interleave(6))
    cp.cond.load(sc["camera_param"], sc["prompt_embeds"], sc["bev_map"], sc["bboxes_3d_data"])
    plan_interp.run(cp.ops)
    assert rel_l2(cp.cond.ctx, G["ctx"].float()) < 5e-3
    assert rel_l2(cp.mid_out, G["mid"]) < 3e-2 and rel_l2(cp.down_out[-1], G["down_last"]) < 3e-2
    with torch.no_grad():
        d, m, ctx = D.controlnet_forward(csd, cfg, lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"])
    up = DN.UNetPlan(cfg, un, CPU, 12, ctx.shape[1], (28, 50))
    up.sample_nchw.copy_(lat.reshape(-1, 4, 28, 50)); up.temb.t.copy_(t.float().repeat_interleave(6)); up.ctx.copy_(ctx)
    for dst, src in zip(up.res_in, d):
        dst.copy_(src)
    up.mid_in.copy_(m)
    plan_interp.run(up.ops)
    per_view = max(rel_l2(up.out_nchw[i], G["eps"][i]) for i in range(12))
    assert per_view < 3e-2, per_view


@pytest.mark.parametrize("do_cfg", [True, False])
def test_sampler_plan_matches_golden_pipeline(tiny, do_cfg):
    cfg, usd, csd, un, cn = tiny
    G = torch.load(os.path.join(GOLD, "tiny_pipeline.pt"))
    si = 0 if do_cfg else 1
    sc = one_scene(scene(cfg, 2, 5), si)
    steps = G["steps"]
    sch = schedulers.DDIMScheduler(); ts = sch.set_timesteps(steps)
    if do_cfg:
        cam, text, bev, boxes = cfg_inputs(D, csd, sc)
        sp = DN.SamplerPlan(cfg, un, cn, CPU, 1, True, 5, (28, 50), num_steps=steps, guidance_scale=G["guidance"])
        gold = G["latents_cfg"][si:si + 1]
    else:       # camera_param=None path: learned uncond camera, CFG forced off, no boxes, zero map
        cam, text, bev, boxes = D.uncond_cam_param(csd, 1, 6), sc["prompt_embeds"], torch.zeros_like(sc["bev_map"]), None
        sp = DN.SamplerPlan(cfg, un, cn, CPU, 1, False, 0, (28, 50), num_steps=steps, guidance_scale=G["guidance"])
        gold = G["latents_textonly"][si:si + 1]
    sp.load_inputs(torch.stack([sc["latents"]] * 6, 1), cam, text, bev, boxes, ts, sch.coefficient_table())
    plan_interp.run(sp.prologue_ops)
    for _ in range(steps):
        plan_interp.run(sp.step_ops, lower_check=False)
    assert sp.step_ctr.item() == steps
    assert rel_l2(sp.latents(), gold) < 4e-2, rel_l2(sp.latents(), gold)
    # every buffer handed out was returned to the pool (no leak that would grow with the step count)
    assert all(v for v in sp.bld.pool.free_list.values())


def test_unipc_sampler_plan_matches_golden_pipeline(tiny):
    """The UniPC step program (one fused cfg+unipc op with device-resident multistep history) vs the reference
    pipeline's latents under diffusers' UniPCMultistepScheduler (tests/golden/tiny_pipeline_unipc.pt)."""
    cfg, usd, csd, un, cn = tiny
    G = torch.load(os.path.join(GOLD, "tiny_pipeline_unipc.pt"))
    sc = one_scene(scene(cfg, 2, 5), 1)
    steps = G["steps"]
    sch = schedulers.UniPCMultistepScheduler(); ts = sch.set_timesteps(steps)
    cam, text, bev, boxes = cfg_inputs(D, csd, sc)
    sp = DN.SamplerPlan(cfg, un, cn, CPU, 1, True, 5, (28, 50), num_steps=steps, guidance_scale=G["guidance"], scheduler_kind="unipc")
    assert sp.step_ops[-1].name == "cfg+unipc"
    sp.load_inputs(torch.stack([sc["latents"]] * 6, 1), cam, text, bev, boxes, ts, sch.coefficient_table())
    plan_interp.run(sp.prologue_ops)
    for _ in range(steps):
        plan_interp.run(sp.step_ops, lower_check=False)
    assert sp.step_ctr.item() == steps
    assert rel_l2(sp.latents(), G["latents_cfg"][1:2]) < 4e-2, rel_l2(sp.latents(), G["latents_cfg"][1:2])
    # a second load_inputs must reset the multistep history (a reused plan starts a fresh trajectory)
    sp.load_inputs(torch.stack([sc["latents"]] * 6, 1), cam, text, bev, boxes, ts, sch.coefficient_table())
    assert sp.step_ctr.item() == 0 and not sp.m1.any() and not sp.x_last.any()


@pytest.mark.parametrize("mode", [1])          # mode 2 (pinned noise prediction): GPU kernel + pipeline tests and the oracle test cover it
def test_given_view_sampler_plan_matches_golden(tiny, mode):
    """Given views inside the fused CFG + DDIM op (MdxDdimDesc.gv_*): mode 1 = re-noise the known views for every model call,
    mode 2 = noise once and pin their noise prediction — vs the reference's given-view pipeline."""
    cfg, usd, csd, un, cn = tiny
    G = torch.load(os.path.join(GOLD, "tiny_pipeline_given_view.pt"))
    si = 0                                    # scene 0: views 0 and 3 given
    sc = one_scene(scene(cfg, 2, 5), si)
    steps = G["steps"]
    sch = schedulers.DDIMScheduler(); ts = sch.set_timesteps(steps)
    cam, text, bev, boxes = cfg_inputs(D, csd, sc)
    sp = DN.SamplerPlan(cfg, un, cn, CPU, 1, True, 5, (28, 50), num_steps=steps, guidance_scale=G["guidance"], given_view_mode=mode)
    cl = given_view_inputs()[si:si + 1]
    mask = torch.tensor([[v is not None for v in r] for r in cl])
    lat = torch.zeros(1, 6, 4, 28, 50)
    for i, r in enumerate(cl):
        for j, v in enumerate(r):
            if v is not None:
                lat[i, j] = v
    sp.load_inputs(torch.stack([sc["latents"]] * 6, 1), cam, text, bev, boxes, ts, sch.coefficient_table(), given_mask=mask, given_latents=lat)
    plan_interp.run(sp.prologue_ops)
    for _ in range(steps):
        plan_interp.run(sp.step_ops, lower_check=False)
    gold = G["latents_every" if mode == 1 else "latents_once"][si:si + 1]
    assert rel_l2(sp.latents(), gold) < 4e-2, rel_l2(sp.latents(), gold)


@pytest.mark.parametrize("mode", [1, 2])
def test_given_view_unipc_sampler_plan_matches_golden(tiny, mode):
    """Given views under UniPC (MdxUniPCDesc.gv_*, ABI 8) — what demo/run_cond_on_view.py really runs: its pipe comes from build_pipe,
    which installs UniPC (magicdrive/misc/test_utils.py:129).  vs the REAL reference given-view pipeline with diffusers'
    UniPCMultistepScheduler (tests/golden/tiny_pipeline_given_view_unipc.pt, tools/make_golden.py givenunipc), both re-noising modes."""
    cfg, usd, csd, un, cn = tiny
    G = torch.load(os.path.join(GOLD, "tiny_pipeline_given_view_unipc.pt"))
    si = 0 if mode == 1 else 1                # scene 0: views 0 and 3 given; scene 1: view 5
    sc = one_scene(scene(cfg, 2, 5), si)
    steps = G["steps"]
    sch = schedulers.UniPCMultistepScheduler(); ts = sch.set_timesteps(steps)
    cam, text, bev, boxes = cfg_inputs(D, csd, sc)
    sp = DN.SamplerPlan(cfg, un, cn, CPU, 1, True, 5, (28, 50), num_steps=steps, guidance_scale=G["guidance"], scheduler_kind="unipc", given_view_mode=mode)
    cl = given_view_inputs()[si:si + 1]
    mask = torch.tensor([[v is not None for v in r] for r in cl])
    lat = torch.zeros(1, 6, 4, 28, 50)
    for i, r in enumerate(cl):
        for j, v in enumerate(r):
            if v is not None:
                lat[i, j] = v
    sp.load_inputs(torch.stack([sc["latents"]] * 6, 1), cam, text, bev, boxes, ts, sch.coefficient_table(), given_mask=mask, given_latents=lat)
    # the first model call sees add_noise(cond, noise, t_0) in the given views (scheduling_unipc_multistep.py add_noise)
    x0 = sp.x.view(1, 6, 28, 50, 4).permute(0, 1, 4, 2, 3)
    gj = 0 if si == 0 else 5
    want = sch.add_noise(lat[0, gj], sc["latents"][0], ts[0])
    assert rel_l2(x0[0, gj], want) < 1e-6
    plan_interp.run(sp.prologue_ops)
    for _ in range(steps):
        plan_interp.run(sp.step_ops, lower_check=False)
    gold = G["latents_every" if mode == 1 else "latents_once"][si:si + 1]
    e = rel_l2(sp.latents(), gold)
    assert e < 4e-2, e
    # sensitivity: the reference's latents WITHOUT given views (same scenes, scheduler, steps) are far from this golden — the test
    # would not pass by ignoring gv_*
    plain = torch.load(os.path.join(GOLD, "tiny_pipeline_unipc.pt"))["latents_cfg"][si:si + 1]
    assert rel_l2(plain, gold) > 10 * e, (rel_l2(plain, gold), e)


def test_cxyz_bbox_mode_module_plan_matches_golden():
    """bbox_embedder mode='cxyz' (the reference CLASS default, bbox_embedder.py:41: 4 points per box, bbox_proj = Linear(4 * 27, .)) with
    minmax_normalize (the class default too): ControlNetPlan context tokens + UNet eps vs the REAL reference modules
    (tests/golden/tiny_forward_cxyz.pt, tools/make_golden.py cxyz)."""
    import copy
    cfg = copy.deepcopy(spec.TINY_CONFIG)
    cfg["controlnet"]["bbox"].update(mode="cxyz", n_corners=4, minmax_normalize=True)
    usd = spec.random_state_dict(spec.unet_param_shapes(cfg), 0)
    csd = spec.random_state_dict(spec.controlnet_param_shapes(cfg), 1)
    assert csd["bbox_embedder.bbox_proj.weight"].shape[1] == 4 * 27 and csd["bbox_embedder.null_pos_feature"].numel() == 108
    G = torch.load(os.path.join(GOLD, "tiny_forward_cxyz.pt"))
    assert abs(G["meta"]["cn_checksum"] - float(sum(v.double().abs().sum() for v in csd.values()))) < 1e-6 * G["meta"]["cn_checksum"]
    un, cn = PackedNet(bf16_round(usd), CPU), PackedNet(bf16_round(csd), CPU)
    sc = scene(cfg, 1, 5)
    boxes = dict(sc["bboxes_3d_data"]); boxes["bboxes"] = boxes["bboxes"][..., :4, :].contiguous() * 20.0
    lat = torch.randn(1, 6, 4, 28, 50, generator=torch.Generator().manual_seed(G["lat_seed"]))
    cp = DN.ControlNetPlan(cfg, cn, CPU, 1, 5, (28, 50))
    cp.sample_nchw.copy_(lat.reshape(6, 4, 28, 50))
    cp.temb.t.copy_(G["timesteps"].float().repeat_interleave(6))
    cp.cond.load(sc["camera_param"], sc["prompt_embeds"], sc["bev_map"], boxes)
    plan_interp.run(cp.ops, lower_check=False)
    ctx = cp.cond.ctx.float()
    e_box = rel_l2(ctx[:, 78:], G["ctx"].float()[:, 78:])
    assert e_box < 2e-2, e_box
    # the 8-corner embedding of the same boxes is a different function: the golden must be sensitive to the mode
    up = DN.UNetPlan(cfg, un, CPU, 6, ctx.shape[1], (28, 50))
    up.sample_nchw.copy_(lat.reshape(6, 4, 28, 50)); up.temb.t.copy_(G["timesteps"].float().repeat_interleave(6)); up.ctx.copy_(cp.cond.ctx)
    for dst, src in zip(up.res_in, cp.down_out):
        dst.copy_(src)
    up.mid_in.copy_(cp.mid_out)
    plan_interp.run(up.ops, lower_check=False)
    e = rel_l2(up.out_nchw, G["eps"].float())
    assert e < 4e-2, e


@pytest.mark.parametrize("mode", ["concat", "self"])
def test_unet_plan_neighboring_attn_modes(tiny, mode):
    """neighboring_attn_type concat / self: the UNet op graph (joint-softmax attention op over 2 / 6 sources, out-bias once) in the CPU
    interpreter vs the real reference's output (tests/golden/tiny_forward_nattn.pt)."""
    cfg0, usd, csd, un, cn = tiny
    G = torch.load(os.path.join(GOLD, "tiny_forward_nattn.pt"))
    cfg = dict(cfg0); cfg["neighboring_attn_type"] = mode
    sc = scene(cfg, 1, 3)
    lat = torch.randn(1, 6, 4, 28, 50, generator=torch.Generator().manual_seed(G["lat_seed"]))
    t = G["timesteps"]
    with torch.no_grad():
        d, m, ctx = D.controlnet_forward(csd, cfg, lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"])
    up = DN.UNetPlan(cfg, un, CPU, 6, ctx.shape[1], (28, 50))
    up.sample_nchw.copy_(lat.reshape(-1, 4, 28, 50)); up.temb.t.copy_(t.float().repeat_interleave(6)); up.ctx.copy_(ctx)
    for dst, src in zip(up.res_in, d):
        dst.copy_(src)
    up.mid_in.copy_(m)
    plan_interp.run(up.ops)
    per_view = max(rel_l2(up.out_nchw[i], G["eps_" + mode][i].float()) for i in range(6))
    assert per_view < 3e-2, per_view
    other = "self" if mode == "concat" else "concat"
    assert min(rel_l2(up.out_nchw[i], G["eps_" + other][i].float()) for i in range(6)) > 2 * per_view


@pytest.mark.parametrize("mode", ["gated", "none"])
def test_unet_plan_zero_module_types(tiny, mode):
    """zero_module_type gated / none: the connector folded into attn4.to_out at pack time (engine.PackedNet.gated_affine), the UNet op graph
    in the CPU interpreter vs the real reference's output (tests/golden/tiny_forward_zmod.pt)."""
    cfg0, _, csd, _, cn = tiny
    G = torch.load(os.path.join(GOLD, "tiny_forward_zmod.pt"))
    cfg = dict(cfg0); cfg["zero_module_type"] = mode
    usd = spec.random_state_dict(spec.unet_param_shapes(cfg), 0)
    un = PackedNet(usd, CPU)
    sc = scene(cfg, 1, 3)
    lat = torch.randn(1, 6, 4, 28, 50, generator=torch.Generator().manual_seed(G["lat_seed"]))
    t = G["timesteps"]
    with torch.no_grad():
        d, m, ctx = D.controlnet_forward(csd, cfg, lat, t, sc["camera_param"], sc["bboxes_3d_data"], sc["prompt_embeds"], sc["bev_map"])
    up = DN.UNetPlan(cfg, un, CPU, 6, ctx.shape[1], (28, 50))
    up.sample_nchw.copy_(lat.reshape(-1, 4, 28, 50)); up.temb.t.copy_(t.float().repeat_interleave(6)); up.ctx.copy_(ctx)
    for dst, src in zip(up.res_in, d):
        dst.copy_(src)
    up.mid_in.copy_(m)
    plan_interp.run(up.ops)
    per_view = max(rel_l2(up.out_nchw[i], G["eps_" + mode][i].float()) for i in range(6))
    assert per_view < 3e-2, per_view
    other = "none" if mode == "gated" else "gated"
    assert min(rel_l2(up.out_nchw[i], G["eps_" + other][i].float()) for i in range(6)) > 2 * per_view


def test_module_plans_hires_plus_map_encoder(tiny):
    """configs[3] shape: 54x96 latents + the ...Plus map encoder (adaptive average pool as one GEMM over pixels) through the op
    graphs vs the real reference's outputs."""
    cfg0, usd, csd, un, cn = tiny
    G = torch.load(os.path.join(GOLD, "tiny_forward_hires.pt"))
    hw = tuple(G["hw"])
    cfg = spec.with_plus_map_embedder(cfg0, hw)
    sc = scene(cfg, 1, 3, hw)
    lat = torch.randn(1, 6, 4, *hw, generator=torch.Generator().manual_seed(G["lat_seed"]))
    t = G["timesteps"]
    cp = DN.ControlNetPlan(cfg, cn, CPU, 1, 3, hw)
    cp.sample_nchw.copy_(lat.reshape(-1, 4, *hw)); cp.temb.t.copy_(t.float().repeat_interleave(6))
    cp.cond.load(sc["camera_param"], sc["prompt_embeds"], sc["bev_map"], sc["bboxes_3d_data"])
    plan_interp.run(cp.ops)
    assert rel_l2(cp.mid_out, G["mid"]) < 3e-2
    assert rel_l2(cp.down_out[0][:, :, ::9, ::12], G["down_first"]) < 3e-2      # carries the map feature (added after conv_in)
    # (the UNet at 54x96 — T0 = 5184 tokens — is checked against the same golden on the GPU: test_e2e_gpu.py, and by the oracle test;
    #  the CPU interpreter needs a minute for it)
    with pytest.raises(ValueError):          # the default encoder cannot produce a 54x96 feature: loud, with the config hint
        DN.ControlNetPlan(cfg0, cn, CPU, 1, 3, hw)


def test_vae_decode_plan_matches_golden():
    """The VAE decode op program (mid-block attention spelled out as GEMM / softmax / GEMM) in the CPU interpreter vs diffusers' output."""
    from magicdrive_amd.vae import VaeDecodePlan
    G = torch.load(os.path.join(GOLD, "tiny_vae_decode.pt"))
    vcfg = spec.VAE_TINY_CONFIG
    sd = spec.random_state_dict(spec.vae_decoder_param_shapes(vcfg), G["weights_seed"])
    plan = VaeDecodePlan(vcfg, PackedNet(sd, CPU), CPU, 2, (7, 13))
    plan.z_in.copy_(torch.randn(2, 4, 7, 13, generator=torch.Generator().manual_seed(G["z_seed"])))
    plan_interp.run(plan.ops)
    img = plan.out_nhwc.permute(0, 3, 1, 2)
    assert rel_l2(img, G["image"].float()) < 3e-2, rel_l2(img, G["image"].float())


def test_vae_encode_plan_matches_golden():
    """Round 4: the VAE ENCODE op program (what demo/run_cond_on_view.py:79-86 calls on its known views: conv_in, four down blocks whose
    stride-2 convs pad only bottom / right, mid block, conv_out, quant_conv) in the CPU interpreter vs diffusers' AutoencoderKL.encode
    (tests/golden/tiny_vae_encode.pt, tools/make_golden.py vaeenc): mean and log-variance of the latent distribution."""
    from magicdrive_amd.vae import VaeEncodePlan
    G = torch.load(os.path.join(GOLD, "tiny_vae_encode.pt"))
    vcfg = spec.VAE_TINY_CONFIG
    sd = spec.random_state_dict(spec.vae_decoder_param_shapes(vcfg), G["weights_seed"])
    sd.update(spec.random_state_dict(spec.vae_encoder_param_shapes(vcfg), G["weights_seed"] + 1000))
    assert abs(float(sum(v.double().abs().sum() for v in sd.values())) - G["checksum"]) < 1e-6 * G["checksum"]
    plan = VaeEncodePlan(vcfg, PackedNet(sd, CPU), CPU, 2, (56, 104))
    # the three downsampling convs are stride 2, pad 0 at the top / left and 1 at the bottom / right (resnet.py:215-217)
    downs = [op for op in plan.ops if getattr(op, "name", "").endswith("downsamplers.0.conv.")]
    assert len(downs) == 3 and all(op.stride == (2, 2) and op.pad == (0, 0) and op.pad_end == (1, 1) for op in downs)
    plan.x_in.copy_(torch.rand(2, 3, 56, 104, generator=torch.Generator().manual_seed(G["x_seed"])) * 2 - 1)
    plan_interp.run(plan.ops)
    mom = plan.moments_nhwc.permute(0, 3, 1, 2)
    assert tuple(mom.shape) == (2, 8, 7, 13)
    e_mean, e_lv = rel_l2(mom[:, :4], G["mean"]), rel_l2(mom[:, 4:], G["logvar"])
    assert e_mean < 3e-2 and e_lv < 3e-2, (e_mean, e_lv)


def test_fused_qkv_op_equals_separate_projections():
    """The level-0 fused q/k/v op (engine.self_like_attention) in the CPU interpreter == the q/k GEMM + batched V^T GEMM it replaces."""
    import magicdrive_amd.ops as O
    g = torch.Generator().manual_seed(0)
    Bv, T, C = 3, 16, 32
    x = torch.randn(Bv * T, C, generator=g).to(torch.bfloat16)
    w = (torch.randn(3 * C, C, generator=g) * C ** -0.5).to(torch.bfloat16)
    qk1 = torch.zeros(Bv * T, 2 * C, dtype=torch.bfloat16); vt1 = torch.zeros(Bv, C, T, dtype=torch.bfloat16)
    qk2 = torch.zeros_like(qk1); vt2 = torch.zeros_like(vt1)
    plan_interp.run([O.Gemm(x, w, qk1, Vt=vt1, vt_from=2 * C, vt_T=T)], lower_check=False)
    plan_interp.run([O.Gemm(x, w[:2 * C], qk2), O.Gemm(w[2 * C:], x.view(Bv, T, C), vt2)], lower_check=False)
    assert torch.equal(qk1, qk2) and torch.equal(vt1, vt2)


def test_step_program_work_is_deduplicated():
    """F_step(224x400, L=32, c=1) ~= 2.325 TF after removing the reference's redundant work;
    folding connector o attn4.to_out into one matrix (engine.py) removes a further 16 C x C GEMMs = 0.027 TF."""
    cfg = spec.SD15_CONFIG
    z = lambda shapes: {k: torch.zeros(1).expand(s) for k, s in shapes.items()}      # shape-only weights

    class ShapeNet(PackedNet):
        def _get(self, tag, keys, fn):
            ck = (tag,) + tuple(keys)
            if ck not in self.cache:
                small = [torch.zeros(self.sd[k].shape) for k in keys]
                self.cache[ck] = fn(*small)
            return self.cache[ck]
    un = ShapeNet(z(spec.unet_param_shapes(cfg)), CPU); cn = ShapeNet(z(spec.controlnet_param_shapes(cfg)), CPU)
    sp = DN.SamplerPlan(cfg, un, cn, CPU, 1, False, 32, (28, 50), num_steps=50)
    f = flops.program_flops(sp.step_ops)["total"] / 1e12
    fp = flops.program_flops(sp.prologue_ops)["total"] / 1e12
    assert abs(f - 2.298) < 0.01, f
    assert abs(fp - 0.045) < 0.005, fp


def test_fp16_sampler_plan_matches_golden_pipeline():
    """torch.float16 models build fp16 plans: every weight packed as fp16, every 16-bit buffer fp16, every op marked MDX_DTYPE_F16
    (the fp16 build of the kernels; the reference samples in fp16, magicdrive/misc/test_utils.py:95) — interpreted on the CPU vs the
    reference pipeline's golden latents."""
    from magicdrive_amd import _lib as L, ops as O
    cfg = spec.TINY_CONFIG
    usd, csd = state_dicts(cfg)
    un, cn = PackedNet(usd, CPU, torch.float16), PackedNet(csd, CPU, torch.float16)
    G = torch.load(os.path.join(GOLD, "tiny_pipeline.pt"))
    sc = one_scene(scene(cfg, 2, 5), 1)
    steps = G["steps"]
    sch = schedulers.DDIMScheduler(); ts = sch.set_timesteps(steps)
    cam, text, bev, boxes = cfg_inputs(D, csd, sc)
    sp = DN.SamplerPlan(cfg, un, cn, CPU, 1, True, 5, (28, 50), num_steps=steps, guidance_scale=G["guidance"])
    assert sp.dtype == torch.float16
    for op in sp.prologue_ops + sp.step_ops:
        ts16 = [v for v in vars(op).values() if isinstance(v, torch.Tensor) and v.element_size() == 2 and v.is_floating_point()]
        assert all(t.dtype == torch.float16 for t in ts16), getattr(op, "name", op)
        if ts16:
            assert O.dtype_code(op) == L.DTYPE_F16
    sp.load_inputs(torch.stack([sc["latents"]] * 6, 1), cam, text, bev, boxes, ts, sch.coefficient_table())
    plan_interp.run(sp.prologue_ops)
    for _ in range(steps):
        plan_interp.run(sp.step_ops, lower_check=False)
    e = rel_l2(sp.latents(), G["latents_cfg"][1:2])
    assert e < 2e-2, e            # fp16 activations (11-bit mantissa): measured ~0.2 %
    print(f"[fp16 plan, CPU interpreter vs reference golden] {e:.4f}")


@pytest.mark.parametrize("dtype,limit", [(torch.bfloat16, 6e-3), (torch.float16, 8e-4)])
def test_fused_layernorm_block_equals_separate_layernorm(monkeypatch, dtype, limit):
    """Level-0 transformer block (C = 320, 6 views x 1400 tokens): the emission with LayerNorm folded into the q/k/v, to_q and (round 6) GEGLU
    projections, the row statistics handed from each producer's store phase to the consumer (ops.Gemm.rowstat -> ln_stats)
    (engine.PackedNet.ln_lin / ln_geglu + ops.Gemm.ln_eps: raw tokens in, W diag(gamma), W beta, column sums) against the plain LayerNorm -> Linear
    emission of the same block, both through the CPU interpreter.  The two differ only in where the 16-bit rounding falls (W' vs x_hat):
    measured 4.1e-3 in bf16 (2^-9 steps), 8x less in fp16 — an algebra or packing error would show at the same size in both."""
    from magicdrive_amd import engine as E, ops as O
    g = torch.Generator().manual_seed(3)
    B, T, C, heads, S = 6, 1400, 320, 8, 20
    pre = "blk."
    sd = {}
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    for n in ("norm1", "norm2", "norm3", "norm4"):
        sd[pre + n + ".weight"] = 1.0 + rn(C, sc=0.3); sd[pre + n + ".bias"] = rn(C, sc=0.3)
    for a in ("attn1", "attn2", "attn4"):
        for k in ("to_q", "to_k", "to_v"):
            sd[f"{pre}{a}.{k}.weight"] = rn(C, C, sc=C ** -0.5)
        sd[f"{pre}{a}.to_out.0.weight"] = rn(C, C, sc=C ** -0.5); sd[f"{pre}{a}.to_out.0.bias"] = rn(C, sc=0.1)
    sd[pre + "connector.weight"] = rn(C, C, sc=C ** -0.5); sd[pre + "connector.bias"] = rn(C, sc=0.1)
    sd[pre + "ff.net.0.proj.weight"] = rn(8 * C, C, sc=C ** -0.5); sd[pre + "ff.net.0.proj.bias"] = rn(8 * C, sc=0.1)
    sd[pre + "ff.net.2.weight"] = rn(C, 4 * C, sc=(4 * C) ** -0.5); sd[pre + "ff.net.2.bias"] = rn(C, sc=0.1)
    x = (rn(B * T, C) * 1.5 + 0.7).to(dtype)          # a non-zero mean: the mean * csum term matters
    Kc = rn(B, S, C).to(dtype); Vtc = torch.zeros(B, C, 24, dtype=dtype); Vtc[:, :, :S] = rn(B, C, S).to(dtype)

    def run(fused):
        monkeypatch.setattr(E.Builder, "fuses_qkv", staticmethod((lambda B_, T_, C_: True) if fused else (lambda B_, T_, C_: False)))
        net = PackedNet(sd, CPU, dtype)
        b = E.Builder(spec.SD15_CONFIG, CPU, B, 6, ws_mb=1, dtype=dtype)
        h = b.pool.get((B * T, C)); h.copy_(x)
        hs = None
        if fused:       # the block's input statistics come from ITS producer (Transformer2DModel.proj_in): here an identity projection that emits them
            hs = b.rowstat(B * T)
            h0 = b.pool.get((B * T, C)); h0.copy_(x)
            b.gemm(h0, torch.eye(C).to(dtype), C, out=h, rowstat=hs, name="producer")
        out = b.transformer_block(net, pre, h, B, T, C, heads, {pre + "attn2.": (Kc, Vtc, S)}, "blk", h_stats=hs)
        kinds = [type(o).__name__ + (":ln" if getattr(o, "ln_eps", 0) > 0 else "") + (":rs" if getattr(o, "rowstat", None) is not None else "")
                 + (":st" if getattr(o, "ln_stats", None) is not None else "") for o in b.ops]
        plan_interp.run(b.ops)
        return out.float().clone(), kinds
    y1, k1 = run(True)
    y0, k0 = run(False)
    # round 6: norm1 / norm2 / norm4 take their row statistics from the projection that wrote their input; norm3 -> GEGLU keeps its own pass by default
    assert k1.count("Gemm:ln:st") == 3 and k1.count("LayerNorm") == 1 and k1.count("Gemm:rs") == 3, k1
    assert k0.count("Gemm:ln") == 0 and k0.count("LayerNorm") == 4 and not any(":rs" in k or ":st" in k for k in k0)
    assert rel_l2(y1, y0) < limit, rel_l2(y1, y0)
    monkeypatch.setattr(E.Builder, "FOLD_NORM3", True)       # the built-but-not-default form: norm3 folded into the GEGLU epilogue as well
    y2, k2 = run(True)
    assert k2.count("Gemm:ln:st") == 4 and k2.count("LayerNorm") == 0 and k2.count("Gemm:rs") == 4, k2
    assert rel_l2(y2, y0) < limit, rel_l2(y2, y0)


def test_forked_step_branches_share_no_buffer(tiny):
    """Round 5, the small-batch operating point: with fork=True the ControlNet and the UNet encoder of a step are emitted with separate buffer pools and
    workspaces so that they may run CONCURRENTLY (SamplerPlan.launch_step: ControlNet on a side stream, joined before the zero-convs).  Checked here on
    the interpreter: (1) no buffer written by one branch is touched by the other (the shared inputs x_in / context K, V / temb tables are read-only
    in both), (2) executing the encoder branch BEFORE the ControlNet gives bit-identical latents to the linear order, for several steps, (3) it is the
    same program as the unforked plan."""
    cfg, usd, csd, un, cn = tiny
    sc = one_scene(scene(cfg, 2, 5), 0)
    steps = 3
    sch = schedulers.DDIMScheduler(); ts = sch.set_timesteps(steps)
    cam, text, bev, boxes = cfg_inputs(D, csd, sc)
    outs = {}
    for name, fork, order in (("linear", False, None), ("forked_linear", True, None), ("forked_enc_first", True, "enc_first")):
        sp = DN.SamplerPlan(cfg, un, cn, CPU, 1, True, 5, (28, 50), num_steps=steps, guidance_scale=2.0, fork=fork)
        sp.load_inputs(torch.stack([sc["latents"]] * 6, 1), cam, text, bev, boxes, ts, sch.coefficient_table())
        plan_interp.run(sp.prologue_ops)
        for _ in range(steps):
            if order is None:
                plan_interp.run(sp.step_ops, lower_check=False)
            else:
                a, b = sp.fork_at
                plan_interp.run(sp.step_ops[a:b], lower_check=False)
                plan_interp.run(sp.step_ops[:a], lower_check=False)
                plan_interp.run(sp.step_ops[b:], lower_check=False)
        outs[name] = sp.latents().clone()
        if fork:
            a, b = sp.fork_at
            assert 0 < a < b < len(sp.step_ops)

            def storages(ops, written_only):
                got = set()
                for op in ops:
                    outs_ = {"C", "Y", "O", "Vt", "x", "x_in", "eps", "ln_scratch", "ws"}
                    for k, v in vars(op).items():
                        if isinstance(v, torch.Tensor) and v.numel() and (not written_only or k in outs_):
                            got.add(v.untyped_storage().data_ptr())
                return got
            w_cn, w_enc = storages(sp.step_ops[:a], True), storages(sp.step_ops[a:b], True)
            assert not (w_cn & storages(sp.step_ops[a:b], False)), "the encoder branch touches a buffer the ControlNet writes"
            assert not (w_enc & storages(sp.step_ops[:a], False)), "the ControlNet touches a buffer the encoder branch writes"
    assert torch.equal(outs["forked_linear"], outs["linear"]) and torch.equal(outs["forked_enc_first"], outs["linear"])


def test_pool_guard_mode_puts_every_buffer_at_the_end_of_its_own_storage():
    """engine.Pool.guard (a debugging aid, tools/batch_sweep.py --guard): each buffer ends within 15 bytes of the end of a storage of its own,
    16-byte aligned, and goes through the free list like any other; off by default."""
    from magicdrive_amd.engine import Pool
    assert Pool.guard is False
    Pool.guard = True
    try:
        pool = Pool(torch.device("cpu"))
        a = pool.get((7, 13, 40))
        b = pool.get((3, 5), torch.float32)
        for t in (a, b):
            es = t.element_size()
            end = t.storage_offset() * es + t.numel() * es
            total = t.untyped_storage().nbytes()
            assert total >= Pool.GUARD_SEGMENT and total % (2 << 20) == 0
            assert 0 <= total - end < 16 and (t.storage_offset() * es) % 16 == 0
            assert t.is_contiguous()
        assert a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()
        big = pool.get((16 << 20,))                                  # 32 MiB of bf16: the segment grows with the buffer
        assert big.untyped_storage().nbytes() == 32 << 20 and big.storage_offset() == 0
        pool.put(a)
        assert pool.get((7 * 13 * 40,)).data_ptr() == a.data_ptr()    # recycled, not re-allocated
    finally:
        Pool.guard = False
