"""-m gpu: the headline configurations at SD-1.5 SIZE against fixtures made by the REAL reference (tools/make_golden.py sd15 /
sd15cfg / sd15hires: /root/reference's StableDiffusionBEVControlNetPipeline.__call__, pipeline_bev_controlnet.py:349-451, and its
BEVControlNetModel / UNet2DConditionModelMultiview forwards, fp32 arithmetic on the bf16-rounded seeded weights).  No CPU oracle runs
here, so the full 50-step loop costs GPU seconds.  Plus the reference's two other shipped resolutions at tiny width.

Tolerances are <= 2x what was measured on MI355X (recorded per test in profiles/r03_parity_measured.jsonl): the bf16 path
differs from the fp32 reference by the activation rounding only (same weights)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import check, parity_log, rel_l2, scene
from magicdrive_amd.networks import spec

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pipe(cfg, dev):
    from magicdrive_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    from magicdrive_amd.networks.unet_addon_rawbox import BEVControlNetModel
    from magicdrive_amd.pipeline.pipeline_bev_controlnet import StableDiffusionBEVControlNetPipeline
    unet = UNet2DConditionModelMultiview.from_config(cfg, 0); cn = BEVControlNetModel.from_config(cfg, 1)
    return StableDiffusionBEVControlNetPipeline(unet=unet, controlnet=cn).to(dev), unet, cn


def _checksum(sd):
    return float(sum(v.to(torch.bfloat16).double().abs().sum() for v in sd.values()))


@pytest.fixture(scope="module")
def sd15_pipe(dev):
    return _pipe(spec.SD15_CONFIG, dev)


def _check_weights(G, unet, cn):
    """The fixture was produced with the bf16-rounded seeded weights: a drifted init must fail here, not as a parity miss."""
    assert abs(_checksum(unet.state_dict()) - G["meta"]["unet_checksum"]) <= 1e-6 * G["meta"]["unet_checksum"]
    assert abs(_checksum(cn.state_dict()) - G["meta"]["cn_checksum"]) <= 1e-6 * G["meta"]["cn_checksum"]


def test_sd15_50step_ddim_loop_vs_reference(dev, sd15_pipe):
    """BASELINE configs[1] — the bench workload — for the full 50 DDIM steps vs the real reference pipeline's latents."""
    pipe, unet, cn = sd15_pipe
    G = torch.load(os.path.join(GOLD, "sd15_loop50.pt"), weights_only=False)
    _check_weights(G, unet, cn)
    cfg = spec.SD15_CONFIG
    sc = scene(cfg, 1, None, (28, 50), zero_map=True)
    trace = {}

    def cb(i, t, lat):
        if (i + 1) in G["trace"]:
            trace[i + 1] = lat.float().cpu().clone()
    out = pipe(prompt=None, image=sc["bev_map"], camera_param=None, height=224, width=400, num_inference_steps=G["steps"], guidance_scale=G["guidance"],
               latents=sc["latents"], prompt_embeds=sc["prompt_embeds"], negative_prompt_embeds=sc["negative_prompt_embeds"],
               output_type="latent", callback=cb, callback_steps=1, bev_controlnet_kwargs={"bboxes_3d_data": None}).images
    torch.cuda.synchronize()
    ref = G["latents"].float()
    per_view = [rel_l2(out[:, v], ref[:, v]) for v in range(6)]
    tr = {k: max(rel_l2(trace[k][:, v], G["trace"][k].float()[:, v]) for v in range(6)) for k in sorted(trace)}
    print(f"[sd15 50-step DDIM vs REAL reference] worst view {max(per_view):.4f}; trace (step: worst view) {tr}")
    parity_log("sd15_50step_ddim_loop_vs_reference", worst_view_rel_l2=max(per_view), all_rel_l2=rel_l2(out, ref),
               trace={str(k): round(v, 5) for k, v in tr.items()}, absmean=ref.abs().mean().item())
    assert torch.isfinite(out).all() and sorted(trace) == sorted(G["trace"])
    check("sd15 50-step DDIM loop vs REAL reference: worst view", max(per_view), 1e-2)        # measured 0.41 %
    check("sd15 50-step DDIM loop vs REAL reference: worst trace point", max(tr.values()), 1e-2)


def test_sd15_cfg_loop_full_conditioning_vs_reference(dev, sd15_pipe):
    """BASELINE configs[2]: camera + 32 padded boxes per view + BEV map + CFG 2.0, 10 DDIM steps vs the real reference pipeline."""
    pipe, unet, cn = sd15_pipe
    G = torch.load(os.path.join(GOLD, "sd15_loop_cfg.pt"), weights_only=False)
    _check_weights(G, unet, cn)
    cfg = spec.SD15_CONFIG
    sc = scene(cfg, 1, 32, (28, 50))
    trace = {}

    def cb(i, t, lat):
        if (i + 1) in G["trace"]:
            trace[i + 1] = lat.float().cpu().clone()
    out = pipe(prompt=None, image=sc["bev_map"], camera_param=sc["camera_param"], height=224, width=400, num_inference_steps=G["steps"],
               guidance_scale=G["guidance"], latents=sc["latents"], prompt_embeds=sc["prompt_embeds"], negative_prompt_embeds=sc["negative_prompt_embeds"],
               output_type="latent", callback=cb, callback_steps=1, bev_controlnet_kwargs={"bboxes_3d_data": sc["bboxes_3d_data"]}).images
    torch.cuda.synchronize()
    ref = G["latents"].float()
    per_view = [rel_l2(out[:, v], ref[:, v]) for v in range(6)]
    tr = {k: max(rel_l2(trace[k][:, v], G["trace"][k].float()[:, v]) for v in range(6)) for k in sorted(trace)}
    print(f"[sd15 10-step CFG loop, camera + boxes + map vs REAL reference] worst view {max(per_view):.4f}; trace {tr}")
    parity_log("sd15_cfg_loop_full_conditioning_vs_reference", worst_view_rel_l2=max(per_view), all_rel_l2=rel_l2(out, ref),
               trace={str(k): round(v, 5) for k, v in tr.items()})
    assert torch.isfinite(out).all()
    check("sd15 10-step CFG loop vs REAL reference: worst view", max(per_view), 1.5e-2)        # measured 0.72 %
    check("sd15 10-step CFG loop vs REAL reference: worst trace point", max(tr.values()), 1.5e-2)
    # View order / conditioning sensitivity: the six views of this fixture start from ONE noise
    # (pipeline_bev_controlnet.py:326) and differ only through camera / boxes / neighbours — by 0.6-1.25 % of the signal, less than the
    # per-view limit above, so a view mix-up would pass it.  The DIFFERENCE to view 0 is what a mix-up changes by O(1) — but in bf16 the
    # arithmetic noise (0.66 % per view, largely UNcorrelated between views) is as large as that difference: measured 0.50-1.06 relative
    # (profiles/r04a_parity_measured.jsonl), so it cannot carry an assertion here.  It is logged; the fp16 build (0.07 % noise) carries the
    # asserted form of this check AND the swapped-camera mutation that proves its sensitivity: tests/test_fp16_gpu.py.
    diff = view_differential(out, ref)
    print(f"[sd15 CFG loop: view differential vs REAL reference, bf16 (report only)] {[round(d, 3) for d in diff]}")
    parity_log("sd15_cfg_loop_view_differential_bf16", worst=max(diff), per_view=[round(d, 4) for d in diff])


def view_differential(out, ref):
    """rel L2 of (out[:, v] - out[:, 0]) against (ref[:, v] - ref[:, 0]) for v = 1..5."""
    out = out.float().cpu(); ref = ref.float().cpu()
    return [rel_l2(out[:, v] - out[:, 0], ref[:, v] - ref[:, 0]) for v in range(1, out.shape[1])]


def sd15_given_view_inputs(hw=(28, 50)):
    """The known views of tests/golden/sd15_loop_given_view.pt (tools/make_golden.py: sd15_given_view_inputs)."""
    g = torch.Generator().manual_seed(78)
    cl = [[None] * 6]
    for j in (0, 3):
        cl[0][j] = torch.randn(4, *hw, generator=g) * 0.8
    return cl


def run_sd15_given_view(pipe_cls_args, dev, G, half=False, camera_param=None):
    from magicdrive_amd.pipeline.pipeline_bev_controlnet_given_view import StableDiffusionBEVControlNetGivenViewPipeline
    unet, cn = pipe_cls_args
    pipe = StableDiffusionBEVControlNetGivenViewPipeline(unet=unet, controlnet=cn).to(dev)
    sc = scene(spec.SD15_CONFIG, 1, 32, (28, 50))
    pe, ne = sc["prompt_embeds"], sc["negative_prompt_embeds"]
    out = pipe(prompt=None, image=sc["bev_map"], camera_param=sc["camera_param"] if camera_param is None else camera_param, height=224, width=400,
               conditional_latents=sd15_given_view_inputs(), conditional_latents_change_every_input=True, num_inference_steps=G["steps"],
               guidance_scale=G["guidance"], latents=sc["latents"], prompt_embeds=pe.half() if half else pe,
               negative_prompt_embeds=ne.half() if half else ne, output_type="latent",
               bev_controlnet_kwargs={"bboxes_3d_data": sc["bboxes_3d_data"]}).images
    torch.cuda.synchronize()
    return out


def test_sd15_given_view_loop_vs_reference(dev, sd15_pipe):
    """A headline-size loop whose views GENUINELY differ: the REAL reference's given-view pipeline (pipeline_bev_controlnet_given_view.py:
    263-296) at SD-1.5 size, camera + 32 boxes + map, CFG 2.0, 10 DDIM steps, views 0 and 3 given and re-noised every step
    (tools/make_golden.py sd15given).  The given views end near their clean latents, O(1) away from the sampled ones, and every sampled
    view has exactly one given neighbour — on a different side for views 1 / 4 than for 2 / 5: the cross-view indexing is live."""
    pipe, unet, cn = sd15_pipe
    G = torch.load(os.path.join(GOLD, "sd15_loop_given_view.pt"), weights_only=False)
    _check_weights(G, unet, cn)
    out = run_sd15_given_view((unet, cn), dev, G)
    ref = G["latents"].float()
    per_view = [rel_l2(out[:, v], ref[:, v]) for v in range(6)]
    spread = [rel_l2(ref[:, v], ref[:, 1]) for v in range(6)]
    print(f"[sd15 10-step given-view loop vs REAL reference] per view {[round(e, 4) for e in per_view]}; reference view-to-view-1 distance {[round(x, 3) for x in spread]}")
    parity_log("sd15_given_view_loop_vs_reference", worst_view_rel_l2=max(per_view), per_view=[round(e, 5) for e in per_view])
    assert spread[0] > 0.5 and spread[3] > 0.5, "the fixture's given views must differ from the sampled ones by O(1)"
    check("sd15 10-step given-view loop vs REAL reference: worst view", max(per_view), 2e-2)


def _module_forward(cfg, G, dev, n_box=3, map_size=200):
    from magicdrive_amd.networks.unet_2d_condition_multiview import UNet2DConditionModelMultiview
    from magicdrive_amd.networks.unet_addon_rawbox import BEVControlNetModel
    hw = tuple(G["hw"])
    unet = UNet2DConditionModelMultiview.from_config(cfg, 0).to(dev); cn = BEVControlNetModel.from_config(cfg, 1).to(dev)
    sc = scene(cfg, 1, n_box, hw, map_size=map_size)
    lat = torch.randn(1, 6, 4, *hw, generator=torch.Generator().manual_seed(G["lat_seed"]))
    t = G["timesteps"]
    down, mid, ctx = cn(lat.to(dev), t.to(dev), sc["camera_param"].to(dev), {k: v.to(dev) for k, v in sc["bboxes_3d_data"].items()},
                        sc["prompt_embeds"].to(dev), sc["bev_map"].to(dev), return_dict=False)
    eps = unet(lat.reshape(-1, 4, *hw).to(dev), t.repeat_interleave(6).to(dev), encoder_hidden_states=ctx,
               down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
    torch.cuda.synchronize()
    return unet, cn, down, mid, eps


def test_sd15_forward_hires_vs_reference(dev):
    """BASELINE configs[3] shape at REAL width: 54x96 latents (T0 = 5184 tokens), ...Plus map encoder, one ControlNet + UNet pass of
    6 views vs the real reference modules."""
    G = torch.load(os.path.join(GOLD, "sd15_forward_hires.pt"), weights_only=False)
    cfg = spec.with_plus_map_embedder(spec.SD15_CONFIG, tuple(G["hw"]))
    unet, cn, down, mid, eps = _module_forward(cfg, G, dev)
    _check_weights(G, unet, cn)
    e = max(rel_l2(eps[i], G["eps"][i].float()) for i in range(6))
    em = rel_l2(mid[:, ::4], G["mid_sub"].float())
    dm = max(abs(x.float().abs().mean().item() / g.item() - 1.0) for x, g in zip(down, G["down_absmean"]))
    print(f"[sd15 54x96 + Plus map encoder vs REAL reference] eps per-view max rel {e:.4f}, mid residual {em:.4f}, down |x| ratio dev {dm:.4f}")
    parity_log("sd15_forward_hires_vs_reference", eps_worst_view_rel_l2=e, mid_rel_l2=em, down_absmean_dev=dm)
    check("sd15 hires forward vs REAL reference: eps per view", e, 2e-2)                       # measured 1.0 %
    check("sd15 hires forward vs REAL reference: mid residual", em, 2.4e-2)                    # measured 1.2 %
    check("sd15 hires forward vs REAL reference: down |x| ratio", dm, 1e-2)


def test_sd15_hires_cfg_loop_vs_reference(dev):
    """BASELINE configs[3] as a LOOP at real width (round 4; VERDICT r3 row g1: "no loop-level golden at 54x96"): 432x768, ...Plus map
    encoder, camera + 3 boxes + BEV map, CFG 2.0, DDIM vs the REAL reference pipeline's latents (tools/make_golden.py sd15hiresloop)."""
    G = torch.load(os.path.join(GOLD, "sd15_loop_hires.pt"), weights_only=False)
    hw = tuple(G["hw"])
    cfg = spec.with_plus_map_embedder(spec.SD15_CONFIG, hw)
    pipe, unet, cn = _pipe(cfg, dev)
    _check_weights(G, unet, cn)
    sc = scene(cfg, 1, 3, hw)
    trace = {}

    def cb(i, t, lat):
        if (i + 1) in G["trace"]:
            trace[i + 1] = lat.float().cpu().clone()
    out = pipe(prompt=None, image=sc["bev_map"], camera_param=sc["camera_param"], height=hw[0] * 8, width=hw[1] * 8, num_inference_steps=G["steps"],
               guidance_scale=G["guidance"], latents=sc["latents"], prompt_embeds=sc["prompt_embeds"], negative_prompt_embeds=sc["negative_prompt_embeds"],
               output_type="latent", callback=cb, callback_steps=1, bev_controlnet_kwargs={"bboxes_3d_data": sc["bboxes_3d_data"]}).images
    torch.cuda.synchronize()
    ref = G["latents"].float()
    assert tuple(out.shape) == tuple(ref.shape) == (1, 6, 4) + hw
    per_view = [rel_l2(out[:, v], ref[:, v]) for v in range(6)]
    tr = {k: max(rel_l2(trace[k][:, v], G["trace"][k].float()[:, v]) for v in range(6)) for k in sorted(trace)}
    print(f"[sd15 432x768 {G['steps']}-step CFG loop vs REAL reference] worst view {max(per_view):.4f}; trace {tr}")
    parity_log("sd15_hires_cfg_loop_vs_reference", worst_view_rel_l2=max(per_view), all_rel_l2=rel_l2(out, ref),
               trace={str(k): round(v, 5) for k, v in tr.items()})
    assert torch.isfinite(out).all() and sorted(trace) == sorted(G["trace"])
    check("sd15 432x768 CFG loop vs REAL reference: worst view", max(per_view), 1.3e-2)        # measured 0.61 %
    check("sd15 432x768 CFG loop vs REAL reference: worst trace point", max(tr.values()), 1.3e-2)


@pytest.mark.parametrize("which", ["272x736", "424x800"])
def test_reference_resolutions_tiny(dev, which):
    """configs/exp/272x736.yaml:15-22 (34x92 latents, ...Plus [34, 92]) and configs/exp/424x800abox0.1_nockpt.yaml:15-17 (53x100 latents,
    400x400 BEV maps through the plain embedder) at tiny width vs the real reference modules."""
    import copy
    G = torch.load(os.path.join(GOLD, f"tiny_forward_{which}.pt"), weights_only=False)
    hw = tuple(G["hw"])
    if which == "272x736":
        cfg = spec.with_plus_map_embedder(spec.TINY_CONFIG, hw); ms = 200
    else:
        cfg = copy.deepcopy(spec.TINY_CONFIG); cfg["controlnet"]["map_size"] = (8, 400, 400); ms = 400
    unet, cn, down, mid, eps = _module_forward(cfg, G, dev, map_size=ms)
    e = max(rel_l2(eps[i], G["eps"][i].float()) for i in range(6))
    em = rel_l2(mid, G["mid"])
    print(f"[{which} vs REAL reference, tiny width] eps per-view max rel {e:.4f}, mid {em:.4f}")
    parity_log(f"reference_resolution_{which}_tiny", eps_worst_view_rel_l2=e, mid_rel_l2=em)
    check(f"{which} tiny vs REAL reference: eps per view", e, 3.6e-2)                           # measured 1.8 %
    check(f"{which} tiny vs REAL reference: mid residual", em, 3e-2)
