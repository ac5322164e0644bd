"""Shared fixtures for the parity tests: seeded tiny/real models, synthetic scenes, oracle runs."""
import torch

from magicdrive_amd import synthetic
from magicdrive_amd.networks import spec


def rel_l2(a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def state_dicts(cfg, seed_unet=0, seed_cn=1):
    return (spec.random_state_dict(spec.unet_param_shapes(cfg), seed_unet),
            spec.random_state_dict(spec.controlnet_param_shapes(cfg), seed_cn))


def bf16_round(sd):
    """Weights as the bf16 model holds them (parity is defined against the oracle on the SAME weights)."""
    return {k: v.to(torch.bfloat16).float() for k, v in sd.items()}


def scene(cfg, n=1, L=5, hw=(28, 50), seed=1234, **kw):
    return synthetic.make_scene_batch(n, seed=seed, ctx_dim=cfg["cross_attention_dim"], max_len=L, latent_hw=hw, **kw)


def cfg_inputs(oracle_D, csd, sc, n_cam=6):
    """[uncond | cond] input halves exactly as the pipeline assembles them (pipeline_bev_controlnet.py:330-343)."""
    nb = sc["latents"].shape[0]
    cam = torch.cat([oracle_D.uncond_cam_param(csd, nb, n_cam), sc["camera_param"]])
    text = torch.cat([sc["negative_prompt_embeds"], sc["prompt_embeds"]])
    bev = torch.cat([sc["bev_map"]] * 2)
    boxes = None
    if sc["bboxes_3d_data"] is not None:
        boxes = {k: torch.cat([torch.zeros_like(v), v]) for k, v in sc["bboxes_3d_data"].items()}
    return cam, text, bev, boxes


def given_view_inputs(hw=(28, 50)):
    """The known views of tests/golden/tiny_pipeline_given_view.pt (same generator as tools/make_golden.py: given_view_inputs)."""
    g = torch.Generator().manual_seed(77)
    cl = [[None] * 6 for _ in range(2)]
    for (i, j) in ((0, 0), (0, 3), (1, 5)):
        cl[i][j] = torch.randn(4, *hw, generator=g) * 0.8
    return cl
