"""Seeded synthetic scenes with the shapes MagicDrive's nuScenes collate_fn produces
(magicdrive/dataset/utils.py:253-352) — there is no dataset offline.

Per scene: prompt / negative-prompt embeddings (77, D), an 8-channel 200x200 BEV map of random
axis-aligned rectangles, camera_param (6, 3, 7) = [K | R | t], and up to L padded 3-D boxes per view
as 8 corners x xyz with classes and masks.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

VIEW_YAW_DEG = (55.0, 0.0, -55.0, -110.0, 180.0, 110.0)   # FL, F, FR, BR, B, BL (configs/dataset/Nuscenes.yaml:19-25)


def _gen(seed: int) -> torch.Generator:
    return torch.Generator().manual_seed(seed)


def bev_map(g: torch.Generator, channels: int = 8, size: int = 200) -> torch.Tensor:
    m = torch.zeros(channels, size, size)
    for c in range(channels):
        n_rect = 12 if c == 0 else 3
        for _ in range(n_rect):
            h = int(torch.randint(8, 60 if c == 0 else 25, (1,), generator=g))
            w = int(torch.randint(8, 60 if c == 0 else 25, (1,), generator=g))
            y = int(torch.randint(0, size - h, (1,), generator=g))
            x = int(torch.randint(0, size - w, (1,), generator=g))
            m[c, y:y + h, x:x + w] = 1.0
    return m


def camera_param(g: torch.Generator, n_cam: int = 6) -> torch.Tensor:
    K = torch.tensor([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]])
    swap = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])   # camera -> lidar axis convention
    out = torch.zeros(n_cam, 3, 7)
    for i in range(n_cam):
        yaw = math.radians(VIEW_YAW_DEG[i % len(VIEW_YAW_DEG)])
        Rz = torch.tensor([[math.cos(yaw), -math.sin(yaw), 0.0], [math.sin(yaw), math.cos(yaw), 0.0], [0.0, 0.0, 1.0]])
        R = Rz @ swap
        t = torch.rand(3, generator=g) * torch.tensor([3.4, 1.0, 0.2]) + torch.tensor([-1.7, -0.5, 1.4])
        out[i, :, :3] = K
        out[i, :, 3:6] = R
        out[i, :, 6] = t
    return out


def boxes(g: torch.Generator, n_cam: int = 6, max_len: int = 32, n_classes: int = 10) -> Dict[str, torch.Tensor]:
    centre = torch.rand(n_cam, max_len, 3, generator=g) * torch.tensor([100.0, 100.0, 3.0]) + torch.tensor([-50.0, -50.0, -2.0])
    size = torch.rand(n_cam, max_len, 3, generator=g) * 9.5 + 0.5
    signs = torch.tensor([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=torch.float32)
    corners = centre[:, :, None, :] + 0.5 * size[:, :, None, :] * signs[None, None]
    classes = torch.randint(0, n_classes, (n_cam, max_len), generator=g)
    n_valid = torch.randint(0, max_len + 1, (n_cam,), generator=g)
    masks = torch.arange(max_len)[None, :] < n_valid[:, None]
    # the dataset pads with zeros / class -1 where mask is 0 (dataset/utils.py:210-240)
    corners = corners * masks[:, :, None, None]
    classes = torch.where(masks, classes, torch.full_like(classes, -1))
    return {"bboxes": corners, "classes": classes, "masks": masks}


def make_scene_batch(batch: int, seed: int = 1234, ctx_dim: int = 768, max_len: Optional[int] = 32, latent_hw=(28, 50),
                     n_cam: int = 6, with_camera: bool = True, zero_map: bool = False, map_size: int = 200) -> Dict[str, object]:
    """Inputs for `batch` scenes; scene i uses seed + i so shards of a batch reproduce the same scenes."""
    pe, ne, maps, cams, bb, cl, mk, lat = [], [], [], [], [], [], [], []
    for i in range(batch):
        g = _gen(seed + i)
        pe.append(torch.randn(77, ctx_dim, generator=g))
        ne.append(torch.randn(77, ctx_dim, generator=g))
        maps.append(torch.zeros(8, map_size, map_size) if zero_map else bev_map(g, size=map_size))
        cams.append(camera_param(g, n_cam))
        if max_len:
            b = boxes(g, n_cam, max_len)
            bb.append(b["bboxes"]); cl.append(b["classes"]); mk.append(b["masks"])
        lat.append(torch.randn(4, *latent_hw, generator=g))
    out: Dict[str, object] = {
        "prompt_embeds": torch.stack(pe), "negative_prompt_embeds": torch.stack(ne),
        "bev_map": torch.stack(maps), "camera_param": torch.stack(cams) if with_camera else None,
        "latents": torch.stack(lat),
        "bboxes_3d_data": {"bboxes": torch.stack(bb), "classes": torch.stack(cl), "masks": torch.stack(mk)} if max_len else None,
    }
    return out
