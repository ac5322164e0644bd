#!/usr/bin/env python
"""Golden vectors for the mm-free input path (magicdrive_amd/dataset/samples.py): synthetic `.pth`-format samples pushed through the
REFERENCE's own demo/helper.py (precompute_cam_ext :495-504, preprocess_fn :507-586, LiDARInstance3DBoxes.corners :150-186),
imported from /root/reference.  demo/helper.py imports cv2 and omegaconf at module level only for its drawing / config helpers; they
are absent here and stubbed (nothing on the preprocessing path touches them).
Writes tests/golden/sample_preprocess.pt = the samples + the reference's outputs.  Run in the authoring container only."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def import_reference_helper():
    for name in ("cv2", "omegaconf"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "omegaconf":
                m.OmegaConf = type("OmegaConf", (), {})
            sys.modules[name] = m
    sys.path.insert(0, os.path.join(REF, "demo"))
    import helper as H
    return H


def make_sample(seed: int, n_box: int):
    """A sample in the demo/data format (demo/readme.md:3-22) with nuScenes-like camera geometry (6 cameras around the ego, the
    camera frame's z looking outward) so that the per-view visibility filter keeps a different subset of boxes per camera."""
    g = torch.Generator().manual_seed(seed)
    K = torch.eye(4).repeat(6, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = 1266.4; K[:, 0, 2] = 816.3; K[:, 1, 2] = 491.5
    l2c = torch.eye(4).repeat(6, 1, 1)
    for v, yaw_deg in enumerate((55.0, 0.0, -55.0, -110.0, 180.0, 110.0)):
        a = np.deg2rad(yaw_deg)
        fwd = np.array([np.cos(a), np.sin(a), 0.0]); left = np.array([-np.sin(a), np.cos(a), 0.0]); up = np.array([0.0, 0.0, 1.0])
        R = np.stack([-left, -up, fwd])                       # lidar -> camera: x right, y down, z forward
        t = torch.rand(3, generator=g).numpy() * np.array([3.4, 1.0, 0.2]) - np.array([1.7, 0.5, -1.4])
        l2c[v, :3, :3] = torch.from_numpy(R).float()
        l2c[v, :3, 3] = torch.from_numpy(-R @ t).float()
    aug = torch.eye(4).repeat(6, 1, 1)
    aug[:, 0, 0] = aug[:, 1, 1] = 0.25; aug[:, 1, 3] = -1.0
    boxes = torch.zeros(n_box, 9)
    if n_box:
        boxes[:, :2] = torch.rand(n_box, 2, generator=g) * 100 - 50
        boxes[:, 2] = torch.rand(n_box, generator=g) * 3 - 2
        boxes[:, 3:6] = torch.rand(n_box, 3, generator=g) * 9.5 + 0.5
        boxes[:, 6] = torch.rand(n_box, generator=g) * 6.28 - 3.14
    return {"img": torch.rand(6, 3, 8, 16, generator=g), "gt_bboxes_3d": boxes, "gt_labels_3d": torch.randint(0, 10, (n_box,), generator=g),
            "gt_masks_bev": (torch.rand(8, 20, 20, generator=g) > 0.7).numpy().astype(np.uint8), "camera_intrinsics": K, "lidar2camera": l2c,
            "img_aug_matrix": aug, "metas": {"location": "boston-seaport", "description": f"Rain, scene {seed}", "timeofday": "day", "token": f"tok{seed}"}}


def main():
    H = import_reference_helper()
    out = {"cases": []}
    for seed, n_box in ((1, 23), (2, 5), (3, 0), (4, 1), (5, 60)):
        s = make_sample(seed, n_box)
        ex = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in s.items()}
        ex["gt_bboxes_3d"] = ex["gt_bboxes_3d"][:, :7]                       # FolderSetWrapper (dataset_wrapper.py:37)
        ex = H.precompute_cam_ext(ex)
        r = H.preprocess_fn(ex, "A driving scene image at {location}. {description}.", None)
        case = {"sample": s, "camera_param": r["camera_param"], "bev": r["bev_map_with_aux"], "captions": r["captions"],
                "boxes": r["kwargs"]["bboxes_3d_data"], "camera2lidar": ex["camera2lidar"], "lidar2image": ex["lidar2image"]}
        if n_box:
            case["corners"] = H.LiDARInstance3DBoxes(ex["gt_bboxes_3d"], box_dim=7, origin=(0.5, 0.5, 0)).corners
        out["cases"].append(case)
        b = case["boxes"]
        print(seed, n_box, None if b is None else (tuple(b["bboxes"].shape), b["masks"].sum(-1).tolist()))
    p = os.path.join(ROOT, "tests", "golden", "sample_preprocess.pt")
    torch.save(out, p)
    print("wrote", p, os.path.getsize(p))


if __name__ == "__main__":
    main()
