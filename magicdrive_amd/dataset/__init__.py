"""mm-free input path: `.pth` samples -> the tensors StableDiffusionBEVControlNetPipeline.__call__ takes."""
from .samples import (box_corners, collate_samples, load_overrides, load_sample, precompute_cam_ext, preprocess_bbox,  # noqa: F401
                      preprocess_fn, FolderSet)
