"""Schedulers of the sampling loop.

DDIMScheduler mirrors third_party/diffusers/src/diffusers/schedulers/scheduling_ddim.py
(config keys :117-190, alphas_cumprod :196-219, set_timesteps :287-323, step :325-445) for the
deterministic eta = 0, epsilon-prediction case the north-star names.  The per-step arithmetic runs
in the fused HIP kernel (mdx_cfg_ddim_step); this class only produces the timestep list and the
fp32 coefficient table {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)} the kernel indexes.

The reference pipeline raises RuntimeError for any scheduler whose step() takes `generator`
(pipeline_bev_controlnet.py:94-97) — which includes diffusers' DDIM.  Ours has no `generator`
parameter (eta = 0 is deterministic), so the guard passes unchanged.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", clip_sample: bool = False, set_alpha_to_one: bool = False,
                 steps_offset: int = 1, prediction_type: str = "epsilon"):
        if prediction_type != "epsilon":
            raise NotImplementedError("only epsilon prediction (SD-1.5) is built")
        if clip_sample:
            raise NotImplementedError("clip_sample=True is not what SD-1.5 / MagicDrive use")
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config):
        keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "clip_sample", "set_alpha_to_one", "steps_offset", "prediction_type")
        get = (lambda k: config[k]) if isinstance(config, dict) else (lambda k: getattr(config, k))
        kw = {}
        for k in keys:
            try:
                kw[k] = get(k)
            except (KeyError, AttributeError):
                pass
        return cls(**kw)

    def set_timesteps(self, num_inference_steps: int, device=None):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts) + self.steps_offset
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def alpha_pair(self, t: int):
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def coefficient_table(self) -> torch.Tensor:
        """fp32 [n_steps, 4]: sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev) per inference step."""
        rows = []
        for t in self.timesteps.tolist():
            a_t, a_p = self.alpha_pair(int(t))
            a_t, a_p = float(a_t), float(a_p)
            rows.append([a_t ** 0.5, (1 - a_t) ** 0.5, a_p ** 0.5, (1 - a_p) ** 0.5])
        return torch.tensor(rows, dtype=torch.float32)

    def step(self, model_output, timestep, sample, eta: float = 0.0, return_dict: bool = True):
        """Single DDIM update through the same fused HIP kernel the pipeline uses (for user code that drives
        the scheduler itself).  CUDA fp32 tensors only — there is no host implementation."""
        if eta != 0.0:
            raise NotImplementedError("eta > 0 needs variance noise; the drop-in builds the deterministic sampler")
        if not (model_output.is_cuda and sample.is_cuda):
            raise RuntimeError("DDIMScheduler.step runs on the GPU kernel; pass CUDA tensors")
        from . import ops as O
        a_t, a_p = self.alpha_pair(int(timestep))
        a_t, a_p = float(a_t), float(a_p)
        dev = sample.device
        coef = torch.tensor([[a_t ** 0.5, (1 - a_t) ** 0.5, a_p ** 0.5, (1 - a_p) ** 0.5]], dtype=torch.float32, device=dev)
        x = sample.detach().to(torch.float32).contiguous().clone()
        eps = model_output.detach().to(torch.float32).contiguous()
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        O.run_ops([O.DdimStep(x.view(-1), eps.view(-1), coef, step)])
        prev = x.to(sample.dtype)
        if not return_dict:
            return (prev,)

        class _Out:
            pass
        o = _Out()
        o.prev_sample = prev
        return o
