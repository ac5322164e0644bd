"""AutoencoderKL — drop-in for the `vae` the reference pipeline decodes with (pipeline_bev_controlnet.py:100-112 -> diffusers
AutoencoderKL.decode, dif:models/autoencoder_kl.py:173-198) and, round 4, encodes the known views of the given-view demo with
(demo/run_cond_on_view.py:79-86: `pipe.vae.encode(x).latent_dist.mean * pipe.vae.config.scaling_factor`; dif:models/autoencoder_kl.py:127-171).

Only what those callers touch is built: `decode(z).sample`, `encode(x).latent_dist` (`.mean`, `.mode()`, `.sample(generator)`, `.logvar`,
`.std`), `.config.scaling_factor`, `.to()`, `.dtype`, `enable_slicing()` (accepted: the op programs already run one scene's views at a
time), `from_pretrained` on the reference checkpoint layout `<sd15>/vae/{config.json, diffusion_pytorch_model.(safetensors|bin)}`.  The
encoder / quant_conv tensors are optional: a decoder-only state dict still loads and `encode` then raises.  The arithmetic is op programs
on libmdx (magicdrive_amd/vae.py); there is no CPU path.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import spec
from ..engine import PackedNet
from ..vae import VaeDecodePlan, VaeEncodePlan
from ..denoiser import PlanCache

CONFIG_NAME = "config.json"
WEIGHTS_ST, WEIGHTS_BIN = "diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"


class DecoderOutput(SimpleNamespace):
    pass


class AutoencoderKLOutput(SimpleNamespace):
    pass


class DiagonalGaussianDistribution:
    """dif:models/vae.py:306-357 on the moments the encode program produced (fp32 [mean | logvar] along dim 1)."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        gdev = generator.device if isinstance(generator, torch.Generator) else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=gdev, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKL:
    def __init__(self, vcfg: Dict, state_dict: Dict[str, torch.Tensor], torch_dtype=torch.bfloat16):
        self.vcfg = dict(vcfg)
        shapes = spec.vae_decoder_param_shapes(self.vcfg)
        missing = [k for k in shapes if k not in state_dict]
        if missing:
            raise KeyError(f"AutoencoderKL: state dict lacks {len(missing)} decoder tensors, e.g. {missing[:4]}")
        for k, shp in shapes.items():
            if tuple(state_dict[k].shape) != tuple(shp):
                raise ValueError(f"AutoencoderKL: {k} has shape {tuple(state_dict[k].shape)}, config implies {tuple(shp)}")
        self._sd = OrderedDict((k, state_dict[k].detach()) for k in shapes)
        # encoder + quant_conv: optional (the sampler only decodes); all of them or none
        enc = spec.vae_encoder_param_shapes(self.vcfg)
        have = [k for k in enc if k in state_dict]
        self.has_encoder = len(have) == len(enc)
        if have and not self.has_encoder:
            raise KeyError(f"AutoencoderKL: state dict holds {len(have)} of the {len(enc)} encoder tensors, e.g. lacks {[k for k in enc if k not in state_dict][:3]}")
        if self.has_encoder:
            for k, shp in enc.items():
                if tuple(state_dict[k].shape) != tuple(shp):
                    raise ValueError(f"AutoencoderKL: {k} has shape {tuple(state_dict[k].shape)}, config implies {tuple(shp)}")
                self._sd[k] = state_dict[k].detach()
        self._dtype = torch_dtype
        self._device = torch.device("cpu")
        self._packed: Optional[PackedNet] = None
        self._plans = PlanCache()
        self.config = SimpleNamespace(**self.vcfg)
        self.max_images_per_pass = 6          # one scene's views per program run (bounds activation memory: 138 MB per 128-ch 224x400 map)

    @classmethod
    def from_config(cls, vcfg: Dict, seed: int = 0, torch_dtype=torch.bfloat16, with_encoder: bool = False):
        sd = spec.random_state_dict(spec.vae_decoder_param_shapes(vcfg), seed)
        if with_encoder:
            sd.update(spec.random_state_dict(spec.vae_encoder_param_shapes(vcfg), seed + 1000))
        return cls(vcfg, sd, torch_dtype)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=torch.bfloat16, subfolder: Optional[str] = None, **unused):
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, CONFIG_NAME)) as f:
            js = json.load(f)
        vcfg = dict(spec.VAE_SD15_CONFIG)
        for k in ("latent_channels", "out_channels", "layers_per_block", "norm_num_groups", "scaling_factor"):
            if k in js and js[k] is not None:
                vcfg[k] = js[k]
        if "block_out_channels" in js:
            vcfg["block_out_channels"] = tuple(js["block_out_channels"])
        if js.get("act_fn", "silu") != "silu":
            raise NotImplementedError(f"AutoencoderKL act_fn={js['act_fn']!r}: SD-1.5's VAE uses silu")
        if os.path.exists(os.path.join(d, WEIGHTS_ST)):
            from safetensors.torch import load_file
            sd = load_file(os.path.join(d, WEIGHTS_ST))
        elif os.path.exists(os.path.join(d, WEIGHTS_BIN)):
            sd = torch.load(os.path.join(d, WEIGHTS_BIN), map_location="cpu")
        else:
            raise FileNotFoundError(f"no {WEIGHTS_ST} / {WEIGHTS_BIN} under {d}")
        # checkpoints older than diffusers 0.17 name the attention projections query/key/value/proj_attn (attention_processor.py:_from_deprecated_attn_block)
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        fixed = {}
        for k, v in sd.items():
            for old, new in ren.items():
                tag = f".attentions.0.{old}."
                if tag in k:
                    k = k.replace(tag, f".attentions.0.{new}.")
                    if v.dim() == 4:
                        v = v.reshape(v.shape[0], v.shape[1])
            fixed[k] = v
        return cls(vcfg, fixed, torch_dtype)

    def save_pretrained(self, path: str, safe_serialization: bool = True):
        os.makedirs(path, exist_ok=True)
        js = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.vcfg.items()}
        js["_class_name"] = "AutoencoderKL"
        with open(os.path.join(path, CONFIG_NAME), "w") as f:
            json.dump(js, f, indent=2)
        sd = {k: v.contiguous() for k, v in self._sd.items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(path, WEIGHTS_ST))
        else:
            torch.save(sd, os.path.join(path, WEIGHTS_BIN))

    # ---- torch-module-like surface ----
    def state_dict(self):
        return self._sd

    def parameters(self):
        return iter(self._sd.values())

    def eval(self):
        return self

    def to(self, *args, **kw):
        """.to(device) / .to(dtype) / .to(device, dtype) like a torch module (dtype = API dtype of decode()'s output)."""
        for a in list(args) + list(kw.values()):
            if isinstance(a, torch.dtype):
                if (a == torch.float16) != (self._dtype == torch.float16):
                    self._packed = None      # the decoder's 16-bit arithmetic type follows the requested dtype
                    self._plans.clear()
                self._dtype = a
            elif isinstance(a, (str, torch.device)):
                dev = torch.device(a)
                if dev != self._device:
                    self._device, self._packed = dev, None
                    self._plans.clear()
        return self

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    def enable_slicing(self):
        """Accepted (val_set_gen.py:78): decode() already runs one scene's views per pass."""

    def disable_slicing(self):
        pass

    def packed(self) -> PackedNet:
        if self._packed is None:
            # arithmetic type: fp16 when the model was asked for in fp16 (what the reference samples in, misc/test_utils.py:95), bf16
            # otherwise (incl. fp32 requests: operands are 16-bit on the MFMA path either way, accumulation is fp32)
            self._packed = PackedNet(self._sd, self._device, torch.float16 if self._dtype == torch.float16 else torch.bfloat16)
        return self._packed

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x (n, 3, H, W) in [-1, 1], H and W multiples of 8 -> AutoencoderKLOutput(latent_dist=DiagonalGaussianDistribution) over
        (n, 4, H / 8, W / 8) latents (un-scaled: callers multiply by config.scaling_factor, demo/run_cond_on_view.py:84)."""
        if not self.has_encoder:
            raise ValueError("this AutoencoderKL was built from a decoder-only state dict: no encoder / quant_conv tensors to encode with")
        if self._device.type != "cuda":
            raise RuntimeError("AutoencoderKL.to('cuda') first: the encoder has no CPU path")
        n, _, H, W = x.shape
        if H % 8 or W % 8:
            raise ValueError(f"image size {H}x{W}: both sides must be multiples of 8 (three stride-2 stages)")
        outs = []
        for i0 in range(0, n, self.max_images_per_pass):
            xi = x[i0:i0 + self.max_images_per_pass]
            key = ("enc", xi.shape[0], H, W)
            plan = self._plans.get(key)
            if plan is None:
                with torch.cuda.device(self._device):
                    plan = VaeEncodePlan(self.vcfg, self.packed(), self._device, xi.shape[0], (H, W))
                    plan.compile()
                self._plans.put(key, plan)
            outs.append(plan.run(xi))
        moments = torch.cat(outs).to(x.dtype if x.dtype.is_floating_point else torch.float32)
        dist = DiagonalGaussianDistribution(moments)
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        """z (n, 4, h, w) (already divided by scaling_factor, as the pipeline does) -> DecoderOutput(sample=(n, 3, 8h, 8w))."""
        if self._device.type != "cuda":
            raise RuntimeError("AutoencoderKL.to('cuda') first: the decoder has no CPU path")
        n, _, h, w = z.shape
        outs = []
        for i0 in range(0, n, self.max_images_per_pass):
            zi = z[i0:i0 + self.max_images_per_pass]
            key = (zi.shape[0], h, w)
            plan = self._plans.get(key)
            if plan is None:
                with torch.cuda.device(self._device):
                    plan = VaeDecodePlan(self.vcfg, self.packed(), self._device, zi.shape[0], (h, w))
                    plan.compile()
                self._plans.put(key, plan)
            outs.append(plan.run(zi).to(z.dtype if z.dtype.is_floating_point else torch.float32))
        sample = torch.cat(outs)
        return DecoderOutput(sample=sample) if return_dict else (sample,)
