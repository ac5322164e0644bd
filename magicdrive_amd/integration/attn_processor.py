"""MdxAttnProcessor — the reference-side binding of ONE libmdx kernel, in the reference's own operator protocol.

diffusers' attention-processor protocol is `processor(attn, hidden_states, encoder_hidden_states=None, attention_mask=None,
temb=None) -> Tensor`, installed with `Attention.set_processor` (third_party/diffusers/src/diffusers/models/
attention_processor.py:176-292); `XFormersAttnProcessor` (:1081-1190) is the instance the reference selects with
`pipe.enable_xformers_memory_efficient_attention()` (magicdrive/misc/test_utils.py:131-132) and whose
`xformers.ops.memory_efficient_attention` call (:1165-1171) this class replaces.  It talks to libmdx.so through ctypes only — plain
pointers and 8-byte fields (include/mdx.h: MdxAttnDesc), no magicdrive_amd Python above the C-ABI — so it is exactly the stub that
would live next to XFormersAttnProcessor in the reference tree.  torch is used for what the reference itself uses it for here:
the q/k/v/out nn.Linear layers of the `Attention` module and device memory.

Not the product path: the sampler hands libmdx whole op programs (magicdrive_amd/denoiser.py); this file exists so that
INTEGRATION.md §2 is executable (tests/test_integration_gpu.py runs it against F.scaled_dot_product_attention).
"""
from __future__ import annotations

import ctypes
import os

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libmdx.so")
_lib = None


class MdxAttnDesc(ctypes.Structure):               # mirrors include/mdx.h MdxAttnDesc field for field
    _fields_ = [(n, ctypes.c_void_p) for n in "Q K Vt O kvmap reserved_p".split()] + \
               [(n, ctypes.c_int64) for n in "B H Tq Tk d nsrc ldq sQ ldk sK ldv sV ldo sO".split()] + \
               [("scale", ctypes.c_double), ("joint", ctypes.c_int64), ("q_prescaled", ctypes.c_int64)]   # q_prescaled = 0: plain Q


def _mdx():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not found: build it with `make -C magicdrive_amd/csrc`; there is no fallback")
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.mdx_attention_bf16.restype = ctypes.c_int
        _lib.mdx_attention_bf16.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.mdx_last_error.restype = ctypes.c_char_p
    return _lib


class MdxAttnProcessor:
    """softmax(Q K^T * scale) V on the fused gfx950 attention kernel; q/k/v/out projections stay the module's own layers."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        if attention_mask is not None:
            raise NotImplementedError("MagicDrive's sampler never passes an attention mask (blocks.py:144-238)")
        if not hidden_states.is_cuda:
            raise RuntimeError("MdxAttnProcessor runs on the GPU kernel; there is no CPU path")
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q = attn.to_q(hidden_states).to(torch.bfloat16).contiguous()             # [B, Tq, C]
        k = attn.to_k(ctx).to(torch.bfloat16).contiguous()                        # [B, Tk, C]
        tk = k.shape[1]
        ldv = (tk + 7) // 8 * 8
        vt = torch.zeros(k.shape[0], k.shape[2], ldv, dtype=torch.bfloat16, device=k.device)
        vt[:, :, :tk] = attn.to_v(ctx).to(torch.bfloat16).transpose(1, 2)         # V^T, kv-padded to a multiple of 8
        o = torch.empty_like(q)
        d = MdxAttnDesc(Q=q.data_ptr(), K=k.data_ptr(), Vt=vt.data_ptr(), O=o.data_ptr(), B=q.shape[0], H=attn.heads,
                        Tq=q.shape[1], Tk=tk, d=q.shape[2] // attn.heads, nsrc=1, ldq=q.stride(1), sQ=q.stride(0),
                        ldk=k.stride(1), sK=k.stride(0), ldv=ldv, sV=vt.stride(0), ldo=o.stride(1), sO=o.stride(0), scale=float(attn.scale))
        with torch.cuda.device(q.device):
            rc = _mdx().mdx_attention_bf16(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream))
        if rc:
            raise RuntimeError((_mdx().mdx_last_error() or b"").decode())
        o = o.to(hidden_states.dtype)
        return attn.to_out[1](attn.to_out[0](o))
# attn.set_processor(MdxAttnProcessor())   — same call site as set_use_memory_efficient_attention_xformers (:176-292)
