"""-m gpu: INTEGRATION.md §2 executed — the reference-side attention-processor binding (ctypes on the C-ABI only) against
torch's F.scaled_dot_product_attention through the same q/k/v/out layers (what diffusers' AttnProcessor2_0 computes,
third_party/diffusers/src/diffusers/models/attention_processor.py:1193-1272)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from magicdrive_amd.integration.attn_processor import MdxAttnProcessor


class Attention(torch.nn.Module):
    """The members of diffusers' `Attention` the processors touch (attention_processor.py:40-167): to_q/to_k/to_v without bias,
    to_out = [Linear, Dropout], heads, scale."""

    def __init__(self, query_dim, cross_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_q = torch.nn.Linear(query_dim, inner, bias=False)
        self.to_k = torch.nn.Linear(cross_dim, inner, bias=False)
        self.to_v = torch.nn.Linear(cross_dim, inner, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(inner, query_dim), torch.nn.Dropout(0.0)])
        self.processor = None

    def set_processor(self, p):
        self.processor = p

    def forward(self, hidden_states, encoder_hidden_states=None):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states)


def sdpa_processor(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
    ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
    B = hidden_states.shape[0]
    sp = lambda t: t.view(B, -1, attn.heads, t.shape[-1] // attn.heads).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(attn.to_q(hidden_states)), sp(attn.to_k(ctx)), sp(attn.to_v(ctx)), scale=attn.scale)
    o = o.transpose(1, 2).reshape(B, -1, attn.heads * o.shape[-1])
    return attn.to_out[1](attn.to_out[0](o))


@pytest.mark.parametrize("C,cross,heads,T,S", [(320, None, 8, 1400, None), (640, 768, 8, 350, 110), (1280, 768, 8, 91, 78)])
def test_attn_processor_binding_matches_sdpa(dev, C, cross, heads, T, S):
    torch.manual_seed(0)
    attn = Attention(C, cross or C, heads, C // heads).to(dev)
    x = torch.randn(6, T, C, device=dev)
    ctx = torch.randn(6, S, cross, device=dev) if cross else None
    with torch.no_grad():
        attn.set_processor(sdpa_processor); ref = attn(x, ctx)
        attn.set_processor(MdxAttnProcessor()); out = attn(x, ctx)
    rel = ((out - ref).norm() / ref.norm()).item()
    assert out.shape == ref.shape and torch.isfinite(out).all() and rel < 1.5e-2, rel      # bf16 q/k/v/p rounding inside the kernel
    with pytest.raises(NotImplementedError):
        MdxAttnProcessor()(attn, x, ctx, attention_mask=torch.ones(1, device=dev))
    with pytest.raises(RuntimeError):
        MdxAttnProcessor()(attn.cpu(), x.cpu())
