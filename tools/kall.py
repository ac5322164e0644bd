#!/usr/bin/env python
"""Launch ONE representative op per kernel family a few times in a single process — the target of the rocprofv3 passes of
tools/pmc_collect.sh (`--kernel-trace --stats` and one `--pmc` group per pass; kernels are told apart by name in the CSVs).
KONE_VIEWS (default 96) = c*b*6 views.  Prints, per case, the kernel the library routed it to and the launch's algorithmic bytes / FLOPs
(what roofline.achieved and the PMC traffic are compared with)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magicdrive_amd import _lib as L, ops as O, packing as PK  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda")
B = int(os.environ.get("KONE_VIEWS", "576"))       # the bench's plan: 192 scenes per call on two streams = 96 scenes x 6 views per step program
REPS = int(os.environ.get("KONE_REPS", "3"))
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(BF)
ws = torch.empty(64 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
cases = []


def conv(name, h, w, cin, cout):
    x = r(B, h, w, cin); wt = r(cout, 3, 3, cin); y = torch.empty(B, h, w, cout, dtype=BF, device=dev); R = r(B, h, w, cout)
    M = B * h * w
    cases.append((name, O.Conv(x, wt, y, bias=torch.randn(cout, device=dev), R=R, ws=ws), 2.0 * M * cout * 9 * cin,
                  dict(read=M * cin * 2 + cout * 9 * cin * 2 + M * cout * 2, write=M * cout * 2)))


def gemm(name, M, N, K, epi=0, res=False):
    A = r(M, K); W = r(N, K); No = N // 2 if epi == 1 else N
    C = torch.empty(M, No, dtype=BF, device=dev)
    cases.append((name, O.Gemm(A, W, C, bias=torch.randn(N, device=dev), R=r(M, No) if res else None, epilogue=epi, ws=ws), 2.0 * M * N * K,
                  dict(read=M * K * 2 + N * K * 2 + (M * No * 2 if res else 0), write=M * No * 2)))


def attn(name, T, C, xview):
    d = C // 8
    qk = r(B, T, 2 * C); vt = torch.zeros(B, C, PK.round_up(T, 8), dtype=BF, device=dev); vt[:, :, :T] = r(B, C, T)
    qk[:, :, :C] = (qk[:, :, :C].float() * (d ** -0.5 * 1.4426950408889634)).to(BF)      # pre-scaled Q, as the engine packs to_q
    o = torch.empty(B, T, C, dtype=BF, device=dev)
    kw = {}
    if xview:
        kw = dict(kvmap=torch.tensor([(i // 6) * 6 + ((i % 6 + s) % 6) for i in range(B) for s in (5, 1)], dtype=torch.int32, device=dev), nsrc=2)
    cases.append((name, O.Attn(qk[:, :, :C], qk[:, :, C:], vt, o, heads=8, Tk=T, scale=d ** -0.5, q_prescaled=True, **kw), (8.0 if xview else 4.0) * B * T * T * C,
                  dict(read=B * T * C * 2 * 3, write=B * T * C * 2)))


conv("conv_28x50_640_640", 28, 50, 640, 640)
conv("conv_28x50_320_320", 28, 50, 320, 320)
conv("conv_14x25_1280_1280", 14, 25, 1280, 1280)
gemm("geglu_L1", B * 350, 5120, 640, epi=1)
gemm("ffout_L0", B * 1400, 320, 1280, res=True)
gemm("geglu_L0", B * 1400, 2560, 320, epi=1)
gemm("out_L0", B * 1400, 320, 320, res=True)
gemm("cc_L1", B * 350, 640, 640, res=True)
gemm("qk_L1", B * 350, 1280, 640)
gemm("geglu_L2", B * 91, 10240, 1280, epi=1)
attn("attn_self_L0", 1400, 320, False)
attn("attn_xview_L0", 1400, 320, True)
# the text-context attention of level 0 (S = 1 + 77 tokens; K / V^T from the prologue): attn2_kernel<40,resident,...>
_q = r(B, 1400, 320); _kc = r(B, 78, 320); _vt = torch.zeros(B, 320, 80, dtype=BF, device=dev); _vt[:, :, :78] = r(B, 320, 78)
_q = (_q.float() * (40 ** -0.5 * 1.4426950408889634)).to(BF); _o = torch.empty(B, 1400, 320, dtype=BF, device=dev)
cases.append(("attn_ctx_L0", O.Attn(_q, _kc, _vt, _o, heads=8, Tk=78, scale=40 ** -0.5, q_prescaled=True), 4.0 * B * 1400 * 78 * 320,
              dict(read=B * 1400 * 320 * 2 + B * 78 * 320 * 2 * 2, write=B * 1400 * 320 * 2)))
x = r(B, 1400, 320); y = torch.empty_like(x)
cases.append(("gn_L0", O.GroupNorm(x, y, torch.ones(320, device=dev), torch.zeros(320, device=dev), 32, 1e-5, True, ws=ws), 0.0,
              dict(read=B * 1400 * 320 * 2 * 2, write=B * 1400 * 320 * 2)))
xl_ = r(B * 1400, 320); yl_ = torch.empty_like(xl_)
cases.append(("ln_L0", O.LayerNorm(xl_, yl_, torch.ones(320, device=dev), torch.zeros(320, device=dev)), 0.0,
              dict(read=B * 1400 * 320 * 2, write=B * 1400 * 320 * 2)))
st = torch.cuda.current_stream().cuda_stream
info = {}
for name, op, fl, by in cases:
    code, desc = op.lower()
    for _ in range(REPS):
        L.call_op(code, desc, st)
    torch.cuda.synchronize()
    info[name] = dict(kernel=(L.lib().mdx_last_kernel() or b"").decode(), flop=fl, alg_read_bytes=by["read"], alg_write_bytes=by["write"])
out = os.environ.get("KALL_INFO")
if out:
    with open(out, "w") as f:
        json.dump(dict(views=B, reps=REPS, build_id=L.build_id(), cases=info), f, indent=1)
print(json.dumps(info))
