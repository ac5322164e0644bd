"""Worker of tests/test_routes_gpu.py::test_forced_routes: the library reads its routing switches (MDX_GEMM_XL, MDX_XL_BN, ...) from the
environment ONCE per process, so every forced route runs in its own interpreter.  Usage: python tests/route_worker.py xl320|xl256|xl160|noxl|geglu320xl|attn_q32|attn_d80|attn_old
Prints ROUTE_WORKER_OK on success; any failure raises."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from magicdrive_amd import _lib as L, ops as O, packing as PK  # noqa: E402
from test_routes_gpu import rnd, close, ws_buf, run_one, conv_ref  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda")


def gemm_case(M, N, K, bias=True, res=False, epi=0, expect=None):
    A = rnd(M, K, seed=1); W = rnd(N, K, scale=K ** -0.5, seed=2)
    b = rnd(N, seed=3, dtype=torch.float32) if bias else None
    R = rnd(M, N, seed=4) if res else None
    C = torch.full((M, N), float("nan"), dtype=BF, device=dev)
    k = run_one(O.Gemm(A, W, C, bias=b, R=R, epilogue=epi, ws=ws_buf()))
    ref = A.float() @ W.float().T
    if bias: ref += b
    if epi == 2: ref = F.silu(ref)
    if res: ref += R.float()
    close(C, ref, name=f"gemm {M}x{N}x{K} ({k})")
    assert expect is None or k == expect, (k, expect)
    return k


def geglu_case(M, F_, K, expect=None):
    A = rnd(M, K, seed=1); W = rnd(2 * F_, K, scale=K ** -0.5, seed=2, dtype=torch.float32); b = rnd(2 * F_, seed=3, dtype=torch.float32)
    Wp, bp = PK.pack_geglu(W.cpu(), b.cpu())
    C = torch.zeros(M, F_, dtype=BF, device=dev)
    k = run_one(O.Gemm(A, Wp.to(dev), C, bias=bp.to(dev), epilogue=L.EPI_GEGLU, ws=ws_buf()))
    h, g = (A.float() @ W.to(BF).float().T + b).chunk(2, dim=-1)
    close(C, h * F.gelu(g), name=f"geglu {M}x{F_}x{K} ({k})")
    assert expect is None or k == expect, (k, expect)


def conv_case(B, H, W, Cin, Cout, stride=(1, 1), res=True, temb=True, expect=None):
    x = rnd(B, H, W, Cin, seed=1)
    w = rnd(Cout, Cin, 3, 3, scale=(Cin * 9) ** -0.5, seed=2, dtype=torch.float32); b = rnd(Cout, seed=3, dtype=torch.float32)
    Ho = (H - 1) // stride[0] + 1; Wo = (W - 1) // stride[1] + 1
    y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=BF, device=dev)
    R = rnd(B, Ho, Wo, Cout, seed=4) if res else None
    tb = rnd(B, Cout, seed=5, dtype=torch.float32) if temb else None
    k = run_one(O.Conv(x, PK.pack_conv_weight(w.cpu()).to(dev), y, bias=b, R=R, temb=tb, temb_b_stride=Cout if temb else 0, stride=stride, pad=(1, 1), ws=ws_buf()))
    close(y, conv_ref(x, w, b, stride, (1, 1), tb, R), name=f"conv {B}x{H}x{W} {Cin}->{Cout} ({k})")
    assert expect is None or k == expect, (k, expect)


mode = sys.argv[1]
if mode.startswith("xl"):
    bn = int(mode[2:])
    assert os.environ.get("MDX_XL_BN") == str(bn) and os.environ.get("MDX_GEMM_XL") == "2"
    g, c = f"gemm_xl_kernel<256x{bn},gemm>", f"gemm_xl_kernel<256x{bn},conv>"
    gemm_case(2000, 640, 640, res=True, expect=g)                 # ragged M (7.8 tiles), N = 2-4 tiles
    gemm_case(777, 324, 128, bias=False, expect=g)                # N % 8 != 0: narrow stores; ragged everything; two slabs
    gemm_case(5000, 320, 64, epi=2, expect=g)                     # one slab, SiLU
    gemm_case(3000, 1280, 960, res=True, expect=g)                # 15 slabs (K < 1024: no automatic split-K)
    conv_case(6, 28, 50, 320, 320, expect=c)                      # level-0 resnet conv
    conv_case(12, 14, 25, 128, 640, res=False, expect=c)          # 350-px images, 2 channel blocks
    conv_case(40, 7, 13, 64, 320, temb=True, expect=c)            # 91-px images: temb slots
    conv_case(100, 4, 7, 128, 160, expect=c)                      # 28-px images: 10 images per tile
    conv_case(6, 28, 50, 64, 320, stride=(2, 2), res=False, temb=False, expect=c)
    if bn == 256:
        geglu_case(3000, 640, 320, expect=g)
elif mode == "noxl":
    assert os.environ.get("MDX_GEMM_XL") == "0"
    # the round-1 main loops at the shapes the review asked for
    gemm_case(8736, 1280, 1280, res=True, expect="gemm_pp_kernel<256x256,gemm>")
    geglu_case(8736, 5120, 1280, expect="gemm_pp_kernel<256x256,gemm>")
    gemm_case(4500, 2560, 1280, res=True, expect="gemm_pp_kernel<256x256,gemm>")          # ragged M
    conv_case(96, 7, 13, 1280, 1280, expect="gemm_pp_kernel<256x256,conv>")                 # 8736 rows, tiles span 3-4 images (temb slots)
    conv_case(24, 14, 25, 1920, 1280, res=False, expect="gemm_pp_kernel<256x256,conv>")
    conv_case(16, 28, 50, 320, 320, expect="conv3x3_kernel")
    conv_case(600, 4, 7, 320, 320, expect="conv3x3_kernel")                                   # 4x7 images
    conv_case(22, 28, 28, 640, 640, expect="conv3x3_kernel")                                  # 28-px rows straddling 128-row tiles
    gemm_case(537600, 320, 320, res=True, expect="gemm_ws_kernel<plain>")                     # bench row count
    geglu_case(26400, 1280, 320, expect="gemm_ws_kernel<geglu>")                              # K = 320 GEGLU on the weight-stationary kernel
elif mode == "geglu320xl":
    assert os.environ.get("MDX_XL_GEGLU320") == "1"
    from route_worker_helpers import geglu_check
    k = geglu_check(26400, 1280, 320)
    assert k == "gemm_xl_kernel<256x256,gemm>", k
elif mode in ("attn_q32", "attn_d80", "attn_old"):
    # attention2.hip's other instantiations (32-query waves; head dim 80) and attention.hip at the same shapes, through the tests
    # of tests/test_kernels_gpu.py (their route assertion follows this process's switches)
    import test_kernels_gpu as T
    assert {"attn_q32": os.environ.get("MDX_ATTN2_QT") == "1", "attn_d80": os.environ.get("MDX_ATTN2_D80") == "1",
            "attn_old": os.environ.get("MDX_ATTN2") == "0"}[mode]
    for case in T.ATTN2_CASES:
        T.test_attention2(dev, *case)
    T.test_attention2_softmax_rescale_branch(dev)
    for case in [(1, 8, 1400, 40), (3, 8, 350, 80), (2, 8, 700, 40)]:
        T.test_attention2_crossview(dev, *case)
else:
    raise SystemExit(f"unknown mode {mode}")
print("ROUTE_WORKER_OK")
