"""ctypes binding of libmdx.so (the C-ABI declared in include/mdx.h).

The product path has NO fallback: if the shared library is missing or does not export the
ABI this module raises at import of the symbol table (`lib()`), and every op wrapper raises
`MdxError` on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDX_LIB_PATH") or os.path.join(_HERE, "libmdx.so")      # MDX_LIB_PATH: A/B a second build (tools only)

ABI_VERSION = 10

# opcodes (mdx.h)
OP_GEMM, OP_CONV, OP_CONV_DIRECT, OP_ATTN, OP_GROUPNORM, OP_LAYERNORM = 1, 2, 3, 4, 5, 6
OP_EW, OP_FOURIER, OP_GATHER, OP_TIMEEMB, OP_DDIM, OP_UNIPC, OP_SOFTMAX = 7, 8, 9, 10, 11, 12, 13
OP_BYTES = 512

EPI_NONE, EPI_GEGLU, EPI_SILU = 0, 1, 2
EW_ADD, EW_COPY, EW_UPSAMPLE, EW_NCHW_TO_NHWC, EW_NHWC_TO_NCHW, EW_SILU, EW_SCALE = 1, 2, 3, 4, 5, 6, 7

P, I, D = C.c_void_p, C.c_int64, C.c_double


def _struct(name: str, fields: List[Tuple[str, object]]):
    return type(name, (C.Structure,), {"_fields_": fields})


def _f(kind, names: str):
    return [(n, kind) for n in names.split()]


MdxGemmDesc = _struct("MdxGemmDesc", _f(P, "A W C R bias temb sel_ptr ws") + _f(I, "M N K lda ldw ldc ldr batch sA sW sC sR "
                      "temb_sel_stride temb_b_stride rows_per_b epilogue splitk c_is_f32 ws_bytes") + _f(P, "Vt") + _f(I, "vt_from vt_T vt_ld vt_stride")
                      + _f(D, "ln_eps") + _f(P, "ln_csum ln_scratch") + _f(P, "rowstat_out") + _f(I, "rowstat_parts") + _f(P, "ln_stats") + _f(I, "ln_stats_parts") + _f(P, "Wq"))
MdxConvDesc = _struct("MdxConvDesc", _f(P, "X Wt Y R bias temb sel_ptr ws") + _f(I, "B Hi Wi Cin Ho Wo Cout kh kw sh sw ph pw "
                      "ldx ldy ldr temb_sel_stride temb_b_stride epilogue splitk ws_bytes reserved1"))
MdxConvDirectDesc = _struct("MdxConvDirectDesc", _f(P, "X Wt Y R bias temb sel_ptr reserved_p") + _f(I, "B Hi Wi Cin Ho Wo Cout kh kw sh sw ph pw "
                            "ldx ldy ldr temb_sel_stride temb_b_stride epilogue x_is_f32 y_is_f32 reserved0"))
MdxAttnDesc = _struct("MdxAttnDesc", _f(P, "Q K Vt O kvmap reserved_p") + _f(I, "B H Tq Tk d nsrc ldq sQ ldk sK ldv sV ldo sO")
                      + _f(D, "scale") + _f(I, "joint q_prescaled"))
MdxGroupNormDesc = _struct("MdxGroupNormDesc", _f(P, "X Y gamma beta") + _f(I, "B HW C G ldx ldy") + _f(D, "eps") + _f(I, "silu") + _f(P, "ws") + _f(I, "ws_bytes"))
MdxLayerNormDesc = _struct("MdxLayerNormDesc", _f(P, "X Y gamma beta") + _f(I, "M C ldx ldy") + _f(D, "eps") + _f(I, "reserved0"))
MdxEwDesc = _struct("MdxEwDesc", _f(P, "X Y ymap xmap") + _f(I, "kind M C ldx ldy B Hi Wi Ho Wo x_is_f32 y_is_f32") + _f(D, "alpha"))
MdxFourierDesc = _struct("MdxFourierDesc", _f(P, "X Y mask null_feat") + _f(I, "n P F ldy"))
MdxGatherDesc = _struct("MdxGatherDesc", _f(P, "T Y idx mask null_row reserved_p") + _f(I, "n C ldt ldy n_rows reserved0"))
MdxTimeEmbDesc = _struct("MdxTimeEmbDesc", _f(P, "t Y") + _f(I, "n dim flip_sin_to_cos ldy") + _f(D, "freq_shift max_period"))
MdxDdimDesc = _struct("MdxDdimDesc", _f(P, "x eps coef step_ptr x_in reserved_p") + _f(I, "n cfg") + _f(D, "guidance") + _f(I, "xin_c xin_ld")
                      + _f(P, "gv_cond gv_noise gv_mask") + _f(I, "gv_mode gv_view_elems gv_last_step"))
MdxUniPCDesc = _struct("MdxUniPCDesc", _f(P, "x eps coef step_ptr x_in x_last m1 m2") + _f(I, "n cfg") + _f(D, "guidance") + _f(I, "xin_c xin_ld")
                       + _f(P, "gv_cond gv_noise gv_mask") + _f(I, "gv_mode gv_view_elems gv_last_step"))

MdxSoftmaxDesc = _struct("MdxSoftmaxDesc", _f(P, "X Y") + _f(I, "rows T ldx ldy") + _f(D, "scale") + _f(I, "reserved0"))
DESC_OF_OP = {
    OP_GEMM: MdxGemmDesc, OP_CONV: MdxConvDesc, OP_CONV_DIRECT: MdxConvDirectDesc, OP_ATTN: MdxAttnDesc,
    OP_GROUPNORM: MdxGroupNormDesc, OP_LAYERNORM: MdxLayerNormDesc, OP_EW: MdxEwDesc, OP_FOURIER: MdxFourierDesc,
    OP_GATHER: MdxGatherDesc, OP_TIMEEMB: MdxTimeEmbDesc, OP_DDIM: MdxDdimDesc, OP_UNIPC: MdxUniPCDesc, OP_SOFTMAX: MdxSoftmaxDesc,
}
DTYPE_BF16, DTYPE_F16 = 0, 1          # MdxOp.dtype: the 16-bit storage / MFMA operand type of an op (mdx.h)
ENTRY_OF_OP = {
    OP_GEMM: "mdx_gemm_bf16", OP_CONV: "mdx_conv2d_bf16", OP_CONV_DIRECT: "mdx_conv2d_direct", OP_ATTN: "mdx_attention_bf16",
    OP_GROUPNORM: "mdx_groupnorm_bf16", OP_LAYERNORM: "mdx_layernorm_bf16", OP_EW: "mdx_elementwise",
    OP_FOURIER: "mdx_fourier_embed", OP_GATHER: "mdx_gather_rows", OP_TIMEEMB: "mdx_timestep_embedding", OP_DDIM: "mdx_cfg_ddim_step",
    OP_UNIPC: "mdx_cfg_unipc_step", OP_SOFTMAX: "mdx_softmax_rows",
}
def entry_name(opcode: int, dtype: int = DTYPE_BF16) -> str:
    """C entry point of an op in the bf16 or the fp16 build (mdx_gemm_bf16 -> mdx_gemm_f16, mdx_elementwise -> mdx_elementwise_f16)."""
    n = ENTRY_OF_OP[opcode]
    if dtype == DTYPE_BF16:
        return n
    return (n[:-5] if n.endswith("_bf16") else n) + "_f16"


# every symbol include/mdx.h declares
EXPORTS = sorted(set(ENTRY_OF_OP.values()) | {entry_name(o, DTYPE_F16) for o in ENTRY_OF_OP} | {
    "mdx_program_run", "mdx_graph_create", "mdx_graph_launch", "mdx_graph_destroy",
    "mdx_abi_version", "mdx_build_id", "mdx_last_error", "mdx_last_kernel", "mdx_device_info", "mdx_set_option", "mdx_get_option", "mdx_option_name"})


class MdxOp(C.Structure):
    _fields_ = [("opcode", I), ("dtype", I), ("desc", C.c_ubyte * (OP_BYTES - 16))]


assert C.sizeof(MdxOp) == OP_BYTES
for _d in DESC_OF_OP.values():
    assert C.sizeof(_d) <= OP_BYTES - 16 and C.sizeof(_d) % 8 == 0


class MdxError(RuntimeError):
    pass


_LIB = None


def lib() -> C.CDLL:
    """Load libmdx.so (once).  Raises if it is missing or exports the wrong ABI — never falls back."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise MdxError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C magicdrive_amd/csrc`. magicdrive_amd has no CPU / PyTorch fallback.")
    l = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(l, name):
            raise MdxError(f"libmdx.so does not export {name}")
    for name in list(ENTRY_OF_OP.values()) + [entry_name(o, DTYPE_F16) for o in ENTRY_OF_OP]:
        fn = getattr(l, name)
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p]
    l.mdx_program_run.restype = C.c_int
    l.mdx_program_run.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    l.mdx_graph_create.restype = C.c_int
    l.mdx_graph_create.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    l.mdx_graph_launch.restype = C.c_int
    l.mdx_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
    l.mdx_graph_destroy.restype = C.c_int
    l.mdx_graph_destroy.argtypes = [C.c_void_p]
    l.mdx_abi_version.restype = C.c_int
    l.mdx_last_error.restype = C.c_char_p
    l.mdx_build_id.restype = C.c_char_p
    l.mdx_last_kernel.restype = C.c_char_p
    l.mdx_device_info.restype = C.c_int
    l.mdx_device_info.argtypes = [C.POINTER(C.c_int64)]
    l.mdx_set_option.restype = C.c_int
    l.mdx_set_option.argtypes = [C.c_char_p, C.c_int64]
    l.mdx_get_option.restype = C.c_int
    l.mdx_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
    l.mdx_option_name.restype = C.c_char_p
    l.mdx_option_name.argtypes = [C.c_int64]
    if l.mdx_abi_version() != ABI_VERSION:
        raise MdxError(f"libmdx.so ABI {l.mdx_abi_version()} != expected {ABI_VERSION}")
    _LIB = l
    return l


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().mdx_last_error()
        raise MdxError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def call_op(opcode: int, desc: C.Structure, stream: int, dtype: int = DTYPE_BF16) -> None:
    """Run a single op descriptor on `stream` (a raw hipStream_t value) in the bf16 or the fp16 build of its kernel."""
    name = entry_name(opcode, dtype)
    fn = getattr(lib(), name)
    check(fn(C.byref(desc), C.c_void_p(stream)), name)


class Program:
    """A flat array of MdxOp built from (opcode, descriptor[, dtype]) tuples; runnable eagerly or as a hipGraph."""

    def __init__(self, ops: List[Tuple]):
        self.n = len(ops)
        self.buf = (MdxOp * max(self.n, 1))()
        for i, t in enumerate(ops):
            code, desc = t[0], t[1]
            self.buf[i].opcode = code
            self.buf[i].dtype = t[2] if len(t) > 2 else DTYPE_BF16
            C.memmove(C.addressof(self.buf[i]) + 16, C.byref(desc), C.sizeof(desc))
        self._graph = C.c_void_p(None)

    def run(self, stream: int) -> None:
        check(lib().mdx_program_run(C.byref(self.buf), self.n, C.c_void_p(stream)), "mdx_program_run")

    def capture(self) -> None:
        if not self._graph:
            check(lib().mdx_graph_create(C.byref(self.buf), self.n, C.byref(self._graph)), "mdx_graph_create")

    def launch(self, stream: int) -> None:
        if not self._graph:
            self.capture()
        check(lib().mdx_graph_launch(self._graph, C.c_void_p(stream)), "mdx_graph_launch")

    def destroy(self) -> None:
        """Release the captured hipGraph (idempotent)."""
        if self._graph and _LIB is not None:
            _LIB.mdx_graph_destroy(self._graph)
        self._graph = C.c_void_p(None)

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def set_option(key: str, value: int) -> None:
    """Set a routing / tuning switch of the library (csrc/options.h) for this process; unknown keys raise MdxError."""
    check(lib().mdx_set_option(key.encode(), int(value)), f"mdx_set_option({key})")


def get_option(key: str) -> int:
    v = C.c_int64(0)
    check(lib().mdx_get_option(key.encode(), C.byref(v)), f"mdx_get_option({key})")
    return int(v.value)


def option_names() -> List[str]:
    out, i = [], 0
    while True:
        n = lib().mdx_option_name(i)
        if not n:
            return out
        out.append(n.decode()); i += 1


class options:
    """`with _lib.options(GEMM_XL=2, XL_BN=160): ...` — set switches for a block, restore the previous values after it."""

    def __init__(self, **kv):
        self.kv = kv
        self.prev: Dict[str, int] = {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.prev[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_option(k, v)
        return False


def build_id() -> str:
    """Source hash of the loaded libmdx.so (mdx_build_id)."""
    return (lib().mdx_build_id() or b"unknown").decode()


def device_info() -> Dict[str, int]:
    out = (C.c_int64 * 3)()
    check(lib().mdx_device_info(out), "mdx_device_info")
    return {"cus": out[0], "clock_khz": out[1], "hbm_bytes": out[2]}
