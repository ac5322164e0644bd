"""CPU: the C-ABI library builds, loads, exports every symbol include/mdx.h declares, and the ctypes mirrors
have the C structs' sizes (no compute calls — there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

from magicdrive_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "mdx.h")


def test_library_loads_and_exports_header_symbols():
    lib = L.lib()
    src = open(HDR).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(mdx_\w+)\s*\(", src, flags=re.M))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mdx_abi_version() == L.ABI_VERSION


def test_ctypes_structs_match_c_sizes(tmp_path):
    names = ["MdxGemmDesc", "MdxConvDesc", "MdxConvDirectDesc", "MdxAttnDesc", "MdxGroupNormDesc", "MdxLayerNormDesc",
             "MdxEwDesc", "MdxFourierDesc", "MdxGatherDesc", "MdxTimeEmbDesc", "MdxDdimDesc", "MdxUniPCDesc", "MdxSoftmaxDesc", "MdxOp"]
    c = tmp_path / "sz.c"
    c.write_text('#include <stdio.h>\n#include "mdx.h"\nint main(){' + "".join(f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    for n, s in zip(names, sizes):
        assert ctypes.sizeof(getattr(L, n)) == s, n


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(L, "_LIB", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libmdx.so")
    with pytest.raises(L.MdxError):
        L.lib()


def test_bad_descriptor_is_rejected_without_a_gpu():
    d = L.MdxGemmDesc()
    rc = L.lib().mdx_gemm_bf16(ctypes.byref(d), None)
    assert rc == -1 and b"null operand" in L.lib().mdx_last_error()


def test_integration_binding_struct_matches_the_abi():
    """The reference-side binding of INTEGRATION.md §2 declares MdxAttnDesc by hand: it must stay byte-compatible with include/mdx.h
    (a stale copy passes garbage in the new trailing fields — caught on the GPU only as a run-time error otherwise)."""
    import ctypes
    from magicdrive_amd import _lib as L
    from magicdrive_amd.integration import attn_processor as AP
    assert ctypes.sizeof(AP.MdxAttnDesc) == ctypes.sizeof(L.MdxAttnDesc)
    assert [f[0] for f in AP.MdxAttnDesc._fields_] == [f[0] for f in L.MdxAttnDesc._fields_]
