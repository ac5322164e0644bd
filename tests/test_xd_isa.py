"""CPU: static guard for csrc/gemm_xd.hip.  The kernel's register loads are inline asm whose completion the compiler does not track; a
register-to-register copy of such a value between the load and the wait that covers it copies stale bits (round 6: phi copies of the weight
fragments at the tile loop's back-edge made ~10 % of the launches wrong).  The source is written so that no such copy exists; this test
compiles it to ISA and checks that: no VGPR -> VGPR moves (DPP row rotates of the epilogue excepted), no scratch, no vmcnt(0) inside the
main loop of any instantiation."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "magicdrive_amd", "csrc", "gemm_xd.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="no hipcc")
def test_xd_kernel_isa_has_no_register_copies_or_scratch():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "gemm_xd.s")
        cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=fast", "-mllvm", "-amdgpu-mfma-vgpr-form",
               "--cuda-device-only", "-S", SRC, "-o", out]
        subprocess.run(cmd, check=True, cwd=os.path.dirname(SRC))
        text = open(out).read()
    kernels = re.findall(r"^(_ZN\w*gemm_xd_kernel\w*):\s*;.*?\n(.*?)\n\s*s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) == 3, [k[0] for k in kernels]
    for name, body in kernels:
        moves = [l.strip() for l in body.split("\n")
                 if re.match(r"\s*v_mov_b(32|64)(_e32|_e64)?\s+v[\[\d]", l) and re.search(r",\s*v[\[\d]", l) and "dpp" not in l and "row_" not in l]
        assert not moves, f"{name}: VGPR -> VGPR copies in the kernel (an asm load's register may be copied before its wait): {moves[:6]}"
        assert "scratch_" not in body, f"{name}: scratch accesses (their reloads wait vmcnt(0) inside the hand-counted pipeline)"
    stats = re.findall(r";\s*ScratchSize:\s*(\d+)", text)
    assert stats and all(int(x) == 0 for x in stats), stats
