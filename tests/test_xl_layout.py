"""CPU: the index arithmetic of gemm_xl.hip (LDS-DMA placement, fragment reads, bank mapping, unit ownership), replayed on the
host with the kernel's own header (magicdrive_amd/csrc/xl_layout.h) by tests/xl_layout_check.cpp."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_xl_layout_model(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "xl_layout_check")
    subprocess.run([gxx, "-O1", "-std=c++17", "-o", exe, os.path.join(HERE, "xl_layout_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "BN=256: ok" in r.stdout and "BN=160: ok" in r.stdout and "BN=320: ok" in r.stdout
