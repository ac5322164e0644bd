"""CPU: the index arithmetic of attention3.hip (permuted K rows -> probabilities that are their own PV operand; V^T fragment addresses; O layout),
replayed on the host with the kernel's own header (magicdrive_amd/csrc/attn3_layout.h) by tests/attn3_layout_check.cpp."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_attn3_layout_model(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "attn3_layout_check")
    subprocess.run([gxx, "-O1", "-std=c++17", "-o", exe, os.path.join(HERE, "attn3_layout_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "attn3 layout: ok" in r.stdout
