"""GPU: the C++ host-side mirror of the reference interface (include/winterfell_hip.hpp), exercised by a C++
test that reads like prover/src/trace/tests.rs.  The binary is built here with g++ against libwf_lde.so (the
product) and liboracle.so (the checker)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_cpp_mirror(orc, capi):
    capi.load()
    csrc = os.path.join(ROOT, "starkpack-winterfell_amd", "csrc")
    odir = os.path.join(ROOT, "oracle")
    exe = os.path.join(ROOT, "tests", "cpp", "test_prover_mirror")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_prover_mirror.cpp"),
                           "-L" + csrc, "-lwf_lde", "-L" + odir, "-loracle",
                           "-Wl,-rpath," + csrc, "-Wl,-rpath," + odir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout
