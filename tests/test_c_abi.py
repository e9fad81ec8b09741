"""include/wf_lde.h is plain C: tests/c/test_abi.c is compiled with gcc -std=c99 -Wall -Wextra -Werror -pedantic and
run -- without a GPU it exercises the host-side entry points and checks that context creation fails loudly (exit 77);
on the GPU box it runs a commitment, a resident commitment and queries through the ABI as a C host would."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "starkpack-winterfell_amd", "csrc")


def build_and_run(capi):
    capi.load()  # builds libwf_lde.so if it is missing
    exe = os.path.join(tempfile.mkdtemp(prefix="wf_abi_"), "test_abi")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "test_abi.c"), "-L", CSRC, "-lwf_lde", f"-Wl,-rpath,{CSRC}", "-o", exe])
    return subprocess.run([exe], capture_output=True, text=True, timeout=300)


def test_header_is_c99_and_host_side_runs_without_a_gpu(capi):
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present: covered by the gpu test")
    out = build_and_run(capi)
    assert out.returncode == 77, out.stdout + out.stderr
    assert "compute skipped" in out.stdout


@pytest.mark.gpu
def test_c_client_commits_and_queries(capi):
    out = build_and_run(capi)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_abi: ok" in out.stdout
