"""CPU: the 32-bit-limb f128 arithmetic of csrc/field.hpp (what the device code runs) against the wide
formulation and a bit-serial product.  Built with plain g++: the functions are __host__ __device__."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_f128_limb_arithmetic(tmp_path):
    exe = str(tmp_path / "test_field_limbs")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "test_field_limbs.cpp")])
    out = subprocess.run([exe, "400000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout
