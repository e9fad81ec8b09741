"""Compile the HIP library in-tree (csrc/libwf_lde.so) for gfx950.  hipcc cross-compiles without a GPU."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libwf_lde.so")


def build(force: bool = False) -> str:
    args = ["make", "-j8", "-C", CSRC, "libwf_lde.so", "libwf_yardstick.so"]
    if force:
        args.insert(1, "-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB


def yardstick_path() -> str:
    """bench.py's measurement helper (isolated rates of the path's instruction sequences); not a product library."""
    return os.path.join(CSRC, "libwf_yardstick.so")


def lib_path() -> str:
    # WF_LDE_LIB: load another build of the same library (tuning experiments: scripts/exp_variants.sh)
    return os.environ.get("WF_LDE_LIB") or LIB
