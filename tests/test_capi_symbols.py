"""The C-ABI library loads without a GPU and exports every function include/wf_lde.h declares (no compute here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "wf_lde.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(capi):
    lib = capi.load()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libwf_lde.so does not export {n}"
    assert set(capi.SYMBOLS) == set(names)
    # every binding declares its argument types: an undeclared size_t argument would travel as a 32-bit int
    no_args = {"wf_last_error", "wf_device_count", "wf_comm_rccl_version", "wf_comm_rccl_path"}
    for n in names:
        assert getattr(lib, n).argtypes is not None or n in no_args, f"capi.load() sets no argtypes for {n}"


def test_size_helpers_and_param_validation_need_no_gpu(capi):
    lib = capi.load()
    p = capi.make_params(capi.F64, 1, 20, 3, 8, 1)
    assert lib.wf_row_width(ctypes.byref(p)) == 8
    assert lib.wf_column_bytes(ctypes.byref(p)) == (1 << 20) * 8
    assert lib.wf_lde_bytes(ctypes.byref(p)) == (1 << 23) * 8 * 8
    assert lib.wf_digests_bytes(ctypes.byref(p)) == (1 << 23) * 32
    q = capi.make_params(capi.F128, 2, 10, 3, 5, 4)
    assert lib.wf_row_width(ctypes.byref(q)) == 16
    assert lib.wf_column_bytes(ctypes.byref(q)) == 1024 * 2 * 16
    assert lib.wf_params_check(ctypes.byref(p), 0) == 0
    assert lib.wf_params_check(ctypes.byref(q), 1) == -16          # constraint commitment takes one "trace"
    bad = capi.make_params(capi.F64, 1, 2, 3, 8, 1)
    assert lib.wf_params_check(ctypes.byref(bad), 0) == -12
    assert b"trace length" in lib.wf_last_error()


def test_no_cpu_fallback_without_a_device(capi):
    """On a box without a GPU the product must fail loudly, not compute on the CPU."""
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(capi.WfError) as e:
        capi.Context(0)
    assert e.value.code == -30


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "starkpack-winterfell_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("no oracle", ""), f"{f} mentions the oracle"


def test_transform_plans(capi):
    """The pass plans of the commitment path (wf_plan_digits, no GPU needed) -- each of these splits was chosen by
    measurement (docs/EXPERIMENTS.md); a change here is a performance change and should be deliberate."""
    F64, F128 = 1, 2
    plan = capi.plan_digits
    assert plan(F64, 10) == [10] and plan(F128, 10) == [10] and plan(F64, 3) == [3]
    assert plan(F64, 11, 1) == [6, 5] and plan(F64, 11, 8) == [6, 5]      # few 2^11-row tiles would idle most CUs
    assert plan(F64, 11, 32) == [11]                                        # many segments: one pass
    assert plan(F64, 20) == [10, 10] and plan(F64, 16) == [8, 8] and plan(F64, 19) == [10, 9]
    assert plan(F64, 21) == [10, 11]                                        # the maximal digit goes last
    assert plan(F64, 22) == [8, 7, 7] and plan(F64, 23) == [8, 8, 7]        # never two full tiles
    # f128 tiles: 2^10 rows are half a CU's LDS in both passes (so 2^20 runs two full passes); 2^11 rows (128 KiB + one
    # 32 KiB table region: all of a CU) only as the single pass of a 2^11-row transform over many segments
    assert plan(F128, 18) == [9, 9] and plan(F128, 19) == [9, 10]
    assert plan(F128, 20) == [10, 10] and plan(F128, 21) == [7, 7, 7] and plan(F128, 22) == [8, 7, 7]
    assert plan(F128, 11, 1) == [6, 5] and plan(F128, 11, 32) == [11]
    assert plan(F64, 32) == [10, 11, 11] and plan(F128, 40) == [10, 10, 10, 10]
    for bad in ((3, 10, 1), (1, 0, 1), (1, 41, 1)):
        with pytest.raises(capi.WfError):
            plan(*bad)
