"""PCIe-inclusive timing of the host-buffer form with pageable vs pinned caller buffers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import starkpack_winterfell_amd.capi as capi

ctx = capi.Context(0)
L = capi.load()
logR, logB, ncol = 20, 3, 8
R, N = 1 << logR, 1 << (logR + logB)
p = capi.make_params(capi.F64, 1, logR, logB, ncol, 1)
rng = np.random.default_rng(1)

def bufs(pin):
    mk = (lambda n, dt: torch.empty(n, dtype=dt, pin_memory=True).numpy()) if pin else (lambda n, dt: np.empty(n, dtype=dt.__str__().replace("torch.", "")))
    cols = [mk(R, torch.int64) for _ in range(ncol)]
    for c in cols:
        c[:] = rng.integers(0, 2**62, size=R, dtype=np.int64)
    polys = [mk(R, torch.int64) for _ in range(ncol)]
    lde = mk(N * 8, torch.int64)
    leaves = mk(N * 32, torch.uint8)
    nodes = mk(N * 32, torch.uint8)
    return cols, polys, lde, leaves, nodes

def run(pin, full):
    cols, polys, lde, leaves, nodes = bufs(pin)
    pa = lambda arrs: (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    root = np.empty(32, dtype=np.uint8)
    def call():
        rc = L.wf_trace_commit(ctx._h, C.byref(p), pa(cols), pa(polys) if full else None, pa([lde]) if full else None,
                               leaves.ctypes.data_as(C.c_void_p), nodes.ctypes.data_as(C.c_void_p), root.ctypes.data_as(C.c_void_p))
        assert rc == 0
    call()
    t0 = time.perf_counter()
    for _ in range(3):
        call()
    return (time.perf_counter() - t0) / 3 * 1e3

for pin in (False, True):
    for full in (True, False):
        print(f"caller buffers {'pinned' if pin else 'pageable'}, {'all outputs' if full else 'leaves+nodes only'}: {run(pin, full):.1f} ms per commitment")
