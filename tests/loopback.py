"""Test helper: W ranks as W threads of one process on the in-process transport of the package (shard.Loopback: the bytes of
the two collectives handed over through host memory) -- one GPU (or none, for the callbacks alone) runs the multi-rank
code paths of libwf_lde.so exactly as RCCL would drive them."""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starkpack_winterfell_amd.shard import Loopback  # noqa: E402,F401  (re-exported for the tests)


def run_ranks(world: int, fn):
    """fn(rank) in `world` threads; returns their results in rank order, re-raises the first failure."""
    results, errors = [None] * world, []

    def body(r):
        try:
            results[r] = fn(r)
        except BaseException as e:  # noqa: BLE001 -- reported to the test below
            errors.append((r, e))

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    if errors:
        raise errors[0][1]
    return results
