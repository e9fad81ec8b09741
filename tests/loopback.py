"""Test helper: an in-process transport for wf_comm -- W ranks as W threads of one process, the bytes of the two
collectives handed over through host memory.  Lets one GPU (or none, for the callbacks alone) run the multi-rank code
paths of libwf_lde.so exactly as RCCL would drive them."""
import threading

import numpy as np


class Loopback:
    def __init__(self, world: int, timeout: float = 120.0):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=timeout)
        self.slots = [None] * world

    def collectives(self, rank: int):
        world = self.world

        def all_gather(mine):
            self.slots[rank] = np.array(mine, copy=True)
            self.barrier.wait()
            out = np.concatenate(self.slots)
            self.barrier.wait()  # nobody overwrites a slot before everybody has read it
            return out

        def all_to_all(mine):
            self.slots[rank] = np.array(mine, copy=True)
            self.barrier.wait()
            n = mine.size // world
            out = np.concatenate([self.slots[s][rank * n:(rank + 1) * n] for s in range(world)])
            self.barrier.wait()
            return out

        return all_gather, all_to_all


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
