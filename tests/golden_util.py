"""Helpers shared by the golden-vector tests: JSON fixtures (canonical integers) <-> in-memory element arrays."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F64_P = 2**64 - 2**32 + 1


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def to_mem(field, ints):
    """canonical python ints -> the reference's in-memory representation (f64: Montgomery u64; f128: (lo,hi))."""
    if field == "f64":
        return np.array([(int(v) << 64) % F64_P for v in ints], dtype=np.uint64)
    out = np.empty((len(ints), 2), dtype=np.uint64)
    for i, v in enumerate(ints):
        v = int(v)
        out[i, 0] = v & 0xFFFFFFFFFFFFFFFF
        out[i, 1] = v >> 64
    return out


def field_id(field):
    return 1 if field == "f64" else 2


def lde_to_mem(field, rows, row_width):
    """golden LDE rows (canonical ints, base columns) -> padded row-major in-memory matrix."""
    n = len(rows)
    if field == "f64":
        out = np.zeros((n, row_width), dtype=np.uint64)
        for j, r in enumerate(rows):
            out[j, :len(r)] = to_mem(field, r)
    else:
        out = np.zeros((n, row_width, 2), dtype=np.uint64)
        for j, r in enumerate(rows):
            out[j, :len(r)] = to_mem(field, r)
    return out


def hexrows(a):
    return [bytes(x).hex() for x in np.asarray(a, dtype=np.uint8).reshape(-1, 32)]
