"""Shared helpers for the tests: limb <-> int conversion on numpy arrays."""
import numpy as np


def limbs_to_int(row):
    v = 0
    for i, l in enumerate(row):
        v |= int(l) << (64 * i)
    return v


def int_to_limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def ints_to_array(vals, n_limbs):
    return np.array([int_to_limbs(v, n_limbs) for v in vals], dtype=np.uint64).reshape(len(vals), n_limbs)


def array_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64)
    return [limbs_to_int(r) for r in arr.reshape(-1, arr.shape[-1])]
