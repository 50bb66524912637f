"""Timing probe for the polynomial entry points (run on the GPU box): divide_by_z_h at the Plonk shape
(degree < 8n, Z_H of n), the 9-wire padded LDE and plain transforms of the same size for comparison."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from plonky_amd import device as dv, synth  # noqa: E402


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dv.init()
    f = 0
    for log_n in (17, 20):
        n = 1 << log_n
        size = 8 * n
        m = dv.to_device(synth.rand_field(f, 1, size - n + 3))  # any polynomial: timing does not depend on divisibility
        out = torch.empty((size, 4), dtype=torch.int64, device="cuda")
        t_div = timeit(lambda: dv.divide_by_z_h_dev(f, m, n, out=out))
        x = dv.to_device(synth.rand_field(f, 2, size))
        t_fwd = timeit(lambda: dv.ntt_dev(f, x, out=out))
        t_inv = timeit(lambda: dv.ntt_dev(f, x, inverse=True, out=out))
        w = dv.to_device(synth.rand_field(f, 3, 9 * n).reshape(9, n, 4))
        ev = torch.empty((9, size, 4), dtype=torch.int64, device="cuda")
        t_lde = timeit(lambda: dv.ntt_padded_dev(f, w, log_n + 3, out=ev), reps=5)
        t_full = timeit(lambda: dv.ntt_dev(f, ev, out=ev), reps=5)
        print("n=2^%d size=2^%d: divide_by_z_h %.3f ms (fwd NTT %.3f + inv NTT %.3f = %.3f) | LDE x9 %.3f ms vs 9 full NTTs %.3f ms"
              % (log_n, log_n + 3, t_div, t_fwd, t_inv, t_fwd + t_inv, t_lde, t_full))


if __name__ == "__main__":
    main()
