"""GPU parity: every pass geometry of k_ntt_pass against the oracle.

The product's plan rule gives a transform of 2^log_n elements passes of at most 2^7 rows (2^20 = 2^7 * 2^7 * 2^6), so the default
suite sees a handful of (rows, columns-per-tile) shapes.  PLK_NTT_PLAN (read when a plan is built) lets a caller choose the pass sizes;
the output is the same function - out[j] = sum in[k] w^(jk), fft.rs:103-156 - whatever the factorisation, and elements have one
representation, so every plan must be bit-identical to the oracle: tall passes (2^8 .. 2^10 rows: 4, 2, 1 columns per tile), short
ones (2^2 .. 2^5 rows), one to four passes, the zero-padded first pass (fft_with_precomputation: fft.rs:60-77), the inverse with
n^-1 folded into the first inter-pass table, over a nine-limb field and the fourteen-limb Bls12377Base.
"""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import plonky_amd as pa
from plonky_amd import lib, synth
from oracle import bigint_ref as br
from oracle import oracle_lib as ol


def _with_plan(plan):
    """Plans are cached per (field, size): drop them so the next transform is planned under `plan` (None: the product's rule)."""
    if plan is None:
        os.environ.pop("PLK_NTT_PLAN", None)
    else:
        os.environ["PLK_NTT_PLAN"] = plan
    lib.check(lib.load().plk_ntt_clear_cache())


@pytest.fixture(autouse=True)
def _restore_plan_rule():
    yield
    _with_plan(None)


PLANS = {
    16: ["8,8", "10,6", "6,10", "9,7", "7,9", "6,5,5", "5,5,6", "2,7,7", "7,7,2", "4,4,4,4", "3,3,5,5", "1,8,7"],
    13: ["7,6", "6,7", "10,3", "3,10", "9,4", "5,4,4", "1,6,6", "2,2,2,7"],
    11: ["10,1", "1,10", "6,5", "4,4,3", "2,9"],
}


@pytest.mark.parametrize("f", [br.TWEEDLEDEE_BASE, br.BLS12_377_BASE], ids=lambda f: f.name)
@pytest.mark.parametrize("log_n", sorted(PLANS))
def test_every_pass_geometry_matches_oracle(f, log_n):
    n = 1 << log_n
    x = synth.rand_field(f.field_id, 0x9A5500 + log_n, n)
    opre = ol.FftPrecomputation(f.field_id, n)
    exp_fwd = opre.fft_with_precomputation_power_of_2(x, threads=4)
    exp_inv = opre.ifft_with_precomputation_power_of_2(x, threads=4)
    short = np.ascontiguousarray(x[: n // 8 - 3])  # zero padding by 8 and a ragged end: the first pass skips three stages
    padded = np.zeros_like(x)
    padded[: short.shape[0]] = short
    exp_pad = opre.fft_with_precomputation_power_of_2(padded, threads=4)
    for plan in PLANS[log_n]:
        _with_plan(plan)
        pre = pa.fft_precompute(f.field_id, n)
        # the plan really is the one asked for: one launch of the pass kernel per pass
        L = lib.load()
        ms, launches = ctypes.c_double(), ctypes.c_uint()
        lib.check(L.plk_ntt_set_profiling(1))
        fwd = pa.fft_with_precomputation_power_of_2(x, pre)
        lib.check(L.plk_ntt_get_timings(ctypes.byref(ms), ctypes.byref(launches)))
        lib.check(L.plk_ntt_set_profiling(0))
        assert launches.value == len(plan.split(",")), (plan, launches.value)
        assert np.array_equal(fwd, exp_fwd), plan
        assert np.array_equal(pa.ifft_with_precomputation_power_of_2(x, pre), exp_inv), plan
        assert np.array_equal(pa.ifft_with_precomputation_power_of_2(fwd, pre), x), plan
        got_pad = np.empty_like(x)
        lib.check(lib.load().plk_ntt_padded(f.field_id, log_n, short.ctypes.data_as(ctypes.c_void_p), short.shape[0],
                                            got_pad.ctypes.data_as(ctypes.c_void_p)))
        assert np.array_equal(got_pad, exp_pad), plan


def test_plan_independence_at_full_size():
    """2^20 TweedledeeBase (BASELINE config 2): the product's (7,7,6) against five other factorisations, limb for limb, forward and
    inverse; the product's output itself is pinned to the oracle by test_gpu_parity.py::test_ntt_2p20_full_size."""
    from plonky_amd import device as dev
    n = 1 << 20
    xh = synth.rand_field(0, 0x9A5520, n)
    dev.init(0)
    x = dev.to_device(xh)
    _with_plan(None)
    ref_f = dev.to_host(dev.ntt_dev(0, x))
    ref_i = dev.to_host(dev.ntt_dev(0, x, inverse=True))
    assert np.array_equal(dev.to_host(dev.ntt_dev(0, dev.to_device(ref_f), inverse=True)), xh)
    for plan in ["7,6,7", "6,7,7", "10,10", "8,8,4", "5,5,5,5", "9,9,2"]:
        _with_plan(plan)
        assert np.array_equal(dev.to_host(dev.ntt_dev(0, x)), ref_f), plan
        assert np.array_equal(dev.to_host(dev.ntt_dev(0, x, inverse=True)), ref_i), plan
