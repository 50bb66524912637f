"""plonky_amd -- MI355X-native NTT + MSM hot path of Plonky (0xPolygonZero/plonky).

Host-side mirror of the reference's hot-path interface (src/fft.rs, src/curve/curve_msm.rs)
on top of the C ABI in include/plonky_hip.h.  See DESIGN.md.
"""
from .api import (  # noqa: F401
    BLS12_377, BLS12_377_BASE, BLS12_377_SCALAR, TWEEDLEDEE, TWEEDLEDEE_BASE, TWEEDLEDUM, TWEEDLEDUM_BASE, PALLAS, PALLAS_BASE, VESTA, VESTA_BASE,
    FftPrecomputation, MsmPrecomputation, fft, fft_precompute, fft_with_precomputation,
    fft_with_precomputation_power_of_2, ifft_with_precomputation_power_of_2, msm_execute, msm_execute_batch, msm_execute_parallel, msm_execute_parallel_projective,
    msm_parallel, msm_precompute, log2_ceil, log2_strict, polynomial_divide_by_z_h, polynomial_mul,
    polynomials_to_values_padded, values_to_polynomials, fold_generators, msm_precompute_table, commitment_precompute, coeffs_vec_to_commitments,
    init_devices, device_count, msm_debug_digits, affine_summation_best, affine_multisummation_best, curve_sum_affine,
)
from .lib import PlonkyHipError  # noqa: F401
