// hostnorm.cpp -- ProjectivePoint::to_affine (curve.rs:206-214) on the HOST, for the one or two points of a call whose results cross PCIe
// in any case (the L_j / R_j of an inner-product-argument round: halo.rs:93-101 takes msm_execute's ProjectivePoints and normalises them on
// the CPU before they go into the transcript).  On the device the inversion is ~35 us of ONE lane at the end of a dependency chain
// (profiles/r05_final_kernel_trace.txt); on a host core it is ~2 us.  Plain C++ (g++): fp.cuh is written to compile outside hipcc
// (tests/test_fp_host.py sweeps the same functions against Python integers).
#include <cstring>

#include "fp.cuh"

namespace plk {

// xyz: count x (x | y | z), R-form canonical limbs, as emit_projective (ecz.cuh) leaves them; zero[k] != 0: ProjectivePoint::ZERO.
// xy: count x (x | y), the unique affine point (zeros for the identity, like emit_affine).
template <class P> static void to_affine_t(unsigned count, const uint8_t* xyz, const uint8_t* zero, uint8_t* xy) {
    constexpr size_t B = (size_t)P::NL * 4;
    for (unsigned k = 0; k < count; ++k) {
        Fe<P> x, y, z;
        memcpy(x.v, xyz + (size_t)k * 3 * B, B);
        memcpy(y.v, xyz + (size_t)k * 3 * B + B, B);
        memcpy(z.v, xyz + (size_t)k * 3 * B + 2 * B, B);
        if (zero[k] || fe_is_zero<P>(z)) {
            memset(xy + (size_t)k * 2 * B, 0, 2 * B);
            continue;
        }
        const Fe<P> zi = fe_inv_safegcd_var<P>(z);  // Montgomery in, Montgomery out (monty.rs:162-166)
        const Fe<P> ax = fe_mul<P>(x, zi), ay = fe_mul<P>(y, zi);
        memcpy(xy + (size_t)k * 2 * B, ax.v, B);
        memcpy(xy + (size_t)k * 2 * B + B, ay.v, B);
    }
}

// curve: PLK_CURVE_* (include/plonky_hip.h).  Returns 0, or -1 for an unknown curve.
int host_projective_to_affine(int curve, unsigned count, const uint8_t* xyz, const uint8_t* zero, uint8_t* xy) {
    switch (curve) {
        case 0: to_affine_t<TweedledeeBaseParams>(count, xyz, zero, xy); return 0;
        case 1: to_affine_t<TweedledumBaseParams>(count, xyz, zero, xy); return 0;
        case 2: to_affine_t<Bls12377BaseParams>(count, xyz, zero, xy); return 0;
        case 3: to_affine_t<PallasBaseParams>(count, xyz, zero, xy); return 0;
        case 4: to_affine_t<VestaBaseParams>(count, xyz, zero, xy); return 0;
    }
    return -1;
}

}  // namespace plk
