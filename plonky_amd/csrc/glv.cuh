// glv.cuh -- splitting a scalar along the curve endomorphism (Gallant-Lambert-Vanstone).
//
// The four prime-order curves have phi((x, y)) = (beta x, y) = [lambda](x, y) (curve.rs:65-69,140-149: HaloCurve::ZETA /
// ZETA_SCALAR; constants derived in tools/gen_glv_params.py).  k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^GLV_BITS turns
// [k] P into [k1] P + [k2] phi(P): the table-free MSM (msm.hip) then runs over 2n points with half-length scalars - the same
// number of additions, but half the windows, i.e. half of the doubling chain that brings the top window's sum into place
// (the latency of a one-shot MSM: 21 * 12 doublings of one point at 255 bits).
//
// Integer arithmetic only:  m1 = floor(k G1 / 2^384), m2 = floor(k G2 / 2^384)  (G = 2^384 |b| / r rounded),
// k1 = k - (+-m1 A1 +- m2 A2),  k2 = -(+-m1 B1 +- m2 B2)  in 256-bit two's complement; the results are small, so the wrap is harmless.
#pragma once
#include <stdint.h>

#include "fp.cuh"
#include "glv_params.cuh"

namespace plk {

// out[t] = word (SKIP + t) of a * b, t < NO  (schoolbook by columns; every column below SKIP is formed for its carry)
template <int NA, int NB, int NO, int SKIP> PLK_DI void glv_mul_words(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    uint64_t carry = 0;
#pragma unroll
    for (int col = 0; col < SKIP + NO; ++col) {
        uint64_t lo = (uint32_t)carry, hi = carry >> 32;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int j = col - i;
            if (j >= 0 && j < NB) {
                const uint64_t p = (uint64_t)a[i] * b[j];
                lo += (uint32_t)p;
                hi += p >> 32;
            }
        }
        if (col >= SKIP) out[col - SKIP] = (uint32_t)lo;
        carry = hi + (lo >> 32);
    }
}
// x += y or x -= y on 256 bits, wrapping
PLK_DI void glv_acc256(uint32_t (&x)[8], const uint32_t (&y)[8], bool subtract) {
    uint64_t c = subtract ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)x[i] + (subtract ? ~y[i] : y[i]);
        x[i] = (uint32_t)c;
        c >>= 32;
    }
}
// two's complement -> magnitude with the sign in bit 255 (the magnitude is below 2^GLV_BITS)
PLK_DI void glv_sign_magnitude(uint32_t (&x)[8]) {
    const bool neg = (x[7] >> 31) != 0;
    if (neg) {
        uint64_t c = 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c += (uint64_t)(~x[i]);
            x[i] = (uint32_t)c;
            c >>= 32;
        }
        x[7] |= 0x80000000u;
    }
}

// k: canonical scalar (8 words, < r).  k1, k2: magnitude below 2^GLV_BITS in the low words, sign in bit 255.
template <class G> PLK_DI void glv_split(const uint32_t (&k)[8], uint32_t (&k1)[8], uint32_t (&k2)[8]) {
    uint32_t m1[5], m2[5];
    glv_mul_words<8, 9, 5, GLV_SHIFT_WORDS>(k, G::G1, m1);
    glv_mul_words<8, 9, 5, GLV_SHIFT_WORDS>(k, G::G2, m2);
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        k1[i] = k[i];
        k2[i] = 0;
    }
    // k1 = k - (a1 m1 + a2 m2),  k2 = -(b1 m1 + b2 m2): a negative coefficient turns the subtraction into an addition
    glv_mul_words<5, 5, 8, 0>(m1, G::A1, t);
    glv_acc256(k1, t, !G::A1_NEG);
    glv_mul_words<5, 5, 8, 0>(m2, G::A2, t);
    glv_acc256(k1, t, !G::A2_NEG);
    glv_mul_words<5, 5, 8, 0>(m1, G::B1, t);
    glv_acc256(k2, t, !G::B1_NEG);
    glv_mul_words<5, 5, 8, 0>(m2, G::B2, t);
    glv_acc256(k2, t, !G::B2_NEG);
    glv_sign_magnitude(k1);
    glv_sign_magnitude(k2);
}

}  // namespace plk
