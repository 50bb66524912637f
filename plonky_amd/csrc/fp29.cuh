// fp29.cuh -- Montgomery multiplication for the 256-bit fields through 29-bit limbs.
//
// Why: on gfx950 v_mad_u64_u32 issues in 4 cycles and every carry-flag instruction
// (v_add_co / v_addc_co) or 64-bit add costs about as much (profiles/r01_microbench_issue_rates.txt),
// so a 32-bit-limb CIOS (monty.rs:67-107 restated on u32) spends 3/4 of its instructions moving
// carries.  With 29-bit limbs a 64-bit accumulator holds a whole column of the schoolbook product
// plus the Montgomery correction terms (<= 18 products < 2^58 each) without overflow, so the
// multiplier chain is nothing but v_mad_u64_u32: one instruction per limb product, one 64-bit
// shift per column.
//
// Interface is unchanged: inputs and output are the reference's representation (8 x u32 = 4 x u64
// little-endian limbs, Montgomery radix R = 2^256, fully reduced).  Internally:
//   * operands are re-sliced into 9 limbs of 29 bits (funnel shifts);
//   * product scanning over columns k = 0..16 with the quotient digits folded in as they become
//     known: q_k = -col_k mod 2^29 for k < 8 (p = 1 mod 2^32, hence also mod 2^29), and a final
//     24-bit digit q_8 = -col_8 mod 2^24, so the total division is by 2^(8*29+24) = 2^256 exactly,
//     the reference's R;
//   * the surviving columns are re-sliced into 8 x u32 and conditionally reduced once (< 2p).
// The value is a*b*R^-1 mod p, fully reduced: the same unique limbs monty_multiply produces.
#pragma once
#include <stdint.h>

namespace plk {

template <class P> struct Mod29 {
    // 29-bit limbs of the modulus, computed at compile time from the 32-bit limbs
    static constexpr uint32_t limb(int j) {
        const int o = 29 * j, w = o >> 5, sh = o & 31;
        uint64_t two = P::MOD[w];
        if (w + 1 < P::NL) two |= (uint64_t)P::MOD[w + 1] << 32;
        return (uint32_t)(two >> sh) & 0x1FFFFFFFu;
    }
};

template <class P> PLK_DI void fe_split29(const uint32_t (&a)[8], uint32_t (&o)[9]) {
    constexpr uint32_t M = 0x1FFFFFFFu;
    o[0] = a[0] & M;
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        const int off = 29 * i, w = off >> 5, sh = off & 31;
        // bits [off, off+29) straddle words w and w+1 (sh is never 0 and sh + 29 > 32 for i in 1..7):
        // a 32-bit funnel shift (v_alignbit_b32).  Written on 32-bit values on purpose - a 64-bit
        // combine makes the compiler fuse the two reads into one 8-byte access and spill the operand.
        o[i] = ((a[w] >> sh) | (a[w + 1] << (32 - sh))) & M;
    }
    o[8] = a[7] >> 8;  // bits 232..255
}

template <class P> PLK_DI void fe_mul29_core(const uint32_t (&a)[8], const uint32_t (&b)[8], uint32_t (&r)[8]) {
    static_assert(P::NL == 8, "29-bit path is for the 256-bit fields");
    static_assert(Mod29<P>::limb(0) == 1u, "needs p = 1 (mod 2^29)");
    constexpr uint32_t M29 = 0x1FFFFFFFu, M24 = 0x00FFFFFFu;
    uint32_t A[9], B[9], q[9];
    fe_split29<P>(a, A);
    fe_split29<P>(b, B);
    uint32_t L[9];      // surviving columns 8..16 after carry propagation (column 8 keeps its low 24 zero bits)
    uint64_t acc = 0;   // running column, carries included
#pragma unroll
    for (int k = 0; k <= 16; ++k) {
        // schoolbook terms of column k
#pragma unroll
        for (int i = 0; i <= 8; ++i) {
            const int j = k - i;
            if (j >= 0 && j <= 8) acc = (uint64_t)A[i] * B[j] + acc;
        }
        // Montgomery terms q_i * p_j, i + j = k, for the digits already known (i < k, i <= 8)
#pragma unroll
        for (int i = 0; i <= 8; ++i) {
            const int j = k - i;
            if (i < k && j >= 1 && j <= 8 && Mod29<P>::limb(j) != 0u) acc = (uint64_t)q[i] * Mod29<P>::limb(j) + acc;
        }
        if (k < 8) {
            q[k] = (0u - (uint32_t)acc) & M29;
            acc += q[k];          // + q_k * p_0, p_0 = 1: the low 29 bits become zero
            acc >>= 29;
        } else if (k == 8) {
            q[8] = (0u - (uint32_t)acc) & M24;
#ifdef __HIPCC__
            // ROCm 7.2 / clang 22 miscompile: the AMDGPU mul24 combine (q8 * small p_j becomes
            // v_mul_u32_u24) strips this mask from q8 for ALL its users, including the 32-bit
            // v_mad_u64_u32 ones (seen for Bls12377Scalar; caught by the parity sweep and
            // reproduced by emulating the emitted ISA).  Make the masked value opaque.
            asm volatile("" : "+v"(q[8]));
#endif
            acc += q[8];          // low 24 bits become zero
            L[0] = (uint32_t)acc & M29;
            acc >>= 29;
        } else {
            L[k - 8] = (uint32_t)acc & M29;
            acc >>= 29;
        }
    }
    // acc now holds what is left above column 16 (value < 2p < 2^256 keeps it tiny)
    const uint32_t top = (uint32_t)acc;
    // result = (L[0] >> 24) + sum_{m=1..8} L[m] * 2^(29 m - 24) + top * 2^(29*9 - 24), < 2p < 2^256.
    // Re-slice into 32-bit words.  Limb m sits at bit s_m = 29 m - 24 (m = 0: only its top 5 bits are
    // non-zero, so the negative position is just a right shift).  Every index below is a constant
    // after unrolling - no loop-carried state, nothing the compiler could turn into an indexed array.
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        uint32_t word = 0;
#pragma unroll
        for (int m = 0; m <= 9; ++m) {
            const int s = 29 * m - 24, lo = 32 * t, hi = 32 * t + 32;
            if (s < hi && s + 29 > lo) {
                const uint32_t v = m <= 8 ? L[m] : top;
                if (s >= lo) word |= v << (s - lo);
                else word |= v >> (lo - s);
            }
        }
        r[t] = word;
    }
}

}  // namespace plk
