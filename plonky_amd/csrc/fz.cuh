// fz.cuh -- lazily reduced field elements on 29-bit limbs: the arithmetic the hot kernels run on.
//
// fp.cuh keeps the reference's representation (32-bit words, R = 2^(32 NL), fully reduced) and is
// what crosses every interface.  Inside a kernel that representation is expensive on gfx950:
// every carry is a 4-cycle instruction and every product has to be brought back below p.  Fz is
// the working form of the inner loops:
//   * NZ = ceil(32 NL / 29) limbs of 29 bits in 32-bit registers (9 for the 256-bit fields, 14 for
//     Bls12377Base), so a 64-bit accumulator sums a whole product column (<= 2 NZ terms < 2^58)
//     with plain v_mad_u64_u32 chains - no carry flags anywhere;
//   * Montgomery radix R' = 2^(29 NZ) (2^261 / 2^406): p / R' <= 2^-7, so a product of two values
//     below 16p comes back below (1 + 256/128) p ... in practice < 2p, without any final
//     conditional subtraction;
//   * values are only congruent mod p ("< k p" for a small tracked k), limbs only almost
//     normalised (< 2^29 + 8); subtraction adds a multiple of p laid out so no limb goes negative.
// Conversions: fz_from_words() re-slices 32-bit words (the value is kept, so data in R-form stays
// in R-form; tables that multiply such data are stored in R'-form, see ntt.hip / msm.hip);
// fz_to_words_canonical() reduces fully and re-slices back, giving the unique representative.
//
// Bounds, stated once.  "limb bound" L: every limb <= L.  mul/sqr accept L <= 2^29 + 2^27 on
// both operands (column sum <= NZ (2^29+2^27)^2 + NZ 2^58 + 2^36 < 2^64 for NZ <= 14) and any
// value < 2^(29 NZ - 3) = R'/8.  mul/sqr return exactly normalised limbs (< 2^29, top limb takes
// the rest) and a value < a b / R' + p.  add/sub return limbs < 2^29 + 8.
#pragma once
#include <stdint.h>

#include "fp.cuh"

namespace plk {

template <class P> struct FzCfg {
    static constexpr int NZ = (32 * P::NL + 28) / 29;
    static constexpr uint32_t M = 0x1FFFFFFFu;
    static constexpr uint32_t plimb(int j) { return Mod29<P>::limb(j); }
    // limb j of (2^k p) in "borrowed" form: every limb below the top carries an extra 2^30 taken from
    // the limb above, so that (a - b + this) has no negative limb for any b with limbs < 2^30.
    static constexpr uint32_t kp_limb(int k, int j) {
        // 2^k p as NZ 29-bit limbs (top limb unbounded)
        uint64_t carry = 0;
        uint32_t c = 0;
        for (int i = 0; i <= j; ++i) {
            uint64_t v = ((uint64_t)plimb(i) << k) + carry;
            if (i == NZ - 1) {
                c = (uint32_t)v;
                carry = 0;
            } else {
                c = (uint32_t)(v & M);
                carry = v >> 29;
            }
        }
        if (j == 0) return c + (1u << 30);
        if (j == NZ - 1) return c - 2u;
        return c + (1u << 30) - 2u;
    }
};

template <class P> struct Fz {
    uint32_t l[FzCfg<P>::NZ];
};

// A code-generation nudge for gfx950, kept because it measured faster (profiles/r01_field_op_costs.txt):
// a modulus limb that is a power of two (2^22 for the Tweedle fields) makes the compiler replace
// q * p_j + acc by a 64-bit shift and a 64-bit add; keeping the constant opaque (in an SGPR) keeps
// it one v_mad_u64_u32 (fz_mul 890 -> 835 cycles).  The opposite experiment - replacing the
// column carry acc >> 29 (v_lshrrev_b64) by v_alignbit_b32 + v_lshrrev_b32 - measured slower.
PLK_DI uint64_t fz_shr29(uint64_t acc) { return acc >> 29; }
PLK_DI uint32_t fz_opaque(uint32_t c) {
#ifdef __HIPCC__
    asm volatile("" : "+s"(c));
#endif
    return c;
}
// Quotient digits without their own addition (round 4).  Column k < NZ of a Montgomery product needs q_k = -acc_k mod 2^29 and
// then (acc_k + q_k) >> 29: the low 29 bits vanish, so that is (acc_k >> 29) + (1 if they were not zero), i.e. (acc_k + M) >> 29 with
// M = 2^29 - 1.  The column adds the CONSTANT M (kept opaque in scalar registers: one v_lshl_add_u64, where "+ q_k" needed the
// digit first) and with acc'_k = acc_k + M the digit is q_k = ~acc'_k & M (-x = ~x + 1 and M = -1 mod 2^29): one v_bitop3_b32
// instead of a subtraction and a mask, no register copies around the 64-bit pair.  fz_mul 201 -> 184 VALU instructions per
// product on gfx950, fz_sqr 173 -> 155 (tools/lab/madd_lab.hip + tools/isa_count.py).  (Starting the column's multiply-add chain
// from M through inline assembly was tried: the compiler still joins two chains with an addition - 183 - and pads with s_nop.)
PLK_DI uint64_t fz_opaque64(uint64_t c) {
#ifdef __HIPCC__
    asm volatile("" : "+s"(c));
#endif
    return c;
}
template <class P> struct FzPow2Limb {
    // index of a modulus limb (j >= 1) that is a power of two, or -1
    static constexpr int index() {
        for (int j = 1; j < FzCfg<P>::NZ; ++j) {
            const uint32_t v = FzCfg<P>::plimb(j);
            if (v != 0u && (v & (v - 1u)) == 0u) return j;
        }
        return -1;
    }
};

// ---- conversions -----------------------------------------------------------------------------
template <class P> PLK_DI Fz<P> fz_from_words(const uint32_t (&a)[P::NL]) {
    constexpr int NZ = FzCfg<P>::NZ;
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        const int off = 29 * i, w = off >> 5, sh = off & 31;
        uint32_t v;
        if (sh == 0) v = a[w];
        else if (sh + 29 <= 32 || w + 1 >= P::NL) v = a[w] >> sh;
        else v = (a[w] >> sh) | (a[w + 1] << (32 - sh));  // 32-bit funnel shift (v_alignbit_b32)
        r.l[i] = v & FzCfg<P>::M;
    }
    return r;
}
template <class P> PLK_DI Fz<P> fz_from_fe(const Fe<P>& a) { return fz_from_words<P>(a.v); }

// exact limb normalisation (sequential carry chain): limbs < 2^29, top limb takes the rest
template <class P> PLK_DI void fz_normalize(Fz<P>& a) {
    constexpr int NZ = FzCfg<P>::NZ;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NZ - 1; ++i) {
        const uint32_t t = a.l[i] + c;
        a.l[i] = t & FzCfg<P>::M;
        c = t >> 29;
    }
    a.l[NZ - 1] += c;
}

// value < 2p, any limbs  ->  the unique representative in the reference's 32-bit words
template <class P> PLK_DI Fe<P> fz_to_fe_canonical(Fz<P> a) {
    constexpr int NZ = FzCfg<P>::NZ;
    fz_normalize<P>(a);
    Fe<P> r;
#pragma unroll
    for (int t = 0; t < P::NL; ++t) {
        uint32_t word = 0;
#pragma unroll
        for (int m = 0; m < NZ; ++m) {
            const int s = 29 * m, lo = 32 * t, hi = 32 * t + 32;
            if (s < hi && s + 32 > lo) {  // the top limb may be wider than 29 bits: treat every limb as 32 wide
                if (s >= lo) word |= a.l[m] << (s - lo);
                else if (lo - s < 32) word |= a.l[m] >> (lo - s);
            }
        }
        r.v[t] = word;
    }
    fe_cond_sub_p<P>(r.v);
    return r;
}

// ---- add / sub -------------------------------------------------------------------------------
// parallel carry pass: limbs <= L < 2^32  ->  limbs < 2^29 + (L >> 29) + 1
template <class P> PLK_DI void fz_carry(Fz<P>& a) {
    constexpr int NZ = FzCfg<P>::NZ;
    uint32_t c[NZ];
#pragma unroll
    for (int i = 0; i < NZ - 1; ++i) c[i] = a.l[i] >> 29;
    a.l[0] &= FzCfg<P>::M;
#pragma unroll
    for (int i = 1; i < NZ - 1; ++i) a.l[i] = (a.l[i] & FzCfg<P>::M) + c[i - 1];
    a.l[NZ - 1] += c[NZ - 2];
}
template <class P> PLK_DI Fz<P> fz_add(const Fz<P>& a, const Fz<P>& b) {
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) r.l[i] = a.l[i] + b.l[i];
    fz_carry<P>(r);
    return r;
}
// a - b + 2^K p.  Needs value(b) <= 2^K p - 2^(29 (NZ - 1) + 2) (any honest bound k p with k < 2^K
// satisfies it) and limbs of b <= 2^30 - 2.  Result value < value(a) + 2^K p, limbs < 2^29 + 8.
template <class P, int K> PLK_DI Fz<P> fz_sub(const Fz<P>& a, const Fz<P>& b) {
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) r.l[i] = a.l[i] - b.l[i] + FzCfg<P>::kp_limb(K, i);
    fz_carry<P>(r);
    return r;
}
template <class P> PLK_DI Fz<P> fz_dbl(const Fz<P>& a) { return fz_add<P>(a, a); }

// ---- add / sub without the carry pass ------------------------------------------------------------
// A value that goes straight into a multiplication by an exactly normalised operand (a table entry: limbs < 2^29) need not
// have its carries moved first.  Column bound of fz_mul with one operand's limbs <= La and the other's < 2^29:
//   NZ La 2^29 (a b) + NZ 2^58 (q p) + 2^36 (carry in)  <  2^64   for  La <= 2.5 * 2^30 + 16  and  NZ <= 10,
// so a sum / difference of up to that limb size is a legal multiplicand (fz_sqr and a product of two such values are not).
// fz_sub_nc borrows 2^B from each limb's upper neighbour: needs limbs of b <= 2^B - 1 (B = 29: b exactly normalised), the same
// value condition as fz_sub, and returns limbs <= limbs(a) + 2^B + 2^29.
template <class P> struct FzNcBound {
    static_assert(FzCfg<P>::NZ <= 10, "limb bound of the carry-free forms is derived for NZ <= 10");
    static constexpr uint32_t MUL_LIMB_MAX = 5u * (1u << 29) + 16u;  // 2.5 * 2^30 + 16
};
// The same bound for any limb count: the largest limb La of one operand of fz_mul whose other operand is exactly normalised
// (< 2^29), from  NZ La 2^29 + NZ 2^58 + 2^37 <= 2^64:  La <= (2^35 - NZ 2^29 - 2^8) / NZ  (3.57 * 2^29 for the 14 limbs of
// Bls12377Base, 6.1 * 2^29 for 9 limbs).  The lazy mixed addition (ecz.cuh) checks its unnormalised operands against it.
template <class P> struct FzLazyBound {
    static constexpr uint64_t MUL_LIMB_MAX = (((uint64_t)1 << 35) - (uint64_t)FzCfg<P>::NZ * ((uint64_t)1 << 29) - 256u) / (uint64_t)FzCfg<P>::NZ;
};
template <class P> PLK_DI Fz<P> fz_add_nc(const Fz<P>& a, const Fz<P>& b) {
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <class P, int K, int B> PLK_DI Fz<P> fz_sub_nc(const Fz<P>& a, const Fz<P>& b) {
    constexpr int NZ = FzCfg<P>::NZ;
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        // limb i of 2^K p, plus the 2^B lent by limb i + 1, minus the one lent to limb i - 1 (so: limbs of b <= 2^B - 2^(B - 29))
        const uint32_t c = FzCfg<P>::kp_limb(K, i) - (i == 0 ? (1u << 30) : i == NZ - 1 ? 0u - 2u : (1u << 30) - 2u);
        constexpr uint32_t lent = 1u << (B - 29);
        const uint32_t k = i == 0 ? c + (1u << B) : i == NZ - 1 ? c - lent : c + (1u << B) - lent;
        r.l[i] = a.l[i] - b.l[i] + k;
    }
    return r;
}

// ---- multiplication --------------------------------------------------------------------------
// Montgomery product a b / R' mod p (lazy): product scanning, quotient digits folded in as they
// become known (q_k = -column_k mod 2^29 because p = 1 mod 2^29).
template <class P> PLK_DI Fz<P> fz_mul(const Fz<P>& a, const Fz<P>& b) {
    constexpr int NZ = FzCfg<P>::NZ;
    constexpr uint32_t M = FzCfg<P>::M;
    static_assert(Mod29<P>::limb(0) == 1u, "needs p = 1 (mod 2^29)");
    uint32_t q[NZ];
    Fz<P> r;
    const uint64_t m64 = fz_opaque64((uint64_t)M);
    uint64_t acc = 0;
    const uint32_t p_pow2 = FzPow2Limb<P>::index() >= 0 ? fz_opaque(FzCfg<P>::plimb(FzPow2Limb<P>::index() >= 0 ? FzPow2Limb<P>::index() : 0)) : 0u;
#pragma unroll
    for (int k = 0; k <= 2 * NZ - 2; ++k) {
        if (k < NZ) acc += m64;  // see fz_opaque64
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (j >= 0 && j < NZ) acc = (uint64_t)a.l[i] * b.l[j] + acc;
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (i < k && j >= 1 && j < NZ && FzCfg<P>::plimb(j) != 0u)
                acc = (uint64_t)q[i] * (j == FzPow2Limb<P>::index() ? p_pow2 : FzCfg<P>::plimb(j)) + acc;
        }
        if (k < NZ) q[k] = ~(uint32_t)acc & M;  // + q_k p_0 (p_0 = 1) is what the M already in acc stands for
        else r.l[k - NZ] = (uint32_t)acc & M;
        acc = fz_shr29(acc);
    }
    r.l[NZ - 1] = (uint32_t)acc;
    return r;
}

// (a b + c d) / R' with ONE Montgomery reduction (round 5): a sum of two products shares its quotient digits - NZ x (non-zero limbs of p)
// multiplier instructions, NZ digit extractions and the second result's carries less than two products and an addition.  The point
// formulas end in such a sum, Y3 = r t - y1 ppp (the subtrahend enters as c = 2^K p - y1).  Column bound
//   NZ (La Lb + Lc Ld) + NZ 2^58 + 2^37 < 2^64:
//   * every operand carried (limbs < 2^29 + 2^27): any NZ <= 14;
//   * NZ <= 9, a and b carried (< 2^29 + 8), d exactly normalised (< 2^29): c may keep limbs up to 2^31 + 2^29 (a carry-free negation).
// Value < (a b + c d) / R' + p; limbs exactly normalised.
template <class P> PLK_DI Fz<P> fz_mul_add2(const Fz<P>& a, const Fz<P>& b, const Fz<P>& c, const Fz<P>& d) {
    constexpr int NZ = FzCfg<P>::NZ;
    constexpr uint32_t M = FzCfg<P>::M;
    static_assert(NZ <= 14, "column bound of the two-product form");
    uint32_t q[NZ];
    Fz<P> r;
    const uint64_t m64 = fz_opaque64((uint64_t)M);
    uint64_t acc = 0;
    const uint32_t p_pow2 = FzPow2Limb<P>::index() >= 0 ? fz_opaque(FzCfg<P>::plimb(FzPow2Limb<P>::index() >= 0 ? FzPow2Limb<P>::index() : 0)) : 0u;
#pragma unroll
    for (int k = 0; k <= 2 * NZ - 2; ++k) {
        if (k < NZ) acc += m64;
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (j >= 0 && j < NZ) {
                acc = (uint64_t)a.l[i] * b.l[j] + acc;
                acc = (uint64_t)c.l[i] * d.l[j] + acc;
            }
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (i < k && j >= 1 && j < NZ && FzCfg<P>::plimb(j) != 0u)
                acc = (uint64_t)q[i] * (j == FzPow2Limb<P>::index() ? p_pow2 : FzCfg<P>::plimb(j)) + acc;
        }
        if (k < NZ) q[k] = ~(uint32_t)acc & M;
        else r.l[k - NZ] = (uint32_t)acc & M;
        acc = fz_shr29(acc);
    }
    r.l[NZ - 1] = (uint32_t)acc;
    return r;
}
// a b - c d (+ 2^K p d): every operand carried, value(c) <= 2^K p - margin (fz_sub); value < (a b + 2^K p d) / R' + p
template <class P, int K> PLK_DI Fz<P> fz_mul_sub2(const Fz<P>& a, const Fz<P>& b, const Fz<P>& c, const Fz<P>& d) {
    Fz<P> z;
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) z.l[i] = 0u;
    return fz_mul_add2<P>(a, b, fz_sub<P, K>(z, c), d);
}

// A SUM of products (and plain values) through one reduction, one pair at a time: 2 NZ - 1 column accumulators (operand scanning), so a
// term is dead as soon as it has been multiplied in - a product-scanning form would have to hold every pair to the end.
//   FzWide w; fz_wide_clear(w); fz_wide_mac(w, a, b); ...; fz_wide_add(w, v); r = fz_wide_reduce(w);   r = (sum a b) / R' + sum v  (mod p)
// Column bound, in units of NZ 2^58 (one product of a carried operand, limbs < 2^29 + 8, with an exactly normalised one, < 2^29): the
// reduction adds one unit + 2^37, so a 64-bit column holds (2^64 - NZ 2^58) / (NZ 2^58) = 6 units for nine limbs (FZ_WIDE_UNITS; a product
// with one operand of limbs < 2^30 - a row as it is loaded - counts two units, with both four).  Real values leave the top limb far
// smaller than the others, so the fullest column has NZ - 1 such terms, not NZ.
// fz_wide_add puts a value in at weight R' (its limbs join the upper columns; limbs below 2^31): the result's value grows by it.
template <class P> struct FzWide {
    uint64_t t[2 * FzCfg<P>::NZ];  // t[2 NZ - 1]: the top limbs of added values
};
constexpr int FZ_WIDE_UNITS = 6;
template <class P> PLK_DI void fz_wide_clear(FzWide<P>& w) {
#pragma unroll
    for (int k = 0; k < 2 * FzCfg<P>::NZ; ++k) w.t[k] = 0;
}
template <class P> PLK_DI void fz_wide_mac(FzWide<P>& w, const Fz<P>& a, const Fz<P>& b) {
    constexpr int NZ = FzCfg<P>::NZ;
#pragma unroll
    for (int i = 0; i < NZ; ++i)
#pragma unroll
        for (int j = 0; j < NZ; ++j) w.t[i + j] = (uint64_t)a.l[i] * b.l[j] + w.t[i + j];
}
template <class P> PLK_DI void fz_wide_add(FzWide<P>& w, const Fz<P>& v) {
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) w.t[FzCfg<P>::NZ + i] += v.l[i];
}
template <class P> PLK_DI Fz<P> fz_wide_reduce(const FzWide<P>& w) {
    constexpr int NZ = FzCfg<P>::NZ;
    constexpr uint32_t M = FzCfg<P>::M;
    uint32_t q[NZ];
    Fz<P> r;
    const uint64_t m64 = fz_opaque64((uint64_t)M);
    uint64_t acc = 0;
    const uint32_t p_pow2 = FzPow2Limb<P>::index() >= 0 ? fz_opaque(FzCfg<P>::plimb(FzPow2Limb<P>::index() >= 0 ? FzPow2Limb<P>::index() : 0)) : 0u;
#pragma unroll
    for (int k = 0; k <= 2 * NZ - 2; ++k) {
        acc += w.t[k];
        if (k < NZ) acc += m64;
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (i < k && j >= 1 && j < NZ && FzCfg<P>::plimb(j) != 0u)
                acc = (uint64_t)q[i] * (j == FzPow2Limb<P>::index() ? p_pow2 : FzCfg<P>::plimb(j)) + acc;
        }
        if (k < NZ) q[k] = ~(uint32_t)acc & M;
        else r.l[k - NZ] = (uint32_t)acc & M;
        acc = fz_shr29(acc);
    }
    r.l[NZ - 1] = (uint32_t)(acc + w.t[2 * NZ - 1]);
    return r;
}

// a^2 / R': the 2 a_i a_j cross terms are formed once from a doubled copy of a (NZ (NZ+1) / 2 products)
template <class P> PLK_DI Fz<P> fz_sqr(const Fz<P>& a) {
    constexpr int NZ = FzCfg<P>::NZ;
    constexpr uint32_t M = FzCfg<P>::M;
    uint32_t q[NZ], a2[NZ];
#pragma unroll
    for (int i = 0; i < NZ; ++i) a2[i] = a.l[i] << 1;
    Fz<P> r;
    const uint64_t m64 = fz_opaque64((uint64_t)M);
    uint64_t acc = 0;
    const uint32_t p_pow2 = FzPow2Limb<P>::index() >= 0 ? fz_opaque(FzCfg<P>::plimb(FzPow2Limb<P>::index() >= 0 ? FzPow2Limb<P>::index() : 0)) : 0u;
#pragma unroll
    for (int k = 0; k <= 2 * NZ - 2; ++k) {
        if (k < NZ) acc += m64;
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (j > i && j < NZ) acc = (uint64_t)a.l[i] * a2[j] + acc;
            if (j == i) acc = (uint64_t)a.l[i] * a.l[i] + acc;
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (i < k && j >= 1 && j < NZ && FzCfg<P>::plimb(j) != 0u)
                acc = (uint64_t)q[i] * (j == FzPow2Limb<P>::index() ? p_pow2 : FzCfg<P>::plimb(j)) + acc;
        }
        if (k < NZ) q[k] = ~(uint32_t)acc & M;
        else r.l[k - NZ] = (uint32_t)acc & M;
        acc = fz_shr29(acc);
    }
    r.l[NZ - 1] = (uint32_t)acc;
    return r;
}

// ---- reduction without a multiplication ----------------------------------------------------------
// value < 2^(29 NZ - 2) with lazy limbs  ->  value below 2p congruent to it, without a multiplication: q = an under-estimate
// of floor(v / p) from the top 30 bits, v - q p by one multiply-subtract per limb.  Replaces the "multiplication by one" that
// only served to reduce the outputs of the last pass (~220 instructions) by ~70.
template <class P> PLK_DI Fz<P> fz_reduce_small(Fz<P> v) {
    constexpr int NZ = FzCfg<P>::NZ;
    fz_normalize<P>(v);
    // top = floor(v / 2^(29 (NZ - 1) - 3)): the top limb (< 2^27 for v < 2^(29 NZ - 2)) and the upper 3 bits of the one below
    const uint32_t top = (v.l[NZ - 1] << 3) | (v.l[NZ - 2] >> 26);
    constexpr uint32_t p_top = (FzCfg<P>::plimb(NZ - 1) << 3) | (FzCfg<P>::plimb(NZ - 2) >> 26);  // floor(p / 2^(29 (NZ - 1) - 3)), >= 2^23
    const uint32_t q = top / (p_top + 1u);  // <= floor(v / p), short of it by at most 1 (division by a constant)
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        acc += (int64_t)v.l[i] - (int64_t)((uint64_t)q * FzCfg<P>::plimb(i));
        if (i < NZ - 1) {
            v.l[i] = (uint32_t)acc & FzCfg<P>::M;
            acc >>= 29;  // arithmetic: a borrow travels as -1
        } else {
            v.l[i] = (uint32_t)acc;
        }
    }
    return v;  // in [0, 2p)
}

// ---- predicates --------------------------------------------------------------------------------
// value == 0 mod p for a value < 2p with exactly normalised limbs (what fz_mul / fz_sqr return)
template <class P> PLK_DI bool fz_is_zero_mod_p(const Fz<P>& a) {
    constexpr int NZ = FzCfg<P>::NZ;
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        z |= a.l[i];
        e |= a.l[i] ^ FzCfg<P>::plimb(i);
    }
    return z == 0 || e == 0;
}

// ---- constants (R'-form), derived at compile time from the modulus ------------------------------
// 2^e mod p as 29-bit limbs, by repeated doubling of 1 (constexpr, host side of the compiler only)
template <class P> struct FzConst {
    static constexpr int NZ = FzCfg<P>::NZ;
    struct Limbs {
        uint32_t l[NZ];
    };
    static constexpr bool geq_p(const Limbs& x) {
        for (int i = NZ - 1; i >= 0; --i) {
            if (x.l[i] != FzCfg<P>::plimb(i)) return x.l[i] > FzCfg<P>::plimb(i);
        }
        return true;
    }
    static constexpr Limbs dbl_mod(Limbs x) {
        uint32_t c = 0;
        for (int i = 0; i < NZ; ++i) {
            uint32_t t = (x.l[i] << 1) + c;
            if (i < NZ - 1) {
                x.l[i] = t & FzCfg<P>::M;
                c = t >> 29;
            } else {
                x.l[i] = t;
            }
        }
        if (geq_p(x)) {
            uint32_t borrow = 0;
            for (int i = 0; i < NZ; ++i) {
                uint32_t t = x.l[i] - FzCfg<P>::plimb(i) - borrow;
                if (i < NZ - 1) {
                    x.l[i] = t & FzCfg<P>::M;
                    borrow = (t >> 31) & 1u;
                } else {
                    x.l[i] = t;
                }
            }
        }
        return x;
    }
    static constexpr Limbs pow2(int e) {
        Limbs x{};
        x.l[0] = 1;
        for (int i = 0; i < e; ++i) x = dbl_mod(x);
        return x;
    }
};

// multiply by this to take an R-form value (x 2^(32 NL)) into R'-form (x 2^(29 NZ)):  2^(2*29 NZ - 32 NL)
template <class P> PLK_DI Fz<P> fz_zero() {
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) r.l[i] = 0;
    return r;
}

template <class P> PLK_DI Fz<P> fz_const_r_to_rprime() {
    constexpr auto c = FzConst<P>::pow2(2 * 29 * FzCfg<P>::NZ - 32 * P::NL);
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) r.l[i] = c.l[i];
    return r;
}
// multiply by this to take an R'-form value back to R-form:  2^(32 NL)
template <class P> PLK_DI Fz<P> fz_const_rprime_to_r() {
    constexpr auto c = FzConst<P>::pow2(32 * P::NL);
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) r.l[i] = c.l[i];
    return r;
}
// multiply by this to take the plain inverse of an R-form INTEGER, (x 2^(32 NL))^-1, to the R'-form of x^-1:  2^(32 NL + 2 * 29 NZ)
template <class P> PLK_DI Fz<P> fz_const_raw_inverse_to_rprime() {
    constexpr auto c = FzConst<P>::pow2(32 * P::NL + 2 * 29 * FzCfg<P>::NZ);
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) r.l[i] = c.l[i];
    return r;
}
// 1 in R'-form: 2^(29 NZ)
template <class P> PLK_DI Fz<P> fz_one_rprime() {
    constexpr auto c = FzConst<P>::pow2(29 * FzCfg<P>::NZ);
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) r.l[i] = c.l[i];
    return r;
}

}  // namespace plk
