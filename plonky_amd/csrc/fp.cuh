// fp.cuh -- prime-field arithmetic for gfx950 on 32-bit limbs (device only).
//
// Replaces, on the device, the reference's u64-limb Montgomery arithmetic:
//   src/field/monty.rs:38-160 (add/sub/neg/CIOS multiply/square, 4 x u64),
//   src/field/bls12_377_base.rs:58-98 (6 x u64 CIOS),
//   src/bigint/bigint_arithmetic.rs:11-55 (cmp / add / sub).
// Values are the reference's in-memory representation: Montgomery form with R = 2^(32*NL)
// (= 2^256 or 2^384), always fully reduced (< p), little-endian limbs, so a u64-limb element
// of the reference is bit-identical to 2 consecutive u32 limbs here and every result of
// fe_add / fe_sub / fe_mul is the unique representative the reference would produce.
//
// CDNA4 mapping: the only wide multiplier is v_mad_u64_u32 (32x32+64 -> 64); every limb
// product below is written as (uint64_t)a*b + c so it lowers to exactly one of them, and the
// carries ride in the upper half of the 64-bit accumulator.  All four in-scope moduli are
// = 1 (mod 2^32), hence -p^-1 = -1 (mod 2^32) and the Montgomery quotient digit is just
// q = -t0: no multiply; the Tweedle moduli (2^254 + 125-bit c) additionally have four zero
// limbs, which the fully unrolled, constexpr-modulus loops fold away (24 instead of 64
// reduction multiplies).  No MFMA: there is no dense contraction in this arithmetic.
#pragma once
#include <stdint.h>

#include "field_params.cuh"

// The arithmetic below is plain C++ so that tests/test_fp_host.py can compile it with g++ and
// sweep it against Python integers without a GPU; under hipcc it is device code.
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define PLK_DI __device__ __forceinline__
#define PLK_DNI __device__ __noinline__
#else
#define PLK_DI inline
#define PLK_DNI inline
#endif

#include "fp29.cuh"

namespace plk {

template <class P> struct Fe {
    uint32_t v[P::NL];
};

template <class P> PLK_DI Fe<P> fe_zero() {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::NL; ++i) r.v[i] = 0;
    return r;
}
template <class P> PLK_DI Fe<P> fe_one() {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::NL; ++i) r.v[i] = P::ONE[i];
    return r;
}
template <class P> PLK_DI bool fe_is_zero(const Fe<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::NL; ++i) o |= a.v[i];
    return o == 0;
}
template <class P> PLK_DI bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::NL; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// r = t - p if t >= p else t   (t < 2p)
template <class P> PLK_DI void fe_cond_sub_p(uint32_t (&t)[P::NL]) {
    constexpr int N = P::NL;
    uint32_t d[N];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t x = (uint64_t)t[i] - P::MOD[i] - borrow;
        d[i] = (uint32_t)x;
        borrow = (x >> 32) & 1;
    }
    bool ge = borrow == 0;
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = ge ? d[i] : t[i];
}

// monty.rs:38-46
template <class P> PLK_DI Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::NL;
    static_assert(P::MOD[N - 1] < 0x80000000u, "needs a spare top bit: a + b must not overflow the limbs");
    Fe<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t x = (uint64_t)a.v[i] + b.v[i] + c;
        r.v[i] = (uint32_t)x;
        c = x >> 32;
    }
    fe_cond_sub_p<P>(r.v);
    return r;
}
// monty.rs:48-56
template <class P> PLK_DI Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::NL;
    uint32_t d[N];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t x = (uint64_t)a.v[i] - b.v[i] - borrow;
        d[i] = (uint32_t)x;
        borrow = (x >> 32) & 1;
    }
    // add p back when the subtraction wrapped
    uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
    Fe<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t x = (uint64_t)d[i] + (P::MOD[i] & mask) + c;
        r.v[i] = (uint32_t)x;
        c = x >> 32;
    }
    return r;
}
// monty.rs:58-64
template <class P> PLK_DI Fe<P> fe_neg(const Fe<P>& a) { return fe_sub<P>(fe_zero<P>(), a); }
template <class P> PLK_DI Fe<P> fe_dbl(const Fe<P>& a) { return fe_add<P>(a, a); }

// Montgomery product a*b*R^-1 mod p, fully reduced.  monty.rs:67-107 / bls12_377_base.rs:58-98,
// restated on 32-bit limbs (same radix R, so the same value).
template <class P> PLK_DI Fe<P> fe_mul_cios(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::NL;
    static_assert(P::MOD[0] == 1u, "q = -t0 shortcut needs p = 1 (mod 2^32)");
    static_assert(P::MOD[N - 1] < 0x80000000u, "t < 2p must fit the limbs");
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t bi = b.v[i];
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            uint64_t x = (uint64_t)a.v[j] * bi + t[j] + c;  // <= (2^32-1)^2 + 2(2^32-1) < 2^64
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        // Invariant: t < 2p < 2^(32N) between rounds, so limb N is just this carry.
        const uint32_t tn = (uint32_t)c;
        // quotient digit: q = -t0 * p^-1 = -t0 (mod 2^32) because p = 1 (mod 2^32)
        const uint32_t q = 0u - t[0];
        c = ((uint64_t)q + t[0]) >> 32;  // q*p0 + t0 with p0 = 1: the low word is 0 by construction
#pragma unroll
        for (int j = 1; j < N; ++j) {
            uint64_t x = (uint64_t)q * P::MOD[j] + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        t[N - 1] = tn + (uint32_t)c;  // cannot wrap: the new t is < 2p
    }
    Fe<P> r;
    uint32_t u[N];
#pragma unroll
    for (int i = 0; i < N; ++i) u[i] = t[i];
    fe_cond_sub_p<P>(u);
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = u[i];
    return r;
}
// 256-bit fields: product scanning on 29-bit limbs (fp29.cuh) -- same value, ~2.3x fewer instructions.
template <class P> PLK_DI Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
    if constexpr (P::NL == 8) {
        Fe<P> r;
        fe_mul29_core<P>(a.v, b.v, r.v);
        fe_cond_sub_p<P>(r.v);
        return r;
    } else {
        return fe_mul_cios<P>(a, b);
    }
}
template <class P> PLK_DI Fe<P> fe_sqr(const Fe<P>& a) { return fe_mul<P>(a, a); }

// Montgomery -> canonical: multiply by the integer 1 (monty.rs:174-177 "to_monty")
template <class P> PLK_DI Fe<P> fe_to_canonical(const Fe<P>& a) {
    Fe<P> one = fe_zero<P>();
    one.v[0] = 1;
    return fe_mul<P>(a, one);
}
// canonical -> Montgomery: multiply by R^2 (monty.rs:169-172 "from_monty")
template <class P> PLK_DI Fe<P> fe_from_canonical(const Fe<P>& a) {
    Fe<P> r2;
#pragma unroll
    for (int i = 0; i < P::NL; ++i) r2.v[i] = P::R2[i];
    return fe_mul<P>(a, r2);
}

// a^(p-2).  The reference inverts with a binary extended Euclid (bigint_inverse.rs:6-55); the
// inverse of a non-zero element is unique, so Fermat gives the identical limbs without the
// data-dependent loop.  fe_inv(0) = 0.
template <class P> PLK_DNI Fe<P> fe_inv(const Fe<P>& a) {
    Fe<P> r = fe_one<P>();
    for (int i = P::BITS - 1; i >= 0; --i) {
        r = fe_sqr<P>(r);
        if ((P::PM2[i >> 5] >> (i & 31)) & 1u) r = fe_mul<P>(r, a);
    }
    return r;
}

// The reference's own inversion: binary extended Euclid (bigint_inverse.rs:6-55, "Algorithm 16")
// followed by a Montgomery multiplication by R^3 (monty.rs:162-166).  Data-dependent control
// flow: meant for single-lane use (the final normalisation of an MSM), where it is ~5x shorter
// than the Fermat chain above.  fe_inv_eea(0) = 0.
template <class P> PLK_DNI Fe<P> fe_inv_eea(const Fe<P>& a) {
    constexpr int N = P::NL;
    if (fe_is_zero<P>(a)) return a;
    uint32_t u[N], v[N], b[N], c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        u[i] = a.v[i];
        v[i] = P::MOD[i];
        b[i] = 0;
        c[i] = 0;
    }
    b[0] = 1;
    auto is_one = [](const uint32_t (&x)[N]) {
        uint32_t o = x[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < N; ++i) o |= x[i];
        return o == 0;
    };
    auto shr1 = [](uint32_t (&x)[N]) {
#pragma unroll
        for (int i = 0; i < N - 1; ++i) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
        x[N - 1] >>= 1;
    };
    auto add_p = [](uint32_t (&x)[N]) {  // x += p (x < p, p has a spare top bit)
        uint64_t cy = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)x[i] + P::MOD[i] + cy;
            x[i] = (uint32_t)t;
            cy = t >> 32;
        }
    };
    auto less = [](const uint32_t (&x)[N], const uint32_t (&y)[N]) {
        // x < y  <=>  x - y borrows.  Fully unrolled borrow chain: an early-exit loop would index the
        // arrays dynamically and push u, v, b, c out of registers into scratch memory.
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) bw = (((uint64_t)x[i] - y[i] - bw) >> 32) & 1;
        return bw != 0;
    };
    auto sub = [](uint32_t (&x)[N], const uint32_t (&y)[N]) {  // x -= y (x >= y)
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)x[i] - y[i] - bw;
            x[i] = (uint32_t)t;
            bw = (t >> 32) & 1;
        }
    };
    while (!is_one(u) && !is_one(v)) {
        while ((u[0] & 1u) == 0) {
            shr1(u);
            if (b[0] & 1u) add_p(b);
            shr1(b);
        }
        while ((v[0] & 1u) == 0) {
            shr1(v);
            if (c[0] & 1u) add_p(c);
            shr1(c);
        }
        if (less(u, v)) {
            sub(v, u);
            if (less(c, b)) add_p(c);
            sub(c, b);
        } else {
            sub(u, v);
            if (less(b, c)) add_p(b);
            sub(b, c);
        }
    }
    Fe<P> r, r3;
    const bool use_b = is_one(u);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        r.v[i] = use_b ? b[i] : c[i];
        r3.v[i] = P::R3[i];
    }
    return fe_mul<P>(r, r3);
}

// x/2 mod p for Montgomery or canonical x alike (used to build n^-1 = 2^-log n)
template <class P> PLK_DI Fe<P> fe_half(const Fe<P>& a) {
    constexpr int N = P::NL;
    uint32_t mask = (uint32_t)0 - (a.v[0] & 1u);
    uint32_t s[N + 1];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t x = (uint64_t)a.v[i] + (P::MOD[i] & mask) + c;
        s[i] = (uint32_t)x;
        c = x >> 32;
    }
    s[N] = (uint32_t)c;
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = (s[i] >> 1) | (s[i + 1] << 31);
    return r;
}

#ifdef __HIPCC__
// ---- global / LDS movement: an element is NL/4 16-byte words (AoS, as the reference stores it) ----
template <class P> PLK_DI Fe<P> fe_load(const uint4* p) {
    Fe<P> r;
#pragma unroll
    for (int k = 0; k < P::NL / 4; ++k) {
        uint4 w = p[k];
        r.v[4 * k + 0] = w.x;
        r.v[4 * k + 1] = w.y;
        r.v[4 * k + 2] = w.z;
        r.v[4 * k + 3] = w.w;
    }
    return r;
}
template <class P> PLK_DI void fe_store(uint4* p, const Fe<P>& a) {
#pragma unroll
    for (int k = 0; k < P::NL / 4; ++k) p[k] = make_uint4(a.v[4 * k], a.v[4 * k + 1], a.v[4 * k + 2], a.v[4 * k + 3]);
}

#endif  // __HIPCC__

}  // namespace plk
