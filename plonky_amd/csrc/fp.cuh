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

// Modular inverse by Bernstein-Yang division steps ("safegcd", half-delta variant) on signed 30-bit limbs,
// 30 steps at a time on the low words with the transition matrix applied to the full numbers afterwards
// (the organisation of libsecp256k1's modinv32, restated for these moduli: p = 1 mod 2^32, so p^-1 mod 2^30 = 1).
// About a tenth of the instructions of the bit-by-bit Euclid above and branch-free, so a wave stays converged.
// Same contract as fe_inv_eea: Montgomery in, Montgomery out (monty.rs:162-166), 0 -> 0.
//
// VAR: the data-dependent form for ONE lane (the normalisation at the end of an MSM): the plain division steps (delta starts at 1),
// runs of them taken at once - the zeros at the bottom of g by a count, then up to six bits of g cancelled by one multiple
// w = -g / f mod 2^k of f (f^-1 = f (2 - f^2) mod 2^6 for odd f), as long as no swap falls due inside the run (k <= eta + 1).
// About a third of the instructions of the fixed-length loop; the matrix and its application to (d, e), (f, g) are the same.
// A wave whose lanes all invert (the fold kernels) keeps the branch-free form: there the longest lane is what everyone pays.
//
// tuning builds (-DPLK_FINAL_TRACE, one translation unit: msm_tail.hip): shader-clock stamps of thread 0 of the first two blocks along
// k_msm_final's chain and inside the normalisation, read back by plk_debug_final_trace (tools/final_trace_probe.py)
#if defined(PLK_FINAL_TRACE) && defined(__HIPCC__)
__device__ unsigned long long g_final_trace[2][16];
#define PLK_FT(i)                                                                          \
    do {                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 2) g_final_trace[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define PLK_FT_NOW() __builtin_amdgcn_s_memtime()
#define g_final_trace_get(i) ((threadIdx.x == 0 && blockIdx.x < 2) ? g_final_trace[blockIdx.x][i] : 0ull)
#define PLK_FT_ADD(i, v)                                                                   \
    do {                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 2) g_final_trace[blockIdx.x][i] += (v);       \
    } while (0)
#else
#define PLK_FT(i) do { } while (0)
#define PLK_FT_NOW() 0ull
#define g_final_trace_get(i) 0ull
#define PLK_FT_ADD(i, v) do { } while (0)
#endif
// MODE 0: the fixed form; 1: VAR; 2: VAR with ONE active lane in the wave (round 5).  The runs of division steps work on the low words
// of f and g alone - a chain of ~15 dependent 32-bit operations per run, three of them multiplications, twelve runs per 30 steps - and on
// the vector unit every one of them waits out the pipeline's latency.  With one lane active the low words, the matrix and eta are uniform
// by construction: they are read into scalar registers (readfirstlane) and the whole inner loop runs on the scalar unit with scalar
// branches; the matrix then multiplies the vector limbs as scalar operands.  profiles/r05_final_kernel_trace.txt has the kernel's stamps.
// RAW: return the plain inverse of the INTEGER a (no Montgomery fix-up) - the caller folds the fix-up into a product it makes anyway.
PLK_DI uint32_t plk_uniform_u32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
#endif
}
template <class P, int MODE, bool RAW = false> PLK_DNI Fe<P> fe_inv_safegcd_impl(const Fe<P>& a) {
    constexpr bool VAR = MODE == 1 || MODE == 2, UNI = MODE >= 2;  // MODE 3: the fixed form with scalar low words
    constexpr int NL = P::NL;
    constexpr int N = (NL * 32 + 29) / 30;          // 9 limbs for 256 bits, 13 for 384
    // fixed form (half-delta steps): 590 steps suffice below 2^256, 886 below 2^384; plain steps (VAR): 741 / 1103 at most
    constexpr int ITER = VAR ? (NL == 8 ? 25 : 37) : (NL == 8 ? 20 : 30);
    constexpr int32_t M30 = (int32_t)(0xffffffffu >> 2);
    if (fe_is_zero<P>(a)) return a;
    int32_t m[N], f[N], g[N], d[N], e[N];
    // 32-bit limbs -> 30-bit limbs
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        uint64_t mm = w < NL ? P::MOD[w] : 0u, gg = w < NL ? a.v[w] : 0u;
        if (w + 1 < NL) {
            mm |= (uint64_t)P::MOD[w + 1] << 32;
            gg |= (uint64_t)a.v[w + 1] << 32;
        }
        m[i] = (int32_t)((uint32_t)(mm >> sh) & (uint32_t)M30);
        g[i] = (int32_t)((uint32_t)(gg >> sh) & (uint32_t)M30);
        f[i] = m[i];
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    int32_t zeta = -1;  // -(delta + 1/2), delta = 1/2
    if constexpr (UNI && RAW) {
        PLK_FT_ADD(13, 0ull - g_final_trace_get(13));
        PLK_FT_ADD(14, 0ull - g_final_trace_get(14));
        PLK_FT_ADD(15, 0ull - g_final_trace_get(15));
    }
    for (int it = 0; it < ITER; ++it) {
        const unsigned long long ft0 = PLK_FT_NOW();
        // 30 division steps on the low words; (u v; q r) is 2^30 times the transition matrix
        uint32_t u = 1, v = 0, q = 0, r = 1, fl = (uint32_t)f[0], gl = (uint32_t)g[0];
        if constexpr (UNI) {
            fl = plk_uniform_u32(fl);
            gl = plk_uniform_u32(gl);
        }
        if constexpr (VAR) {
            // here zeta is eta = -delta of the plain steps (same start: -1)
            int left = 30;
            for (;;) {
                const int zeros = __builtin_ctz(gl | (0xffffffffu << left));  // at most `left`
                gl >>= zeros;
                u <<= zeros;
                v <<= zeros;
                zeta -= zeros;
                left -= zeros;
                if (left == 0) break;
                if (zeta < 0) {  // g odd, delta > 0: (f, g) <- (g, -f), and the matrix rows with them
                    zeta = -zeta;
                    uint32_t t = fl; fl = gl; gl = 0u - t;
                    t = u; u = q; q = 0u - t;
                    t = v; v = r; r = 0u - t;
                }
                // up to min(eta + 1, left, 6) steps without a swap: g += w f clears that many low bits
                const int limit = zeta + 1 > left ? left : zeta + 1;
                const uint32_t mask = (0xffffffffu >> (32 - limit)) & 63u;
                const uint32_t w = (fl * gl * (fl * fl - 2u)) & mask;
                gl += fl * w;
                q += u * w;
                r += v * w;
            }
        } else
#pragma unroll
        for (int i = 0; i < 30; ++i) {
            uint32_t c1 = (uint32_t)(zeta >> 31);           // zeta < 0
            const uint32_t c2 = (uint32_t)0 - (gl & 1u);     // g odd
            const uint32_t x = (fl ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // +-f, +-u, +-v
            gl += x & c2;
            q += y & c2;
            r += z & c2;
            c1 &= c2;
            zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;       // -zeta - 2 or zeta - 1
            fl += gl & c1;
            u += q & c1;
            v += r & c1;
            gl >>= 1;
            u <<= 1;
            v <<= 1;
        }
        const unsigned long long ft1 = PLK_FT_NOW();
        if constexpr (UNI && RAW) PLK_FT_ADD(13, ft1 - ft0);
        const int64_t tu = (int32_t)u, tv = (int32_t)v, tq = (int32_t)q, tr = (int32_t)r;
        // (d, e) <- t (d, e) / 2^30 mod p: a multiple of p makes the low 30 bits vanish first
        {
            const int32_t sd = d[N - 1] >> 31, se = e[N - 1] >> 31;
            int32_t md = ((int32_t)tu & sd) + ((int32_t)tv & se);
            int32_t me = ((int32_t)tq & sd) + ((int32_t)tr & se);
            int64_t cd = tu * d[0] + tv * e[0];
            int64_t ce = tq * d[0] + tr * e[0];
            md -= (int32_t)(((uint32_t)cd + (uint32_t)md) & (uint32_t)M30);  // p^-1 mod 2^30 = 1
            me -= (int32_t)(((uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
            cd += (int64_t)m[0] * md;
            ce += (int64_t)m[0] * me;
            cd >>= 30;
            ce >>= 30;
#pragma unroll
            for (int i = 1; i < N; ++i) {
                cd += tu * d[i] + tv * e[i] + (int64_t)m[i] * md;
                ce += tq * d[i] + tr * e[i] + (int64_t)m[i] * me;
                d[i - 1] = (int32_t)cd & M30;
                e[i - 1] = (int32_t)ce & M30;
                cd >>= 30;
                ce >>= 30;
            }
            d[N - 1] = (int32_t)cd;
            e[N - 1] = (int32_t)ce;
        }
        // (f, g) <- t (f, g) / 2^30, exact
        {
            int64_t cf = tu * f[0] + tv * g[0];
            int64_t cg = tq * f[0] + tr * g[0];
            cf >>= 30;
            cg >>= 30;
#pragma unroll
            for (int i = 1; i < N; ++i) {
                cf += tu * f[i] + tv * g[i];
                cg += tq * f[i] + tr * g[i];
                f[i - 1] = (int32_t)cf & M30;
                g[i - 1] = (int32_t)cg & M30;
                cf >>= 30;
                cg >>= 30;
            }
            f[N - 1] = (int32_t)cf;
            g[N - 1] = (int32_t)cg;
        }
        // ITER covers the worst case (590 / 886 division steps); a random element is done after 17-18 of the 20 (25-26 of the 30)
        // iterations (measured on the host over 20 000 elements per field).  Once g = 0 further steps change nothing (g even: no
        // swap, the matrix is 2^30 times the identity on d), so the loop may stop: lanes of a wave that are done wait for the
        // others at no cost, and the one-lane inversion at the end of every MSM (k_msm_final) is ~12 % shorter.
        int32_t gz = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) gz |= g[i];
#ifdef PLK_INV_STATS
        if (gz == 0) { ++plk_inv_stats[it]; }
#endif
        if constexpr (UNI) gz = (int32_t)plk_uniform_u32((uint32_t)gz);
        if constexpr (UNI && RAW) {
            PLK_FT_ADD(14, PLK_FT_NOW() - ft1);
            PLK_FT_ADD(15, 1ull);
        }
        if (gz == 0) break;
    }
    // g = 0, f = +-1 and d = +-a^-1 in (-2p, p): fix the sign, bring into [0, p)
    {
        const int32_t neg = f[N - 1] >> 31;
        int32_t add = d[N - 1] >> 31;
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = ((d[i] + (m[i] & add)) ^ neg) - neg;
#pragma unroll
        for (int i = 0; i + 1 < N; ++i) {
            d[i + 1] += d[i] >> 30;
            d[i] &= M30;
        }
        add = d[N - 1] >> 31;
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] += m[i] & add;
#pragma unroll
        for (int i = 0; i + 1 < N; ++i) {
            d[i + 1] += d[i] >> 30;
            d[i] &= M30;
        }
    }
    // 30-bit limbs -> 32-bit limbs, then x R^3 / R: (a R)^-1 R^2 = a^-1 R
    Fe<P> r, r3;
#pragma unroll
    for (int w = 0; w < NL; ++w) {
        const int bit = 32 * w, i = bit / 30, sh = bit % 30;
        uint64_t acc = (uint64_t)(uint32_t)d[i] >> sh;
        if (i + 1 < N) acc |= (uint64_t)(uint32_t)d[i + 1] << (30 - sh);
        if (i + 2 < N) acc |= (uint64_t)(uint32_t)d[i + 2] << (60 - sh);
        r.v[w] = (uint32_t)acc;
        r3.v[w] = P::R3[w];
    }
    if constexpr (RAW) return r;
    return fe_mul<P>(r, r3);
}

template <class P> PLK_DI Fe<P> fe_inv_safegcd(const Fe<P>& a) { return fe_inv_safegcd_impl<P, 0>(a); }
template <class P> PLK_DI Fe<P> fe_inv_safegcd_var(const Fe<P>& a) { return fe_inv_safegcd_impl<P, 1>(a); }
// ONE active lane in the wave (see MODE 2 above); _raw: the inverse of the integer, without the Montgomery fix-up
// What the single-lane callers use.  tools/lab/inv_latency.hip (profiles/r05_inversion_latency.txt), ticks per inversion on one lane of a lone
// wave: MODE 1 75 k, MODE 0 86 k, MODE 2 82 k, MODE 3 107 k - the scalar unit does not run the dependent low-word chain any faster than
// the vector unit does, so the forms with scalar low words stay tuning options (-DPLK_ONE_LANE_INV_MODE=2 / 3).
#ifndef PLK_ONE_LANE_INV_MODE
#define PLK_ONE_LANE_INV_MODE 1
#endif
template <class P> PLK_DI Fe<P> fe_inv_safegcd_one_lane(const Fe<P>& a) { return fe_inv_safegcd_impl<P, PLK_ONE_LANE_INV_MODE>(a); }
template <class P> PLK_DI Fe<P> fe_inv_safegcd_one_lane_raw(const Fe<P>& a) { return fe_inv_safegcd_impl<P, PLK_ONE_LANE_INV_MODE, true>(a); }

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
