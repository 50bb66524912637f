// tables.cuh -- device helpers shared by the table-building kernels of ntt.hip and poly.hip.
#pragma once
#include "fp.cuh"
#include "fz.cuh"

namespace plk {

// R-form (x 2^256) -> canonical R'-form words (x 2^(29 NZ)): the form every multiplier TABLE is stored in.  The pass
// kernel multiplies R-form data by R'-form factors on 29-bit limbs (fz.cuh) and the product
// x 2^256 * w 2^261 / 2^261 stays in the reference's R-form.
template <class P> PLK_DI Fe<P> to_rprime(const Fe<P>& v) {
    return fz_to_fe_canonical<P>(fz_mul<P>(fz_from_fe<P>(v), fz_const_r_to_rprime<P>()));
}

// prod_b pw[base_off + b]^(bit b of e), b < log_t: a power of w from its table of repeated squares (R-form)
template <class P> PLK_DI Fe<P> pow_from_table(const uint4* pw, int base_off, uint64_t e, int log_t) {
    Fe<P> r = fe_one<P>();
    for (int b = 0; b < log_t; ++b)
        if ((e >> b) & 1) r = fe_mul<P>(r, fe_load<P>(pw + (base_off + b) * (P::NL / 4)));
    return r;
}

// x^e by square-and-multiply, R-form
template <class P> PLK_DI Fe<P> fe_pow_u64(Fe<P> x, uint64_t e) {
    Fe<P> r = fe_one<P>();
    while (e) {
        if (e & 1) r = fe_mul<P>(r, x);
        x = fe_sqr<P>(x);
        e >>= 1;
    }
    return r;
}

// Limb form in global memory, used for the data between two passes (values below 2p with exactly normalised limbs, as the exit
// multiplication leaves them), for the inter-pass twiddle tables and for the partial sums of the quotient numerator: the NZ 29-bit
// limbs of an element in NZ consecutive words - no re-slicing between 32-bit words and 29-bit limbs on the way.
//   * PACKED (nine limbs, the 256-bit fields; round 5): 36 B per element, three 12-byte accesses (global_load / store_dwordx3 only
//     need word alignment).  Until round 5 the nine words were padded to twelve (48 B, three 16-byte accesses): the same number of
//     memory instructions for 4/3 of the bytes - and a transform's passes are memory phase + compute phase + write-back, not
//     overlapped when a pass is one round of workgroups (PLK_LIMB_PACKED=0 builds the padded form; A/B in profiles/r05_ntt_packed.txt).
//   * 14 limbs (Bls12377Base): padded to 16 words, four 16-byte accesses.
// (A planar layout with 4-byte accesses was measured in round 2: 10 % faster on a batch of 9 transforms, 5 % slower on a single one,
// whose passes are one round of workgroups and therefore sensitive to the number of memory instructions in flight.)
#ifndef PLK_LIMB_PACKED
#define PLK_LIMB_PACKED 1
#endif
constexpr bool limb_packed(int nz) { return PLK_LIMB_PACKED && nz % 3 == 0; }
constexpr int limb_words(int nz) { return limb_packed(nz) ? nz : ((nz + 3) / 4) * 4; }  // words per element in limb form
template <class P> constexpr int limb_u4() { return (FzCfg<P>::NZ + 3) / 4; }
struct PlkWords3 {  // 12 bytes, word-aligned (a 3-vector type would be padded to 16)
    uint32_t x, y, z;
};
static_assert(sizeof(PlkWords3) == 12 && alignof(PlkWords3) == 4, "three packed words");
template <class P> PLK_DI Fz<P> limbs_load(const uint32_t* __restrict__ base, size_t e) {
    constexpr int NZ = FzCfg<P>::NZ;
    Fz<P> r;
    if constexpr (limb_packed(NZ)) {
        const PlkWords3* p = reinterpret_cast<const PlkWords3*>(base + e * NZ);
#pragma unroll
        for (int i = 0; i < NZ / 3; ++i) {
            const PlkWords3 v = p[i];
            r.l[3 * i] = v.x; r.l[3 * i + 1] = v.y; r.l[3 * i + 2] = v.z;
        }
    } else {
        constexpr int U = limb_u4<P>();
        const uint4* p = reinterpret_cast<const uint4*>(base) + e * U;
        uint32_t w[4 * U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const uint4 v = p[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int l = 0; l < NZ; ++l) r.l[l] = w[l];
    }
    return r;
}
template <class P> PLK_DI void limbs_store(uint32_t* __restrict__ base, size_t e, const Fz<P>& v) {
    constexpr int NZ = FzCfg<P>::NZ;
    if constexpr (limb_packed(NZ)) {
        PlkWords3* p = reinterpret_cast<PlkWords3*>(base + e * NZ);
#pragma unroll
        for (int i = 0; i < NZ / 3; ++i) {
            PlkWords3 w;
            w.x = v.l[3 * i]; w.y = v.l[3 * i + 1]; w.z = v.l[3 * i + 2];
            p[i] = w;
        }
    } else {
        constexpr int U = limb_u4<P>();
        uint4* p = reinterpret_cast<uint4*>(base) + e * U;
        uint32_t w[4 * U];
#pragma unroll
        for (int l = 0; l < 4 * U; ++l) w[l] = l < NZ ? v.l[l] : 0u;
#pragma unroll
        for (int i = 0; i < U; ++i) p[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
    }
}
constexpr size_t limb_bytes(size_t elems, int nz) { return elems * (size_t)limb_words(nz) * 4; }

template <class P> PLK_DI Fe<P> fe_const(const uint32_t (&c)[P::NL]) {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::NL; ++i) r.v[i] = c[i];
    return r;
}

}  // namespace plk
