// serial.hip -- the reference's canonical byte encodings on the device (fixture interchange).
//
// Reference path                                                        here
//   ToBytes / FromBytes for F            serialization.rs:17-31     ->  k_field_to_bytes / k_field_from_bytes
//     to_canonical_u8_vec / from_canonical_u8_vec   field.rs:67-102     (little-endian bytes of the canonical limbs; "Out of range")
//   ToBytes / FromBytes for AffinePoint  serialization.rs:33-72     ->  k_point_to_bytes / k_point_from_bytes
//     mask byte = zero | (y odd) << 1, then x; decompression through Field::square_root (field.rs:440-472, Tonelli-Shanks)
// Records are fixed size: BYTES per field element (32, or 48 for Bls12377Base), 1 + BYTES per point (the reference's reader
// stops after the mask byte of the identity; a record here simply leaves the rest unused).
#include "common.h"
#include "fp.cuh"
#include "tables.cuh"

namespace plk {

template <class P> PLK_DI bool canonical_in_range(const Fe<P>& c) {  // is_valid_canonical_u64: value < ORDER
    for (int i = P::NL - 1; i >= 0; --i) {
        if (c.v[i] != P::MOD[i]) return c.v[i] < P::MOD[i];
    }
    return false;
}
template <class P> PLK_DI Fe<P> load_bytes(const uint8_t* p) {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::NL; ++i) r.v[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    return r;
}
template <class P> PLK_DI void store_bytes(uint8_t* p, const Fe<P>& c) {
#pragma unroll
    for (int i = 0; i < P::NL; ++i) {
        p[4 * i] = (uint8_t)c.v[i];
        p[4 * i + 1] = (uint8_t)(c.v[i] >> 8);
        p[4 * i + 2] = (uint8_t)(c.v[i] >> 16);
        p[4 * i + 3] = (uint8_t)(c.v[i] >> 24);
    }
}

template <class P> __global__ void k_field_to_bytes(const uint4* __restrict__ x, size_t count, uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    store_bytes<P>(out + i * P::NL * 4, fe_to_canonical<P>(fe_load<P>(x + i * (P::NL / 4))));
}
template <class P> __global__ void k_field_from_bytes(const uint8_t* __restrict__ in, size_t count, uint4* __restrict__ out, unsigned* __restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Fe<P> c = load_bytes<P>(in + i * P::NL * 4);
    if (!canonical_in_range<P>(c)) {
        atomicAdd(bad, 1u);  // "Out of range" (field.rs:100)
        fe_store<P>(out + i * (P::NL / 4), fe_zero<P>());
        return;
    }
    fe_store<P>(out + i * (P::NL / 4), fe_from_canonical<P>(c));
}

// x^e for a multi-limb exponent (little-endian 32-bit words)
template <class P> PLK_DI Fe<P> fe_pow_limbs(const Fe<P>& x, const uint32_t (&e)[P::NL]) {
    Fe<P> r = fe_one<P>();
    bool started = false;
    for (int i = P::NL - 1; i >= 0; --i)
        for (int b = 31; b >= 0; --b) {
            if (started) r = fe_sqr<P>(r);
            if ((e[i] >> b) & 1u) {
                r = started ? fe_mul<P>(r, x) : x;
                started = true;
            }
        }
    return r;
}
// Field::square_root (field.rs:440-472): Tonelli-Shanks with z = g^T, T = (p - 1) / 2^TWO_ADICITY.  Returns false for a
// non-residue (the reference tests Euler's criterion first; here the same fact falls out of the loop: b = a^T has order
// dividing 2^(adicity - 1) exactly when a is a square).
template <class P> PLK_DNI bool fe_sqrt(const Fe<P>& a, Fe<P>& root) {
    if (fe_is_zero<P>(a)) {
        root = a;
        return true;
    }
    // (T - 1) / 2 from the modulus: T = (p - 1) >> adicity is odd
    uint32_t e[P::NL];
    {
        uint32_t t[P::NL];
        for (int i = 0; i < P::NL; ++i) t[i] = P::MOD[i];
        t[0] -= 1u;  // p is odd
        constexpr int sh = P::TWO_ADICITY + 1;  // (T - 1) / 2 = (p - 1) >> (adicity + 1), T odd
        for (int i = 0; i < P::NL; ++i) {
            const int src = i + sh / 32, bit = sh % 32;
            uint32_t lo = src < P::NL ? t[src] : 0u, hi = src + 1 < P::NL ? t[src + 1] : 0u;
            e[i] = bit ? (lo >> bit) | (hi << (32 - bit)) : lo;
        }
    }
    Fe<P> z = fe_const<P>(P::ROOT_2ADIC);
    Fe<P> w = fe_pow_limbs<P>(a, e);
    Fe<P> x = fe_mul<P>(w, a);
    Fe<P> b = fe_mul<P>(x, w);
    const Fe<P> one = fe_one<P>();
    int v = P::TWO_ADICITY;
    while (!fe_eq<P>(b, one)) {
        int k = 0;
        Fe<P> b2k = b;
        while (!fe_eq<P>(b2k, one)) {
            b2k = fe_sqr<P>(b2k);
            ++k;
            if (k >= v) return false;  // not a square
        }
        const int j = v - k - 1;
        w = z;
        for (int s = 0; s < j; ++s) w = fe_sqr<P>(w);
        z = fe_sqr<P>(w);
        b = fe_mul<P>(b, z);
        x = fe_mul<P>(x, w);
        v = k;
    }
    root = x;
    return true;
}

// serialization.rs:33-45
template <class P> __global__ void k_point_to_bytes(const uint4* __restrict__ xy, const uint8_t* __restrict__ zero, size_t count, uint8_t* __restrict__ out) {
    constexpr int W = P::NL / 4, B = P::NL * 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Fe<P> x = fe_to_canonical<P>(fe_load<P>(xy + i * 2 * W)), y = fe_to_canonical<P>(fe_load<P>(xy + i * 2 * W + W));
    const uint8_t mask = (uint8_t)(((zero && zero[i]) ? 1 : 0) | ((y.v[0] & 1u) ? 2 : 0));
    out[i * (B + 1)] = mask;
    store_bytes<P>(out + i * (B + 1) + 1, x);
}
// serialization.rs:47-72; status[i]: 0 ok, 1 "Out of range", 2 "Invalid x coordinate"
template <class P>
__global__ void __launch_bounds__(64) k_point_from_bytes(const uint8_t* __restrict__ in, size_t count, uint32_t b_coeff, uint4* __restrict__ out_xy,
                                                         uint8_t* __restrict__ out_zero, uint8_t* __restrict__ status) {
    constexpr int W = P::NL / 4, B = P::NL * 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint8_t mask = in[i * (B + 1)];
    Fe<P> x = fe_zero<P>(), y = fe_zero<P>();
    uint8_t st = 0, zf = 0;
    if (mask & 1) {
        zf = 1;
    } else {
        const Fe<P> c = load_bytes<P>(in + i * (B + 1) + 1);
        if (!canonical_in_range<P>(c)) {
            st = 1;
        } else {
            x = fe_from_canonical<P>(c);
            Fe<P> bc = fe_zero<P>();
            bc.v[0] = b_coeff;
            const Fe<P> rhs = fe_add<P>(fe_mul<P>(fe_sqr<P>(x), x), fe_from_canonical<P>(bc));  // x^3 + A x + B, A = 0 on every in-scope curve
            Fe<P> r;
            if (!fe_sqrt<P>(rhs, r)) {
                st = 2;
                x = fe_zero<P>();
            } else {
                const Fe<P> rc = fe_to_canonical<P>(r);
                y = ((rc.v[0] & 1u) == (uint32_t)((mask & 2) >> 1)) ? r : fe_neg<P>(r);
            }
        }
    }
    fe_store<P>(out_xy + i * 2 * W, x);
    fe_store<P>(out_xy + i * 2 * W + W, y);
    out_zero[i] = zf;
    status[i] = st;
}

int field_bytes_impl(int field, int from_bytes, const void* d_in, size_t count, void* d_out, unsigned* d_bad, hipStream_t stream) {
    if (count == 0) return PLK_OK;
    const unsigned blocks = (unsigned)((count + 127) / 128);
    switch (field) {
#define CASE(ID, P)                                                                                                   \
    case ID:                                                                                                          \
        if (from_bytes) k_field_from_bytes<P><<<blocks, 128, 0, stream>>>((const uint8_t*)d_in, count, (uint4*)d_out, d_bad); \
        else k_field_to_bytes<P><<<blocks, 128, 0, stream>>>((const uint4*)d_in, count, (uint8_t*)d_out);             \
        break;
        CASE(PLK_FIELD_TWEEDLEDEE_BASE, TweedledeeBaseParams)
        CASE(PLK_FIELD_TWEEDLEDUM_BASE, TweedledumBaseParams)
        CASE(PLK_FIELD_BLS12_377_SCALAR, Bls12377ScalarParams)
        CASE(PLK_FIELD_BLS12_377_BASE, Bls12377BaseParams)
        CASE(PLK_FIELD_PALLAS_BASE, PallasBaseParams)
        CASE(PLK_FIELD_VESTA_BASE, VestaBaseParams)
#undef CASE
        default: return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

// curve constants B (A = 0): tweedledee_curve.rs:11-12 (5), tweedledum_curve.rs:11-13 (7), bls12_377_curve.rs:14-15 (1)
// pallas_curve.rs:12, vesta_curve.rs:12 (5)
static uint32_t curve_b(int curve) { return curve == PLK_CURVE_TWEEDLEDUM ? 7u : curve == PLK_CURVE_BLS12_377 ? 1u : 5u; }

int point_bytes_impl(int curve, int from_bytes, const void* d_in, const void* d_zero, size_t count, void* d_out, void* d_out_zero, void* d_status,
                     hipStream_t stream) {
    if (count == 0) return PLK_OK;
    switch (curve) {
#define CASE(ID, P)                                                                                                               \
    case ID:                                                                                                                      \
        if (from_bytes)                                                                                                           \
            k_point_from_bytes<P><<<(unsigned)((count + 63) / 64), 64, 0, stream>>>((const uint8_t*)d_in, count, curve_b(curve), (uint4*)d_out, \
                                                                                   (uint8_t*)d_out_zero, (uint8_t*)d_status);    \
        else                                                                                                                      \
            k_point_to_bytes<P><<<(unsigned)((count + 127) / 128), 128, 0, stream>>>((const uint4*)d_in, (const uint8_t*)d_zero, count, (uint8_t*)d_out); \
        break;
        CASE(PLK_CURVE_TWEEDLEDEE, TweedledeeBaseParams)
        CASE(PLK_CURVE_TWEEDLEDUM, TweedledumBaseParams)
        CASE(PLK_CURVE_BLS12_377, Bls12377BaseParams)
        CASE(PLK_CURVE_PALLAS, PallasBaseParams)
        CASE(PLK_CURVE_VESTA, VestaBaseParams)
#undef CASE
        default: return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

}  // namespace plk
