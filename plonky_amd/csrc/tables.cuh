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
        if ((e >> b) & 1) r = fe_mul<P>(r, fe_load<P>(pw + (base_off + b) * 2));
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

template <class P> PLK_DI Fe<P> fe_const(const uint32_t (&c)[P::NL]) {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::NL; ++i) r.v[i] = c[i];
    return r;
}

}  // namespace plk
