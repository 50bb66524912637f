// ecj_lab.cuh -- NOT part of the library (round 5: built into the fold kernels, bit-identical, measured slower, dropped - profiles/NOTES.md;
// include with -I plonky_amd/csrc).  Jacobian coordinates (X, Y, Z: x = X / Z^2, y = Y / Z^3) on the lazily reduced limbs of fz.cuh, for the kernels that
// DOUBLE more than they add: the generator folds of the opening argument (fold.hip) run one chain of ~128 doublings per output with
// 64-192 mixed additions beside it.  A doubling here is 4 squarings + 4 products with 6 reductions (dbl-2009-l, a = 0, with 2 X Y^2
// and 2 Y Z as products and the closing E (D - X3) - 8 Y^4 through one reduction: 729 multiplier instructions on nine limbs) where
// the XYZZ form of the MSM (ecz.cuh) makes 981; a mixed addition is 1233 against 1143 (madd-2007-bl; XYZZ keeps the cheaper addition,
// which is what a bucket accumulation is made of).  Same group elements, and every caller normalises: same canonical results.
// Bounds (units of p; limbs carried below 2^29 + 8 unless "exactly normalised" = a product): X < 8, Y < 4, Z < 2 on entry of both
// operations; a doubling leaves X < 5.2, Y < 1.5, Z < 1.2, a mixed addition X < 7.8, Y < 1.9, Z < 1.2.  Every product operand < 16p.
#pragma once
#include "ecz.cuh"

namespace plk {

template <class FP> struct JacZ {
    Fz<FP> x, y, z;
    bool inf;
};
template <class FP> PLK_DI JacZ<FP> jac_identity() {
    JacZ<FP> r;
    r.x = r.y = r.z = fz_zero<FP>();
    r.inf = true;
    return r;
}

// 2 a
template <class FP> PLK_DI JacZ<FP> jac_dbl(const JacZ<FP>& a) {
    if (a.inf) return a;
    JacZ<FP> r;
    const Fz<FP> A = fz_sqr<FP>(a.x);                                  // X^2 < 1.5
    const Fz<FP> B = fz_sqr<FP>(a.y);                                  // Y^2 < 1.13
    const Fz<FP> B2 = fz_dbl<FP>(B), B4 = fz_dbl<FP>(B2);              // < 2.3, < 4.5
    const Fz<FP> S = fz_mul<FP>(a.x, B4);                              // 4 X Y^2 < 8 * 4.5 / 128 + 1 = 1.3   (D of dbl-2009-l)
    const Fz<FP> E = fz_add<FP>(fz_dbl<FP>(A), A);                     // 3 X^2 < 4.5
    const Fz<FP> F = fz_sqr<FP>(E);                                    // < 1.16
    r.x = fz_sub<FP, 2>(F, fz_dbl<FP>(S));                             // F - 2 D < 1.16 + 4 = 5.2     (2 D < 2.6 <= 4p - margin)
    const Fz<FP> dx = fz_sub<FP, 3>(S, r.x);                           // D - X3 < 1.3 + 8 = 9.3
    // Y3 = E (D - X3) - 8 Y^4 = E (D - X3) + (8p - 4 Y^2) (2 Y^2) through one reduction: (4.5 * 9.3 + 8 * 2.3) / 128 + 1 < 1.5
    r.y = fz_mul_add2<FP>(E, dx, fz_sub<FP, 3>(fz_zero<FP>(), B4), B2);
    r.z = fz_mul<FP>(fz_dbl<FP>(a.y), a.z);                            // 2 Y Z < 8 * 2 / 128 + 1 = 1.13
    r.inf = fz_is_zero_mod_p<FP>(r.z);                                 // a point of order two
    return r;
}

// acc += (x2, y2), an affine point: x2, y2 < 2p (a negated y is 2p - y), limbs below 2^29 + 8
template <class FP> PLK_DI void jac_madd(JacZ<FP>& acc, const Fz<FP>& x2, const Fz<FP>& y2) {
    if (acc.inf) {
        acc.x = x2;
        acc.y = y2;
        acc.z = fz_one_rprime<FP>();
        acc.inf = false;
        return;
    }
    const Fz<FP> zz = fz_sqr<FP>(acc.z);                               // < 1.04
    const Fz<FP> u2 = fz_mul<FP>(x2, zz);                              // < 1.02
    const Fz<FP> s2 = fz_mul<FP>(y2, fz_mul<FP>(acc.z, zz));           // < 1.02
    const Fz<FP> h = fz_sub<FP, 3>(u2, acc.x);                         // U2 - X1 < 1.02 + 8 = 9.02
    const Fz<FP> hh = fz_sqr<FP>(h);                                   // < 1.64
    const Fz<FP> i4 = fz_dbl<FP>(fz_dbl<FP>(hh));                      // I = 4 H^2 < 6.6
    const Fz<FP> j = fz_mul<FP>(h, i4);                                // < 9.02 * 6.6 / 128 + 1 = 1.47
    const Fz<FP> r = fz_dbl<FP>(fz_sub<FP, 2>(s2, acc.y));             // 2 (S2 - Y1) < 2 (1.02 + 4) = 10.04
    const Fz<FP> v = fz_mul<FP>(acc.x, i4);                            // < 8 * 6.6 / 128 + 1 = 1.42
    const Fz<FP> rr = fz_sqr<FP>(r);                                   // < 1.79
    const Fz<FP> z3 = fz_mul<FP>(fz_dbl<FP>(acc.z), h);                // 2 Z1 H < 2.4 * 9.02 / 128 + 1 = 1.17
    if (fz_is_zero_mod_p<FP>(z3)) {
        // H = 0 mod p: the operands share x.  Same point -> double it; opposite points -> the identity.
        if (fz_is_zero_mod_p<FP>(rr)) {
            JacZ<FP> p;
            p.x = x2;
            p.y = y2;
            p.z = fz_one_rprime<FP>();
            p.inf = false;
            acc = jac_dbl<FP>(p);
        } else {
            acc.inf = true;
        }
        return;
    }
    const Fz<FP> x3 = fz_sub<FP, 2>(fz_sub<FP, 1>(rr, j), fz_dbl<FP>(v));  // r^2 - J - 2 V < 1.79 + 2 + 4 = 7.8
    const Fz<FP> vx = fz_sub<FP, 3>(v, x3);                            // < 1.42 + 8 = 9.42
    // Y3 = r (V - X3) - 2 Y1 J through one reduction: (10.04 * 9.42 + 8 * 1.47) / 128 + 1 < 1.9
    acc.y = fz_mul_add2<FP>(r, vx, fz_sub<FP, 3>(fz_zero<FP>(), fz_dbl<FP>(acc.y)), j);
    acc.x = x3;
    acc.z = z3;
}

#ifdef __HIPCC__
// back to the reference's form, then to_affine (curve.rs:206-214): x = X / Z^2, y = Y / Z^3, on the working limbs (the plain inverse of
// the integer Z x 2^(32 NL) returns through one product by a constant that also carries the Montgomery fix-up: emit_affine, ecz.cuh)
template <class FP> PLK_DI void jac_emit_affine(const JacZ<FP>& acc, uint4* out_xy, uint8_t* out_zero) {
    constexpr int W = FP::NL / 4;
    const Fz<FP> back = fz_const_rprime_to_r<FP>();
    Fe<FP> zr = fe_zero<FP>();
    if (!acc.inf) zr = fz_to_fe_canonical<FP>(fz_mul<FP>(acc.z, back));
    if (acc.inf || fe_is_zero<FP>(zr)) {
        fe_store<FP>(out_xy, fe_zero<FP>());
        fe_store<FP>(out_xy + W, fe_zero<FP>());
        *out_zero = 1;
        return;
    }
    const Fz<FP> i1 = fz_mul<FP>(fz_from_fe<FP>(fe_inv_safegcd_impl<FP, 0, true>(zr)), fz_const_raw_inverse_to_rprime<FP>());  // 1 / Z
    const Fz<FP> i2 = fz_sqr<FP>(i1);
    const Fz<FP> xa = fz_mul<FP>(acc.x, i2), ya = fz_mul<FP>(acc.y, fz_mul<FP>(i2, i1));
    fe_store<FP>(out_xy, fz_to_fe_canonical<FP>(fz_mul<FP>(xa, back)));
    fe_store<FP>(out_xy + W, fz_to_fe_canonical<FP>(fz_mul<FP>(ya, back)));
    *out_zero = 0;
}
#endif

}  // namespace plk
