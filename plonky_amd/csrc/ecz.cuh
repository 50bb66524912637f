// ecz.cuh -- XYZZ mixed addition on lazily reduced 29-bit-limb coordinates (fz.cuh): the inner
// loop of the MSM bucket accumulation.  Same formulas as ec.cuh (EFD madd-2008-s, mdbl-2008-s-1,
// a = 0), same group law as the reference's P + A addition (curve_adds.rs:50-90); coordinates are
// in R'-form (x 2^(29 NZ)) and only congruent mod p.  Value bounds (multiples of p) are tracked in
// the comments; a Montgomery product of values a p and b p comes back below (a b / 128 + 1) p for
// the 255-bit fields (p / R' <= 2^-7), far less for the others.
//
// Invariant of an accumulator:  X < 8p, Y < 4p, ZZ < 2p, ZZZ < 2p, limbs < 2^29 + 8,
// or inf = true (the identity; the reference's `zero` flag, curve.rs:176-181).
#pragma once
#include "fz.cuh"
#ifdef __HIPCC__
#include "ec.cuh"
#endif

namespace plk {

// (PLK_FT: the shader-clock stamps of tuning builds, fp.cuh)
template <class FP> struct XyzzZ {
    Fz<FP> x, y, zz, zzz;
    bool inf;
};

// -y for a canonical y (< p): 2p - y, in (p, 2p]
template <class FP> PLK_DI Fz<FP> fz_neg_canonical(const Fz<FP>& y) { return fz_sub<FP, 1>(fz_zero<FP>(), y); }

// 2 * (x, y), affine operand with x, y < 2p  (y = 0 mod p gives the identity)
// Operands by value (they travel in VGPRs) and result by value: nothing of the hot loop has its
// address taken, so the accumulator and the current point stay in registers.
template <class FP> PLK_DNI XyzzZ<FP> xyzzz_mdbl(Fz<FP> x, Fz<FP> y) {
    XyzzZ<FP> r;
    Fz<FP> u = fz_dbl<FP>(y);                                // < 4
    Fz<FP> v = fz_sqr<FP>(u);                                // < 2
    Fz<FP> w = fz_mul<FP>(u, v);                             // < 2
    Fz<FP> s = fz_mul<FP>(x, v);                             // < 2
    Fz<FP> xx = fz_sqr<FP>(x);                               // < 2
    Fz<FP> m = fz_add<FP>(fz_dbl<FP>(xx), xx);               // < 6
    Fz<FP> mm = fz_sqr<FP>(m);                               // < 2
    r.x = fz_sub<FP, 2>(mm, fz_dbl<FP>(s));                  // < 2 + 4 = 6   (2s < 4 <= 4p - margin)
    Fz<FP> t = fz_sub<FP, 3>(s, r.x);                        // < 2 + 8 = 10
    r.y = fz_mul_sub2<FP, 1>(m, t, w, y);                    // m t - w y through one reduction (fz.cuh): (6 * 10 + 2 * 2) / 128 + 1 < 1.6
    r.zz = v;
    r.zzz = w;
    r.inf = fz_is_zero_mod_p<FP>(v);
    return r;
}

// ---- the mixed addition of the bucket accumulation ------------------------------------------------------------------------
// Round 4 put it on an instruction diet (tools/lab/madd_lab.hip; SQ_INSTS_VALU per addition 2446 -> ~2290):
//   * the products lost their "+ q_k" additions (fz_mul / fz_sqr, fz.cuh);
//   * Y may stay UNCARRIED between additions ("lazy Y": limbs <= 3 * 2^29 - 3 instead of < 2^29 + 8).  Inside an addition Y only
//     meets a subtraction that borrows 2^31 per limb and a product with the exactly normalised PPP, both of which accept such
//     limbs; xyzzz_settle() moves the carries before the point leaves the loop (a store, a full addition);
//   * -y of a table entry is formed without a carry pass (limbs <= 2^30): it goes straight into the product with the exactly
//     normalised ZZZ;
//   * "P = 0 mod p" (equal or opposite points) is screened by ZZ3's lowest limb - for a non-zero ZZ3 < 2p it is 0 or p_0 = 1 with
//     probability 2^-28 - before the full comparison.
// (A branch-free form - the formulas run on an identity accumulator's stale coordinates and the result is selected - was built
// and measured in the ISA: the register copies at the merge fall from 90 to 38 per iteration, but the entry must stay live to the
// end: 198 registers, two waves per SIMD, or 55 scratch accesses in the loop when held to three.  The branch stays; the caller
// marks a new bucket with acc.inf = true instead of clearing 36 registers.)
// acc: X < 8p, ZZ < 2p, ZZZ < 2p with limbs < 2^29 + 8 (ZZ, ZZZ exactly normalised after the first addition: they are products);
// Y < 4p with limbs <= 3 * 2^29 - 3.  (x2, y2): x2 < 2p exactly normalised limbs; y2 < 2p, limbs <= 2^30.
template <class FP> PLK_DI void xyzzz_madd_lazy(XyzzZ<FP>& acc, const Fz<FP>& x2, const Fz<FP>& y2) {
    static_assert(3ull * (1ull << 29) <= FzLazyBound<FP>::MUL_LIMB_MAX, "a lazy Y must be a legal multiplicand of an exactly normalised operand");
    if (acc.inf) {
        acc.x = x2;
        acc.y = y2;
        acc.zz = fz_one_rprime<FP>();
        acc.zzz = acc.zz;
        acc.inf = false;
        return;
    }
    Fz<FP> u2 = fz_mul<FP>(x2, acc.zz);                      // < 2
    Fz<FP> s2 = fz_mul<FP>(y2, acc.zzz);                     // < 2      (y2 limbs <= 2^30 against an exactly normalised ZZZ... or < 2^29 + 8: fz_mul's general bound)
    Fz<FP> p = fz_sub<FP, 4>(u2, acc.x);                     // < 2 + 16 = 18      (X < 8 <= 16 - margin)
    Fz<FP> r = fz_sub_nc<FP, 2, 31>(s2, acc.y);              // < 2 + 4 = 6        (Y < 3.6; its limbs <= 3 * 2^29 - 3 <= 2^31 - 4)
    fz_carry<FP>(r);                                         // limbs <= 2^29 + 2^31 + 2^29 -> < 2^29 + 8
    Fz<FP> pp = fz_sqr<FP>(p);                               // < 18^2/128 + 1 < 3.6
    Fz<FP> ppp = fz_mul<FP>(p, pp);                          // < 18*3.6/128 + 1 < 1.6
    Fz<FP> q = fz_mul<FP>(acc.x, pp);                        // < 8*3.6/128 + 1 < 1.3
    Fz<FP> rr = fz_sqr<FP>(r);                               // < 36/128 + 1 < 1.3
    Fz<FP> zz3 = fz_mul<FP>(acc.zz, pp);                     // < 1.1
    if (zz3.l[0] <= 1u && fz_is_zero_mod_p<FP>(zz3)) {
        // p = 0 mod p: the operands share x.  Same point -> double it; opposite points -> identity.
        if (fz_is_zero_mod_p<FP>(rr)) {
            Fz<FP> yc = y2;
            fz_carry<FP>(yc);
            acc = xyzzz_mdbl<FP>(x2, yc);  // rare, out of line
        } else {
            acc.inf = true;
        }
        return;
    }
    // carries are moved only where a later step needs them (fz_add_nc / fz_sub_nc, fz.cuh): rr - ppp and 2q are consumed by
    // the subtraction that forms x3 (limbs <= 2^29 + 2^30 and <= 2^30 - 2), t only by the product r t (limbs <= 2^31, r carried)
    Fz<FP> x3 = fz_sub_nc<FP, 2, 30>(fz_sub_nc<FP, 1, 29>(rr, ppp), fz_add_nc<FP>(q, q));  // (1.3 + 2) + 4 < 7.3 < 8
    fz_carry<FP>(x3);                                                   // limbs <= 3 * 2^30 -> < 2^29 + 8
    Fz<FP> t;                                                           // < 1.3 + 8 = 9.3
    if constexpr (FzCfg<FP>::NZ <= 10) t = fz_sub_nc<FP, 3, 30>(q, x3);  // limbs <= 2^31 <= FzNcBound::MUL_LIMB_MAX
    else t = fz_sub<FP, 3>(q, x3);                                      // 14 limbs: the column sums have no room for it
    // Y3 = r t - Y1 PPP, both products exactly normalised: the difference keeps its carries (limbs <= 3 * 2^29 - 3)
    // round 5: through ONE reduction (fz_mul_add2): y1 enters as 4p - y1 (limbs <= 2^31 + 2^29, carried on 14 limbs)
    {
        Fz<FP> yn = fz_sub_nc<FP, 2, 31>(fz_zero<FP>(), acc.y);
        if constexpr (FzCfg<FP>::NZ <= 9) fz_carry<FP>(t);
        else fz_carry<FP>(yn);
        acc.y = fz_mul_add2<FP>(r, t, yn, ppp);                        // (6 * 9.3 + 8 * 1.6) / 128 + 1 < 1.6, exactly normalised
    }
    acc.x = x3;
    acc.zz = zz3;
    acc.zzz = fz_mul<FP>(acc.zzz, ppp);                                 // < 1.1
}
// the carries of a lazy Y moved: the point is inside the invariant every other routine expects (limbs < 2^29 + 8)
template <class FP> PLK_DI void xyzzz_settle(XyzzZ<FP>& acc) { fz_carry<FP>(acc.y); }

// acc += (x2, y2); x2 < p canonical, y2 < 2p (a negated canonical y is 2p - y); limbs < 2^29 + 8 in and out
template <class FP> PLK_DI void xyzzz_madd(XyzzZ<FP>& acc, const Fz<FP>& x2, const Fz<FP>& y2) {
    xyzzz_madd_lazy<FP>(acc, x2, y2);
    xyzzz_settle<FP>(acc);
}
// acc += +-(x, y) for a table entry in the interface form (R'-form words, canonical): conversion, conditional negation without a
// carry pass (2p - y: limbs <= 2^30 - 2), lazy addition.  The caller settles acc before it leaves the loop.
template <class FP> PLK_DI void xyzzz_madd_entry(XyzzZ<FP>& acc, const Fe<FP>& x, const Fe<FP>& y, bool negate) {
    const Fz<FP> xz = fz_from_fe<FP>(x), yp = fz_from_fe<FP>(y);
    const Fz<FP> yn = fz_sub_nc<FP, 1, 29>(fz_zero<FP>(), yp);
    Fz<FP> yz;
#pragma unroll
    for (int i = 0; i < FzCfg<FP>::NZ; ++i) yz.l[i] = negate ? yn.l[i] : yp.l[i];
    xyzzz_madd_lazy<FP>(acc, xz, yz);
}

// 2 * a, XYZZ operand within the accumulator invariant (EFD dbl-2008-s-1, a = 0)
template <class FP> PLK_DI XyzzZ<FP> xyzzz_dbl(const XyzzZ<FP>& a) {
    if (a.inf) return a;
    XyzzZ<FP> r;
    Fz<FP> u = fz_dbl<FP>(a.y);                              // < 8
    Fz<FP> v = fz_sqr<FP>(u);                                // < 1.5
    Fz<FP> w = fz_mul<FP>(u, v);                             // < 1.1
    Fz<FP> s = fz_mul<FP>(a.x, v);                           // < 1.1
    Fz<FP> xx = fz_sqr<FP>(a.x);                             // < 1.5
    Fz<FP> m = fz_add<FP>(fz_dbl<FP>(xx), xx);               // < 4.5
    Fz<FP> mm = fz_sqr<FP>(m);                               // < 1.2
    r.x = fz_sub<FP, 2>(mm, fz_dbl<FP>(s));                  // < 1.2 + 4 = 5.2
    Fz<FP> t = fz_sub<FP, 3>(s, r.x);                        // < 9.1
    r.y = fz_mul_sub2<FP, 1>(m, t, w, a.y);                  // m t - w y through one reduction: (4.5 * 9.1 + 2 * 4) / 128 + 1 < 1.4
    r.zz = fz_mul<FP>(v, a.zz);
    r.zzz = fz_mul<FP>(w, a.zzz);
    r.inf = fz_is_zero_mod_p<FP>(r.zz);                      // 2-torsion point
    return r;
}

// a + b, both XYZZ within the invariant (EFD add-2008-s + the exceptional cases of curve_adds.rs:5-48)
template <class FP> PLK_DI XyzzZ<FP> xyzzz_add(const XyzzZ<FP>& a, const XyzzZ<FP>& b) {
    if (a.inf) return b;
    if (b.inf) return a;
    Fz<FP> u1 = fz_mul<FP>(a.x, b.zz);                       // < 1.2
    Fz<FP> u2 = fz_mul<FP>(b.x, a.zz);
    Fz<FP> s1 = fz_mul<FP>(a.y, b.zzz);
    Fz<FP> s2 = fz_mul<FP>(b.y, a.zzz);
    Fz<FP> p = fz_sub<FP, 1>(u2, u1);                        // < 3.2
    Fz<FP> r = fz_sub<FP, 1>(s2, s1);                        // < 3.2
    Fz<FP> pp = fz_sqr<FP>(p);                               // < 1.1
    Fz<FP> ppp = fz_mul<FP>(p, pp);
    Fz<FP> q = fz_mul<FP>(u1, pp);
    Fz<FP> rr = fz_sqr<FP>(r);
    XyzzZ<FP> o;
    o.zz = fz_mul<FP>(fz_mul<FP>(a.zz, b.zz), pp);
    if (fz_is_zero_mod_p<FP>(o.zz)) {
        if (fz_is_zero_mod_p<FP>(rr)) return xyzzz_dbl<FP>(a);
        o = a;
        o.inf = true;
        return o;
    }
    o.x = fz_sub<FP, 2>(fz_sub<FP, 1>(rr, ppp), fz_dbl<FP>(q));        // < 7.3
    Fz<FP> t = fz_sub<FP, 3>(q, o.x);                                  // < 9.3
    o.y = fz_mul_sub2<FP, 1>(r, t, s1, ppp);                           // r t - s1 ppp through one reduction: < 1.3
    o.zzz = fz_mul<FP>(fz_mul<FP>(a.zzz, b.zzz), ppp);
    o.inf = false;
    return o;
}

template <class FP> PLK_DI XyzzZ<FP> xyzzz_identity() {
    XyzzZ<FP> r;
    r.x = r.y = r.zz = r.zzz = fz_zero<FP>();
    r.inf = true;
    return r;
}

#ifdef __HIPCC__
// Exchange format between the MSM kernels: X, Y, ZZ, ZZZ in R'-form, canonical, packed in 32-bit
// words (the layout of Xyzz<FP>); the identity is ZZ = 0.
template <class FP> PLK_DI void xyzzz_store_packed(uint4* dst, const XyzzZ<FP>& a) {
    Xyzz<FP> o = xyzz_identity<FP>();
    if (!a.inf) {
        const Fz<FP> one = fz_one_rprime<FP>();  // the extra product brings every coordinate below 2p
        o.x = fz_to_fe_canonical<FP>(fz_mul<FP>(a.x, one));
        o.y = fz_to_fe_canonical<FP>(fz_mul<FP>(a.y, one));
        o.zz = fz_to_fe_canonical<FP>(fz_mul<FP>(a.zz, one));
        o.zzz = fz_to_fe_canonical<FP>(fz_mul<FP>(a.zzz, one));
    }
    xyzz_store<FP>(dst, o);
}
template <class FP> PLK_DI XyzzZ<FP> xyzzz_load_packed(const uint4* src) {
    const Xyzz<FP> p = xyzz_load<FP>(src);
    XyzzZ<FP> r;
    r.x = fz_from_fe<FP>(p.x);
    r.y = fz_from_fe<FP>(p.y);
    r.zz = fz_from_fe<FP>(p.zz);
    r.zzz = fz_from_fe<FP>(p.zzz);
    r.inf = fe_is_zero<FP>(p.zz);
    return r;
}
// the same through volatile loads: a point another workgroup of the running kernel has just written (after a device-scope fence)
template <class FP> PLK_DI XyzzZ<FP> xyzzz_load_packed_volatile(const uint4* src) {
    constexpr int NW = 4 * FP::NL;
    const volatile uint32_t* w = reinterpret_cast<const volatile uint32_t*>(src);
    Xyzz<FP> p;
#pragma unroll
    for (int i = 0; i < FP::NL; ++i) {
        p.x.v[i] = w[i];
        p.y.v[i] = w[FP::NL + i];
        p.zz.v[i] = w[2 * FP::NL + i];
        p.zzz.v[i] = w[3 * FP::NL + i];
    }
    (void)NW;
    XyzzZ<FP> r;
    r.x = fz_from_fe<FP>(p.x);
    r.y = fz_from_fe<FP>(p.y);
    r.zz = fz_from_fe<FP>(p.zz);
    r.zzz = fz_from_fe<FP>(p.zzz);
    r.inf = fe_is_zero<FP>(p.zz);
    return r;
}
// back to the reference's form, then ProjectivePoint::to_affine (curve.rs:206-214)
// ONE_LANE: the caller is a single lane (the end of an MSM) - the inversion takes its data-dependent form (fp.cuh)
template <class FP, bool ONE_LANE = false> PLK_DI void emit_affine(const XyzzZ<FP>& acc, uint4* out_xy, uint8_t* out_zero) {
    constexpr int W = FP::NL / 4;
    if constexpr (ONE_LANE) {
        // Round 5: the normalisation stays on the working limbs.  Only ZZZ goes back to words (the inversion works on the integer); its
        // plain inverse returns through ONE product by a constant that also carries the Montgomery fix-up, 1 / Z = ZZ / ZZZ, x = X / ZZ,
        // y = Y / ZZZ are products on 29-bit limbs, and the two coordinates alone are taken back to the reference's form: 8 products +
        // 3 conversions where the word-form route (below) made 4 + 4 and 5 products on 32-bit words with their carry chains.
        // Same field values, unique representatives: the same bits (profiles/r05_final_kernel_trace.txt: a third of k_msm_final).
        const Fz<FP> back = fz_const_rprime_to_r<FP>();
        Fe<FP> zr = fe_zero<FP>();
        if (!acc.inf) zr = fz_to_fe_canonical<FP>(fz_mul<FP>(acc.zzz, back));  // ZZZ x 2^(32 NL), canonical
        if (acc.inf || fe_is_zero<FP>(zr)) {
            fe_store<FP>(out_xy, fe_zero<FP>());
            fe_store<FP>(out_xy + W, fe_zero<FP>());
            *out_zero = 1;
            return;
        }
        PLK_FT(10);
        const Fe<FP> raw = fe_inv_safegcd_one_lane_raw<FP>(zr);
        PLK_FT(11);
        const Fz<FP> i3 = fz_mul<FP>(fz_from_fe<FP>(raw), fz_const_raw_inverse_to_rprime<FP>());  // 1 / ZZZ
        const Fz<FP> iz = fz_mul<FP>(acc.zz, i3);                                                                                    // 1 / Z
        const Fz<FP> xa = fz_mul<FP>(acc.x, fz_sqr<FP>(iz)), ya = fz_mul<FP>(acc.y, i3);
        fe_store<FP>(out_xy, fz_to_fe_canonical<FP>(fz_mul<FP>(xa, back)));
        fe_store<FP>(out_xy + W, fz_to_fe_canonical<FP>(fz_mul<FP>(ya, back)));
        *out_zero = 0;
        PLK_FT(12);
        return;
    }
    Xyzz<FP> r = xyzz_identity<FP>();
    if (!acc.inf) {
        const Fz<FP> back = fz_const_rprime_to_r<FP>();
        r.x = fz_to_fe_canonical<FP>(fz_mul<FP>(acc.x, back));
        r.y = fz_to_fe_canonical<FP>(fz_mul<FP>(acc.y, back));
        r.zz = fz_to_fe_canonical<FP>(fz_mul<FP>(acc.zz, back));
        r.zzz = fz_to_fe_canonical<FP>(fz_mul<FP>(acc.zzz, back));
    }
    Fe<FP> x, y;
    bool ident = xyzz_to_affine<FP, ONE_LANE>(r, x, y);
    fe_store<FP>(out_xy, x);
    fe_store<FP>(out_xy + W, y);
    *out_zero = ident ? 1 : 0;
}

// The reference's own return type: msm_execute[_parallel] hands back a ProjectivePoint (curve_msm.rs:102-157 ends in `y`, no
// normalisation; curve.rs:175-181: x / z, y / z and a `zero` flag).  From XYZZ (x = X / ZZ, y = Y / ZZZ): (X ZZZ : Y ZZ : ZZ ZZZ) - six
// products where emit_affine makes an inversion (a third of k_msm_final).  R-form, canonical limbs, x | y | z.
template <class FP> PLK_DI void emit_projective(const XyzzZ<FP>& acc, uint4* out_xyz, uint8_t* out_zero) {
    constexpr int W = FP::NL / 4;
    const Fz<FP> back = fz_const_rprime_to_r<FP>();
    Fe<FP> z = fe_zero<FP>();
    if (!acc.inf) z = fz_to_fe_canonical<FP>(fz_mul<FP>(fz_mul<FP>(acc.zz, acc.zzz), back));
    if (acc.inf || fe_is_zero<FP>(z)) {  // ProjectivePoint::ZERO
        fe_store<FP>(out_xyz, fe_zero<FP>());
        fe_store<FP>(out_xyz + W, fe_zero<FP>());
        fe_store<FP>(out_xyz + 2 * W, fe_zero<FP>());
        *out_zero = 1;
        return;
    }
    fe_store<FP>(out_xyz, fz_to_fe_canonical<FP>(fz_mul<FP>(fz_mul<FP>(acc.x, acc.zzz), back)));
    fe_store<FP>(out_xyz + W, fz_to_fe_canonical<FP>(fz_mul<FP>(fz_mul<FP>(acc.y, acc.zz), back)));
    fe_store<FP>(out_xyz + 2 * W, z);
    *out_zero = 0;
}

// the other lane's point (xor butterfly inside a wave)
template <class FP> PLK_DI XyzzZ<FP> xyzzz_shfl_xor(const XyzzZ<FP>& a, int mask) {
    XyzzZ<FP> r;
#pragma unroll
    for (int i = 0; i < FzCfg<FP>::NZ; ++i) {
        r.x.l[i] = __shfl_xor(a.x.l[i], mask);
        r.y.l[i] = __shfl_xor(a.y.l[i], mask);
        r.zz.l[i] = __shfl_xor(a.zz.l[i], mask);
        r.zzz.l[i] = __shfl_xor(a.zzz.l[i], mask);
    }
    r.inf = __shfl_xor((int)a.inf, mask) != 0;
    return r;
}
#endif

}  // namespace plk
