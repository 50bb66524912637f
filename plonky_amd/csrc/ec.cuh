// ec.cuh -- short-Weierstrass group law on the device, a = 0 curves, XYZZ coordinates.
//
// Replaces, on the device, the reference's point arithmetic:
//   src/curve/curve.rs:74-137,176-278   AffinePoint / ProjectivePoint (homogeneous X:Y:Z + zero flag)
//   src/curve/curve_adds.rs:5-128       P+P, P+A, A+A with the zero / doubling / inverse branches
// The reference accumulates in homogeneous projective coordinates; any coordinate system that
// implements the same group law yields the same group element, and parity is defined on
// ProjectivePoint::to_affine() (curve.rs:206-214), which is unique.  XYZZ (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2, identity <=> ZZ = 0) is used because the mixed addition that dominates bucket
// accumulation costs 8M + 2S and needs no inversion (EFD madd-2008-s, add-2008-s, dbl-2008-s-1).
// Every in-scope curve has A = 0 (tweedledee_curve.rs:11, tweedledum_curve.rs:11,
// bls12_377_curve.rs:14); B never appears in the formulas.
// All exceptional cases the reference branches on (curve_adds.rs:12-33,57-75,99-113) are
// handled: identity operands, P + P (falls to doubling), P + (-P) (identity); 2-torsion points
// (y = 0, possible on BLS12-377 G1 whose cofactor is even) double to the identity.
#pragma once
#include "fp.cuh"
#include "glv_params.cuh"

namespace plk {

struct TweedledeeCurve {
    static constexpr int CURVE_ID = 0;
    using FP = TweedledeeBaseParams;   // coordinates  (tweedledee_curve.rs:8)
    using SP = TweedledumBaseParams;   // scalars      (tweedledee_curve.rs:9)
    using Glv = TweedledeeGlv;         // HaloCurve    (tweedledee_curve.rs:21-37)
};
struct TweedledumCurve {
    static constexpr int CURVE_ID = 1;
    using FP = TweedledumBaseParams;
    using SP = TweedledeeBaseParams;
    using Glv = TweedledumGlv;
};
struct Bls12377Curve {
    static constexpr int CURVE_ID = 2;
    using FP = Bls12377BaseParams;     // bls12_377_curve.rs:11
    using SP = Bls12377ScalarParams;   // bls12_377_curve.rs:12
    using Glv = NoGlv;                 // not a HaloCurve in the reference, and G1 has a cofactor: inputs need not lie in the r-subgroup
};

struct PallasCurve {
    static constexpr int CURVE_ID = 3;
    using FP = PallasBaseParams;       // pallas_curve.rs:8
    using SP = VestaBaseParams;        // pallas_curve.rs:9
    using Glv = PallasGlv;
};
struct VestaCurve {
    static constexpr int CURVE_ID = 4;
    using FP = VestaBaseParams;
    using SP = PallasBaseParams;
    using Glv = VestaGlv;
};

template <class FP> struct Xyzz {
    Fe<FP> x, y, zz, zzz;
};

template <class FP> PLK_DI Xyzz<FP> xyzz_identity() {
    Xyzz<FP> r;
    r.x = fe_zero<FP>();
    r.y = fe_zero<FP>();
    r.zz = fe_zero<FP>();
    r.zzz = fe_zero<FP>();
    return r;
}
template <class FP> PLK_DI bool xyzz_is_identity(const Xyzz<FP>& p) { return fe_is_zero<FP>(p.zz); }

template <class FP> PLK_DI Xyzz<FP> xyzz_from_affine(const Fe<FP>& x, const Fe<FP>& y) {
    Xyzz<FP> r;
    r.x = x;
    r.y = y;
    r.zz = fe_one<FP>();
    r.zzz = fe_one<FP>();
    return r;
}

// 2 * (x, y) for an affine point (EFD mdbl-2008-s-1, a = 0).  y = 0 gives the identity.
template <class FP> __device__ __noinline__ Xyzz<FP> xyzz_mdbl(const Fe<FP>& x, const Fe<FP>& y) {
    Fe<FP> u = fe_dbl<FP>(y);
    Fe<FP> v = fe_sqr<FP>(u);
    Fe<FP> w = fe_mul<FP>(u, v);
    Fe<FP> s = fe_mul<FP>(x, v);
    Fe<FP> xx = fe_sqr<FP>(x);
    Fe<FP> m = fe_add<FP>(fe_dbl<FP>(xx), xx);
    Xyzz<FP> r;
    r.x = fe_sub<FP>(fe_sqr<FP>(m), fe_dbl<FP>(s));
    r.y = fe_sub<FP>(fe_mul<FP>(m, fe_sub<FP>(s, r.x)), fe_mul<FP>(w, y));
    r.zz = v;
    r.zzz = w;
    if (fe_is_zero<FP>(v)) r = xyzz_identity<FP>();
    return r;
}

// 2 * p (EFD dbl-2008-s-1, a = 0)
template <class FP> __device__ __noinline__ Xyzz<FP> xyzz_dbl(const Xyzz<FP>& p) {
    if (xyzz_is_identity<FP>(p)) return p;
    Fe<FP> u = fe_dbl<FP>(p.y);
    Fe<FP> v = fe_sqr<FP>(u);
    Fe<FP> w = fe_mul<FP>(u, v);
    Fe<FP> s = fe_mul<FP>(p.x, v);
    Fe<FP> xx = fe_sqr<FP>(p.x);
    Fe<FP> m = fe_add<FP>(fe_dbl<FP>(xx), xx);
    Xyzz<FP> r;
    r.x = fe_sub<FP>(fe_sqr<FP>(m), fe_dbl<FP>(s));
    r.y = fe_sub<FP>(fe_mul<FP>(m, fe_sub<FP>(s, r.x)), fe_mul<FP>(w, p.y));
    r.zz = fe_mul<FP>(v, p.zz);
    r.zzz = fe_mul<FP>(w, p.zzz);
    if (fe_is_zero<FP>(r.zz)) r = xyzz_identity<FP>();
    return r;
}

// acc += (x2, y2), the affine operand is never the identity (EFD madd-2008-s + exceptional cases;
// the P+A branches of curve_adds.rs:50-90)
template <class FP> PLK_DI void xyzz_madd(Xyzz<FP>& acc, const Fe<FP>& x2, const Fe<FP>& y2) {
    if (xyzz_is_identity<FP>(acc)) {
        acc = xyzz_from_affine<FP>(x2, y2);
        return;
    }
    Fe<FP> u2 = fe_mul<FP>(x2, acc.zz);
    Fe<FP> s2 = fe_mul<FP>(y2, acc.zzz);
    Fe<FP> p = fe_sub<FP>(u2, acc.x);
    Fe<FP> r = fe_sub<FP>(s2, acc.y);
    if (fe_is_zero<FP>(p)) {
        // same x: either the same point (double) or opposite points (identity)
        if (fe_is_zero<FP>(r)) acc = xyzz_mdbl<FP>(x2, y2);
        else acc = xyzz_identity<FP>();
        return;
    }
    Fe<FP> pp = fe_sqr<FP>(p);
    Fe<FP> ppp = fe_mul<FP>(p, pp);
    Fe<FP> q = fe_mul<FP>(acc.x, pp);
    Fe<FP> x3 = fe_sub<FP>(fe_sub<FP>(fe_sqr<FP>(r), ppp), fe_dbl<FP>(q));
    Fe<FP> y3 = fe_sub<FP>(fe_mul<FP>(r, fe_sub<FP>(q, x3)), fe_mul<FP>(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul<FP>(acc.zz, pp);
    acc.zzz = fe_mul<FP>(acc.zzz, ppp);
}

// a + b, both XYZZ (EFD add-2008-s + exceptional cases; curve_adds.rs:5-48)
template <class FP> __device__ __noinline__ Xyzz<FP> xyzz_add(const Xyzz<FP>& a, const Xyzz<FP>& b) {
    if (xyzz_is_identity<FP>(a)) return b;
    if (xyzz_is_identity<FP>(b)) return a;
    Fe<FP> u1 = fe_mul<FP>(a.x, b.zz);
    Fe<FP> u2 = fe_mul<FP>(b.x, a.zz);
    Fe<FP> s1 = fe_mul<FP>(a.y, b.zzz);
    Fe<FP> s2 = fe_mul<FP>(b.y, a.zzz);
    Fe<FP> p = fe_sub<FP>(u2, u1);
    Fe<FP> r = fe_sub<FP>(s2, s1);
    if (fe_is_zero<FP>(p)) {
        if (fe_is_zero<FP>(r)) return xyzz_dbl<FP>(a);
        return xyzz_identity<FP>();
    }
    Fe<FP> pp = fe_sqr<FP>(p);
    Fe<FP> ppp = fe_mul<FP>(p, pp);
    Fe<FP> q = fe_mul<FP>(u1, pp);
    Xyzz<FP> o;
    o.x = fe_sub<FP>(fe_sub<FP>(fe_sqr<FP>(r), ppp), fe_dbl<FP>(q));
    o.y = fe_sub<FP>(fe_mul<FP>(r, fe_sub<FP>(q, o.x)), fe_mul<FP>(s1, ppp));
    o.zz = fe_mul<FP>(fe_mul<FP>(a.zz, b.zz), pp);
    o.zzz = fe_mul<FP>(fe_mul<FP>(a.zzz, b.zzz), ppp);
    return o;
}

// ProjectivePoint::to_affine (curve.rs:206-214): returns true for the identity.
// x = X/ZZ, y = Y/ZZZ with one inversion: 1/Z = ZZ/ZZZ, 1/ZZ = (1/Z)^2.
// The inversion is the branch-free division-step one (fp.cuh): a tenth of the instructions of the reference's
// bit-by-bit Euclid (bigint_inverse.rs:6-55) or of a Fermat chain, for one lane and for a full wave alike.
template <class FP, bool SINGLE_LANE = false> PLK_DI bool xyzz_to_affine(const Xyzz<FP>& p, Fe<FP>& x, Fe<FP>& y) {
    if (xyzz_is_identity<FP>(p)) {
        x = fe_zero<FP>();
        y = fe_zero<FP>();
        return true;
    }
    Fe<FP> i3;
    if constexpr (SINGLE_LANE) i3 = fe_inv_safegcd_one_lane<FP>(p.zzz);  // ONE active lane in the wave: the data-dependent (variable-time) division steps, fp.cuh MODE 1; the scalar-unit forms (MODE 2 / 3) are build options that measured slower
    else i3 = fe_inv_safegcd<FP>(p.zzz);
    Fe<FP> iz = fe_mul<FP>(p.zz, i3);
    Fe<FP> izz = fe_sqr<FP>(iz);
    x = fe_mul<FP>(p.x, izz);
    y = fe_mul<FP>(p.y, i3);
    return false;
}

// ---- memory formats ----
// Affine table entry: x limbs then y limbs, 2 * NL u32.  The identity is flagged in the (always
// clear) top bit of y's top limb - internal to the device tables, never crosses the C ABI.
constexpr uint32_t AFF_IDENTITY_BIT = 0x80000000u;

template <class FP> PLK_DI void affine_store(uint4* dst, const Fe<FP>& x, const Fe<FP>& y, bool identity) {
    constexpr int W = FP::NL / 4;
    Fe<FP> yy = y;
    if (identity) yy.v[FP::NL - 1] |= AFF_IDENTITY_BIT;
    fe_store<FP>(dst, x);
    fe_store<FP>(dst + W, yy);
}
template <class FP> PLK_DI bool affine_load(const uint4* src, Fe<FP>& x, Fe<FP>& y) {
    constexpr int W = FP::NL / 4;
    x = fe_load<FP>(src);
    y = fe_load<FP>(src + W);
    bool identity = (y.v[FP::NL - 1] & AFF_IDENTITY_BIT) != 0;
    y.v[FP::NL - 1] &= ~AFF_IDENTITY_BIT;
    return identity;
}
template <class FP> PLK_DI void xyzz_store(uint4* dst, const Xyzz<FP>& p) {
    constexpr int W = FP::NL / 4;
    fe_store<FP>(dst, p.x);
    fe_store<FP>(dst + W, p.y);
    fe_store<FP>(dst + 2 * W, p.zz);
    fe_store<FP>(dst + 3 * W, p.zzz);
}
template <class FP> PLK_DI Xyzz<FP> xyzz_load(const uint4* src) {
    constexpr int W = FP::NL / 4;
    Xyzz<FP> p;
    p.x = fe_load<FP>(src);
    p.y = fe_load<FP>(src + W);
    p.zz = fe_load<FP>(src + 2 * W);
    p.zzz = fe_load<FP>(src + 3 * W);
    return p;
}

}  // namespace plk
