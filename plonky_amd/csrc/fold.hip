// fold.hip -- the generator fold of an inner-product-argument round:
//   G'_i = [a] G_lo_i + [b] G_hi_i   for every i, the same two scalars for all pairs
// (halo.rs:119-123: `msm_parallel(&[u_j_inv, u_j], &[g_lo_i, g_hi_i], 4)` per pair, n/2 pairs per round).
//
// One lane per pair, lazy 29-bit-limb arithmetic (ecz.cuh).  The two scalars are recoded ONCE on the host
// into their joint sparse form (Solinas): digit pairs in {-1,0,1}^2 with on average half of the columns
// empty; every lane then walks the same columns (no divergence: the branch on the digit pair is uniform),
// doubling its accumulator and adding one of +-P, +-Q, +-(P+Q), +-(P-Q).  P+Q and P-Q are affine and share
// one inversion (same denominator x_Q - x_P).  ~255 doublings + ~128 mixed additions + 2 inversions per pair.
// The result is the unique affine point, as everywhere on this boundary.
//
// On the four prime-order curves the two scalars are first split along the endomorphism (glv.cuh): a = a1 + a2 lambda,
// b = b1 + b2 lambda with half-length parts, G' = [a1] P + [a2] phi(P) + [b1] Q + [b2] phi(Q): ~130 doublings and two joint
// sparse forms, (a1, a2) over (P, phi P) and (b1, b2) over (Q, phi Q), ~130 additions - 30 % fewer field multiplications.
// phi(P) = (beta x, y) and P + phi(P) = -phi^2(P) = (beta^2 x, -y) cost one multiplication when they are used; only
// P - phi(P) and Q - phi(Q) are real additions (one shared inversion).
#include <cstdlib>
#include <vector>

#include "common.h"
#include "ec.cuh"
#include "ecz.cuh"
#include "glv.cuh"
#include "tables.cuh"

namespace plk {

constexpr int FOLD_MAX_COLS = 264;

struct FoldDigits {
    int plus_lo;                    // 1: out_i = lo_i + [b] hi_i (lo is added once at the end, the scalar a is not used)
    int cols;                       // number of columns, most significant first
    int8_t d[FOLD_MAX_COLS];        // (da + 1) * 3 + (db + 1): 4 = empty column
    int8_t e[FOLD_MAX_COLS];        // GLV form: d = the pair (a1, a2) over (P, phi P), e = the pair (b1, b2) over (Q, phi Q)
};

template <class FP> struct AffZ {  // affine point in R'-form, canonical; ident = the identity
    Fz<FP> x, y;
    bool ident;
};

// R-form affine pair -> P + Q and P - Q (affine, R'-form) with one inversion
template <class FP> PLK_DI void sum_and_diff(const Fe<FP>& px, const Fe<FP>& py, bool pi, const Fe<FP>& qx, const Fe<FP>& qy, bool qi, AffZ<FP>& P,
                                             AffZ<FP>& Q, AffZ<FP>& S, AffZ<FP>& D) {
    auto rp = [](const Fe<FP>& v) { return fz_from_fe<FP>(to_rprime<FP>(v)); };
    P.x = rp(px); P.y = rp(py); P.ident = pi;
    Q.x = rp(qx); Q.y = rp(qy); Q.ident = qi;
    if (pi || qi) {
        // P + Q = the other one (or nothing), P - Q = P or -Q
        S = pi ? Q : P;
        D = qi ? P : Q;
        if (!qi) D.y = fz_from_fe<FP>(to_rprime<FP>(fe_neg<FP>(qy)));
        S.ident = pi && qi;
        D.ident = pi && qi;
        return;
    }
    const Fe<FP> dx = fe_sub<FP>(qx, px);
    if (fe_is_zero<FP>(dx)) {
        // same x: Q = P or Q = -P.  The non-trivial one of P + Q / P - Q is 2P (tangent), the other the identity.
        const bool same = fe_is_zero<FP>(fe_sub<FP>(qy, py));
        AffZ<FP> two;
        if (fe_is_zero<FP>(py)) {
            two = P;
            two.ident = true;  // 2-torsion
        } else {
            // lambda = 3 x^2 / 2 y  (a = 0, curve.rs:112-135)
            const Fe<FP> xx = fe_sqr<FP>(px);
            const Fe<FP> lam = fe_mul<FP>(fe_add<FP>(fe_dbl<FP>(xx), xx), fe_inv_safegcd<FP>(fe_dbl<FP>(py)));
            const Fe<FP> x3 = fe_sub<FP>(fe_sqr<FP>(lam), fe_dbl<FP>(px));
            const Fe<FP> y3 = fe_sub<FP>(fe_mul<FP>(lam, fe_sub<FP>(px, x3)), py);
            two.x = rp(x3); two.y = rp(y3); two.ident = false;
        }
        AffZ<FP> none = P;
        none.ident = true;
        S = same ? two : none;
        D = same ? none : two;
        return;
    }
    const Fe<FP> inv = fe_inv_safegcd<FP>(dx);
    const Fe<FP> lam_s = fe_mul<FP>(fe_sub<FP>(qy, py), inv);                  // (yQ - yP) / (xQ - xP)
    const Fe<FP> lam_d = fe_mul<FP>(fe_neg<FP>(fe_add<FP>(qy, py)), inv);      // (-yQ - yP) / (xQ - xP)
    const Fe<FP> xs = fe_sub<FP>(fe_sub<FP>(fe_sqr<FP>(lam_s), px), qx);
    const Fe<FP> ys = fe_sub<FP>(fe_mul<FP>(lam_s, fe_sub<FP>(px, xs)), py);
    const Fe<FP> xd = fe_sub<FP>(fe_sub<FP>(fe_sqr<FP>(lam_d), px), qx);
    const Fe<FP> yd = fe_sub<FP>(fe_mul<FP>(lam_d, fe_sub<FP>(px, xd)), py);
    S.x = rp(xs); S.y = rp(ys); S.ident = false;
    D.x = rp(xd); D.y = rp(yd); D.ident = false;
}

template <class C>
__global__ void __launch_bounds__(128) k_fold_pairs(const uint4* __restrict__ lo, const uint8_t* __restrict__ lo_zero, const uint4* __restrict__ hi,
                                                    const uint8_t* __restrict__ hi_zero, size_t m, const FoldDigits* __restrict__ dgp,
                                                    uint4* __restrict__ out_xy, uint8_t* __restrict__ out_zero) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const Fe<FP> px = fe_load<FP>(lo + i * 2 * W), py = fe_load<FP>(lo + i * 2 * W + W);
    const Fe<FP> qx = fe_load<FP>(hi + i * 2 * W), qy = fe_load<FP>(hi + i * 2 * W + W);
    AffZ<FP> P, Q, S, D;
    sum_and_diff<FP>(px, py, lo_zero ? lo_zero[i] != 0 : false, qx, qy, hi_zero ? hi_zero[i] != 0 : false, P, Q, S, D);
    XyzzZ<FP> acc = xyzzz_identity<FP>();
    const int cols = dgp->cols;
    for (int k = 0; k < cols; ++k) {
        acc = xyzzz_dbl<FP>(acc);
        const int code = dgp->d[k];  // uniform over the grid
        if (code == 4) continue;
        const int da = code / 3 - 1, db = code % 3 - 1;
        // +-P, +-Q, +-(P + Q) when the signs agree, +-(P - Q) when they differ; the table holds the version whose
        // first non-zero digit is positive.  The digit pair is uniform, so these are scalar selects: the four table
        // points stay in registers (picking one through a reference would push them to scratch memory).
        const bool use_q = da == 0, use_s = da != 0 && da == db, use_d = da != 0 && db != 0 && da != db;
        const bool neg = da != 0 ? da < 0 : db < 0;
        Fz<FP> tx = P.x, ty = P.y;
        bool tid = P.ident;
        if (use_q) { tx = Q.x; ty = Q.y; tid = Q.ident; }
        if (use_s) { tx = S.x; ty = S.y; tid = S.ident; }
        if (use_d) { tx = D.x; ty = D.y; tid = D.ident; }
        if (tid) continue;
        if (neg) ty = fz_neg_canonical<FP>(ty);
        xyzzz_madd<FP>(acc, tx, ty);
    }
    if (dgp->plus_lo && !P.ident) xyzzz_madd<FP>(acc, P.x, P.y);
    emit_affine<FP>(acc, out_xy + i * 2 * W, out_zero + i);
}

// ---- the same fold along the endomorphism -------------------------------------------------------------------------------
// P - phi(P) for an affine P = (x, y), x != 0 (a prime-order curve has no point with x = 0: those have order 3):
// slope (-y - y) / (beta x - x); `inv` = 1 / ((beta - 1) x) in R-form
template <class FP> PLK_DI void diff_with_phi(const Fe<FP>& x, const Fe<FP>& y, const Fe<FP>& beta_x, const Fe<FP>& inv, AffZ<FP>& D) {
    const Fe<FP> lam = fe_mul<FP>(fe_neg<FP>(fe_dbl<FP>(y)), inv);
    const Fe<FP> x3 = fe_sub<FP>(fe_sub<FP>(fe_sqr<FP>(lam), x), beta_x);
    const Fe<FP> y3 = fe_sub<FP>(fe_mul<FP>(lam, fe_sub<FP>(x, x3)), y);
    D.x = fz_from_fe<FP>(to_rprime<FP>(x3));
    D.y = fz_from_fe<FP>(to_rprime<FP>(y3));
    D.ident = false;
}
// one digit pair (u0, u1) of the joint sparse form over (T, phi T): the point to add and whether it is negated.
//   (1, 0) T   (0, 1) phi T = (beta x, y)   (1, 1) T + phi T = (beta^2 x, -y)   (1, -1) T - phi T = D
template <class FP>
PLK_DI void glv_fold_add(XyzzZ<FP>& acc, int code, const AffZ<FP>& T, const AffZ<FP>& D, const Fz<FP>& beta, const Fz<FP>& beta2) {
    if (code == 4 || T.ident) return;  // the digit is uniform over the grid, the identity flag is not: the addition below is masked per lane
    const int u0 = code / 3 - 1, u1 = code % 3 - 1;
    bool neg = u0 != 0 ? u0 < 0 : u1 < 0;
    Fz<FP> tx = T.x, ty = T.y;
    if (u0 == 0) {
        tx = fz_mul<FP>(T.x, beta);
    } else if (u1 != 0 && u0 == u1) {
        tx = fz_mul<FP>(T.x, beta2);
        neg = !neg;
    } else if (u1 != 0) {
        tx = D.x;
        ty = D.y;
    }
    if (neg) ty = fz_neg_canonical<FP>(ty);
    xyzzz_madd<FP>(acc, tx, ty);
}
template <class C>
__global__ void __launch_bounds__(128) k_fold_pairs_glv(const uint4* __restrict__ lo, const uint8_t* __restrict__ lo_zero, const uint4* __restrict__ hi,
                                                        const uint8_t* __restrict__ hi_zero, size_t m, const FoldDigits* __restrict__ dgp,
                                                        uint4* __restrict__ out_xy, uint8_t* __restrict__ out_zero) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    if constexpr (C::Glv::ENABLED) {
        const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= m) return;
        const Fe<FP> px = fe_load<FP>(lo + i * 2 * W), py = fe_load<FP>(lo + i * 2 * W + W);
        const Fe<FP> qx = fe_load<FP>(hi + i * 2 * W), qy = fe_load<FP>(hi + i * 2 * W + W);
        Fe<FP> bc;
#pragma unroll
        for (int k = 0; k < FP::NL; ++k) bc.v[k] = C::Glv::BETA[k];
        const Fe<FP> beta_r = fe_from_canonical<FP>(bc);
        auto rp = [](const Fe<FP>& v) { return fz_from_fe<FP>(to_rprime<FP>(v)); };
        AffZ<FP> P, Q, DP, DQ;
        P.x = rp(px); P.y = rp(py); P.ident = lo_zero ? lo_zero[i] != 0 : false;
        Q.x = rp(qx); Q.y = rp(qy); Q.ident = hi_zero ? hi_zero[i] != 0 : false;
        DP = P;
        DQ = Q;
        {
            // 1 / ((beta - 1) x_P) and 1 / ((beta - 1) x_Q) from one inversion; an identity operand lends the value 1
            const Fe<FP> bm1 = fe_sub<FP>(beta_r, fe_one<FP>());
            const Fe<FP> dp = P.ident ? fe_one<FP>() : fe_mul<FP>(bm1, px), dq = Q.ident ? fe_one<FP>() : fe_mul<FP>(bm1, qx);
            const Fe<FP> inv = fe_inv_safegcd<FP>(fe_mul<FP>(dp, dq));
            if (!P.ident) diff_with_phi<FP>(px, py, fe_mul<FP>(beta_r, px), fe_mul<FP>(inv, dq), DP);
            if (!Q.ident) diff_with_phi<FP>(qx, qy, fe_mul<FP>(beta_r, qx), fe_mul<FP>(inv, dp), DQ);
        }
        const Fz<FP> beta = rp(beta_r), beta2 = rp(fe_sqr<FP>(beta_r));
        XyzzZ<FP> acc = xyzzz_identity<FP>();
        const int cols = dgp->cols;
        for (int k = 0; k < cols; ++k) {
            acc = xyzzz_dbl<FP>(acc);
            glv_fold_add<FP>(acc, dgp->d[k], P, DP, beta, beta2);
            glv_fold_add<FP>(acc, dgp->e[k], Q, DQ, beta, beta2);
        }
        if (dgp->plus_lo && !P.ident) xyzzz_madd<FP>(acc, P.x, P.y);
        emit_affine<FP>(acc, out_xy + i * 2 * W, out_zero + i);
    }
}

// joint sparse form (Solinas) of two canonical scalars (8 x 32-bit limbs each), least significant column first
PLK_DI int joint_sparse_form(const uint32_t* a, const uint32_t* b, int8_t* ua, int8_t* ub) {
    // 9 limbs so that the shifts never lose a bit; the carries d0, d1 in {0, 1} are added on the fly
    uint32_t k0[9], k1[9];
    for (int i = 0; i < 8; ++i) {
        k0[i] = a[i];
        k1[i] = b[i];
    }
    k0[8] = k1[8] = 0;
    int d0 = 0, d1 = 0, n = 0;
    for (;;) {
        uint32_t nz = (uint32_t)d0 | (uint32_t)d1;
        for (int i = 0; i < 9; ++i) nz |= k0[i] | k1[i];
        if (nz == 0) break;
        const uint32_t l0 = (k0[0] + (uint32_t)d0) & 7u, l1 = (k1[0] + (uint32_t)d1) & 7u;
        int u0 = 0, u1 = 0;
        if (l0 & 1) {
            u0 = 2 - (int)(l0 & 3);
            if ((l0 == 3 || l0 == 5) && (l1 & 3) == 2) u0 = -u0;
        }
        if (l1 & 1) {
            u1 = 2 - (int)(l1 & 3);
            if ((l1 == 3 || l1 == 5) && (l0 & 3) == 2) u1 = -u1;
        }
        if (2 * d0 == 1 + u0) d0 = 1 - d0;
        if (2 * d1 == 1 + u1) d1 = 1 - d1;
        for (int i = 0; i < 8; ++i) {
            k0[i] = (k0[i] >> 1) | (k0[i + 1] << 31);
            k1[i] = (k1[i] >> 1) | (k1[i + 1] << 31);
        }
        k0[8] >>= 1;
        k1[8] >>= 1;
        ua[n] = (int8_t)u0;
        ub[n] = (int8_t)u1;
        ++n;
    }
    return n;
}

struct ScalarPair {
    uint32_t a[8], b[8];  // Montgomery form in the scalar field
};
// one thread: Montgomery -> canonical (to_canonical_u64_vec), joint sparse form, most significant column first
// d_sc (optional): the two scalars in device memory (2 x 8 words, Montgomery) instead of `sp` - an inner-product argument keeps
// its running scale and the squared challenge there (halo.hip).  plus_lo: the scalar a is replaced by "add lo once at the end".
template <class C> __global__ void k_fold_digits(ScalarPair sp, const uint32_t* __restrict__ d_sc, FoldDigits* out, int use_glv, int plus_lo) {
    using SP = typename C::SP;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fe<SP> a, b;
    for (int i = 0; i < 8; ++i) {
        a.v[i] = d_sc ? d_sc[i] : sp.a[i];
        b.v[i] = d_sc ? d_sc[8 + i] : sp.b[i];
    }
    a = plus_lo ? fe_zero<SP>() : fe_to_canonical<SP>(a);
    b = fe_to_canonical<SP>(b);
    out->plus_lo = plus_lo;
    int8_t ua[FOLD_MAX_COLS], ub[FOLD_MAX_COLS];
    if (C::Glv::ENABLED && use_glv) {
        // two joint sparse forms of half-length parts, signs applied to the digits, aligned at the least significant column
        int8_t va[FOLD_MAX_COLS], vb[FOLD_MAX_COLS];
        uint32_t a1[8], a2[8], b1[8], b2[8];
        if constexpr (C::Glv::ENABLED) {
            glv_split<typename C::Glv>(a.v, a1, a2);
            glv_split<typename C::Glv>(b.v, b1, b2);
        }
        const int sa1 = (a1[7] >> 31) ? -1 : 1, sa2 = (a2[7] >> 31) ? -1 : 1, sb1 = (b1[7] >> 31) ? -1 : 1, sb2 = (b2[7] >> 31) ? -1 : 1;
        a1[7] &= 0x7fffffffu; a2[7] &= 0x7fffffffu; b1[7] &= 0x7fffffffu; b2[7] &= 0x7fffffffu;
        const int na = joint_sparse_form(a1, a2, ua, ub), nb = joint_sparse_form(b1, b2, va, vb);
        const int n = na > nb ? na : nb;
        out->cols = n;
        for (int k = 0; k < n; ++k) {
            const int col = n - 1 - k;  // most significant first
            const int u0 = col < na ? ua[col] * sa1 : 0, u1 = col < na ? ub[col] * sa2 : 0;
            const int v0 = col < nb ? va[col] * sb1 : 0, v1 = col < nb ? vb[col] * sb2 : 0;
            out->d[k] = (int8_t)((u0 + 1) * 3 + (u1 + 1));
            out->e[k] = (int8_t)((v0 + 1) * 3 + (v1 + 1));
        }
    } else {
        const int n = joint_sparse_form(a.v, b.v, ua, ub);
        out->cols = n;
        for (int k = 0; k < n; ++k) out->d[k] = (int8_t)((ua[n - 1 - k] + 1) * 3 + (ub[n - 1 - k] + 1));
    }
}

template <class C>
static int fold_pairs_t(size_t m, const void* d_lo, const void* d_lo_zero, const void* d_hi, const void* d_hi_zero, const uint64_t* a_mont,
                        const uint64_t* b_mont, void* d_out_xy, void* d_out_zero, hipStream_t stream, const void* d_scalars, int plus_lo) {
    static_assert(C::SP::NL == 8, "scalar fields are 256-bit");
    if (m == 0) return PLK_OK;
    ScalarPair sp{};
    for (int i = 0; i < 4 && !d_scalars; ++i) {
        sp.a[2 * i] = (uint32_t)a_mont[i];
        sp.a[2 * i + 1] = (uint32_t)(a_mont[i] >> 32);
        sp.b[2 * i] = (uint32_t)b_mont[i];
        sp.b[2 * i + 1] = (uint32_t)(b_mont[i] >> 32);
    }
    FoldDigits* d_dg = (FoldDigits*)scratch_acquire(sizeof(FoldDigits), stream);
    if (!d_dg) return PLK_ERR_OOM;
    const bool use_glv = C::Glv::ENABLED && !getenv("PLK_FOLD_NO_GLV");
    k_fold_digits<C><<<1, 64, 0, stream>>>(sp, (const uint32_t*)d_scalars, d_dg, use_glv ? 1 : 0, plus_lo);
    if (use_glv)
        k_fold_pairs_glv<C><<<(unsigned)((m + 127) / 128), 128, 0, stream>>>((const uint4*)d_lo, (const uint8_t*)d_lo_zero, (const uint4*)d_hi,
                                                                             (const uint8_t*)d_hi_zero, m, d_dg, (uint4*)d_out_xy, (uint8_t*)d_out_zero);
    else
        k_fold_pairs<C><<<(unsigned)((m + 127) / 128), 128, 0, stream>>>((const uint4*)d_lo, (const uint8_t*)d_lo_zero, (const uint4*)d_hi,
                                                                         (const uint8_t*)d_hi_zero, m, d_dg, (uint4*)d_out_xy, (uint8_t*)d_out_zero);
    hipError_t e = hipGetLastError();
    scratch_release(d_dg, stream);
    if (e != hipSuccess) return set_error(PLK_ERR_HIP, "fold launch failed: %s", hipGetErrorString(e));
    return PLK_OK;
}

// d_scalars (optional): the two scalars as 2 x 32 bytes of device memory (then a_mont / b_mont are not read).  plus_lo:
// out_i = lo_i + [b] hi_i.
int curve_fold_pairs_dev_impl(int curve, size_t m, const void* d_lo, const void* d_lo_zero, const void* d_hi, const void* d_hi_zero,
                              const uint64_t* a_mont, const uint64_t* b_mont, void* d_out_xy, void* d_out_zero, hipStream_t stream, const void* d_scalars,
                              int plus_lo) {
    if (!d_scalars && (!a_mont || !b_mont)) return set_error(PLK_ERR_INVALID_ARG, "null scalar");
    if (m && (!d_lo || !d_hi || !d_out_xy || !d_out_zero)) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: return fold_pairs_t<TweedledeeCurve>(m, d_lo, d_lo_zero, d_hi, d_hi_zero, a_mont, b_mont, d_out_xy, d_out_zero, stream, d_scalars, plus_lo);
        case PLK_CURVE_TWEEDLEDUM: return fold_pairs_t<TweedledumCurve>(m, d_lo, d_lo_zero, d_hi, d_hi_zero, a_mont, b_mont, d_out_xy, d_out_zero, stream, d_scalars, plus_lo);
        case PLK_CURVE_BLS12_377: return fold_pairs_t<Bls12377Curve>(m, d_lo, d_lo_zero, d_hi, d_hi_zero, a_mont, b_mont, d_out_xy, d_out_zero, stream, d_scalars, plus_lo);
        case PLK_CURVE_PALLAS: return fold_pairs_t<PallasCurve>(m, d_lo, d_lo_zero, d_hi, d_hi_zero, a_mont, b_mont, d_out_xy, d_out_zero, stream, d_scalars, plus_lo);
        case PLK_CURVE_VESTA: return fold_pairs_t<VestaCurve>(m, d_lo, d_lo_zero, d_hi, d_hi_zero, a_mont, b_mont, d_out_xy, d_out_zero, stream, d_scalars, plus_lo);
    }
    return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
}

}  // namespace plk
