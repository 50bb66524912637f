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

// ---- 2^r-to-1 fold ------------------------------------------------------------------------------------------------------
// r rounds of an inner-product argument folded at once (halo.hip keeps L_j / R_j of those rounds on the caller's commitment
// tables, so the folded generators of the rounds in between are never needed):
//   out_i = G_i + sum_{t = 1 .. T - 1} [s_t] G_{i + t n_out},   T = 2^r, the same T - 1 scalars for every i
// (the scaled form of halo.rs:119-123 applied r times: s_t = the product of u_k^2 over the rounds k in which index i + t n_out fell
// into the upper half).  Straus: ONE chain of ~128 doublings per output shared by all its inputs, and per input the joint sparse
// form of its scalar's two endomorphism halves over (Q, phi Q) - ~64 additions.  Sequential pair folds pay the doubling chain per
// output of every round (n/2 + n/4 + ... chains instead of n/T): 47 % fewer field multiplications at r = 3.
// Two kernels: k_fold_multi_prep writes, per input, Q and Q - phi(Q) as affine points in R'-form (pairs of inputs share an
// inversion); k_fold_multi_glv streams them (64 coalesced bytes per addition) - a lane holds its accumulator and one operand,
// not the 2 x 4 precomputed points of the pair kernel.  Workgroups of ONE wave: at 2^16 outputs (one wave per SIMD on average) the waves
// of two-wave workgroups were seen to share SIMDs while others idle - 2.85 ms against 1.82 ms for the 4-to-1 fold 2^18 -> 2^16
// (tools/lab/mul_latency.hip shows the same for a bare chain of doublings: 3.6 against 2.2 us per step).
constexpr int FOLD_MULTI_MAX_LOG = 4;
constexpr int FOLD_MULTI_MAX = 1 << FOLD_MULTI_MAX_LOG;
struct MultiDigits {
    int cols;                                       // columns, LEAST significant first
    int pad[3];
    uint32_t beta[2][16];                           // beta, beta^2 as Fz limbs (R'-form)
    int8_t d[FOLD_MULTI_MAX - 1][FOLD_MAX_COLS];    // input t (1 .. T - 1): (u0 + 1) * 3 + (u1 + 1) over (Q, phi Q); 4 = empty column
};

// thread t of one block: the digits of scalar t.  ratios: 2^r_bits scalars (Montgomery), the scalar of input t at index bitrev_r(t)
template <class C> __global__ void __launch_bounds__(64) k_fold_multi_digits(const uint32_t* __restrict__ ratios, int r_bits, MultiDigits* out) {
    using SP = typename C::SP;
    using FP = typename C::FP;
    if constexpr (C::Glv::ENABLED) {
        __shared__ int s_cols;
        const int t = threadIdx.x, T = 1 << r_bits;
        if (t == 0) {
            s_cols = 0;
            Fe<FP> bc;
#pragma unroll
            for (int k = 0; k < FP::NL; ++k) bc.v[k] = C::Glv::BETA[k];
            const Fe<FP> beta_r = fe_from_canonical<FP>(bc);
            const Fz<FP> b1 = fz_from_fe<FP>(to_rprime<FP>(beta_r)), b2 = fz_from_fe<FP>(to_rprime<FP>(fe_sqr<FP>(beta_r)));
            for (int k = 0; k < FzCfg<FP>::NZ; ++k) {
                out->beta[0][k] = b1.l[k];
                out->beta[1][k] = b2.l[k];
            }
        }
        __syncthreads();
        if (t >= 1 && t < T) {
            int rev = 0;
            for (int k = 0; k < r_bits; ++k) rev |= ((t >> k) & 1) << (r_bits - 1 - k);
            Fe<SP> s;
            for (int i = 0; i < 8; ++i) s.v[i] = ratios[rev * 8 + i];
            s = fe_to_canonical<SP>(s);
            uint32_t k1[8], k2[8];
            glv_split<typename C::Glv>(s.v, k1, k2);
            const int s1 = (k1[7] >> 31) ? -1 : 1, s2 = (k2[7] >> 31) ? -1 : 1;
            k1[7] &= 0x7fffffffu;
            k2[7] &= 0x7fffffffu;
            int8_t ua[FOLD_MAX_COLS], ub[FOLD_MAX_COLS];
            const int n = joint_sparse_form(k1, k2, ua, ub);
            for (int k = 0; k < FOLD_MAX_COLS; ++k) out->d[t - 1][k] = k < n ? (int8_t)((ua[k] * s1 + 1) * 3 + (ub[k] * s2 + 1)) : (int8_t)4;
            atomicMax(&s_cols, n);
        }
        __syncthreads();
        if (t == 0) out->cols = s_cols;
    }
}

// input j of `count` (points first + j of g): pp[j] = the point, dp[j] = point - phi(point), affine, x then y, canonical R'-form words.
// Lane l takes inputs l and l + ceil(count / 2): one inversion for both.  An identity input leaves its slots unwritten (never read).
template <class C>
__global__ void __launch_bounds__(64) k_fold_multi_prep(const uint4* __restrict__ g, const uint8_t* __restrict__ gz, size_t first, size_t count,
                                                         uint4* __restrict__ pp, uint4* __restrict__ dp) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    if constexpr (C::Glv::ENABLED) {
        const size_t half = (count + 1) / 2;
        const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (l >= half) return;
        const size_t j1 = l, j2 = l + half;
        const bool has2 = j2 < count;
        const size_t a1 = first + j1, a2 = first + (has2 ? j2 : j1);
        const Fe<FP> px = fe_load<FP>(g + a1 * 2 * W), py = fe_load<FP>(g + a1 * 2 * W + W);
        const Fe<FP> qx = fe_load<FP>(g + a2 * 2 * W), qy = fe_load<FP>(g + a2 * 2 * W + W);
        const bool pi = gz ? gz[a1] != 0 : false, qi = !has2 || (gz ? gz[a2] != 0 : false);
        Fe<FP> bc;
#pragma unroll
        for (int k = 0; k < FP::NL; ++k) bc.v[k] = C::Glv::BETA[k];
        const Fe<FP> beta_r = fe_from_canonical<FP>(bc);
        const Fe<FP> bm1 = fe_sub<FP>(beta_r, fe_one<FP>());
        const Fe<FP> dpv = pi ? fe_one<FP>() : fe_mul<FP>(bm1, px), dqv = qi ? fe_one<FP>() : fe_mul<FP>(bm1, qx);
        const Fe<FP> inv = fe_inv_safegcd<FP>(fe_mul<FP>(dpv, dqv));
        auto emit = [&](const Fe<FP>& x, const Fe<FP>& y, const Fe<FP>& inv_x, size_t j) {
            // Q - phi(Q): slope (-y - y) / (beta x - x)
            const Fe<FP> lam = fe_mul<FP>(fe_neg<FP>(fe_dbl<FP>(y)), inv_x);
            const Fe<FP> x3 = fe_sub<FP>(fe_sub<FP>(fe_sqr<FP>(lam), x), fe_mul<FP>(beta_r, x));
            const Fe<FP> y3 = fe_sub<FP>(fe_mul<FP>(lam, fe_sub<FP>(x, x3)), y);
            fe_store<FP>(pp + j * 2 * W, to_rprime<FP>(x));
            fe_store<FP>(pp + j * 2 * W + W, to_rprime<FP>(y));
            fe_store<FP>(dp + j * 2 * W, to_rprime<FP>(x3));
            fe_store<FP>(dp + j * 2 * W + W, to_rprime<FP>(y3));
        };
        if (!pi) emit(px, py, fe_mul<FP>(inv, dqv), j1);
        if (!qi) emit(qx, qy, fe_mul<FP>(inv, dpv), j2);
    }
}

template <class C>
__global__ void __launch_bounds__(64) k_fold_multi_glv(const uint4* __restrict__ g, const uint8_t* __restrict__ gz, size_t n_out, int T,
                                                        const uint4* __restrict__ pp, const uint4* __restrict__ dp, const MultiDigits* __restrict__ dg,
                                                        uint4* __restrict__ out_xy, uint8_t* __restrict__ out_zero) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    constexpr int NZ = FzCfg<FP>::NZ;
    if constexpr (C::Glv::ENABLED) {
        const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n_out) return;
        uint32_t ident = 0;  // bit t: input t is the identity
        if (gz)
            for (int t = 0; t < T; ++t) ident |= gz[i + (size_t)t * n_out] ? (1u << t) : 0u;
        XyzzZ<FP> acc = xyzzz_identity<FP>();
        const int cols = dg->cols;
        for (int k = cols - 1; k >= 0; --k) {
            acc = xyzzz_dbl<FP>(acc);
            for (int t = 1; t < T; ++t) {
                const int code = dg->d[t - 1][k];  // uniform over the grid
                if (code == 4) continue;
                if ((ident >> t) & 1u) continue;   // per lane: the addition below is masked
                const int u0 = code / 3 - 1, u1 = code % 3 - 1;
                // (1, 0) Q   (0, 1) phi Q = (beta x, y)   (1, 1) Q + phi Q = (beta^2 x, -y)   (1, -1) Q - phi Q
                const bool diff = u0 != 0 && u1 != 0 && u0 != u1;
                const uint4* src = (diff ? dp : pp) + ((size_t)(t - 1) * n_out + i) * 2 * W;
                Fz<FP> tx = fz_from_fe<FP>(fe_load<FP>(src)), ty = fz_from_fe<FP>(fe_load<FP>(src + W));
                bool neg = u0 != 0 ? u0 < 0 : u1 < 0;
                if (u0 == 0 || (u1 != 0 && u0 == u1)) {
                    const int which = u0 == 0 ? 0 : 1;
                    Fz<FP> b;
#pragma unroll
                    for (int q = 0; q < NZ; ++q) b.l[q] = dg->beta[which][q];
                    tx = fz_mul<FP>(tx, b);
                    if (which == 1) neg = !neg;
                }
                if (neg) ty = fz_neg_canonical<FP>(ty);
                xyzzz_madd<FP>(acc, tx, ty);
            }
        }
        if (!(ident & 1u)) {
            const Fz<FP> x0 = fz_from_fe<FP>(to_rprime<FP>(fe_load<FP>(g + i * 2 * W))), y0 = fz_from_fe<FP>(to_rprime<FP>(fe_load<FP>(g + i * 2 * W + W)));
            xyzzz_madd<FP>(acc, x0, y0);
        }
        emit_affine<FP>(acc, out_xy + i * 2 * W, out_zero + i);
    }
}

template <class C>
static int fold_multi_t(size_t n_out, int r_bits, const void* d_g, const void* d_gz, const void* d_ratios, void* d_out_xy, void* d_out_zero, hipStream_t stream) {
    if constexpr (!C::Glv::ENABLED) {
        return set_error(PLK_ERR_INVALID_ARG, "the 2^r-to-1 fold runs along the curve endomorphism: not on this curve");
    } else {
        constexpr size_t PT = (size_t)2 * C::FP::NL * 4;
        const int T = 1 << r_bits;
        const size_t count = (size_t)(T - 1) * n_out;
        uint8_t* work = (uint8_t*)scratch_acquire(2 * count * PT + sizeof(MultiDigits) + 256, stream);
        if (!work) return PLK_ERR_OOM;
        uint4* pp = (uint4*)work;
        uint4* dp = (uint4*)(work + count * PT);
        MultiDigits* dg = (MultiDigits*)(work + ((2 * count * PT + 255) & ~(size_t)255));
        k_fold_multi_digits<C><<<1, 64, 0, stream>>>((const uint32_t*)d_ratios, r_bits, dg);
        const size_t half = (count + 1) / 2;
        k_fold_multi_prep<C><<<(unsigned)((half + 63) / 64), 64, 0, stream>>>((const uint4*)d_g, (const uint8_t*)d_gz, n_out, count, pp, dp);
        k_fold_multi_glv<C><<<(unsigned)((n_out + 63) / 64), 64, 0, stream>>>((const uint4*)d_g, (const uint8_t*)d_gz, n_out, T, pp, dp, dg,
                                                                                 (uint4*)d_out_xy, (uint8_t*)d_out_zero);
        hipError_t e = hipGetLastError();
        scratch_release(work, stream);
        if (e != hipSuccess) return set_error(PLK_ERR_HIP, "fold launch failed: %s", hipGetErrorString(e));
        return PLK_OK;
    }
}

// out_i = g_i + sum_{t = 1 .. 2^r_bits - 1} [ratio_t] g_{i + t n_out}, i < n_out; d_ratios: 2^r_bits scalars in device memory
// (Montgomery, scalar field), the scalar of input t at index bitrev(t) (entry 0 is not read).  In place (d_out = d_g) is fine.
int curve_fold_multi_dev_impl(int curve, size_t n_out, int r_bits, const void* d_g, const void* d_gz, const void* d_ratios, void* d_out_xy,
                              void* d_out_zero, hipStream_t stream) {
    if (r_bits < 1 || r_bits > FOLD_MULTI_MAX_LOG) return set_error(PLK_ERR_INVALID_ARG, "fold of 2^%d inputs per output: 1 <= r <= %d", r_bits, FOLD_MULTI_MAX_LOG);
    if (n_out == 0) return PLK_OK;
    if (!d_g || !d_ratios || !d_out_xy || !d_out_zero) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: return fold_multi_t<TweedledeeCurve>(n_out, r_bits, d_g, d_gz, d_ratios, d_out_xy, d_out_zero, stream);
        case PLK_CURVE_TWEEDLEDUM: return fold_multi_t<TweedledumCurve>(n_out, r_bits, d_g, d_gz, d_ratios, d_out_xy, d_out_zero, stream);
        case PLK_CURVE_BLS12_377: return fold_multi_t<Bls12377Curve>(n_out, r_bits, d_g, d_gz, d_ratios, d_out_xy, d_out_zero, stream);
        case PLK_CURVE_PALLAS: return fold_multi_t<PallasCurve>(n_out, r_bits, d_g, d_gz, d_ratios, d_out_xy, d_out_zero, stream);
        case PLK_CURVE_VESTA: return fold_multi_t<VestaCurve>(n_out, r_bits, d_g, d_gz, d_ratios, d_out_xy, d_out_zero, stream);
    }
    return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
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
