// ecz_coop.cuh -- XYZZ doubling / addition computed by a QUAD of four adjacent lanes.
//
// The reduction tail of an MSM (plane sums, doubling the planes into place, the last additions) is a
// handful of points and a long dependent chain: one wave's field multiplication takes ~1.1 us whatever
// else the GPU does, so the chain is latency, not throughput.  A point operation is 9-14 multiplications
// but only 3-4 deep.  Here the four lanes of a quad hold identical copies of the operands, each lane
// performs ONE of the (up to four) independent products of a round on its own operand pair, and the
// products are broadcast back inside the quad with DPP quad_perm moves: a doubling costs 3 multiplication
// latencies instead of 9, an addition 4 instead of 14.  Same formulas, same value bounds and the same
// exceptional cases as ecz.cuh (every decision depends on quad-uniform data, so a quad never diverges).
#pragma once
#include "ecz.cuh"

namespace plk {

template <int SRC> PLK_DI uint32_t quad_bcast_u32(uint32_t v) {
    uint32_t r = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, SRC * 0x55, 0xf, 0xf, true);  // quad_perm:[SRC,SRC,SRC,SRC]
    // Keep the move a move: ROCm 7.2's DPP combiner folds it into the consuming VALU instruction and gets
    // a - b wrong when both operands are broadcasts of the same register (observed on gfx950: only the lane
    // that owns the subtrahend computed the right difference; tools/lab/dpp_combine_repro.hip and plk_selftest_quad pin it).
    asm volatile("" : "+v"(r));
    return r;
}
template <class FP, int SRC> PLK_DI Fz<FP> quad_bcast(const Fz<FP>& v) {
    Fz<FP> r;
#pragma unroll
    for (int i = 0; i < FzCfg<FP>::NZ; ++i) r.l[i] = quad_bcast_u32<SRC>(v.l[i]);
    return r;
}
// operand of lane ql (0..3) of the quad
template <class FP> PLK_DI Fz<FP> quad_select(int ql, const Fz<FP>& a0, const Fz<FP>& a1, const Fz<FP>& a2, const Fz<FP>& a3) {
    Fz<FP> r;
#pragma unroll
    for (int i = 0; i < FzCfg<FP>::NZ; ++i) {
        const uint32_t lo = (ql & 1) ? a1.l[i] : a0.l[i];
        const uint32_t hi = (ql & 1) ? a3.l[i] : a2.l[i];
        r.l[i] = (ql & 2) ? hi : lo;
    }
    return r;
}

// 2 * a; every lane of the quad holds a and receives the result (formulas and bounds of xyzzz_dbl)
template <class FP> PLK_DI XyzzZ<FP> xyzzz_dbl_q(const XyzzZ<FP>& a, int ql) {
    if (a.inf) return a;
    XyzzZ<FP> o;
    const Fz<FP> u = fz_dbl<FP>(a.y);                                    // < 8
    // round 1: u^2 | x^2
    Fz<FP> op = quad_select<FP>(ql, u, a.x, u, a.x);
    Fz<FP> r = fz_mul<FP>(op, op);
    const Fz<FP> v = quad_bcast<FP, 0>(r), xx = quad_bcast<FP, 1>(r);   // < 1.5
    const Fz<FP> m = fz_add<FP>(fz_dbl<FP>(xx), xx);                     // < 4.5
    // round 2: u v | x v | m^2 | v zz
    r = fz_mul<FP>(quad_select<FP>(ql, u, a.x, m, v), quad_select<FP>(ql, v, v, m, a.zz));
    const Fz<FP> w = quad_bcast<FP, 0>(r), s = quad_bcast<FP, 1>(r), mm = quad_bcast<FP, 2>(r);
    o.zz = quad_bcast<FP, 3>(r);
    o.x = fz_sub<FP, 2>(mm, fz_dbl<FP>(s));                              // < 5.2
    const Fz<FP> t = fz_sub<FP, 3>(s, o.x);                              // < 9.1
    // round 3: m t | w y | w zzz
    r = fz_mul<FP>(quad_select<FP>(ql, m, w, w, w), quad_select<FP>(ql, t, a.y, a.zzz, a.zzz));
    o.y = fz_sub<FP, 1>(quad_bcast<FP, 0>(r), quad_bcast<FP, 1>(r));     // < 3.4
    o.zzz = quad_bcast<FP, 2>(r);
    o.inf = fz_is_zero_mod_p<FP>(o.zz);                                  // 2-torsion point
    return o;
}

// a + b; every lane of the quad holds both operands and receives the result (formulas and bounds of xyzzz_add)
template <class FP> PLK_DI XyzzZ<FP> xyzzz_add_q(const XyzzZ<FP>& a, const XyzzZ<FP>& b, int ql) {
    if (a.inf) return b;
    if (b.inf) return a;
    // round 1: x1 zz2 | x2 zz1 | y1 zzz2 | y2 zzz1
    Fz<FP> r = fz_mul<FP>(quad_select<FP>(ql, a.x, b.x, a.y, b.y), quad_select<FP>(ql, b.zz, a.zz, b.zzz, a.zzz));
    const Fz<FP> u1 = quad_bcast<FP, 0>(r), u2 = quad_bcast<FP, 1>(r), s1 = quad_bcast<FP, 2>(r), s2 = quad_bcast<FP, 3>(r);
    const Fz<FP> p = fz_sub<FP, 1>(u2, u1);                              // < 3.2
    const Fz<FP> rr_in = fz_sub<FP, 1>(s2, s1);                          // < 3.2
    // round 2: p^2 | r^2 | zz1 zz2 | zzz1 zzz2
    r = fz_mul<FP>(quad_select<FP>(ql, p, rr_in, a.zz, a.zzz), quad_select<FP>(ql, p, rr_in, b.zz, b.zzz));
    const Fz<FP> pp = quad_bcast<FP, 0>(r), rr = quad_bcast<FP, 1>(r), zz12 = quad_bcast<FP, 2>(r), zzz12 = quad_bcast<FP, 3>(r);
    // round 3: p pp | u1 pp | zz12 pp
    r = fz_mul<FP>(quad_select<FP>(ql, p, u1, zz12, zz12), pp);
    const Fz<FP> ppp = quad_bcast<FP, 0>(r), q = quad_bcast<FP, 1>(r);
    XyzzZ<FP> o;
    o.zz = quad_bcast<FP, 2>(r);
    if (fz_is_zero_mod_p<FP>(o.zz)) {
        // p = 0 mod p: same x.  Same point -> double it; opposite points -> identity.
        if (fz_is_zero_mod_p<FP>(rr)) return xyzzz_dbl_q<FP>(a, ql);
        o = a;
        o.inf = true;
        return o;
    }
    o.x = fz_sub<FP, 2>(fz_sub<FP, 1>(rr, ppp), fz_dbl<FP>(q));          // < 7.3
    const Fz<FP> t = fz_sub<FP, 3>(q, o.x);                              // < 9.3
    // round 4: r t | s1 ppp | zzz12 ppp
    r = fz_mul<FP>(quad_select<FP>(ql, rr_in, s1, zzz12, zzz12), quad_select<FP>(ql, t, ppp, ppp, ppp));
    o.y = fz_sub<FP, 1>(quad_bcast<FP, 0>(r), quad_bcast<FP, 1>(r));     // < 3.3
    o.zzz = quad_bcast<FP, 2>(r);
    o.inf = false;
    return o;
}

// sum over `quads` adjacent quads of a wave (a power of two <= 16); every lane ends with the total of its group
template <class FP> PLK_DI XyzzZ<FP> wave_sum_q(XyzzZ<FP> v, int quads, int ql) {
    for (int m = 4; m < 4 * quads; m <<= 1) v = xyzzz_add_q<FP>(v, xyzzz_shfl_xor<FP>(v, m), ql);
    return v;
}

}  // namespace plk
