// ntt.hip -- natural-order in / natural-order out NTT over the reference's six prime fields (tuned for the 256-bit ones), gfx950.
//
// Replaces the reference's src/fft.rs:
//   fft_precompute                         fft.rs:47-59    -> NttPlan (device twiddle tables)
//   fft_with_precomputation_power_of_2     fft.rs:103-156  -> ntt_dev_impl(inverse = 0)
//   ifft_with_precomputation_power_of_2    fft.rs:82-101   -> ntt_dev_impl(inverse = 1)
//   reverse_index_bits (both of them)      fft.rs:8-26,120,155 -> absorbed into the tile indexing
// The reference runs log n radix-2 layers with a barrier and two full array copies per layer.
// Here the transform is a multi-pass decomposition n = A_1 * A_2 * ... * A_m (A_t <= 2^7 for
// strided passes): pass t performs, for every contiguous block of N_t = A_t * S_t elements and
// every residual index r < S_t, an A_t-point NTT over the stride-S_t column, multiplies by the
// inter-pass twiddle w_{N_t}^(r k) and stores in place; the last pass works on contiguous
// blocks and scatters to the digit-reversed final position.  Each workgroup owns a
// [A_t x Q] tile (1024 elements, 36 KiB of LDS as nine limb planes) so that every global access is a run of
// Q * 32 B (>= 256 B) contiguous bytes; the A_t-point NTTs run inside LDS, two stages per round trip
// (radix-4 steps in registers), with the stage twiddles staged in LDS (decimation in time on lazily reduced
// 29-bit limbs whose carries are moved twice per step, fz.cuh; input placed bit-reversed by the load indexing,
// output in natural order).  Between two passes the data is in limb form (48 B per element) in a scratch buffer.
// out[j] = sum_k in[k] w^(jk), w = primitive_root_of_unity(log n) (field.rs:429-435): the same
// function the reference computes, and field elements have a unique representation, so the
// limbs are bit-identical.  iNTT = the same passes with w^-1 tables and n^-1 folded into the
// first inter-pass twiddle table (instead of the index-reversal trick of fft.rs:90-99).
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"
#include "fp.cuh"
#include "fz.cuh"
#include "tables.cuh"

namespace plk {

#ifndef PLK_NTT_TILE_LOG
#define PLK_NTT_TILE_LOG 10
#endif
constexpr int TILE_LOG = PLK_NTT_TILE_LOG;  // elements per workgroup tile
constexpr int TILE = 1 << TILE_LOG;
constexpr int NTT_THREADS = TILE / 4;        // one radix-4 group per thread and stage pair
constexpr int MAX_PASSES = 6;
constexpr int NTT_STAGGER_DEFAULT = 0;  // phases of a one-round pass (k_ntt_pass), sleep quanta per phase; 0: off
constexpr size_t NTT_LDS_MAX = 160 * 1024;  // per workgroup on gfx950
constexpr int INNER_LOG = TILE_LOG;      // inner twiddle table: w_TILE^e, e < TILE / 2
// 16-byte words per element in the caller's buffers and in the R-form / R'-form word tables: 2 for the 256-bit fields, 3 for Bls12377Base
template <class P> constexpr int EU() { return P::NL / 4; }

struct NttPassArgs {
    int log_n;        // whole transform
    int log_a;        // this pass: NTT length A = 2^log_a
    int log_q;        // columns per tile, Q = 2^log_q
    int log_nt;       // N_t: size of the contiguous sub-problem blocks of this pass
    int log_s;        // S_t = N_t / A_t
    int first;        // 1: first pass (reads the caller's input)
    int last;         // 1: last pass (contiguous blocks in, digit-reversed scatter out)
    int n_prev;       // number of earlier passes (last pass only)
    int prev_log[MAX_PASSES];  // their log sizes a_1..a_{m-1}
    int scale;        // 1: multiply outputs by *scale_ptr (single-pass inverse)
    int skip;         // first pass of a zero-padded transform: the first `skip` stages only replicate (see tile_stages)
    int tw_global;    // 1: the stage twiddles do not fit in LDS next to the tile: they are read from the inner table (L1 / L2)
    int shuffle;      // 1: the first two radix-4 steps of a tile are joined by wave shuffles instead of an LDS round trip
    int stagger;      // > 0: a pass that is ONE round of workgroups starts them in phases (see k_ntt_pass): sleep quanta (~0.5 us) per phase
    int stagger_div;  // workgroups per phase (the number of CUs)
};

// ---------------------------------------------------------------------------------------------
// table generation
// ---------------------------------------------------------------------------------------------
// pw[b]      = w^(2^b),   b < log_t, w = primitive 2^log_t-th root (ROOT_2ADIC^(2^(adicity-log_t)))
// pw[32+b]   = w^-(2^b)
// pw[64]     = 2^-log_n in Montgomery form (n^-1)              -- all of the above in R-form
// pw[65]     = 1 and pw[66] = n^-1 in R'-form (the form every twiddle TABLE is stored in: the pass
//              kernel multiplies R-form data by R'-form twiddles on 29-bit limbs, fz.cuh, and the
//              product x 2^256 * w 2^261 / 2^261 stays in the reference's R-form)
template <class P> __global__ void k_ntt_pow2(uint4* pw, int log_t, int log_n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fe<P> w;
#pragma unroll
    for (int i = 0; i < P::NL; ++i) w.v[i] = P::ROOT_2ADIC[i];
    for (int i = 0; i < P::TWO_ADICITY - log_t; ++i) w = fe_sqr<P>(w);
    Fe<P> cur = w, winv = fe_one<P>();
    for (int b = 0; b < log_t; ++b) {
        fe_store<P>(pw + b * EU<P>(), cur);
        winv = fe_mul<P>(winv, cur);  // w^(2^log_t - 1) = w^-1
        cur = fe_sqr<P>(cur);
    }
    cur = winv;
    for (int b = 0; b < log_t; ++b) {
        fe_store<P>(pw + (32 + b) * EU<P>(), cur);
        cur = fe_sqr<P>(cur);
    }
    Fe<P> ninv = fe_one<P>();
    for (int i = 0; i < log_n; ++i) ninv = fe_half<P>(ninv);
    fe_store<P>(pw + 64 * EU<P>(), ninv);
    fe_store<P>(pw + 65 * EU<P>(), fz_to_fe_canonical<P>(fz_one_rprime<P>()));
    fe_store<P>(pw + 66 * EU<P>(), to_rprime<P>(ninv));
}

// inner table: tw[e] = w_1024^(+-e), e < 512, expressed through the 2^log_t-th root
template <class P> __global__ void k_ntt_fill_inner(uint4* tw, const uint4* pw, int log_t, int inverse) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (1 << (INNER_LOG - 1))) return;
    uint64_t ex = (uint64_t)e << (log_t - INNER_LOG);
    fe_store<P>(tw + e * EU<P>(), to_rprime<P>(pow_from_table<P>(pw, inverse ? 32 : 0, ex, log_t)));
}
// outer table of a pass: W[k * S + r] = w_{N_t}^(+- r k) (* n^-1 when scale != 0), R'-form, limb form
template <class P> __global__ void k_ntt_fill_outer(uint32_t* tw, const uint4* pw, int log_t, int log_nt, int log_s, int inverse, int scale) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ((size_t)1 << log_nt)) return;
    uint64_t r = idx & (((uint64_t)1 << log_s) - 1);
    uint64_t k = idx >> log_s;
    uint64_t ex = (r * k) & (((uint64_t)1 << log_nt) - 1);
    ex <<= (log_t - log_nt);
    Fe<P> v = pow_from_table<P>(pw, inverse ? 32 : 0, ex, log_t);
    if (scale) v = fe_mul<P>(v, fe_load<P>(pw + 64 * EU<P>()));
    limbs_store<P>(tw, idx, fz_from_fe<P>(to_rprime<P>(v)));
}

// The reference's own table (FftPrecomputation::subgroups_rev, fft.rs:28-59), flat: element 2^i - 1 + k is entry k of layer i =
// g_i^(bitrev_i(k)) with g_i = primitive_root_of_unity(i) = w^(2^(log_t - i)) (field.rs:429-435), R-form as the reference stores it
PLK_DI uint32_t bitrev_u32(uint32_t x, int bits) { return bits == 0 ? 0u : (__brev(x) >> (32 - bits)); }
template <class P> __global__ void __launch_bounds__(256) k_ntt_reference_table(uint4* __restrict__ out, const uint4* __restrict__ pw, int log_t, int log_n) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ((size_t)2 << log_n) - 1) return;
    const int i = 63 - __clzll((unsigned long long)(e + 1));
    const uint32_t k = (uint32_t)(e + 1 - ((size_t)1 << i));
    const uint64_t ex = (uint64_t)bitrev_u32(k, i) << (log_t - i);
    fe_store<P>(out + e * EU<P>(), pow_from_table<P>(pw, 0, ex, log_t));
}

// ---------------------------------------------------------------------------------------------
// the pass kernel
// ---------------------------------------------------------------------------------------------
PLK_DI uint32_t bitrev(uint32_t x, int bits) { return bits == 0 ? 0u : (__brev(x) >> (32 - bits)); }

// LDS layout: limb-major planes, plane l of the tile at s_mem[l * TILE + e], then the stage twiddles at
// s_mem[NZ * TILE + l * (A/2) + e]: consecutive lanes touch consecutive dwords (conflict-free ds_*_b32).
template <class P> PLK_DI Fz<P> lds_load(const uint32_t* base, int stride, int idx) {
    Fz<P> r;
#pragma unroll
    for (int l = 0; l < FzCfg<P>::NZ; ++l) r.l[l] = base[l * stride + idx];
    return r;
}
template <class P> PLK_DI void lds_store(uint32_t* base, int stride, int idx, const Fz<P>& v) {
#pragma unroll
    for (int l = 0; l < FzCfg<P>::NZ; ++l) base[l * stride + idx] = v.l[l];
}

// The tile itself, with an optional bank skew (tuning bit 3 of PLK_NTT_VARIANT): element i of a plane sits at word i + (i >> 5), so the
// groups of Q lanes that a stage with a small half-size (and the bit-reversed placement of the inputs) put 32 words apart fall on
// different banks instead of the same Q ones.
#ifndef PLK_NTT_VARIANT
#define PLK_NTT_VARIANT 5
#endif
#if PLK_NTT_VARIANT & 8
constexpr int DAT_STRIDE = TILE + TILE / 32;
PLK_DI int dat_index(int i) { return i + (i >> 5); }
#else
constexpr int DAT_STRIDE = TILE;
PLK_DI int dat_index(int i) { return i; }
#endif
template <class P> PLK_DI Fz<P> dat_load(const uint32_t* base, int idx) { return lds_load<P>(base, DAT_STRIDE, dat_index(idx)); }
template <class P> PLK_DI void dat_store(uint32_t* base, int idx, const Fz<P>& v) { lds_store<P>(base, DAT_STRIDE, dat_index(idx), v); }

// b^i from a two-level geometric table (R'-form): hi[i >> 10] * lo[i & 1023], < 1.01p
template <class P> PLK_DI Fz<P> geom_pow(const void* lo, const void* hi, size_t i) {
    const Fz<P> l = fz_from_fe<P>(fe_load<P>((const uint4*)lo + (i & ((1u << NTT_POW_LO_LOG) - 1)) * EU<P>()));
    const Fz<P> h = fz_from_fe<P>(fe_load<P>((const uint4*)hi + (i >> NTT_POW_LO_LOG) * EU<P>()));
    return fz_mul<P>(l, h);
}

// One pass = for every tile: A-point NTTs on Q columns, decimation in time on lazily reduced 29-bit
// limbs (fz.cuh).  DIT is chosen for its bounds: the butterfly (a, b) -> (a + w b, a - w b + 2p) only
// ADDS 2p per stage to a value (the product w b is always < 1.2p), so no reduction is needed inside
// a tile (< (2 log A + 1) p after the last stage, far below R' = 128p); the input goes to the
// bit-reversed LDS slot (free: it is just the store index), the output comes out in natural order.

// where a tile lives: which transform of the batch, and its place in this pass's index space
struct TileGeom {
    size_t b;         // transform of the batch
    size_t blk_base;  // first element of the contiguous sub-problem block (non-last passes)
    size_t r0;        // first column (non-last passes)
    size_t ol0;       // first output-low index (last pass)
};
PLK_DI TileGeom tile_geom(const NttPassArgs& a, size_t t) {
    const size_t tiles_per = ((size_t)1 << a.log_n) >> (a.log_a + a.log_q);
    TileGeom g;
    g.b = t / tiles_per;
    const size_t tile = t % tiles_per;
    g.blk_base = g.r0 = g.ol0 = 0;
    if (!a.last) {
        const size_t rtiles = ((size_t)1 << a.log_s) >> a.log_q;
        g.r0 = (tile % rtiles) << a.log_q;
        g.blk_base = (tile / rtiles) << a.log_nt;
    } else {
        g.ol0 = tile << a.log_q;
    }
    return g;
}
// input index (within the transform) of tile element e = (p, q)
PLK_DI size_t tile_in_index(const NttPassArgs& a, const TileGeom& t, int e) {
    const int q = e & ((1 << a.log_q) - 1), p = e >> a.log_q;
    if (!a.last) return t.blk_base + ((size_t)p << a.log_s) + t.r0 + q;
    // column q of the tile is the block whose output-low index is ol0 + q; blocks are stored
    // with k_1 most significant (kprev), the output wants k_1 least significant
    size_t ol = t.ol0 + q, kprev = 0;
    for (int s = 0; s < a.n_prev; ++s) {
        kprev = (kprev << a.prev_log[s]) | (ol & (((size_t)1 << a.prev_log[s]) - 1));
        ol >>= a.prev_log[s];
    }
    return (kprev << a.log_a) + p;
}
// LDS slot of tile element e on load: row bitrev(p)
PLK_DI int tile_in_slot(const NttPassArgs& a, int e) {
    const int q = e & ((1 << a.log_q) - 1), p = e >> a.log_q;
    return ((int)bitrev((uint32_t)p, a.log_a) << a.log_q) + q;
}
// with a.skip > 0 the loop runs over SLOTS: slot row s takes the input whose slot row is s with the low `skip` bits cleared
// (2^skip slots share one input: the replication the skipped stages would have performed)
PLK_DI int tile_slot_source(const NttPassArgs& a, int e) {
    const int q = e & ((1 << a.log_q) - 1), s = e >> a.log_q;
    const int p = (int)bitrev((uint32_t)(s & ~((1 << a.skip) - 1)), a.log_a);
    return (p << a.log_q) + q;
}
// output index (within the transform) of tile element e = (k, q) on store
PLK_DI size_t tile_out_index(const NttPassArgs& a, const TileGeom& t, int e) {
    const int q = e & ((1 << a.log_q) - 1), k = e >> a.log_q;
    if (!a.last) return t.blk_base + ((size_t)k << a.log_s) + t.r0 + q;
    return (t.ol0 + q) + ((size_t)k << (a.log_n - a.log_a));
}
// index into the outer twiddle table of a non-last pass
PLK_DI size_t tile_tw_index(const NttPassArgs& a, const TileGeom& t, int e) {
    const int q = e & ((1 << a.log_q) - 1), k = e >> a.log_q;
    return ((size_t)k << a.log_s) + t.r0 + q;
}

// the stages of a tile in LDS: h = 1, 2, .., A/2, two at a time (radix-4 in registers: half the LDS round
// trips and barriers of a radix-2 sweep, same multiplications).  Ends with a barrier.
// `first_stage` > 0: the tile starts at that stage (zero-padded input: when only the rows p < A / 2^k of a column are non-zero,
// the bit-reversed placement puts them at every 2^k-th slot and the first k butterfly stages pair every value with a zero:
// (a, 0) -> (a, a).  They are pure replication, done by the loads; polynomials_to_values_padded has k = 3).
// one radix-4 step in registers.  x[0..3]: the elements at rows i0, i0 + st, i0 + 2 st, i0 + 3 st; on return the outputs for
// the same rows.  FIRST: the step with half-size h = 1 (unit twiddles w_2^0, w_4^0: one multiplication, exactly normalised inputs).
template <class P, bool FIRST>
PLK_DI void radix4_step(Fz<P> (&x)[4], const Fz<P>& wa, const Fz<P>& wb0, const Fz<P>& wb1) {
    if constexpr (FzCfg<P>::NZ > 10) {
        // Bls12377Base (14 limbs): the column sums of a product have no room for uncarried operands (FzLazyBound: 3.57 * 2^29), so
        // every sum and difference moves its carries and LDS holds limbs below 2^29 + 8.  Values grow by 2p per stage as below,
        // far under R' / 8 = 2^403.  No caller of the reference transforms over this field (it is a `Field`, so fft::<Bls12377Base>
        // type-checks, bls12_377_base.rs:18-262); the path is kept simple rather than tuned.
        if constexpr (!FIRST) {
            x[1] = fz_mul<P>(x[1], wa);
            x[3] = fz_mul<P>(x[3], wa);
        }
        const Fz<P> y0 = fz_add<P>(x[0], x[1]), y1 = fz_sub<P, 1>(x[0], x[1]);
        Fz<P> y2 = fz_add<P>(x[2], x[3]), y3 = fz_sub<P, 1>(x[2], x[3]);
        y3 = fz_mul<P>(y3, wb1);
        if constexpr (!FIRST) y2 = fz_mul<P>(y2, wb0);
        x[1] = fz_add<P>(y1, y3);
        x[3] = fz_sub<P, 1>(y1, y3);
        x[0] = fz_add<P>(y0, y2);
        // FIRST: y2 = x2 + x3 is a sum of two canonical values, below 2p but not below p: it is taken off 4p
        if constexpr (FIRST) x[2] = fz_sub<P, 2>(y0, y2);
        else x[2] = fz_sub<P, 1>(y0, y2);
        return;
    }
    if constexpr (!FIRST) {
        // Carries are moved twice per step instead of eight times (fz_add_nc / fz_sub_nc): LDS holds limbs up to
        // MUL_LIMB_MAX = 2.5 * 2^30 + 16; x1 and x3 go straight into a multiplication by a table entry; x0 and x2 are
        // carried (limbs < 2^29 + 8) and every sum below stays within the bound:
        //   y0, y2 <= 2^29 + 8 + 2^29;   y1, y3 <= 2^29 + 8 + 2^30 (borrow 2^29 + limb of 2p);   outputs <= y + 2^30.
        // stage with half-size h: pairs (x0, x1), (x2, x3), twiddle w_{2h}^j for both
        fz_carry<P>(x[0]);
        fz_carry<P>(x[2]);
        x[1] = fz_mul<P>(x[1], wa);
        x[3] = fz_mul<P>(x[3], wa);
        const Fz<P> y0 = fz_add_nc<P>(x[0], x[1]), y1 = fz_sub_nc<P, 1, 29>(x[0], x[1]);
        Fz<P> y2 = fz_add_nc<P>(x[2], x[3]), y3 = fz_sub_nc<P, 1, 29>(x[2], x[3]);
        // stage with half-size 2h: pairs (y0, y2) with w_{4h}^j and (y1, y3) with w_{4h}^(j+h)
        y3 = fz_mul<P>(y3, wb1);
        x[1] = fz_add_nc<P>(y1, y3);
        x[3] = fz_sub_nc<P, 1, 29>(y1, y3);
        y2 = fz_mul<P>(y2, wb0);
        x[0] = fz_add_nc<P>(y0, y2);
        x[2] = fz_sub_nc<P, 1, 29>(y0, y2);
    } else {
        // first step of a tile: the inputs are exactly normalised (canonical, or products below 1.01p), the
        // twiddles w_2^0 and w_4^0 are 1: one multiplication; y2 = x2 + x3 is below 2.1p with limbs <= 2^30 - 2
        const Fz<P> y0 = fz_add_nc<P>(x[0], x[1]), y1 = fz_sub_nc<P, 1, 29>(x[0], x[1]);
        const Fz<P> y2 = fz_add_nc<P>(x[2], x[3]);
        Fz<P> y3 = fz_sub_nc<P, 1, 29>(x[2], x[3]);
        y3 = fz_mul<P>(y3, wb1);
        x[1] = fz_add_nc<P>(y1, y3);           // <= 2^29 + 2^30 + 2^29
        x[3] = fz_sub_nc<P, 1, 29>(y1, y3);    // <= 2^29 + 2^30 + 2^30
        x[0] = fz_add_nc<P>(y0, y2);           // <= 2^31
        x[2] = fz_sub_nc<P, 2, 30>(y0, y2);    // <= 2^30 + 2^30 + 2^29
    }
}
// Exchange between the four lanes {l, l + d, l + 2d, l + 3d} of a wave (t = the lane's place among them): on entry lane t holds
// o[0..3], on return x[k] = (lane k's) o[t] - a 4 x 4 transposition in two butterfly rounds of wave shuffles, two elements
// each way per round.  This is what carries the outputs of the first radix-4 step of a tile to the lanes of the second one
// WITHOUT the LDS round trip and its barrier ("wavefront butterfly shuffles").
template <class P> PLK_DI void quad_transpose(Fz<P> (&v)[4], int t, int d) {
    constexpr int NZ = FzCfg<P>::NZ;
    const bool b0 = (t & 1) != 0, b1 = (t & 2) != 0;
#pragma unroll
    for (int l = 0; l < NZ; ++l) {
        // round A (distance d): keep one of each pair (v0, v1), (v2, v3), trade the other
        const uint32_t s0 = b0 ? v[0].l[l] : v[1].l[l], s1 = b0 ? v[2].l[l] : v[3].l[l];
        const uint32_t r0 = (uint32_t)__shfl_xor((int)s0, d), r1 = (uint32_t)__shfl_xor((int)s1, d);
        const uint32_t a0 = b0 ? r0 : v[0].l[l], a1 = b0 ? v[1].l[l] : r0, a2 = b0 ? r1 : v[2].l[l], a3 = b0 ? v[3].l[l] : r1;
        // round B (distance 2d): pairs (a0, a2), (a1, a3)
        const uint32_t u0 = b1 ? a0 : a2, u1 = b1 ? a1 : a3;
        const uint32_t q0 = (uint32_t)__shfl_xor((int)u0, 2 * d), q1 = (uint32_t)__shfl_xor((int)u1, 2 * d);
        v[0].l[l] = b1 ? q0 : a0;
        v[1].l[l] = b1 ? q1 : a1;
        v[2].l[l] = b1 ? a2 : q0;
        v[3].l[l] = b1 ? a3 : q1;
    }
}

template <class P>
PLK_DI void tile_stages(uint32_t* s_dat, const uint32_t* s_tw, int tid, int log_a, int log_q, int tile_elems, int first_stage, const uint4* __restrict__ g_tw,
                        bool tw_global, bool shuffle) {
    const int Q = 1 << log_q, half_a = (1 << log_a) >> 1;
    // w_A^e, e < A / 2
    auto stage_tw = [&](int e) -> Fz<P> {
        if (TILE_LOG >= 12 && tw_global)  // tuning builds only (tools/gpu/ntt_variants.sh): a 1024-element tile always has room
            return fz_from_fe<P>(fe_load<P>(g_tw + ((size_t)e << (INNER_LOG - log_a)) * EU<P>()));
        return lds_load<P>(s_tw, half_a, e);
    };
    int log_h = first_stage;
    // The first FOUR stages without an LDS round trip in the middle: step h = 1 in registers, its outputs handed to the lanes
    // of step h = 4 by wave shuffles (the four lanes of a group are Q apart), step h = 4 in registers, then LDS.
    if (shuffle && log_h == 0 && log_a >= 4 && (tile_elems >> 2) == NTT_THREADS && (4 << log_q) <= 64) {
        const int q = tid & (Q - 1), pq = tid >> log_q;
        Fz<P> x[4];
        {
            const int i0 = ((pq << 2) << log_q) + q, st = Q;
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = dat_load<P>(s_dat, i0 + k * st);
            const Fz<P> wb1 = stage_tw(1 << (log_a - 2));
            radix4_step<P, true>(x, wb1, wb1, wb1);
        }
        quad_transpose<P>(x, pq & 3, Q);
        {
            const int h = 4, j = pq & 3, blk = pq >> 2;
            const int i0 = (((blk << 4) + j) << log_q) + q, st = h << log_q;
            const Fz<P> wa = stage_tw(j << (log_a - 3)), wb1 = stage_tw((j + h) << (log_a - 4)), wb0 = stage_tw(j << (log_a - 4));
            radix4_step<P, false>(x, wa, wb0, wb1);
            // every lane of the workgroup has read its step-1 inputs before any of these stores can land: the exchange above
            // is wave-wide, but lanes of OTHER waves may still be loading - they load rows 4 pq .. 4 pq + 3 of their own pq only,
            // and the rows written here, 16 blk + j + 4 k, belong to the four pq of THIS lane group: no overlap across waves
#pragma unroll
            for (int k = 0; k < 4; ++k) dat_store<P>(s_dat, i0 + k * st, x[k]);
        }
        __syncthreads();
        log_h = 4;
    }
    for (; log_h + 1 < log_a; log_h += 2) {
        const int h = 1 << log_h;
        for (int qd = tid; qd < (tile_elems >> 2); qd += NTT_THREADS) {
            const int q = qd & (Q - 1), pq = qd >> log_q;
            const int j = pq & (h - 1), blk = pq >> log_h;
            const int i0 = (((blk << (log_h + 2)) + j) << log_q) + q, st = h << log_q;
            Fz<P> x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = dat_load<P>(s_dat, i0 + k * st);
            if (log_h > 0) {
                const Fz<P> wa = stage_tw(j << (log_a - 1 - log_h));
                const Fz<P> wb1 = stage_tw((j + h) << (log_a - 2 - log_h));
                const Fz<P> wb0 = stage_tw(j << (log_a - 2 - log_h));
                radix4_step<P, false>(x, wa, wb0, wb1);
            } else {
                const Fz<P> wb1 = stage_tw((j + h) << (log_a - 2 - log_h));
                radix4_step<P, true>(x, wb1, wb1, wb1);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) dat_store<P>(s_dat, i0 + k * st, x[k]);
        }
        __syncthreads();
    }
    if (log_h < log_a) {  // odd log A: one radix-2 stage left
        const int h = 1 << log_h;
        for (int bf = tid; bf < (tile_elems >> 1); bf += NTT_THREADS) {
            const int q = bf & (Q - 1), pb = bf >> log_q;
            const int j = pb & (h - 1), blk = pb >> log_h;
            const int i0 = (((blk << (log_h + 1)) + j) << log_q) + q, i1 = i0 + (h << log_q);
            Fz<P> x = dat_load<P>(s_dat, i0);
            Fz<P> t = dat_load<P>(s_dat, i1);
            if (log_h > 0) {
                fz_carry<P>(x);
                t = fz_mul<P>(t, stage_tw(j << (log_a - 1 - log_h)));  // w_{2h}^j, product < 1.2p
            }
            // log_h == 0: a two-point transform of exactly normalised inputs
            if constexpr (FzCfg<P>::NZ > 10) {  // 14 limbs: carried sums (see radix4_step)
                dat_store<P>(s_dat, i0, fz_add<P>(x, t));
                dat_store<P>(s_dat, i1, fz_sub<P, 1>(x, t));
            } else {
                dat_store<P>(s_dat, i0, fz_add_nc<P>(x, t));
                dat_store<P>(s_dat, i1, fz_sub_nc<P, 1, 29>(x, t));  // t < 2p - margin in both cases
            }
        }
        __syncthreads();
    }
}

// element from global memory into its working form; hooks of the first pass: zero padding, x_i *= b^i
template <class P, bool HOOKS> PLK_DI Fz<P> tile_ingest(const NttPassArgs& a, const NttHooks& hk, const Fe<P>& v, size_t g) {
    Fz<P> x = fz_from_fe<P>(v);
    if constexpr (HOOKS) {
        if (a.first) {
            if (g >= hk.in_len) return fz_zero<P>();
            if (hk.in_lo) x = fz_mul<P>(x, geom_pow<P>(hk.in_lo, hk.in_hi, g));
        }
    }
    return x;
}
// what every output gets on its way out: the outer twiddle of a non-last pass (tw), or the factor 1 / n^-1 and the hooks of
// the last pass.  Returns a value below 2p with exactly normalised limbs.
template <class P, bool HOOKS>
PLK_DI Fz<P> tile_emit(const NttPassArgs& a, const NttHooks& hk, Fz<P> v, const Fz<P>& tw, const Fz<P>& scale, size_t g) {
    if (!a.last) return fz_mul<P>(v, tw);
    bool have = a.scale != 0;
    Fz<P> mult = scale;  // n^-1 (R'-form) when a.scale
    if constexpr (HOOKS) {
        // g is the natural output index; every factor is an R'-form value below 2p
        if (hk.out_tab) {
            const Fz<P> t = fz_from_fe<P>(fe_load<P>((const uint4*)hk.out_tab + (g & hk.out_mask) * EU<P>()));
            mult = have ? fz_mul<P>(t, mult) : t;
            have = true;
        }
        if (hk.out_lo) {
            const Fz<P> pw = geom_pow<P>(hk.out_lo, hk.out_hi, g);
            mult = have ? fz_mul<P>(mult, pw) : pw;
            have = true;
        }
    }
    if (have) return fz_mul<P>(v, mult);
    // fz_reduce_small estimates the quotient from the modulus' top 26 bits above limb NZ - 2; Bls12377Base (377 bits in 14 limbs) leaves
    // three there, so its outputs are reduced by a multiplication by one (R'-form) instead
    if constexpr (FzCfg<P>::NZ > 10) return fz_mul<P>(v, fz_one_rprime<P>());
    else return fz_reduce_small<P>(v);
}

// One tile per workgroup; any tile shape (transforms shorter than a tile included).  IN_LIMBS / OUT_LIMBS: the pass reads /
// writes the library's own scratch buffer in limb form (every pass but the first / the last); the caller's buffers are the
// reference's 32-byte elements.  `in` and `out` may be the same buffer (a tile reads all of its elements before it writes).
// tuning builds (tools/gpu/r05_ntt_variants.sh): bit 0 the loads of a tile requested together, bit 1 the inter-pass twiddles likewise,
// bit 2 the register allocation held to four waves per SIMD.  Measured on one lease (profiles/r05_ntt_variants.txt): bit 0 is worth 1-2.5 %
// on multi-round launches (nine 2^20 transforms 98.0 -> 96.4 us each, 2^23 971 -> 948 us) and costs 1.7-2.3 us on a lone 2^18 / 2^19
// transform; bit 1 gains nothing anywhere and costs a lone 2^18 transform 4 us - the memory latency of a tile is NOT what a pass waits for.
#ifndef PLK_NTT_VARIANT
#define PLK_NTT_VARIANT 5
#endif
#if PLK_NTT_VARIANT & 4
// (the 14 limbs of Bls12377Base: a 56 KiB tile, two workgroups per CU at most - the registers of two waves per SIMD are theirs)
#define PLK_NTT_BOUNDS __launch_bounds__(NTT_THREADS, FzCfg<P>::NZ > 10 ? 2 : 4)
#else
#define PLK_NTT_BOUNDS __launch_bounds__(NTT_THREADS)
#endif
template <class P, bool HOOKS, bool IN_LIMBS, bool OUT_LIMBS>
__global__ void PLK_NTT_BOUNDS k_ntt_pass(const void* in, void* out, const uint4* __restrict__ inner_tw,
                                                          const uint32_t* __restrict__ outer_tw, const uint4* __restrict__ scale_ptr, NttPassArgs a,
                                                          NttHooks hk) {
    constexpr int NZ = FzCfg<P>::NZ;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    const int tid = threadIdx.x;
    const int log_a = a.log_a, log_q = a.log_q;
    const int tile_elems = (1 << log_a) << log_q;
    const int half_a = (1 << log_a) >> 1;
    const size_t n = (size_t)1 << a.log_n;
    uint32_t* s_dat = s_mem;
    uint32_t* s_tw = s_mem + NZ * DAT_STRIDE;

    // A pass whose workgroups all fit the GPU at once (a single 2^20 transform: 1024 tiles, four per CU) would run in lockstep - every
    // workgroup loads, then every workgroup computes, then every workgroup stores: ~20 us of memory phases beside ~21 us of issue
    // (SQ_WAVE_CYCLES, round 4: 39.5 % of the wave cycles parked).  The k-th workgroup of a CU therefore starts k phases late: the first
    // ones get the memory system to themselves and compute while the others load; in a multi-round launch (batches, 2^23) the rounds
    // stagger by themselves and a.stagger is 0.
    if (a.stagger > 0) {
        const int ph = (int)(blockIdx.x / (unsigned)a.stagger_div);
        for (int i = 0; i < ph * a.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }
    const TileGeom tg = tile_geom(a, blockIdx.x);
    // The loads of a thread's EPT = 4 elements are ISSUED before any of them is waited for (round 5; until round 4: load, wait, LDS store,
    // four round trips behind each other).
    constexpr int EPT = TILE / NTT_THREADS;
    const bool by_slot = HOOKS && !IN_LIMBS && a.skip > 0;
    Fe<P> scale_raw = fe_zero<P>();  // the last pass's factor (1 or n^-1): requested with the tile, used when it is written out
    if (a.last) scale_raw = fe_load<P>(scale_ptr);
#if PLK_NTT_VARIANT & 1
    size_t gin[EPT];
    Fz<P> xin[EPT];
    Fe<P> vin[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        // a thread past the end of a short tile reads the tile's last element again (and drops it): no branch around the loads
        const int e = min(tid + k * NTT_THREADS, tile_elems - 1);
        gin[k] = tile_in_index(a, tg, by_slot ? tile_slot_source(a, e) : e);
        if constexpr (IN_LIMBS) {
            xin[k] = limbs_load<P>((const uint32_t*)in, tg.b * n + gin[k]);
        } else {
            const uint4* inb = (const uint4*)in + tg.b * ((HOOKS && a.first) ? hk.in_stride : n) * EU<P>();
            if constexpr (HOOKS) {
                vin[k] = fe_zero<P>();
                if (!a.first || gin[k] < hk.in_len) vin[k] = fe_load<P>(inb + gin[k] * EU<P>());
            } else {
                vin[k] = fe_load<P>(inb + gin[k] * EU<P>());
            }
        }
    }
    // stage twiddles (R'-form): w_A^e = inner[e * (1024 / A)], e < A/2
    for (int e = tid; e < ((TILE_LOG >= 12 && a.tw_global) ? 0 : half_a); e += NTT_THREADS) {
        const Fe<P> w = fe_load<P>(inner_tw + ((size_t)e << (INNER_LOG - log_a)) * EU<P>());
        lds_store<P>(s_tw, half_a, e, fz_from_fe<P>(w));
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * NTT_THREADS;
        if (e < tile_elems) {
            Fz<P> x;
            if constexpr (IN_LIMBS) x = xin[k];
            else x = tile_ingest<P, HOOKS>(a, hk, vin[k], gin[k]);
            dat_store<P>(s_dat, by_slot ? e : tile_in_slot(a, e), x);
        }
    }
#else
    // stage twiddles (R'-form): w_A^e = inner[e * (1024 / A)], e < A/2
    for (int e = tid; e < ((TILE_LOG >= 12 && a.tw_global) ? 0 : half_a); e += NTT_THREADS) {
        const Fe<P> w = fe_load<P>(inner_tw + ((size_t)e << (INNER_LOG - log_a)) * EU<P>());
        lds_store<P>(s_tw, half_a, e, fz_from_fe<P>(w));
    }
    for (int e = tid; e < tile_elems; e += NTT_THREADS) {
        const size_t g = tile_in_index(a, tg, by_slot ? tile_slot_source(a, e) : e);
        Fz<P> x;
        if constexpr (IN_LIMBS) {
            x = limbs_load<P>((const uint32_t*)in, tg.b * n + g);
        } else {
            const uint4* inb = (const uint4*)in + tg.b * ((HOOKS && a.first) ? hk.in_stride : n) * EU<P>();
            Fe<P> v = fe_zero<P>();
            if (!HOOKS || !a.first || g < hk.in_len) v = fe_load<P>(inb + g * EU<P>());
            x = tile_ingest<P, HOOKS>(a, hk, v, g);
        }
        dat_store<P>(s_dat, by_slot ? e : tile_in_slot(a, e), x);
    }
#endif
    __syncthreads();
    tile_stages<P>(s_dat, s_tw, tid, log_a, log_q, tile_elems, (HOOKS && !IN_LIMBS) ? a.skip : 0, inner_tw, a.tw_global != 0, a.shuffle != 0);
    const Fz<P> scale = fz_from_fe<P>(scale_raw);
#if PLK_NTT_VARIANT & 2
    // the inter-pass twiddles of this thread's four outputs: requested together, one round trip
    Fz<P> otw[EPT];
    if (!a.last) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) otw[k] = limbs_load<P>(outer_tw, tile_tw_index(a, tg, min(tid + k * NTT_THREADS, tile_elems - 1)));
    } else {
#pragma unroll
        for (int k = 0; k < EPT; ++k) otw[k] = fz_zero<P>();
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * NTT_THREADS;
        if (e < tile_elems) {
            const size_t g = tile_out_index(a, tg, e);
            const Fz<P> r = tile_emit<P, HOOKS>(a, hk, dat_load<P>(s_dat, e), otw[k], scale, g);
            if constexpr (OUT_LIMBS) limbs_store<P>((uint32_t*)out, tg.b * n + g, r);
            else fe_store<P>((uint4*)out + (tg.b * n + g) * EU<P>(), fz_to_fe_canonical<P>(r));
        }
    }
#else
    for (int e = tid; e < tile_elems; e += NTT_THREADS) {
        const size_t g = tile_out_index(a, tg, e);
        Fz<P> tw = fz_zero<P>();
        if (!a.last) tw = limbs_load<P>(outer_tw, tile_tw_index(a, tg, e));
        const Fz<P> r = tile_emit<P, HOOKS>(a, hk, dat_load<P>(s_dat, e), tw, scale, g);
        if constexpr (OUT_LIMBS) limbs_store<P>((uint32_t*)out, tg.b * n + g, r);
        else fe_store<P>((uint4*)out + (tg.b * n + g) * EU<P>(), fz_to_fe_canonical<P>(r));
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// plans + cache
// ---------------------------------------------------------------------------------------------
struct NttPlan {
    int field = 0, log_n = 0, device = 0;
    std::vector<int> pass_log;            // a_1 .. a_m
    void* pw = nullptr;                   // 65 elements (see k_ntt_pow2)
    void* inner[2] = {nullptr, nullptr};  // forward / inverse stage twiddles
    std::vector<void*> outer[2];          // per non-last pass
    ~NttPlan() {
        if (pw) (void)hipFree(pw);
        for (int d = 0; d < 2; ++d) {
            if (inner[d]) (void)hipFree(inner[d]);
            for (void* p : outer[d])
                if (p) (void)hipFree(p);
        }
    }
};

static std::vector<int> plan_passes(int log_n) {
    std::vector<int> v;
    // debugging / tuning override: PLK_NTT_PLAN="7,7,6" (must sum to log_n, else ignored)
    if (const char* e = getenv("PLK_NTT_PLAN")) {
        int sum = 0;
        for (const char* p = e; *p;) {
            int x = atoi(p);
            if (x <= 0 || x > TILE_LOG) { v.clear(); sum = -1; break; }
            v.push_back(x);
            sum += x;
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
        if (sum == log_n && (int)v.size() <= MAX_PASSES) return v;
        v.clear();
    }
    if (log_n <= TILE_LOG) {
        v.push_back(log_n);
        return v;
    }
    const int MAX_A = 7;  // strided passes: Q = 1024 / A >= 8 columns = 256-byte runs
    int m = (log_n + MAX_A - 1) / MAX_A;
    int base = log_n / m, extra = log_n % m;
    for (int t = 0; t < m; ++t) v.push_back(base + (t < extra ? 1 : 0));
    return v;
}

// optional per-launch timing of the pass kernel (HIP events on the launch stream), for bench.py's roofline
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;  // recorded, not yet read
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_free;

static bool prof_begin(hipStream_t stream, std::pair<hipEvent_t, hipEvent_t>& ev) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on) return false;
    if (!g_prof_free.empty()) {
        ev = g_prof_free.back();
        g_prof_free.pop_back();
    } else {
        if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) return false;
    }
    (void)hipEventRecord(ev.first, stream);
    return true;
}
static void prof_end(hipStream_t stream, const std::pair<hipEvent_t, hipEvent_t>& ev) {
    (void)hipEventRecord(ev.second, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_events.push_back(ev);
}

int ntt_set_profiling_impl(int enable) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = enable != 0;
    return PLK_OK;
}
int ntt_get_timings_impl(double* sum_ms, unsigned* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double tot = 0;
    unsigned cnt = 0;
    for (auto& ev : g_prof_events) {
        float ms = 0;
        if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
            tot += ms;
            ++cnt;
        }
        g_prof_free.push_back(ev);
    }
    g_prof_events.clear();
    if (sum_ms) *sum_ms = tot;
    if (launches) *launches = cnt;
    return PLK_OK;
}

static std::mutex g_plan_mu;
static std::map<std::tuple<int, int, int>, std::shared_ptr<NttPlan>> g_plans;

template <class P> static int build_plan_t(NttPlan& pl) {
    const int log_n = pl.log_n;
    const int log_t = log_n > INNER_LOG ? log_n : INNER_LOG;
    if (log_t > P::TWO_ADICITY) return set_error(PLK_ERR_TWO_ADICITY, "log_n %d exceeds the field's 2-adicity %d", log_n, P::TWO_ADICITY);
    PLK_HIP_TRY(hipMalloc(&pl.pw, 67 * (size_t)P::NL * 4));
    k_ntt_pow2<P><<<1, 64>>>((uint4*)pl.pw, log_t, log_n);
    PLK_HIP_TRY(hipGetLastError());
    const int m = (int)pl.pass_log.size();
    for (int dir = 0; dir < 2; ++dir) {
        PLK_HIP_TRY(hipMalloc(&pl.inner[dir], (size_t)(1 << (INNER_LOG - 1)) * P::NL * 4));
        k_ntt_fill_inner<P><<<((1 << (INNER_LOG - 1)) + 255) / 256, 256>>>((uint4*)pl.inner[dir], (const uint4*)pl.pw, log_t, dir);
        PLK_HIP_TRY(hipGetLastError());
        int log_nt = log_n;
        for (int t = 0; t + 1 < m; ++t) {
            const int log_s = log_nt - pl.pass_log[t];
            void* tab = nullptr;
            PLK_HIP_TRY(hipMalloc(&tab, limb_bytes((size_t)1 << log_nt, FzCfg<P>::NZ)));
            pl.outer[dir].push_back(tab);
            const size_t cnt = (size_t)1 << log_nt;
            k_ntt_fill_outer<P><<<(unsigned)((cnt + 255) / 256), 256>>>((uint32_t*)tab, (const uint4*)pl.pw, log_t, log_nt, log_s, dir,
                                                                        (dir == 1 && t == 0) ? 1 : 0);
            PLK_HIP_TRY(hipGetLastError());
            log_nt = log_s;
        }
    }
    PLK_HIP_TRY(hipDeviceSynchronize());
    return PLK_OK;
}

static int get_plan(int field, unsigned log_n, std::shared_ptr<NttPlan>& out) {
    PLK_TRY(ensure_device());
    int dev = 0;
    PLK_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto key = std::make_tuple(dev, field, (int)log_n);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) {
        out = it->second;
        return PLK_OK;
    }
    auto pl = std::make_shared<NttPlan>();
    pl->field = field;
    pl->log_n = (int)log_n;
    pl->device = dev;
    pl->pass_log = plan_passes((int)log_n);
    int rc;
    switch (field) {
        case PLK_FIELD_TWEEDLEDEE_BASE: rc = build_plan_t<TweedledeeBaseParams>(*pl); break;
        case PLK_FIELD_TWEEDLEDUM_BASE: rc = build_plan_t<TweedledumBaseParams>(*pl); break;
        case PLK_FIELD_BLS12_377_SCALAR: rc = build_plan_t<Bls12377ScalarParams>(*pl); break;
        case PLK_FIELD_BLS12_377_BASE: rc = build_plan_t<Bls12377BaseParams>(*pl); break;
        case PLK_FIELD_PALLAS_BASE: rc = build_plan_t<PallasBaseParams>(*pl); break;
        case PLK_FIELD_VESTA_BASE: rc = build_plan_t<VestaBaseParams>(*pl); break;
        default: return set_error(PLK_ERR_INVALID_ARG, "field %d has no NTT entry point", field);
    }
    if (rc != PLK_OK) return rc;
    g_plans[key] = pl;
    out = pl;
    return PLK_OK;
}

int ntt_precompute_impl(int field, unsigned log_n) {
    if (log_n > 30) return set_error(PLK_ERR_TWO_ADICITY, "log_n %u too large (max 30)", log_n);
    std::shared_ptr<NttPlan> pl;
    return get_plan(field, log_n, pl);
}

int ntt_clear_cache_impl() {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    g_plans.clear();
    return PLK_OK;
}

template <class P>
static int run_plan_t(const NttPlan& pl, int inverse, unsigned batch, const void* d_in, void* d_out, const NttHooks* hooks, hipStream_t stream) {
    const int log_n = pl.log_n;
    const int m = (int)pl.pass_log.size();
    const int dir = inverse ? 1 : 0;
    if (log_n == 0 && !hooks) {
        if (d_in != d_out) PLK_HIP_TRY(hipMemcpyAsync(d_out, d_in, (size_t)batch * P::NL * 4, hipMemcpyDeviceToDevice, stream));
        return PLK_OK;
    }
    // The last pass is a transposition (reads contiguous blocks, writes digit-reversed), so with
    // m >= 2 passes it cannot run in place: passes 1..m-1 work in a library-owned scratch
    // buffer and the last pass writes the caller's output.
    void* scratch = nullptr;
    if (m >= 2) {
        scratch = scratch_acquire(limb_bytes((size_t)batch << log_n, FzCfg<P>::NZ), stream);
        if (!scratch) return PLK_ERR_OOM;
    }
    int log_nt = log_n;
    const void* src = d_in;
    int rc = PLK_OK;
    for (int t = 0; t < m; ++t) {
        NttPassArgs a{};
        a.log_n = log_n;
        a.log_a = pl.pass_log[t];
        a.log_nt = log_nt;
        a.log_s = log_nt - a.log_a;
        a.first = (t == 0) ? 1 : 0;
        a.last = (t == m - 1) ? 1 : 0;
        a.log_q = TILE_LOG - a.log_a;
        if (a.last) {
            const int log_blocks = log_n - a.log_a;  // number of contiguous blocks
            if (a.log_q > log_blocks) a.log_q = log_blocks;
            a.n_prev = m - 1;
            for (int s = 0; s < m - 1; ++s) a.prev_log[s] = pl.pass_log[s];
        } else if (a.log_q > a.log_s) {
            a.log_q = a.log_s;
        }
        a.scale = (inverse && m == 1) ? 1 : 0;
        a.skip = 0;
        if (hooks && a.first && hooks->in_len > 0 && !hooks->in_lo) {
            // zero padding by 2^k: the stored coefficients end before n / 2^k
            int k = 0;
            while (k < a.log_a && hooks->in_len <= (((size_t)1 << log_n) >> (k + 1))) ++k;
            a.skip = k;
        }
        const size_t tiles = (((size_t)1 << log_n) >> (a.log_a + a.log_q)) * batch;
        const void* outer = a.last ? nullptr : pl.outer[dir][t];
        void* dst = a.last ? d_out : scratch;
        std::pair<hipEvent_t, hipEvent_t> pev;
        const bool prof = prof_begin(stream, pev);
        size_t lds_bytes = ((size_t)FzCfg<P>::NZ * DAT_STRIDE + (size_t)FzCfg<P>::NZ * ((size_t)1 << a.log_a) / 2) * 4;
        static const int shuffle_mode = getenv("PLK_NTT_SHUFFLE") ? atoi(getenv("PLK_NTT_SHUFFLE")) : 0;
        a.shuffle = shuffle_mode;
        a.tw_global = 0;
        if (lds_bytes > NTT_LDS_MAX) {
            a.tw_global = 1;
            lds_bytes = (size_t)FzCfg<P>::NZ * DAT_STRIDE * 4;
        }
        // one round of workgroups (at most four 39 KB tiles per CU) and more than one per CU: start them in phases
        {
            static const int cus = [] {
                int dev = 0, c = 256;
                (void)hipGetDevice(&dev);
                (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
                return c > 0 ? c : 256;
            }();
            const char* se = getenv("PLK_NTT_STAGGER");  // sleep quanta (~0.5 us) per phase; tuning
            const int quanta = se ? atoi(se) : NTT_STAGGER_DEFAULT;
            a.stagger = (quanta > 0 && tiles > (size_t)cus && tiles <= (size_t)4 * cus) ? quanta : 0;
            a.stagger_div = cus;
        }
        const uint4* scale = (const uint4*)pl.pw + (a.scale ? 66 : 65) * EU<P>();
        const bool use_hooks = hooks && (a.first || a.last);
        const NttHooks hk = use_hooks ? *hooks : NttHooks{};
        const unsigned tl = (unsigned)tiles;
        const uint32_t* otw = (const uint32_t*)outer;
        // the first pass reads the caller's elements, the last one writes them; everything in between is limb form in scratch
        // (more than 64 KiB of dynamic LDS - a one-pass transform over the 14 limbs of Bls12377Base: 56 KiB tile + up to 28 KiB of stage
        // twiddles - has to be asked for per kernel)
#define PLK_NTT_LAUNCH(H, IL, OL)                                                                                                             \
    do {                                                                                                                                      \
        if (lds_bytes > 64 * 1024)                                                                                                            \
            (void)hipFuncSetAttribute((const void*)k_ntt_pass<P, H, IL, OL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);     \
        k_ntt_pass<P, H, IL, OL><<<tl, NTT_THREADS, lds_bytes, stream>>>(src, dst, (const uint4*)pl.inner[dir], otw, scale, a, hk);           \
    } while (0)
        if (a.first && a.last) {
            if (use_hooks) PLK_NTT_LAUNCH(true, false, false); else PLK_NTT_LAUNCH(false, false, false);
        } else if (a.first) {
            if (use_hooks) PLK_NTT_LAUNCH(true, false, true); else PLK_NTT_LAUNCH(false, false, true);
        } else if (a.last) {
            if (use_hooks) PLK_NTT_LAUNCH(true, true, false); else PLK_NTT_LAUNCH(false, true, false);
        } else {
            PLK_NTT_LAUNCH(false, true, true);
        }
#undef PLK_NTT_LAUNCH
        if (prof) prof_end(stream, pev);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            rc = set_error(PLK_ERR_HIP, "ntt pass launch failed: %s", hipGetErrorString(e));
            break;
        }
        src = dst;
        log_nt = a.log_s;
    }
    if (scratch) scratch_release(scratch, stream);
    return rc;
}

static int ntt_dispatch(int field, unsigned log_n, int inverse, unsigned batch, const void* d_in, void* d_out, const NttHooks* hooks,
                        hipStream_t stream) {
    if (!d_in || !d_out) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    if (batch == 0) return PLK_OK;
    if (log_n > 30) return set_error(PLK_ERR_TWO_ADICITY, "log_n %u too large (max 30)", log_n);
    std::shared_ptr<NttPlan> pl;
    PLK_TRY(get_plan(field, log_n, pl));
    switch (field) {
        case PLK_FIELD_TWEEDLEDEE_BASE: return run_plan_t<TweedledeeBaseParams>(*pl, inverse, batch, d_in, d_out, hooks, stream);
        case PLK_FIELD_TWEEDLEDUM_BASE: return run_plan_t<TweedledumBaseParams>(*pl, inverse, batch, d_in, d_out, hooks, stream);
        case PLK_FIELD_BLS12_377_SCALAR: return run_plan_t<Bls12377ScalarParams>(*pl, inverse, batch, d_in, d_out, hooks, stream);
        case PLK_FIELD_BLS12_377_BASE:
            // only the plain transform and its zero-padded form (fft_with_precomputation, fft.rs:61-80); the polynomial callers' hooks
            // (coset factors, denominators) belong to the circuit's scalar fields
            if (hooks && (hooks->in_lo || hooks->out_tab || hooks->out_lo)) return set_error(PLK_ERR_INVALID_ARG, "field %d has no polynomial entry points", field);
            return run_plan_t<Bls12377BaseParams>(*pl, inverse, batch, d_in, d_out, hooks, stream);
        case PLK_FIELD_PALLAS_BASE: return run_plan_t<PallasBaseParams>(*pl, inverse, batch, d_in, d_out, hooks, stream);
        case PLK_FIELD_VESTA_BASE: return run_plan_t<VestaBaseParams>(*pl, inverse, batch, d_in, d_out, hooks, stream);
    }
    return set_error(PLK_ERR_INVALID_ARG, "field %d has no NTT entry point", field);
}

int ntt_dev_impl(int field, unsigned log_n, int inverse, unsigned batch, const void* d_in, void* d_out, hipStream_t stream) {
    return ntt_dispatch(field, log_n, inverse, batch, d_in, d_out, nullptr, stream);
}

int ntt_dev_hooked_impl(int field, unsigned log_n, int inverse, unsigned batch, const void* d_in, void* d_out, const NttHooks& hooks,
                        hipStream_t stream) {
    return ntt_dispatch(field, log_n, inverse, batch, d_in, d_out, &hooks, stream);
}

int ntt_reference_table_dev_impl(int field, unsigned log_n, void* d_out, hipStream_t stream) {
    if (field_limbs(field) < 0) return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    if (log_n > 30) return set_error(PLK_ERR_TWO_ADICITY, "log_n %u too large (max 30)", log_n);
    if (!d_out) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    std::shared_ptr<NttPlan> pl;
    PLK_TRY(get_plan(field, log_n, pl));  // PLK_ERR_TWO_ADICITY beyond the field's 2-adicity (field.rs:430)
    const int log_t = (int)log_n > INNER_LOG ? (int)log_n : INNER_LOG;
    const size_t total = ((size_t)2 << log_n) - 1;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    switch (field) {
#define CASE(ID, P) case ID: k_ntt_reference_table<P><<<blocks, 256, 0, stream>>>((uint4*)d_out, (const uint4*)pl->pw, log_t, (int)log_n); break;
        CASE(PLK_FIELD_TWEEDLEDEE_BASE, TweedledeeBaseParams)
        CASE(PLK_FIELD_TWEEDLEDUM_BASE, TweedledumBaseParams)
        CASE(PLK_FIELD_BLS12_377_SCALAR, Bls12377ScalarParams)
        CASE(PLK_FIELD_BLS12_377_BASE, Bls12377BaseParams)
        CASE(PLK_FIELD_PALLAS_BASE, PallasBaseParams)
        CASE(PLK_FIELD_VESTA_BASE, VestaBaseParams)
#undef CASE
    }
    PLK_HIP_TRY(hipGetLastError());
    // the plan (and its power table) must outlive the kernel: the cache may be cleared by another thread
    PLK_HIP_TRY(hipStreamSynchronize(stream));
    return PLK_OK;
}

int ntt_plan_pow_table(int field, unsigned log_n, const void** pw, int* log_t, std::shared_ptr<const void>* hold) {
    std::shared_ptr<NttPlan> pl;
    PLK_TRY(get_plan(field, log_n, pl));
    *pw = pl->pw;  // owned by the plan: alive while the cache or *hold references it
    if (hold) *hold = pl;
    *log_t = (int)log_n > INNER_LOG ? (int)log_n : INNER_LOG;
    return PLK_OK;
}

}  // namespace plk
