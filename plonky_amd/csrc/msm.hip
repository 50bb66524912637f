// msm.hip -- multi-scalar multiplication sum_i s_i * G_i on gfx950 (bucket method on window tables).
//
// Replaces the reference's src/curve/curve_msm.rs (+ curve_summations.rs, curve_adds.rs):
//   msm_precompute / precompute_single_generator  curve_msm.rs:27-52  -> k_msm_table (device tables)
//   to_digits                                     curve_msm.rs:159-180 -> ord_digit (signed, carry based; recomputed where used)
//   digit_occurrences scatter (serial in the ref) curve_msm.rs:117-126 -> two-level LDS partition (k_ord_*)
//   per-digit affine multi-summation              curve_msm.rs:131-145 -> k_msm_accumulate (XYZZ mixed adds, equal chunks)
//   serial Yao tail  u += acc[d]; y += u          curve_msm.rs:149-154 -> k_msm_assemble + k_msm_gsum/lsum + k_msm_planes + k_msm_final
//   msm_execute / msm_execute_parallel            curve_msm.rs:63-157  -> msm_execute_dev_impl
//   msm_parallel (precompute + execute, one use)  curve_msm.rs:54-61   -> the table-free mode (PLK_MSM_TABLE_FREE)
// Same mathematical structure as the reference (Yao's method over per-generator power tables
// [2^(c j)] G_i, one bucket per digit value, result = sum_d d * bucket_d) with two MI355X-first
// changes: (1) digits are signed (carry-based integer recoding, never s -> r - s, so it is valid
// on BLS12-377 G1 whose cofactor is even): half the buckets for the same window; (2) the serial
// running sum is replaced by a two-level weighting (row / column sums of the bucket grid) followed by
// bit-plane tree sums (sum_d d B_d = sum_p 2^p sum_{d: bit p} B_d), so the tail is O(log) deep instead of
// 2 * 2^w sequential additions and costs 2 additions per bucket - which is what makes windows of 18-20 bits
// (13 additions per scalar at 2^20) affordable.  The result is returned as
// the unique affine point (to_affine, curve.rs:206-214), on which parity is defined.
//
// Work decomposition: every (scalar i, window j) with a non-zero digit is one *entry* that adds
// +-table[j*n + i] into bucket |d|-1.  Entries are counting-sorted by bucket; the sorted list is cut
// into equal chunks and one lane accumulates one chunk (storing a piece at every bucket boundary), so the
// load per lane is the same whatever the digit distribution (a skewed witness cannot serialise the kernel).
//
// Table-free mode (generators used once): only the generators themselves are stored; window j gets its own
// bucket range [j 2^(c-1), (j+1) 2^(c-1)), the same kernels run over all windows at once, every window's
// sum (plane sums over its buckets up to 12 bits, over its row / column sums above) is doubled into place (2^(c j)) by a
// quad and k_msm_combine adds the windows.  On the prime-order curves the scalars are split along the endomorphism first
// (glv.cuh): 2n points [G.., phi(G)..], half-length scalars, half the windows - half of that doubling chain.
// Batches: every MSM of a group has its own workspace and the group shares one reduction (msm_execute_dev_impl).
// The reduction kernels run on quads of lanes (ecz_coop.cuh): they are chains of point operations, i.e. latency.
#include <mutex>
#include <type_traits>
#include <vector>

#include "msm_dev.cuh"
#include "glv.cuh"
#include "ecz_coop.cuh"
#include "tables.cuh"

namespace plk {

// comb.hip: small fixed-base MSMs without buckets (a table of every multiple a signed 4-bit digit can ask for)
struct CombPlan;
int comb_build(int curve, size_t n, const void* d_base0, const void* d_chain, hipStream_t stream, CombPlan** out);
void comb_free(CombPlan* p);
int comb_execute(const CombPlan* p, unsigned batch, const void* const* d_scalars, const uint64_t* first, const uint64_t* count, void* d_out_xy, void* d_out_zero,
                 hipStream_t stream);
// msm_acc.hip: the bucket accumulation kernel (its own translation unit)
template <class C> int msm_accumulate_blocks_per_cu();
template <class C>
void msm_launch_accumulate(unsigned blocks, hipStream_t stream, const void* tab, const void* sorted, const void* off, void* p_start, void* p_head,
                           void* head_live, void* head_bucket, void* live_list, uint32_t* live_count, uint32_t buckets, const uint32_t* dyn_chunk, int wshift,
                           uint32_t n_sub, uint32_t tab_entries);
// msm_order.hip: digits + the two-level bucket ordering; msm_tail.hip: the reduction
template <class C> int msm_launch_glv_split(const void* d_scalars, size_t n, void* halves, hipStream_t stream);
template <class C> int msm_launch_order_stage(int stage, const OrdCfg& o, const OrdBuffers& b, hipStream_t stream);
template <class C> int msm_launch_digits(const void* d_scalars, size_t n, const OrdCfg& o, void* d_digits, hipStream_t stream);
template <class C> int msm_launch_reduce_stage(int stage, const TailGeom& g, const TailBatch& tb, hipStream_t stream);
constexpr size_t COMB_MAX_N = (size_t)1 << 15;  // generators up to which a tabled context with an automatic window is a comb (1 GB of table at 2^15)
constexpr size_t COMB_AUTO_MAX_N = (size_t)1 << 12;  // ... and up to which it is chosen without being asked for (measured faster than the bucket method)
constexpr int COMB_WINDOW = 4, COMB_WINDOWS = 64;



// ---------------------------------------------------------------------------------------------
// table construction: tab[j*n + i] = [2^(c j)] G_i, affine  (curve_msm.rs:40-52)
// ---------------------------------------------------------------------------------------------
// The table is what the accumulation kernel multiplies with, so it is stored in the working form of
// that kernel (ecz.cuh / fz.cuh): coordinates in R'-form (x 2^(29 NZ)), canonical, packed in the
// same 32-bit words.  The generators arrive in the reference's R-form.
// One lane per generator, on the lazy arithmetic of the accumulation kernel (ecz.cuh): c doublings per window,
// then back to affine with the division-step inversion (about a quarter of a window's work).
// glv != 0 (table-free mode on the curves with the endomorphism, glv.cuh): tab[n + i] = phi(G_i) = (beta x, y).
// The generators may come in two pieces: n_main points at `bases` (+ optional identity flags) followed by n - n_main points at
// `extra` (an IPA round appends H and U' to the half of G it multiplies, halo.rs:86-93).
template <class C>
__global__ void __launch_bounds__(128) k_msm_table(const uint4* __restrict__ bases, const uint8_t* __restrict__ base_zero, uint4* __restrict__ tab,
                                                   size_t n, int c, int windows, int glv, size_t n_main, const uint4* __restrict__ extra) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // R-form (the reference's) -> canonical R'-form, the form the table is stored and consumed in
    const uint4* src = i < n_main ? bases + i * 2 * W : extra + (i - n_main) * 2 * W;
    const Fe<FP> x_in = fe_load<FP>(src);
    Fe<FP> xr = to_rprime<FP>(x_in), yr = to_rprime<FP>(fe_load<FP>(src + W));
    bool ident = (base_zero && i < n_main) ? base_zero[i] != 0 : false;
    affine_store<FP>(tab + i * 2 * W, xr, yr, ident);
    if constexpr (C::Glv::ENABLED) {
        if (glv) {
            Fe<FP> beta;
#pragma unroll
            for (int k = 0; k < FP::NL; ++k) beta.v[k] = C::Glv::BETA[k];
            const Fe<FP> xphi = to_rprime<FP>(fe_mul<FP>(x_in, fe_from_canonical<FP>(beta)));
            affine_store<FP>(tab + (n + i) * 2 * W, xphi, yr, ident);
        }
    }
    const Fz<FP> one = fz_one_rprime<FP>();
    for (int j = 1; j < windows; ++j) {
        if (!ident) {
            XyzzZ<FP> p;
            p.x = fz_from_fe<FP>(xr);
            p.y = fz_from_fe<FP>(yr);
            p.zz = one;
            p.zzz = one;
            p.inf = false;
            for (int k = 0; k < c; ++k) p = xyzzz_dbl<FP>(p);
            ident = p.inf;
            if (!ident) {
                // x = X / ZZ, y = Y / ZZZ with 1 / Z = ZZ / ZZZ (xyzz_to_affine, ec.cuh), all in R'-form
                const Fe<FP> zzz_r = fz_to_fe_canonical<FP>(fz_mul<FP>(p.zzz, fz_const_rprime_to_r<FP>()));
                const Fz<FP> i3 = fz_from_fe<FP>(to_rprime<FP>(fe_inv_safegcd<FP>(zzz_r)));
                const Fz<FP> iz = fz_mul<FP>(p.zz, i3);
                const Fz<FP> izz = fz_sqr<FP>(iz);
                xr = fz_to_fe_canonical<FP>(fz_mul<FP>(p.x, izz));
                yr = fz_to_fe_canonical<FP>(fz_mul<FP>(p.y, i3));
            }
        }
        affine_store<FP>(tab + ((size_t)j * n + i) * 2 * W, xr, yr, ident);
    }
}

// The reference's own table (MsmPrecomputation::powers_per_generator, curve_msm.rs:16-52): the device table
// [j][i] in R'-form becomes [i][j] in the reference's Montgomery form, with AffinePoint::zero flags.
template <class C>
__global__ void __launch_bounds__(256) k_msm_table_export(const uint4* __restrict__ tab, size_t n, int digits, uint4* __restrict__ out_xy,
                                                          uint8_t* __restrict__ out_zero) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // output slot i * digits + j
    if (e >= n * (size_t)digits) return;
    const size_t i = e / digits, j = e % digits;
    Fe<FP> x, y;
    const bool ident = affine_load<FP>(tab + (j * n + i) * 2 * W, x, y);
    const Fz<FP> back = fz_const_rprime_to_r<FP>();
    x = ident ? fe_zero<FP>() : fz_to_fe_canonical<FP>(fz_mul<FP>(fz_from_fe<FP>(x), back));
    y = ident ? fe_zero<FP>() : fz_to_fe_canonical<FP>(fz_mul<FP>(fz_from_fe<FP>(y), back));
    fe_store<FP>(out_xy + e * 2 * W, x);
    fe_store<FP>(out_xy + e * 2 * W + W, y);
    out_zero[e] = ident ? 1 : 0;
}

// Short generator lists (the frozen generators of an inner-product argument: 2^14 + 2 points): one lane per generator is a
// dependency chain, and in k_msm_table a quarter of it is the inversion that brings every window back to affine before the next
// c doublings.  Here the chain only doubles - c (windows - 1) doublings without a normalisation in between, every window's
// XYZZ point parked in scratch memory - and a second kernel normalises all (window, generator) pairs side by side:
// 1.75 -> 1.2 ms for 2^14 + 2 generators at c = 13 with a lane per generator, 0.65 ms on quads in one-wave workgroups (a chain of quad
// doublings takes 3.6 us per step in two-wave workgroups at this occupancy, 2.2 us in one-wave ones: tools/lab/mul_latency.hip).  Same points, so the same (unique) affine table entries.
template <class C>
__global__ void __launch_bounds__(64) k_msm_table_chain(const uint4* __restrict__ bases, const uint8_t* __restrict__ base_zero, uint4* __restrict__ tab,
                                                         uint4* __restrict__ raw, size_t n, int c, int windows, size_t n_main, const uint4* __restrict__ extra) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    constexpr int RU = raw_u4<FP>();
    // a quad per generator (ecz_coop.cuh): a doubling is 3 multiplications deep instead of 9
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int ql = threadIdx.x & 3;
    if (i >= n) return;
    const uint4* src = i < n_main ? bases + i * 2 * W : extra + (i - n_main) * 2 * W;
    const Fe<FP> xr = to_rprime<FP>(fe_load<FP>(src)), yr = to_rprime<FP>(fe_load<FP>(src + W));
    const bool ident = (base_zero && i < n_main) ? base_zero[i] != 0 : false;
    if (ql == 0) affine_store<FP>(tab + i * 2 * W, xr, yr, ident);
    XyzzZ<FP> p = xyzzz_identity<FP>();
    if (!ident) {
        p.x = fz_from_fe<FP>(xr);
        p.y = fz_from_fe<FP>(yr);
        p.zz = fz_one_rprime<FP>();
        p.zzz = p.zz;
        p.inf = false;
    }
    for (int j = 1; j < windows; ++j) {
        for (int k = 0; k < c; ++k) p = xyzzz_dbl_q<FP>(p, ql);
        if (ql == 0) xyzzz_store_raw<FP>(raw + ((size_t)(j - 1) * n + i) * RU, p);
    }
}
template <class C>
__global__ void __launch_bounds__(64) k_msm_table_norm(const uint4* __restrict__ raw, uint4* __restrict__ tab, size_t n, int windows) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    constexpr int RU = raw_u4<FP>();
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)(windows - 1) * n) return;
    const XyzzZ<FP> p = xyzzz_load_raw<FP>(raw + t * RU);
    Fe<FP> xr = fe_zero<FP>(), yr = fe_zero<FP>();
    if (!p.inf) {
        // x = X / ZZ, y = Y / ZZZ with 1 / Z = ZZ / ZZZ (xyzz_to_affine, ec.cuh), all in R'-form
        const Fe<FP> zzz_r = fz_to_fe_canonical<FP>(fz_mul<FP>(p.zzz, fz_const_rprime_to_r<FP>()));
        const Fz<FP> i3 = fz_from_fe<FP>(to_rprime<FP>(fe_inv_safegcd<FP>(zzz_r)));
        const Fz<FP> iz = fz_mul<FP>(p.zz, i3);
        const Fz<FP> izz = fz_sqr<FP>(iz);
        xr = fz_to_fe_canonical<FP>(fz_mul<FP>(p.x, izz));
        yr = fz_to_fe_canonical<FP>(fz_mul<FP>(p.y, i3));
    }
    affine_store<FP>(tab + (n + t) * 2 * W, xr, yr, p.inf);   // entry (j, i) of the table sits at j n + i = n + t
}


template <class FP> PLK_DI Xyzz<FP> block_sum(Xyzz<FP> v, uint4* s_pts) {
    constexpr int W = FP::NL / 4;
    const int tid = threadIdx.x;
    xyzz_store<FP>(s_pts + tid * 4 * W, v);
    __syncthreads();
    for (int d = blockDim.x >> 1; d >= 1; d >>= 1) {
        if (tid < d) {
            v = xyzz_add<FP>(v, xyzz_load<FP>(s_pts + (tid + d) * 4 * W));
            xyzz_store<FP>(s_pts + tid * 4 * W, v);
        }
        __syncthreads();
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// small utilities: sum of k affine points; synthetic generators G0 + (first + i) D
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(64) k_sum_affine(const uint4* __restrict__ pts, const uint8_t* __restrict__ zero, size_t k, uint4* __restrict__ out_xy,
                                                   uint8_t* __restrict__ out_zero) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    extern __shared__ __attribute__((aligned(16))) uint4 s_pts[];
    Xyzz<FP> acc = xyzz_identity<FP>();
    for (size_t i = threadIdx.x; i < k; i += blockDim.x) {
        if (zero && zero[i]) continue;
        Fe<FP> x = fe_load<FP>(pts + i * 2 * W), y = fe_load<FP>(pts + i * 2 * W + W);
        xyzz_madd<FP>(acc, x, y);
    }
    acc = block_sum<FP>(acc, s_pts);
    if (threadIdx.x == 0) {
        Fe<FP> x, y;
        bool ident = xyzz_to_affine<FP, true>(acc, x, y);
        fe_store<FP>(out_xy, x);
        fe_store<FP>(out_xy + W, y);
        *out_zero = ident ? 1 : 0;
    }
}

// Multi-GPU exchange (plonky_hip.h): every rank's results travel as one packed record of `slots` points then
// `slots` identity flags.  Block v produces vector v: a whole vector (v < whole * world) is rank v % world's slot v / world,
// a sharded one is the sum over the ranks of slot whole + (v - whole * world).
template <class C>
__global__ void __launch_bounds__(64) k_combine_partials(const uint8_t* __restrict__ gathered, size_t rec_bytes, unsigned world, unsigned slots,
                                                         unsigned whole, uint4* __restrict__ out_xy, uint8_t* __restrict__ out_zero) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    extern __shared__ __attribute__((aligned(16))) uint4 s_pts[];
    const unsigned v = blockIdx.x;
    const bool is_whole = v < whole * world;
    const unsigned slot = is_whole ? v / world : whole + (v - whole * world);
    const unsigned r0 = is_whole ? v % world : 0, r1 = is_whole ? r0 + 1 : world;
    Xyzz<FP> acc = xyzz_identity<FP>();
    // The records were written by OTHER devices (peer copies over xGMI, multi.hip), by RCCL or through the host, into a buffer this
    // device may have read before (the scratch pool hands it out again): every word is read at SYSTEM scope, past this device's
    // caches - a few hundred bytes per rank, so the price is nothing, and the hand-over does not depend on what a kernel boundary
    // invalidates (each XCD has its own L2; MI355X_MICROARCH.md).  Records are 16-byte aligned (msm_partials_bytes).
    auto word = [](const uint8_t* p) { return __hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    for (unsigned r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const uint8_t* rec = gathered + (size_t)r * rec_bytes;
        const size_t flag_at = (size_t)slots * 2 * W * 16 + slot;
        if ((word(rec + (flag_at & ~(size_t)3)) >> (8 * (flag_at & 3))) & 0xffu) continue;
        const uint8_t* pt = rec + (size_t)slot * 2 * W * 16;
        Fe<FP> x, y;
#pragma unroll
        for (int i = 0; i < FP::NL; ++i) {
            x.v[i] = word(pt + 4 * i);
            y.v[i] = word(pt + 4 * (FP::NL + i));
        }
        xyzz_madd<FP>(acc, x, y);
    }
    acc = block_sum<FP>(acc, s_pts);
    if (threadIdx.x == 0) {
        Fe<FP> x, y;
        const bool ident = xyzz_to_affine<FP, true>(acc, x, y);
        fe_store<FP>(out_xy + (size_t)v * 2 * W, x);
        fe_store<FP>(out_xy + (size_t)v * 2 * W + W, y);
        out_zero[v] = ident ? 1 : 0;
    }
}

template <class C>
__global__ void __launch_bounds__(128) k_gen_bases(const uint4* __restrict__ g0d, uint4* __restrict__ out, size_t n, uint64_t first) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<FP> gx = fe_load<FP>(g0d), gy = fe_load<FP>(g0d + W), dx = fe_load<FP>(g0d + 2 * W), dy = fe_load<FP>(g0d + 3 * W);
    // (first + i) * D by double-and-add from the top bit, then + G0
    uint64_t m = first + i;
    Xyzz<FP> acc = xyzz_identity<FP>();
    for (int b = 63; b >= 0; --b) {
        acc = xyzz_dbl<FP>(acc);
        if ((m >> b) & 1) xyzz_madd<FP>(acc, dx, dy);
    }
    xyzz_madd<FP>(acc, gx, gy);
    Fe<FP> x, y;
    bool ident = xyzz_to_affine<FP>(acc, x, y);
    (void)ident;  // G0 + m D is the identity only for one m in the whole group; callers use small m
    fe_store<FP>(out + i * 2 * W, x);
    fe_store<FP>(out + i * 2 * W + W, y);
}

// ---------------------------------------------------------------------------------------------
// self-test: the quad arithmetic (ecz_coop.cuh) against the one-lane arithmetic (ecz.cuh) on the same operands
// ---------------------------------------------------------------------------------------------
// same group element: x1 zz2 == x2 zz1 and y1 zzz2 == y2 zzz1 (the projective equality of curve.rs:280-302)
template <class FP> PLK_DI bool xyzzz_same(const XyzzZ<FP>& a, const XyzzZ<FP>& b) {
    if (a.inf || b.inf) return a.inf == b.inf;
    const Fe<FP> l1 = fz_to_fe_canonical<FP>(fz_mul<FP>(a.x, b.zz)), r1 = fz_to_fe_canonical<FP>(fz_mul<FP>(b.x, a.zz));
    const Fe<FP> l2 = fz_to_fe_canonical<FP>(fz_mul<FP>(a.y, b.zzz)), r2 = fz_to_fe_canonical<FP>(fz_mul<FP>(b.y, a.zzz));
    bool ok = true;
    for (int i = 0; i < FP::NL; ++i) ok = ok && (l1.v[i] == r1.v[i]) && (l2.v[i] == r2.v[i]);
    return ok;
}
template <class C>
__global__ void __launch_bounds__(256) k_selftest_quad(const uint4* __restrict__ pts, uint32_t n, uint32_t* __restrict__ mismatches) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    const uint32_t quad = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int ql = threadIdx.x & 3;
    const uint32_t i = quad % n, j = (quad * 7u + 3u) % n;
    const Fz<FP> k = fz_const_r_to_rprime<FP>();
    auto load_pt = [&](uint32_t idx, Fz<FP>& x, Fz<FP>& y) {
        x = fz_from_fe<FP>(fz_to_fe_canonical<FP>(fz_mul<FP>(fz_from_fe<FP>(fe_load<FP>(pts + (size_t)idx * 2 * W)), k)));
        y = fz_from_fe<FP>(fz_to_fe_canonical<FP>(fz_mul<FP>(fz_from_fe<FP>(fe_load<FP>(pts + (size_t)idx * 2 * W + W)), k)));
    };
    Fz<FP> xi, yi, xj, yj;
    load_pt(i, xi, yi);
    load_pt(j, xj, yj);
    XyzzZ<FP> a = xyzzz_identity<FP>(), b = xyzzz_identity<FP>();
    xyzzz_madd<FP>(a, xi, yi);
    a = xyzzz_dbl<FP>(a);           // 2 P_i, zz != 1
    xyzzz_madd<FP>(b, xj, yj);
    xyzzz_madd<FP>(b, xi, yi);      // P_j + P_i (or 2 P_i / identity when the indices collide)
    XyzzZ<FP> na = a;
    na.y = fz_sub<FP, 2>(fz_zero<FP>(), a.y);  // -a, y < 4p
    bool ok = true;
    // sum over the 16 quads of the wave against a serial sum of the same 16 points (whole wave active)
    bool wave_ok;
    {
        XyzzZ<FP> tot = wave_sum_q<FP>(a, 16, ql);
        XyzzZ<FP> ser = xyzzz_identity<FP>();
        for (int q = 0; q < 16; ++q) {
            XyzzZ<FP> t = a;  // lane 4q of this wave holds that quad's a
            const int src = 4 * q;
#pragma unroll
            for (int l = 0; l < FzCfg<FP>::NZ; ++l) {
                t.x.l[l] = __shfl(a.x.l[l], src);
                t.y.l[l] = __shfl(a.y.l[l], src);
                t.zz.l[l] = __shfl(a.zz.l[l], src);
                t.zzz.l[l] = __shfl(a.zzz.l[l], src);
            }
            t.inf = __shfl((int)a.inf, src) != 0;
            ser = xyzzz_add<FP>(ser, t);
        }
        wave_ok = xyzzz_same<FP>(tot, ser);
    }
    switch (quad & 7u) {
        case 0: ok = xyzzz_same<FP>(xyzzz_add_q<FP>(a, b, ql), xyzzz_add<FP>(a, b)); break;
        case 1: ok = xyzzz_same<FP>(xyzzz_dbl_q<FP>(a, ql), xyzzz_dbl<FP>(a)); break;
        case 2: ok = xyzzz_same<FP>(xyzzz_add_q<FP>(a, a, ql), xyzzz_dbl<FP>(a)); break;          // doubling inside the addition
        case 3: ok = xyzzz_add_q<FP>(a, na, ql).inf; break;                                          // opposite points
        case 4: ok = xyzzz_same<FP>(xyzzz_add_q<FP>(xyzzz_identity<FP>(), b, ql), b); break;
        case 5: ok = xyzzz_same<FP>(xyzzz_add_q<FP>(b, xyzzz_identity<FP>(), ql), b); break;
        case 6: ok = xyzzz_same<FP>(xyzzz_dbl_q<FP>(xyzzz_dbl_q<FP>(b, ql), ql), xyzzz_dbl<FP>(xyzzz_dbl<FP>(b))); break;
        default: ok = wave_ok;
    }
    if (!ok) atomicAdd(mismatches + (quad & 7u), 1u);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
}  // namespace plk

// per-execution device workspace; a batched execution owns one per MSM of a group
struct MsmWork {
    void* tmp = nullptr;       // uint2 (code, entry id) ordered by coarse bin
    void* sorted = nullptr;    // (entry id << 1 | negative) ordered by bucket
    void* cnt1 = nullptr;      // [nbins][nt1]
    void* meta = nullptr;      // bin_total[1024] | bin_base[1025] | seg_base[1025] | done counter
    void* cnt2 = nullptr;      // [segment][fine]
    void* off = nullptr;       // off[buckets + 1]
    void* p_start = nullptr;   // raw pieces, one per bucket
    void* p_head = nullptr;    // raw pieces, one per accumulation lane
    void* head_live = nullptr; // one byte per accumulation lane
    void* bucket = nullptr;    // packed points: operands of the plane sums
    void* heavy = nullptr;     // heavy-bucket work list (see k_msm_heavy_list)
    void* heavy_part = nullptr;
    void* line_part = nullptr; // two-level tail: row / column partial sums
    void* plane_part = nullptr;
    void* win_pts = nullptr;   // the per-window results
    void* slab = nullptr;      // the one allocation all of the above point into
    size_t cap[15] = {};       // bytes of each part above, in the order of msm_work_parts (a context that is rebound to fewer points keeps its layout)
    bool pooled = false;       // slab comes from the library's scratch pool (table-free contexts: built and dropped per call)
    bool ready = false;
    // executions on different streams share the workspace: the next user waits for the previous one's last kernel
    hipEvent_t ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool used = false;
    void release() {
        if (slab && pooled) plk::scratch_release(slab, last_stream);  // stream-ordered: the next taker waits for our last kernel
        else if (slab) (void)hipFree(slab);
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;
        slab = nullptr;
        for (void** p : {&tmp, &sorted, &cnt1, &cnt2, &meta, &off, &p_start, &p_head, &head_live, &bucket, &heavy, &heavy_part, &line_part, &plane_part, &win_pts}) *p = nullptr;
        ready = false;
        used = false;
    }
};

struct plk_msm_ctx {
    int curve = 0;
    int device = 0;
    size_t n = 0;
    int c = 0;          // window bits
    int windows = 0;    // ceil((BITS + 1) / c)
    uint32_t buckets = 0;  // bucket slots: 2^(c-1) with tables; windows * 2^(c-1) (rounded up to whole partition bins) without
    uint32_t wbuckets = 0; // 2^(c-1): buckets per window
    bool table_free = false;  // no window tables: every window has its own buckets and is doubled into place at the end
    uint32_t chunk = 24;   // entries per accumulation lane
    // tail geometry
    bool glv = false;        // table-free mode on a curve with the endomorphism: 2n points, half-length scalars (glv.cuh)
    size_t n_eff = 0;        // points the kernels see: 2n with glv, else n
    bool two_level = false;  // tabled mode with many buckets: row / column sums first
    int L = 0, H = 0;        // bucket grid 2^H x 2^L
    int g_log = 0, lpl_log = 0, lpb_log = 0;
    int tail_windows = 1;    // windows seen by the plane kernels (2 in two-level mode: columns, rows)
    uint32_t tail_wbuckets = 0;
    int tail_shift = 0;      // doublings between consecutive tail windows
    int planes = 0;
    int plane_blocks = 1;  // blocks (parts) per plane
    size_t max_lanes = 0;
    // device memory
    void* tab = nullptr;
    size_t tab_cap = 0;      // bytes allocated for the table
    hipStream_t tab_stream = nullptr;  // table-free: the stream the (pooled) table was built on
    plk::OrdCfg ord{};
    uint32_t heavy_cap = 0;
    std::vector<MsmWork> ws;   // ws[0] at precompute; a batched execution allocates one per MSM of a group (<= TAIL_MAX)
    size_t ws_bytes = 0;       // size of one workspace slab
    std::mutex mu;             // one enqueue at a time per context
    hipStream_t tail_stream = nullptr;  // batched executions: the reduction of vector k runs here, under the accumulation of vector k + 1
    hipEvent_t ev_tail = nullptr;
    std::vector<hipEvent_t> ev_acc;
    std::vector<hipStream_t> fork_streams;  // small batched executions: the ordering + accumulation of vector k >= 1 runs on fork_streams[k - 1]
    // optional per-kernel timing (HIP events on the launch stream) for bench.py's roofline
    bool profiling = false;
    static constexpr int N_STAGES = 7;  // order: count + scan | scatter | bins; accumulate; heavy + assemble + lines; planes; final
    std::vector<std::vector<hipEvent_t>> prof_sets;  // each N_STAGES + 1 events, recorded
    std::vector<std::vector<hipEvent_t>> prof_free;
    // A context built while the library runs over several devices (plk_init_devices; multi.hip) owns one context with the full
    // tables on every other logical device (peers[d - 1]) and, on every device d including this one, a context over that device's
    // contiguous share of the generators with the window such a share deserves (shards[d]): whole vectors of a batch run on the
    // full tables, a single MSM runs sharded by base range.
    std::vector<plk_msm_ctx*> peers, shards;
    // A tabled context over few generators (automatic window; <= COMB_AUTO_MAX_N by default, PLK_MSM_COMB) is a COMB (comb.hip): no window tables,
    // no workspaces, no bucket method - executions are mixed additions of table entries and a tree over the lanes' sums.
    plk::CombPlan* comb = nullptr;
    bool auto_window = false;
    bool many_heads = false;  // the call in progress holds a bucket share (set and cleared under `mu` by msm_execute_dev_impl)
    bool out_projective = false;  // the call in progress returns ProjectivePoints (likewise)
    ~plk_msm_ctx() {
        if (comb) plk::comb_free(comb);
        for (auto* v : {&peers, &shards})
            for (plk_msm_ctx* sub : *v)
                if (sub) {
                    (void)hipSetDevice(sub->device);
                    delete sub;
                }
        (void)hipSetDevice(device);
        if (tab && table_free) {
            // pooled: hand it back ordered after the last kernel that read it (several user streams: wait for them here)
            hipStream_t last = tab_stream;
            int users = 0;
            for (MsmWork& w : ws)
                if (w.used) {
                    if (users++ && w.last_stream != last) (void)hipStreamSynchronize(last);
                    last = w.last_stream;
                }
            plk::scratch_release(tab, last);
        } else if (tab) {
            (void)hipFree(tab);
        }
        for (MsmWork& w : ws) w.release();
        if (tail_stream) {
            (void)hipStreamSynchronize(tail_stream);
            plk::stream_pool_release(tail_stream);
        }
        if (ev_tail) (void)hipEventDestroy(ev_tail);
        for (hipEvent_t e : ev_acc) (void)hipEventDestroy(e);
        for (hipStream_t st : fork_streams) {
            (void)hipStreamSynchronize(st);
            plk::stream_pool_release(st);
        }
        for (auto* v : {&prof_sets, &prof_free})
            for (auto& set : *v)
                for (hipEvent_t e : set) (void)hipEventDestroy(e);
    }
};

namespace plk {

static int scalar_bits(int curve) { return curve == PLK_CURVE_BLS12_377 ? 253 : 255; }
static int ilog2_ceil(uint64_t v) {
    int b = 0;
    while (((uint64_t)1 << b) < v) ++b;
    return b;
}

static int choose_window(size_t n, int curve) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) ++lg;
    const int bits = scalar_bits(curve) + 1;
    auto digits = [&](int c) { return (bits + c - 1) / c; };
    auto top_bits = [&](int c) { return bits - (digits(c) - 1) * c; };
    int c;
    if (lg >= 14) {
        // From 2^14 generators on the window minimises a count of field multiplications: digits(c) mixed additions per scalar
        // (10 each) + two full additions per bucket in the reduction (14 each) + a tenth on top of the accumulation when the TOP
        // WINDOW IS SHORT (fewer than c / 2 bits: every scalar's top digit lands in a handful of buckets, which go through the
        // heavy-bucket path - ~120 us of workgroup-wide sums whatever the size; 24 windows of 11 leave the top one 3 bits, 18
        // of 15 one bit).  Measured in round 3 (profiles/r03_commit9_scaling.txt, r03_window_sweeps.txt): 2^14: 13 (0.41 ms
        // against 0.52 at 11), 2^16 / 2^17 / 2^18: 16 (0.53 against 0.68 at 14; 0.88 against 1.01 at 18), 2^19 (BLS12-377): 17,
        // 2^20 and up: 20 (13 additions per scalar, 2^19 buckets: round 2).  Round 6 (the list-driven assembly, tools/gpu/r06_small_msm2.sh,
        // profiles/r06_small_msm_windows.txt): at 2^14 the count is no longer the measure - every stage is a chain of a few point operations -
        // and 16 wins (two vectors: 0.442 ms against 0.481 at 13; one bucket piece per lane, so no k_msm_assemble tree), with a smaller table.
        double best = 0;
        c = 0;
        for (int t = 10; t <= MSM_MAX_WINDOW - 1; ++t) {
            const double acc = 10.0 * (double)n * digits(t);
            const double cost = acc + 28.0 * (double)((size_t)1 << (t - 1)) + (2 * top_bits(t) < t ? 0.1 * acc : 0.0);
            if (c == 0 || cost < best) {
                best = cost;
                c = t;
            }
        }
        if (lg == 14) {
            c = 16;
            if (const char* e = getenv("PLK_MSM_WINDOW_2P14")) c = atoi(e);  // A/B of this size alone (the IPA's frozen generators)
        }
    } else {
        c = lg - 4;
        if (c < 3) c = 3;
        // the smallest window with the same number of digits (fewer buckets for the same additions) ...
        while (c > 3 && digits(c - 1) == digits(c)) --c;
        // ... unless that leaves the top window short: then the next width whose top window holds at least half a window
        for (int t = c; t <= c + 4 && t <= 16; ++t)
            if (2 * top_bits(t) >= t) {
                c = t;
                break;
            }
    }
    if (const char* e = getenv("PLK_MSM_WINDOW")) c = atoi(e);
    if (c < 3) c = 3;
    if (c > MSM_MAX_WINDOW) c = MSM_MAX_WINDOW;
    return c;
}

// accumulation lanes the GPU runs at once (for the chunk size: whole rounds of lanes, no ragged last round)
template <class C> static size_t accumulate_slots() {
    static std::mutex mu;
    static size_t slots[64] = {};  // per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    size_t& s = slots[dev & 63];
    if (s == 0) {
        int per_cu = 0, cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        per_cu = msm_accumulate_blocks_per_cu<C>();
        s = (size_t)per_cu * ACC_THREADS * (size_t)cus;
    }
    return s;
}

// head_live[] (bytes) and head_bucket[] (words) share one workspace part: the words start at this offset
static size_t head_lanes_padded(size_t max_lanes) { return (max_lanes + ACC_THREADS + 15) & ~(size_t)15; }
static uint32_t* head_bucket_of(const plk_msm_ctx* ctx, const MsmWork& w);
static uint32_t* live_list_of(const plk_msm_ctx* ctx, const MsmWork& w);

// One slab per workspace: a single hipMalloc / hipFree instead of a dozen (they dominate a one-shot msm_parallel).
constexpr int MSM_WORK_PARTS = 15;
struct WorkPart { void** p; size_t bytes; };
template <class C> static void msm_work_parts(const plk_msm_ctx* ctx, MsmWork& w, WorkPart* parts) {
    using FP = typename C::FP;
    const size_t packed_bytes = (size_t)4 * FP::NL * 4;
    const size_t raw_bytes = (size_t)raw_u4<FP>() * 16;
    const size_t entries = ctx->n_eff * ctx->windows;
    const int bucket_windows = ctx->tail_windows;
    // packed operands of the plane sums: the buckets themselves, or (two-level tail) the column and row sums
    const size_t tail_slots = ctx->two_level ? (size_t)bucket_windows * ctx->tail_wbuckets : (size_t)ctx->buckets;
    const WorkPart src[MSM_WORK_PARTS] = {
        {&w.tmp, entries * 8 + 16},
        {&w.sorted, entries * 4 + 16},
        {&w.cnt1, (size_t)ctx->ord.nbins * ctx->ord.nt1 * 4},
        {&w.cnt2, ((entries / ORD_SEG + ctx->ord.nbins + 1) << ctx->ord.fine_bits) * 4},
        {&w.meta, (size_t)(1024 + 1025 + 1025 + 8) * 4},
        {&w.off, ((size_t)ctx->buckets + 2) * 4},
        {&w.p_start, (size_t)ctx->buckets * raw_bytes},
        {&w.p_head, (ctx->max_lanes + 1) * raw_bytes},
        // one flag byte per lane, then the lanes' head buckets (4 bytes each), then the list of live lanes (a counter word + 4 bytes each)
        {&w.head_live, 9 * head_lanes_padded(ctx->max_lanes)},
        {&w.bucket, tail_slots * packed_bytes},
        {&w.heavy, (size_t)(2 + 3 * ctx->heavy_cap) * 4},
        {&w.heavy_part, (size_t)ctx->heavy_cap * raw_bytes},
        {&w.line_part, ctx->two_level ? (size_t)2 * (ctx->buckets >> ctx->g_log) * raw_bytes : 0},
        {&w.plane_part, (size_t)bucket_windows * ctx->planes * ctx->plane_blocks * packed_bytes},
        {&w.win_pts, bucket_windows > 1 ? (size_t)bucket_windows * packed_bytes : 0},
    };
    for (int k = 0; k < MSM_WORK_PARTS; ++k) parts[k] = src[k];
}
template <class C> static int msm_alloc_work(plk_msm_ctx* ctx, MsmWork& w, hipStream_t stream) {
    WorkPart parts[MSM_WORK_PARTS];
    msm_work_parts<C>(ctx, w, parts);
    size_t total = 0;
    for (int k = 0; k < MSM_WORK_PARTS; ++k) {
        if (parts[k].bytes < w.cap[k]) parts[k].bytes = w.cap[k];  // reserved for the other generator counts of a rebound context
        total += (parts[k].bytes + 255) & ~(size_t)255;
    }
    ctx->ws_bytes = total + 256;
    // A table-free context lives for one call (msm_parallel, an IPA round): its memory comes from the scratch pool, because
    // hipMalloc + hipFree of a few hundred MB cost as much as a tenth of the MSM itself (0.4 ms of 3.8 at 2^20).
    w.pooled = ctx->table_free;
    if (w.pooled) {
        w.slab = scratch_acquire(total + 256, stream);
        if (!w.slab) return PLK_ERR_OOM;
        w.last_stream = stream;
    } else {
        PLK_HIP_TRY(hipMalloc(&w.slab, total + 256));
    }
    uint8_t* cur = (uint8_t*)w.slab;
    for (int k = 0; k < MSM_WORK_PARTS; ++k) {
        *parts[k].p = parts[k].bytes ? cur : nullptr;
        w.cap[k] = parts[k].bytes;
        cur += (parts[k].bytes + 255) & ~(size_t)255;
    }
    // the "last block" counter of k_ord_scan1 and the heavy-list counters start at zero and are left at zero by their users
    // (on the caller's stream: a non-blocking stream is not ordered after the null stream)
    PLK_HIP_TRY(hipMemsetAsync(w.meta, 0, (size_t)(1024 + 1025 + 1025 + 8) * 4, stream));
    PLK_HIP_TRY(hipMemsetAsync(w.heavy, 0, 8, stream));
    PLK_HIP_TRY(hipEventCreateWithFlags(&w.ev, hipEventDisableTiming));
    w.ready = true;
    return PLK_OK;
}

// entries per accumulation lane and what follows from it (lanes, heavy-bucket capacity, lanes per bucket in k_msm_assemble)
template <class C> static void msm_configure_lanes(plk_msm_ctx* ctx) {
    const size_t entries = ctx->n_eff * ctx->windows;
    // entries per accumulation lane: whole rounds of the lanes the GPU holds, at most 72 entries each (longer chunks: fewer pieces)
    {
        const size_t slots = accumulate_slots<C>();
        const double per_slot = (double)entries / (double)slots;
        size_t rounds = (size_t)(per_slot / 72.0 + 0.999);
        if (rounds < 1) rounds = 1;
        size_t ch = (size_t)(per_slot / (double)rounds + 0.999);
        if (ch < 8) ch = 8;
        if (ch > 96) ch = 96;
        ctx->chunk = (uint32_t)ch;
        if (const char* e = getenv("PLK_MSM_SLICE")) {
            int v = atoi(e);
            if (v >= 2 && v <= 4096) ctx->chunk = (uint32_t)v;
        }
    }
    ctx->max_lanes = entries / ctx->chunk + 2;
    // at most max_lanes / HEAVY_HEADS heavy buckets, max_lanes / HEAVY_CHUNK + that many chunk items
    ctx->heavy_cap = (uint32_t)(ctx->max_lanes / HEAVY_HEADS + ctx->max_lanes / HEAVY_CHUNK + 2);
    // lanes per bucket in k_msm_assemble: from the expected number of head pieces per bucket
    {
        const double heads = (double)entries / (double)ctx->buckets / (double)ctx->chunk;
        ctx->lpb_log = heads > 6.0 ? 3 : heads > 2.0 ? 2 : 0;
    }
}

template <class C> static void msm_launch_table(plk_msm_ctx* ctx, const void* d_bases, const void* d_zero, const void* d_extra, size_t n_extra,
                                                hipStream_t stream) {
    const size_t n = ctx->n;
    if (!n) return;
    static const bool no_split = getenv("PLK_MSM_TABLE_FUSED") != nullptr;
    if (!ctx->table_free && ctx->windows > 1 && n <= ((size_t)1 << 15) && !no_split) {
        constexpr size_t RAW = (size_t)raw_u4<typename C::FP>() * 16;
        const size_t entries = (size_t)(ctx->windows - 1) * n;
        if (uint4* raw = (uint4*)scratch_acquire(entries * RAW, stream)) {
            k_msm_table_chain<C><<<(unsigned)((4 * n + 63) / 64), 64, 0, stream>>>((const uint4*)d_bases, (const uint8_t*)d_zero, (uint4*)ctx->tab, raw, n,
                                                                                 ctx->c, ctx->windows, n - n_extra, (const uint4*)d_extra);
            k_msm_table_norm<C><<<(unsigned)((entries + 63) / 64), 64, 0, stream>>>(raw, (uint4*)ctx->tab, n, ctx->windows);
            scratch_release(raw, stream);
            return;
        }
        (void)hipGetLastError();  // no scratch memory: the fused kernel needs none
    }
    k_msm_table<C><<<(unsigned)((n + 127) / 128), 128, 0, stream>>>((const uint4*)d_bases, (const uint8_t*)d_zero, (uint4*)ctx->tab, n, ctx->c,
                                                                   ctx->table_free ? 1 : ctx->windows, ctx->glv ? 1 : 0, n - n_extra, (const uint4*)d_extra);
}

static int msm_configure(plk_msm_ctx* ctx, int curve, size_t n, unsigned window_bits, bool table_free);
template <class C>
static int msm_precompute_t(plk_msm_ctx* ctx, const void* d_bases, const void* d_zero, const void* d_extra, size_t n_extra, hipStream_t stream,
                            const size_t* also_n, int also_count) {
    using FP = typename C::FP;
    const size_t pt_bytes = (size_t)2 * FP::NL * 4;
    // Few generators with an automatic window are a COMB (comb.hip) - chosen BY SIZE since round 5: measured in round 4
    // (profiles/r04_comb_small_msm.txt) an execution takes 0.19 / 0.25 ms against 0.30 / 0.33 for the bucket method at 2^10 / 2^12
    // generators, the same at 2^14 and more at 2^15 (eight dependent gathers + additions per lane and two trees of full additions are
    // latency too), and its table costs 2-4 x the window tables to build: up to COMB_AUTO_MAX_N = 2^12 generators it is the default.
    // PLK_MSM_COMB=0 turns it off, PLK_MSM_COMB=1 forces it up to COMB_MAX_N (the parity suites run through it that way too).  The frozen
    // generators of an opening argument (2^14 + 2, explicit window) stay on the bucket method.
    const char* comb_env = getenv("PLK_MSM_COMB");
    const int comb_mode = comb_env ? atoi(comb_env) : -1;
    const bool want_comb = comb_mode > 0 || (comb_mode < 0 && ctx->n <= COMB_AUTO_MAX_N);
    if (!ctx->table_free && ctx->auto_window && ctx->n >= 1 && ctx->n <= COMB_MAX_N && want_comb) {
        // the doubling chain [2^(4 j)] G_i on quads (window 0 affine in `base0`, the others raw in `raw`), then comb.hip turns every
        // window's point into its multiples 1 .. 8
        const size_t n = ctx->n;
        constexpr size_t RAW = (size_t)raw_u4<FP>() * 16;
        uint4* base0 = (uint4*)scratch_acquire(n * pt_bytes + 16, stream);
        uint4* raw = (uint4*)scratch_acquire((size_t)(COMB_WINDOWS - 1) * n * RAW + 16, stream);
        int rc = (base0 && raw) ? PLK_OK : PLK_ERR_OOM;
        if (rc == PLK_OK) {
            k_msm_table_chain<C><<<(unsigned)((4 * n + 63) / 64), 64, 0, stream>>>((const uint4*)d_bases, (const uint8_t*)d_zero, base0, raw, n, COMB_WINDOW,
                                                                                 COMB_WINDOWS, n - n_extra, (const uint4*)d_extra);
            if (hipGetLastError() != hipSuccess) rc = set_error(PLK_ERR_HIP, "comb chain launch failed");
        }
        if (rc == PLK_OK) rc = comb_build(ctx->curve, n, base0, raw, stream, &ctx->comb);
        if (raw) scratch_release(raw, stream);
        if (base0) scratch_release(base0, stream);
        // The comb's table is ~20 x the window tables it replaces (32 KiB per generator, 48 for BLS12-377: 128-192 MiB at 2^12).  A host
        // that holds many small contexts can run out of HBM on it: when the comb was the library's own choice and its
        // table cannot be had, the context falls back to the bucket method below instead of failing.
        if (rc == PLK_ERR_OOM && comb_mode < 0) {
            (void)hipGetLastError();
            ctx->comb = nullptr;
        } else {
            PLK_TRY(rc);
            ctx->c = COMB_WINDOW;
            ctx->windows = COMB_WINDOWS;
            PLK_HIP_TRY(hipStreamSynchronize(stream));
            return PLK_OK;
        }
    }
    ctx->ws.resize(1);
    size_t tab_min = 0;
    if (also_count > 0) {
        // the context will be rebound to these generator counts (msm_rebind_dev_impl): every part of the workspace is
        // sized for the largest need over all of them - the parts are not monotonic in n (the window, hence the bucket and
        // tile counts, changes with it)
        const size_t n_own = ctx->n;
        MsmWork probe;
        WorkPart parts[MSM_WORK_PARTS];
        for (int a = 0; a < also_count; ++a) {
            if (msm_configure(ctx, ctx->curve, also_n[a], 0, ctx->table_free) != PLK_OK) continue;
            msm_configure_lanes<C>(ctx);
            msm_work_parts<C>(ctx, probe, parts);
            for (int k = 0; k < MSM_WORK_PARTS; ++k)
                if (parts[k].bytes > ctx->ws[0].cap[k]) ctx->ws[0].cap[k] = parts[k].bytes;
            if (ctx->n_eff * pt_bytes + 16 > tab_min) tab_min = ctx->n_eff * pt_bytes + 16;
        }
        PLK_TRY(msm_configure(ctx, ctx->curve, n_own, 0, ctx->table_free));
    }
    const size_t entries = ctx->n_eff * ctx->windows;
    msm_configure_lanes<C>(ctx);
    if (ctx->table_free) {
        ctx->tab_cap = ctx->n_eff * pt_bytes + 16;
        if (ctx->tab_cap < tab_min) ctx->tab_cap = tab_min;
        ctx->tab = scratch_acquire(ctx->tab_cap, stream);
        if (!ctx->tab) return PLK_ERR_OOM;
        ctx->tab_stream = stream;
    } else {
        ctx->tab_cap = entries * pt_bytes + 16;
        PLK_HIP_TRY(hipMalloc(&ctx->tab, ctx->tab_cap));
    }
    PLK_TRY(msm_alloc_work<C>(ctx, ctx->ws[0], stream));
    msm_launch_table<C>(ctx, d_bases, d_zero, d_extra, n_extra, stream);
    PLK_HIP_TRY(hipGetLastError());
    PLK_HIP_TRY(hipStreamSynchronize(stream));
    return PLK_OK;
}

// A table-free context re-used for another (smaller or equal) generator set: new geometry, same memory, no allocation, no
// synchronisation - the rounds of an inner-product argument halve their generators every time (halo.rs:63-124).
template <class C>
static int msm_rebind_t(plk_msm_ctx* ctx, const void* d_bases, const void* d_zero, const void* d_extra, size_t n_extra, hipStream_t stream) {
    using FP = typename C::FP;
    const size_t pt_bytes = (size_t)2 * FP::NL * 4;
    msm_configure_lanes<C>(ctx);
    if (ctx->n_eff * pt_bytes + 16 > ctx->tab_cap) return set_error(PLK_ERR_INVALID_ARG, "rebind: %zu points do not fit the context's table", ctx->n_eff);
    MsmWork& w = ctx->ws[0];
    WorkPart parts[MSM_WORK_PARTS];
    MsmWork probe;  // only its field addresses are used
    msm_work_parts<C>(ctx, probe, parts);
    for (int k = 0; k < MSM_WORK_PARTS; ++k)
        if (parts[k].bytes > w.cap[k])
            return set_error(PLK_ERR_INVALID_ARG, "rebind: workspace part %d needs %zu bytes, the context holds %zu", k, parts[k].bytes, w.cap[k]);
    if (w.used && w.last_stream != stream) PLK_HIP_TRY(hipStreamWaitEvent(stream, w.ev, 0));
    msm_launch_table<C>(ctx, d_bases, d_zero, d_extra, n_extra, stream);
    PLK_HIP_TRY(hipGetLastError());
    ctx->tab_stream = stream;
    w.last_stream = stream;
    return PLK_OK;
}

// table-free window: windows * 2^(c-1) bucket slots, long chunks wanted.  A top window of one or two bits (131 = 13 * 10 + 1)
// would put every scalar's top digit into a handful of buckets: a neighbouring width is taken instead.
static int choose_window_table_free(size_t n, int bits) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) ++lg;
    int c = lg - 5;
    if (c < 3) c = 3;
    if (c > MSM_TF_MAX_WINDOW) c = MSM_TF_MAX_WINDOW;
    auto top = [&](int w) { return bits - ((bits + w - 1) / w - 1) * w; };
    if (top(c) < 3) {
        if (c + 1 <= MSM_TF_MAX_WINDOW && top(c + 1) >= 3) c = c + 1;
        else if (c - 1 >= 3 && top(c - 1) >= 3) c = c - 1;
    }
    if (const char* e = getenv("PLK_MSM_WINDOW_TF")) c = atoi(e);
    if (c < 3) c = 3;
    if (c > MSM_TF_MAX_WINDOW) c = MSM_TF_MAX_WINDOW;
    return c;
}

// window, ordering configuration and tail geometry of a context over n generators (no device work, no allocation)
static int msm_configure(plk_msm_ctx* ctx, int curve, size_t n, unsigned window_bits, bool table_free) {
    // table-free mode on the prime-order curves: split every scalar along the endomorphism (glv.cuh) - 2n points, half the windows
    const bool glv = table_free && n > 0 && curve != PLK_CURVE_BLS12_377 && !getenv("PLK_MSM_NO_GLV");
    const size_t n_eff = glv ? 2 * n : n;
    int c = window_bits ? (int)window_bits : (table_free ? choose_window_table_free(n_eff ? n_eff : 1, (glv ? GLV_BITS : scalar_bits(curve)) + 1) : choose_window(n ? n : 1, curve));
    if (c < 2 || c > MSM_MAX_WINDOW) return set_error(PLK_ERR_INVALID_ARG, "window_bits %d outside [2, %d]", c, MSM_MAX_WINDOW);
    const int windows = ((glv ? GLV_BITS : scalar_bits(curve)) + 1 + c - 1) / c;
    if (table_free && c > MSM_TF_MAX_WINDOW)
        return set_error(PLK_ERR_INVALID_ARG, "table-free mode: window_bits %d above %d", c, MSM_TF_MAX_WINDOW);
    if (table_free) {
        // bit-plane reduction over the buckets themselves up to 12 bits, the two-level reduction per window above
        const size_t slot_limit = c - 1 >= 12 ? (size_t)ORD_MAX_BINS << ORD_MAX_FINE : 65536;
        const int window_limit = c - 1 >= 12 ? COMBINE_THREADS / 8 : COMBINE_THREADS / 4;
        if (((size_t)windows << (c - 1)) > slot_limit || windows > window_limit)
            return set_error(PLK_ERR_INVALID_ARG, "table-free mode: window_bits %d gives %d windows x %d buckets (limits: %zu slots, %d windows)", c, windows,
                             1 << (c - 1), slot_limit, window_limit);
    }
    if (n_eff * (size_t)windows >= ((size_t)1 << 31))
        return set_error(PLK_ERR_INVALID_ARG, "n * windows = %zu entries exceeds 2^31", n_eff * (size_t)windows);
    ctx->table_free = table_free;
    ctx->curve = curve;
    ctx->n = n;
    ctx->glv = glv;
    ctx->n_eff = n_eff;
    ctx->c = c;
    ctx->windows = windows;
    ctx->wbuckets = 1u << (c - 1);
    {
        // partition geometry from the number of bucket slots: <= 512 coarse bins (one workgroup each at level 2), the rest fine
        const uint32_t want = ctx->table_free ? ctx->wbuckets * (uint32_t)ctx->windows : ctx->wbuckets;
        const int bits = ilog2_ceil(want);
        // up to 2^10 bucket slots: ONE level - the coarse bins are the buckets, the first level's output is the bucket order
        int coarse = bits <= 10 ? bits : 9;
        if (bits - coarse > ORD_MAX_FINE) coarse = bits - ORD_MAX_FINE;
        OrdCfg& o = ctx->ord;
        o.c = c;
        o.windows = windows;
        o.window_buckets = ctx->table_free ? ctx->wbuckets : 0u;
        o.fine_bits = bits - coarse;
        o.nbins = (int)((want + (1u << o.fine_bits) - 1) >> o.fine_bits);
        ctx->buckets = (uint32_t)o.nbins << o.fine_bits;
        o.spt = (uint32_t)(ORD_TILE / windows);
        if (o.spt > (uint32_t)ORD_THREADS) o.spt = ORD_THREADS;
        o.sub = n_eff >= ((size_t)1 << 16) ? 4 : 1;
        o.nt1 = (uint32_t)((n_eff + (size_t)o.spt * o.sub - 1) / ((size_t)o.spt * o.sub));
        if (o.nt1 == 0) o.nt1 = 1;
        o.raw_signed = glv ? 1 : 0;
        o.entries_cap = (uint32_t)(n_eff * (size_t)windows);
        o.ent_stride = (uint32_t)n_eff;
        o.ent_first = 0;
        // round 6: the tile-major level 1 with the bins taken from the LOW bits of the bucket number (OrdCfg::perm) - tabled contexts whose
        // buckets split into at least as many fine as coarse bits (c = 19 .. 21: the 2^19 generators and up that get such windows), tiles
        // of 1024 scalars, records of at most 16 windows.  PLK_MSM_ORDER_V1 keeps round 5's kernels (A/B, tests/test_gpu_knobs.py).
        static const bool order_v1 = getenv("PLK_MSM_ORDER_V1") != nullptr;
        // ... and a bin's expected share of the entries fits the LDS of k_ord_bin_sort with 15 % to spare (2^20 scalars of 13 windows over 512
        // bins: 26.6 k of 32 k; larger problems keep round 5's kernels, hot bins of a skewed vector take the segmented ones).
        const bool bins_fit = (double)n_eff * windows / (double)o.nbins * 1.15 <= (double)ORD2_BIN_CAP;
        o.bin_lo = 0;
        o.bin_hi = (uint32_t)o.nbins;
        o.perm = (!order_v1 && !table_free && coarse == 9 && o.fine_bits >= coarse && o.nbins == (1 << coarse) && o.sub == 4 && o.spt * o.sub == 1024u &&
                  windows <= 16 && o.nt1 <= 2048u && bins_fit)
                     ? 1
                     : 0;
    }
    // tail geometry
    ctx->two_level = c - 1 >= 12;
    ctx->L = ctx->H = ctx->g_log = ctx->lpl_log = 0;
    if (ctx->two_level) {
        ctx->L = (c - 1) / 2;
        ctx->H = c - 1 - ctx->L;
        if (ctx->ord.perm) {
            // the bucket slots are numbered [coarse bin = LOW bits of the bucket | fine = its high bits]: the weighting splits where the
            // ordering does (TailGeom::transposed)
            ctx->L = c - 1 - ctx->ord.fine_bits;
            ctx->H = ctx->ord.fine_bits;
        }
        ctx->g_log = c - 1 >= 17 ? 3 : 2;
        if (const char* e = getenv("PLK_MSM_GLOG")) ctx->g_log = atoi(e);
        if (ctx->g_log > ctx->L) ctx->g_log = ctx->L;
        if (ctx->g_log < 0) ctx->g_log = 0;
        const int longest = ctx->H - ctx->g_log;  // log2 of the partials per column (rows have L - g_log <= that)
        ctx->lpl_log = longest < 4 ? longest : 4;  // quads per line
        ctx->tail_windows = ctx->table_free ? 2 * ctx->windows : 2;  // per real window: its column sums, then its row sums
        ctx->tail_wbuckets = 1u << ctx->H;
        ctx->tail_shift = c;  // between real windows (table-free mode); the row sums of a window weigh 2^L more (k_msm_final)
        ctx->planes = ctx->H;  // weights up to 2^H - 1 (rows) / 2^L (columns): plane H - 1 is the top one for rows; columns need bit L <= H - 1 or L == H
        if (ctx->L == ctx->H) ctx->planes = ctx->H + 1;  // column weight 2^L = 2^H needs plane H
    } else {
        ctx->tail_windows = ctx->table_free ? ctx->windows : 1;
        ctx->tail_wbuckets = ctx->wbuckets;
        ctx->tail_shift = c;
        ctx->planes = c;
    }
    ctx->plane_blocks = 1;
    while (ctx->plane_blocks < MSM_MAX_PLANE_PARTS && (uint32_t)ctx->plane_blocks * 512u < ctx->tail_wbuckets &&
           ctx->planes * ctx->plane_blocks * 4 <= FINAL_THREADS)  // after doubling: planes * parts / 2 quads in the final block
        ctx->plane_blocks *= 2;
    return PLK_OK;
}

int msm_precompute_dev_impl(int curve, size_t n, const void* d_bases, const void* d_zero, unsigned window_bits, unsigned flags, hipStream_t stream,
                            plk_msm_ctx** out_ctx, const void* d_extra, size_t n_extra, const size_t* also_n, int also_count) {
    if (!out_ctx) return set_error(PLK_ERR_INVALID_ARG, "null out_ctx");
    *out_ctx = nullptr;
    if (curve_limbs(curve) < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (n > n_extra && !d_bases) return set_error(PLK_ERR_INVALID_ARG, "null bases");
    if (n_extra > n || (n_extra && !d_extra)) return set_error(PLK_ERR_INVALID_ARG, "bad extra generators");
    PLK_TRY(ensure_device());
    int dev = 0;
    PLK_HIP_TRY(hipGetDevice(&dev));
    auto* ctx = new plk_msm_ctx();
    ctx->device = dev;
    ctx->auto_window = window_bits == 0 && also_count == 0;
    int rc = msm_configure(ctx, curve, n, window_bits, (flags & PLK_MSM_TABLE_FREE) != 0);
    if (rc == PLK_OK) switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: rc = msm_precompute_t<TweedledeeCurve>(ctx, d_bases, d_zero, d_extra, n_extra, stream, also_n, also_count); break;
        case PLK_CURVE_TWEEDLEDUM: rc = msm_precompute_t<TweedledumCurve>(ctx, d_bases, d_zero, d_extra, n_extra, stream, also_n, also_count); break;
        case PLK_CURVE_PALLAS: rc = msm_precompute_t<PallasCurve>(ctx, d_bases, d_zero, d_extra, n_extra, stream, also_n, also_count); break;
        case PLK_CURVE_VESTA: rc = msm_precompute_t<VestaCurve>(ctx, d_bases, d_zero, d_extra, n_extra, stream, also_n, also_count); break;
        default: rc = msm_precompute_t<Bls12377Curve>(ctx, d_bases, d_zero, d_extra, n_extra, stream, also_n, also_count); break;
    }
    if (rc != PLK_OK) {
        delete ctx;
        return rc;
    }
    *out_ctx = ctx;
    return PLK_OK;
}

// see msm_rebind_t.  n counts the extra generators (the last n_extra of the n points come from d_extra).
int msm_rebind_dev_impl(plk_msm_ctx* ctx, size_t n, const void* d_bases, const void* d_zero, const void* d_extra, size_t n_extra, hipStream_t stream) {
    if (!ctx || !ctx->table_free || ctx->ws.empty()) return set_error(PLK_ERR_INVALID_ARG, "rebind needs a table-free context");
    if (n_extra > n || (n > n_extra && !d_bases) || (n_extra && !d_extra)) return set_error(PLK_ERR_INVALID_ARG, "bad generators");
    PLK_TRY(ensure_device());
    std::lock_guard<std::mutex> lk(ctx->mu);
    const int curve = ctx->curve;
    PLK_TRY(msm_configure(ctx, curve, n, 0, true));
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: return msm_rebind_t<TweedledeeCurve>(ctx, d_bases, d_zero, d_extra, n_extra, stream);
        case PLK_CURVE_TWEEDLEDUM: return msm_rebind_t<TweedledumCurve>(ctx, d_bases, d_zero, d_extra, n_extra, stream);
        case PLK_CURVE_PALLAS: return msm_rebind_t<PallasCurve>(ctx, d_bases, d_zero, d_extra, n_extra, stream);
        case PLK_CURVE_VESTA: return msm_rebind_t<VestaCurve>(ctx, d_bases, d_zero, d_extra, n_extra, stream);
        default: return msm_rebind_t<Bls12377Curve>(ctx, d_bases, d_zero, d_extra, n_extra, stream);
    }
}

static uint32_t* head_bucket_of(const plk_msm_ctx* ctx, const MsmWork& w) {
    return (uint32_t*)((uint8_t*)w.head_live + head_lanes_padded(ctx->max_lanes));
}
static uint32_t* live_list_of(const plk_msm_ctx* ctx, const MsmWork& w) {
    return (uint32_t*)((uint8_t*)w.head_live + 5 * head_lanes_padded(ctx->max_lanes));
}
static TailSlot tail_slot(const plk_msm_ctx* ctx, const MsmWork& w, void* d_out_xy, void* d_out_zero) {
    TailSlot t;
    t.off = (const uint32_t*)w.off;
    t.p_start = (uint4*)w.p_start;
    t.p_head = (const uint4*)w.p_head;
    t.head_live = (const uint8_t*)w.head_live;
    t.head_bucket = head_bucket_of(ctx, w);
    t.live_list = live_list_of(ctx, w);
    t.live_count = (uint32_t*)w.meta + (1024 + 1025 + 1025 + 3);
    t.bucket = (uint4*)w.bucket;
    t.heavy = (uint32_t*)w.heavy;
    t.heavy_part = (uint4*)w.heavy_part;
    t.line_part = (uint4*)w.line_part;
    t.plane_part = (uint4*)w.plane_part;
    t.win_pts = (uint4*)w.win_pts;
    t.final_done = (uint32_t*)w.meta + (1024 + 1025 + 1025 + 1);
    t.dyn_chunk = (const uint32_t*)w.meta + (1024 + 1025 + 1025 + 2);
    t.out_xy = (uint4*)d_out_xy;
    t.out_zero = (uint8_t*)d_out_zero;
    t.projective = ctx->out_projective ? 1 : 0;
    return t;
}

// pieces -> buckets -> (row / column sums ->) bit-plane sums -> result, for the tb.count MSMs of a batch at once (msm_tail.hip)
template <class C, class Mark>
static int msm_reduce_t(plk_msm_ctx* ctx, const TailBatch& tb, hipStream_t stream, Mark&& mark) {
    TailGeom g;
    g.buckets = ctx->buckets; g.heavy_cap = ctx->heavy_cap; g.tail_wbuckets = ctx->tail_wbuckets; g.max_lanes = (uint32_t)ctx->max_lanes;
    g.lpb_log = ctx->lpb_log; g.two_level = ctx->two_level ? 1 : 0; g.L = ctx->L; g.H = ctx->H; g.g_log = ctx->g_log; g.lpl_log = ctx->lpl_log;
    g.table_free = ctx->table_free ? 1 : 0; g.windows = ctx->windows; g.tail_windows = ctx->tail_windows; g.plane_blocks = ctx->plane_blocks;
    g.planes = ctx->planes; g.tail_shift = ctx->tail_shift; g.transposed = ctx->ord.perm; g.many_heads = ctx->many_heads ? 1 : 0;
    for (int stage = 0; stage < 3; ++stage) {
        PLK_TRY(msm_launch_reduce_stage<C>(stage, g, tb, stream));
        mark();
    }
    return PLK_OK;
}


// phases: 1 = bucket ordering, 2 = accumulation, 4 = reduction; 7 = the whole MSM on one stream
constexpr int PH_ORDER = 1, PH_ACC = 2, PH_REDUCE = 4, PH_ALL = 7;
template <class C>
static int msm_execute_t(plk_msm_ctx* ctx, MsmWork& w, const void* d_scalars, void* d_out_xy, void* d_out_zero, hipStream_t stream,
                         int phases = PH_ALL, size_t first = 0, size_t count = (size_t)-1, uint32_t bucket_part = 0, uint32_t bucket_parts = 1) {
    // count != -1: the scalars belong to generators first .. first + count - 1 (tabled contexts: the table index of an entry is its id)
    const bool ranged = count != (size_t)-1;
    const size_t n = ranged ? count : ctx->n_eff;
    const uint32_t buckets = ctx->buckets;
    OrdCfg o = ctx->ord;
    if (ranged) {
        o.nt1 = (uint32_t)((n + (size_t)o.spt * o.sub - 1) / ((size_t)o.spt * o.sub));
        if (o.nt1 == 0) o.nt1 = 1;
        o.ent_first = (uint32_t)first;
    }
    if (bucket_parts > 1) {  // this execution's range of the coarse bins (the larger ranges first, like shard_bounds)
        const uint32_t nb = (uint32_t)o.nbins, base = nb / bucket_parts, rem = nb % bucket_parts;
        o.bin_lo = bucket_part * base + (bucket_part < rem ? bucket_part : rem);
        o.bin_hi = o.bin_lo + base + (bucket_part < rem ? 1u : 0u);
    }
    uint32_t* off = (uint32_t*)w.off;
    uint32_t* bin_total = (uint32_t*)w.meta;
    uint32_t* bin_base = bin_total + 1024;
    uint32_t* seg_base = bin_base + 1025;
    uint32_t* done_counter = seg_base + 1025;
    // the workspace may still be in use by an execution enqueued on another stream
    if (w.used && w.last_stream != stream) PLK_HIP_TRY(hipStreamWaitEvent(stream, w.ev, 0));
    std::vector<hipEvent_t> ev;
    if (ctx->profiling && phases == PH_ALL) {
        if (!ctx->prof_free.empty()) {
            ev = ctx->prof_free.back();
            ctx->prof_free.pop_back();
        } else {
            ev.resize(plk_msm_ctx::N_STAGES + 1);
            for (auto& e : ev) PLK_HIP_TRY(hipEventCreate(&e));
        }
    }
    int stage = 0;
    auto mark = [&]() {
        if (!ev.empty()) (void)hipEventRecord(ev[stage], stream);
        ++stage;
    };
    mark();
    if (phases & PH_ORDER) {
        void* halves = nullptr;
        if (ctx->glv) {
            // the two half scalars of every scalar (stream-ordered scratch: handed back once the ordering kernels are enqueued)
            halves = scratch_acquire(n * 32, stream);
            if (!halves) return PLK_ERR_OOM;
            PLK_TRY(msm_launch_glv_split<C>(d_scalars, ctx->n, halves, stream));
            d_scalars = halves;
        }
        // the pooled buffers of this call go back on every path out of it
        struct Guard {
            void*& halves;
            std::vector<hipEvent_t>& ev;
            plk_msm_ctx* ctx;
            hipStream_t stream;
            bool armed = true;
            ~Guard() {
                if (!armed) return;
                if (halves) scratch_release(halves, stream);
                if (!ev.empty()) ctx->prof_free.push_back(ev);
            }
        } guard{halves, ev, ctx, stream};
        OrdBuffers ob;
        ob.scalars = d_scalars; ob.n = n; ob.cnt1 = w.cnt1; ob.tmp = w.tmp; ob.sorted = w.sorted; ob.cnt2 = w.cnt2; ob.off = off;
        ob.bin_total = bin_total; ob.bin_base = bin_base; ob.seg_base = seg_base; ob.done_counter = done_counter;
        ob.chunk = ctx->chunk; ob.lanes = (uint32_t)(ctx->max_lanes > 2 ? ctx->max_lanes - 2 : 1); ob.buckets = buckets;
        PLK_TRY(msm_launch_order_stage<C>(0, o, ob, stream));
        mark();
        PLK_TRY(msm_launch_order_stage<C>(1, o, ob, stream));
        mark();
        PLK_TRY(msm_launch_order_stage<C>(2, o, ob, stream));
        PLK_HIP_TRY(hipGetLastError());
        guard.armed = false;
        if (halves) scratch_release(halves, stream);
        mark();
    } else {
        stage += 3;
    }
    if (phases & PH_ACC) {
        // the entry count is only known on the device: launch for the upper bound, lanes past it exit
        const unsigned ablocks = (unsigned)((ctx->max_lanes + ACC_THREADS - 1) / ACC_THREADS);
        msm_launch_accumulate<C>(ablocks, stream, ctx->tab, w.sorted, off, w.p_start, w.p_head, w.head_live, head_bucket_of(ctx, w), live_list_of(ctx, w), done_counter + 3, buckets, done_counter + 2,
                                 ctx->table_free ? ctx->c - 1 : 31, ctx->table_free ? (uint32_t)n : 0u,
                                 // tabled: sorted[] entries index the table; table-free: the table holds the n_eff points, sorted[] the entries
                                 ctx->table_free ? (uint32_t)ctx->n_eff : o.entries_cap);
        PLK_HIP_TRY(hipGetLastError());
    }
    mark();
    if (phases & PH_REDUCE) {
        TailBatch tb;
        tb.count = 1;
        tb.s[0] = tail_slot(ctx, w, d_out_xy, d_out_zero);
        PLK_TRY(msm_reduce_t<C>(ctx, tb, stream, mark));
        if (!ev.empty()) ctx->prof_sets.push_back(ev);
    }
    return PLK_OK;
}

static void work_done(MsmWork& w, hipStream_t stream) {
    (void)hipEventRecord(w.ev, stream);
    w.last_stream = stream;
    w.used = true;
}

// ready (optional): one event per scalar vector; vector b is not touched before ready[b] has completed (the host-pointer entry
// point copies vector b + 1 through PCIe while vector b is being reduced)
int msm_execute_dev_impl(plk_msm_ctx* ctx, unsigned batch, const void* d_scalars, size_t n_scalars, void* d_out_xy, void* d_out_zero, hipStream_t stream,
                         hipEvent_t* ready, const MsmParts* parts, unsigned out_flags) {
    // out_flags bit 0: results as the reference's un-normalised ProjectivePoint, x | y | z (3 L limbs per vector), not the affine point
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    if (parts) {
        // vector b: parts->count[b] scalars at parts->scalars[b] for the generators parts->first[b] .. (plk_msm_execute_parts_dev)
        if (ctx->table_free) return set_error(PLK_ERR_INVALID_ARG, "a sub-range of the generators needs a tabled context");
        if (!parts->first || !parts->count || !parts->scalars) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
        if ((parts->bucket_part == nullptr) != (parts->bucket_parts == nullptr)) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
        for (unsigned b = 0; b < batch && parts->bucket_parts; ++b) {
            if (parts->bucket_parts[b] > 1 && (ctx->comb || parts->bucket_part[b] >= parts->bucket_parts[b] || parts->bucket_parts[b] > (uint32_t)ctx->ord.nbins))
                return set_error(PLK_ERR_INVALID_ARG, "vector %u: bucket range %u of %u over %d coarse bins%s", b, parts->bucket_part[b], parts->bucket_parts[b],
                                 ctx->ord.nbins, ctx->comb ? " (a comb context has no buckets)" : "");
        }
        for (unsigned b = 0; b < batch; ++b) {
            if (parts->first[b] > ctx->n || parts->count[b] > ctx->n - parts->first[b])
                return set_error(PLK_ERR_SIZE_MISMATCH, "vector %u covers generators %llu .. +%llu but the precomputation holds %zu", b,
                                 (unsigned long long)parts->first[b], (unsigned long long)parts->count[b], ctx->n);
            if (parts->count[b] && !parts->scalars[b]) return set_error(PLK_ERR_INVALID_ARG, "null scalars in batch slot %u", b);
        }
    } else if (n_scalars != ctx->n) {
        return set_error(PLK_ERR_SIZE_MISMATCH, "scalars.len() = %zu but the precomputation holds %zu generators (curve_msm.rs:67)", n_scalars, ctx->n);
    }
    if (batch == 0) return PLK_OK;
    if ((!parts && ctx->n && !d_scalars) || !d_out_xy || !d_out_zero) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    PLK_HIP_TRY(hipSetDevice(ctx->device));  // a context works on the device it was built on, whichever device the thread last used
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t L = (size_t)curve_limbs(ctx->curve);
    ctx->many_heads = false;
    for (unsigned b = 0; b < batch && parts && parts->bucket_parts; ++b) ctx->many_heads = ctx->many_heads || parts->bucket_parts[b] > 1;
    const bool projective = (out_flags & 1u) != 0;
    ctx->out_projective = projective;
    const size_t out_stride = (projective ? 3 : 2) * L * 8;
    if (ctx->comb) {
        if (projective) return set_error(PLK_ERR_INVALID_ARG, "a comb context returns affine points (the caller expands them)");
        // few generators: additions of table entries and a tree, two launches for the whole batch (comb.hip)
        std::vector<const void*> ptr(batch);
        std::vector<uint64_t> first(batch), count(batch);
        for (unsigned b = 0; b < batch; ++b) {
            if (ready) PLK_HIP_TRY(hipStreamWaitEvent(stream, ready[b], 0));
            ptr[b] = parts ? parts->scalars[b] : (const void*)((const uint8_t*)d_scalars + (size_t)b * ctx->n * 32);
            first[b] = parts ? parts->first[b] : 0;
            count[b] = parts ? parts->count[b] : ctx->n;
        }
        return comb_execute(ctx->comb, batch, ptr.data(), first.data(), count.data(), d_out_xy, d_out_zero, stream);
    }
    auto run_one = [&](unsigned b, MsmWork& w, hipStream_t st, int phases) -> int {
        if (ready && (phases & PH_ORDER)) PLK_HIP_TRY(hipStreamWaitEvent(st, ready[b], 0));
        const uint8_t* sc = parts ? (const uint8_t*)parts->scalars[b] : (const uint8_t*)d_scalars + (size_t)b * ctx->n * 32;
        const size_t first = parts ? (size_t)parts->first[b] : 0, count = parts ? (size_t)parts->count[b] : (size_t)-1;
        const uint32_t bp = (parts && parts->bucket_parts) ? parts->bucket_part[b] : 0u, bps = (parts && parts->bucket_parts) ? parts->bucket_parts[b] : 1u;
        uint8_t* oxy = (uint8_t*)d_out_xy + (size_t)b * out_stride;
        uint8_t* oz = (uint8_t*)d_out_zero + b;
        switch (ctx->curve) {
            case PLK_CURVE_TWEEDLEDEE: return msm_execute_t<TweedledeeCurve>(ctx, w, sc, oxy, oz, st, phases, first, count, bp, bps);
            case PLK_CURVE_TWEEDLEDUM: return msm_execute_t<TweedledumCurve>(ctx, w, sc, oxy, oz, st, phases, first, count, bp, bps);
            case PLK_CURVE_PALLAS: return msm_execute_t<PallasCurve>(ctx, w, sc, oxy, oz, st, phases, first, count, bp, bps);
            case PLK_CURVE_VESTA: return msm_execute_t<VestaCurve>(ctx, w, sc, oxy, oz, st, phases, first, count, bp, bps);
            default: return msm_execute_t<Bls12377Curve>(ctx, w, sc, oxy, oz, st, phases, first, count, bp, bps);
        }
    };
    static const bool no_batching = getenv("PLK_MSM_NO_OVERLAP") != nullptr;  // every MSM of a batch start to end, one by one
    if (batch == 1 || ctx->profiling || no_batching) {
        for (unsigned b = 0; b < batch; ++b) PLK_TRY(run_one(b, ctx->ws[0], stream, PH_ALL));
        work_done(ctx->ws[0], stream);
        return PLK_OK;
    }
    // Several scalar vectors against the same generators (commit_polynomials, plonk_util.rs:215-231), in groups of up
    // to TAIL_MAX: every MSM of a group has its own workspace, ordering and accumulation run one MSM after the other,
    // and the latency-bound reduction runs ONCE for the whole group - it is a chain of point operations on few points,
    // so nine of them cost little more than one.
    unsigned group = batch < (unsigned)TAIL_MAX ? batch : (unsigned)TAIL_MAX;
    const size_t budget = (size_t)16 << 30;  // bytes of workspace a batch may hold
    while (group > 2 && ctx->ws_bytes * group > budget) --group;
    while (ctx->ws.size() < group) {
        ctx->ws.emplace_back();
        int rc;
        switch (ctx->curve) {
            case PLK_CURVE_TWEEDLEDEE: rc = msm_alloc_work<TweedledeeCurve>(ctx, ctx->ws.back(), stream); break;
            case PLK_CURVE_TWEEDLEDUM: rc = msm_alloc_work<TweedledumCurve>(ctx, ctx->ws.back(), stream); break;
            case PLK_CURVE_PALLAS: rc = msm_alloc_work<PallasCurve>(ctx, ctx->ws.back(), stream); break;
            case PLK_CURVE_VESTA: rc = msm_alloc_work<VestaCurve>(ctx, ctx->ws.back(), stream); break;
            default: rc = msm_alloc_work<Bls12377Curve>(ctx, ctx->ws.back(), stream); break;
        }
        if (rc != PLK_OK) {
            ctx->ws.back().release();
            ctx->ws.pop_back();
            (void)hipGetLastError();
            group = (unsigned)ctx->ws.size();  // make do with what fits (at least the workspace of the precomputation)
            break;
        }
    }
    auto reduce = [&](const TailBatch& tb, hipStream_t st) -> int {
        auto nomark = [] {};
        switch (ctx->curve) {
            case PLK_CURVE_TWEEDLEDEE: return msm_reduce_t<TweedledeeCurve>(ctx, tb, st, nomark);
            case PLK_CURVE_TWEEDLEDUM: return msm_reduce_t<TweedledumCurve>(ctx, tb, st, nomark);
            case PLK_CURVE_PALLAS: return msm_reduce_t<PallasCurve>(ctx, tb, st, nomark);
            case PLK_CURVE_VESTA: return msm_reduce_t<VestaCurve>(ctx, tb, st, nomark);
            default: return msm_reduce_t<Bls12377Curve>(ctx, tb, st, nomark);
        }
    };
    // Pipelined reductions: the reduction of a vector is mostly latency (chains on few points) plus
    // 0.1 ms of full-width row / column sums; on a second stream it runs under the ordering and accumulation of the NEXT vector
    // instead of after the last one.  The caller's stream waits for the second stream before the call returns.
    // Measured (profiles/r03_commit9_scaling.txt): SLOWER than one shared reduction at the end - 12.85 against 12.38 ms for nine
    // 2^20 vectors, 2.52 against 2.21 ms for a rank's share of them at eight ranks: the accumulation is sized as exactly one round
    // of lanes, and every slot a reduction workgroup takes sends part of that round into a second one.  Off unless asked for.
    static const bool pipeline_tails = getenv("PLK_MSM_TAIL_PIPELINE") != nullptr;
    if (pipeline_tails) {
        if (!ctx->tail_stream && !(ctx->tail_stream = stream_pool_acquire())) return PLK_ERR_HIP;
        if (!ctx->ev_tail) PLK_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_tail, hipEventDisableTiming));
        while (ctx->ev_acc.size() < group) {
            hipEvent_t e = nullptr;
            PLK_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->ev_acc.push_back(e);
        }
        for (unsigned g0 = 0; g0 < batch; g0 += group) {
            const unsigned cnt = batch - g0 < group ? batch - g0 : group;
            // the workspaces of the previous group are free again when its reductions are done
            if (g0) {
                PLK_HIP_TRY(hipEventRecord(ctx->ev_tail, ctx->tail_stream));
                PLK_HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_tail, 0));
            }
            for (unsigned k = 0; k < cnt; ++k) {
                const unsigned b = g0 + k;
                PLK_TRY(run_one(b, ctx->ws[k], stream, PH_ORDER | PH_ACC));
                PLK_HIP_TRY(hipEventRecord(ctx->ev_acc[k], stream));
                PLK_HIP_TRY(hipStreamWaitEvent(ctx->tail_stream, ctx->ev_acc[k], 0));
                TailBatch tb;
                tb.count = 1;
                tb.s[0] = tail_slot(ctx, ctx->ws[k], (uint8_t*)d_out_xy + (size_t)b * out_stride, (uint8_t*)d_out_zero + b);
                PLK_TRY(reduce(tb, ctx->tail_stream));
            }
        }
        PLK_HIP_TRY(hipEventRecord(ctx->ev_tail, ctx->tail_stream));
        PLK_HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_tail, 0));
        for (unsigned k = 0; k < group && k < ctx->ws.size(); ++k) work_done(ctx->ws[k], stream);
        return PLK_OK;
    }
    // Small MSMs (the rounds of an inner-product argument over frozen generators: 2^14 points, two vectors): ordering and
    // accumulation are six short kernels per vector that leave most of the GPU idle - the vectors of a group run them side by side
    // on streams of their own and meet again for the shared reduction (0.52 -> 0.42 ms for such a round).
    static const bool no_fork = getenv("PLK_MSM_NO_FORK") != nullptr;
    const bool fork = !no_fork && group > 1 && ctx->n_eff * (size_t)ctx->windows <= ((size_t)1 << 21);
    if (fork) {
        if (!ctx->ev_tail) PLK_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_tail, hipEventDisableTiming));
        while (ctx->ev_acc.size() < group) {
            hipEvent_t e = nullptr;
            PLK_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->ev_acc.push_back(e);
        }
        while (ctx->fork_streams.size() + 1 < group) {
            hipStream_t st = stream_pool_acquire();
            if (!st) return PLK_ERR_HIP;
            ctx->fork_streams.push_back(st);
        }
    }
    for (unsigned g0 = 0; g0 < batch; g0 += group) {
        const unsigned cnt = batch - g0 < group ? batch - g0 : group;
        TailBatch tb;
        tb.count = (int)cnt;
        if (fork && cnt > 1) PLK_HIP_TRY(hipEventRecord(ctx->ev_tail, stream));  // the scalars (and the previous group's reduction) are ready
        for (unsigned k = 0; k < cnt; ++k) {
            const unsigned b = g0 + k;
            hipStream_t st = stream;
            if (fork && k > 0) {
                st = ctx->fork_streams[k - 1];
                PLK_HIP_TRY(hipStreamWaitEvent(st, ctx->ev_tail, 0));
            }
            PLK_TRY(run_one(b, ctx->ws[k], st, PH_ORDER | PH_ACC));
            if (st != stream) {
                PLK_HIP_TRY(hipEventRecord(ctx->ev_acc[k], st));
                PLK_HIP_TRY(hipStreamWaitEvent(stream, ctx->ev_acc[k], 0));
            }
            tb.s[k] = tail_slot(ctx, ctx->ws[k], (uint8_t*)d_out_xy + (size_t)b * out_stride, (uint8_t*)d_out_zero + b);
        }
        int rc;
        auto nomark = [] {};
        switch (ctx->curve) {
            case PLK_CURVE_TWEEDLEDEE: rc = msm_reduce_t<TweedledeeCurve>(ctx, tb, stream, nomark); break;
            case PLK_CURVE_TWEEDLEDUM: rc = msm_reduce_t<TweedledumCurve>(ctx, tb, stream, nomark); break;
            case PLK_CURVE_PALLAS: rc = msm_reduce_t<PallasCurve>(ctx, tb, stream, nomark); break;
            case PLK_CURVE_VESTA: rc = msm_reduce_t<VestaCurve>(ctx, tb, stream, nomark); break;
            default: rc = msm_reduce_t<Bls12377Curve>(ctx, tb, stream, nomark); break;
        }
        PLK_TRY(rc);
        for (unsigned k = 0; k < cnt; ++k) work_done(ctx->ws[k], stream);
    }
    return PLK_OK;
}

// affine results as ProjectivePoints with z = 1 (contexts and paths that normalise anyway: combs, device groups)
template <class FP> __global__ void k_affine_to_projective(const uint4* __restrict__ xy, const uint8_t* __restrict__ zero, uint4* __restrict__ xyz, unsigned batch) {
    constexpr int W = FP::NL / 4;
    const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const bool ident = zero[b] != 0;
    fe_store<FP>(xyz + (size_t)b * 3 * W, ident ? fe_zero<FP>() : fe_load<FP>(xy + (size_t)b * 2 * W));
    fe_store<FP>(xyz + (size_t)b * 3 * W + W, ident ? fe_zero<FP>() : fe_load<FP>(xy + (size_t)b * 2 * W + W));
    fe_store<FP>(xyz + (size_t)b * 3 * W + 2 * W, ident ? fe_zero<FP>() : fe_one<FP>());
}
int msm_affine_to_projective_impl(int curve, unsigned batch, const void* d_xy, const void* d_zero, void* d_xyz, hipStream_t stream) {
    if (batch == 0) return PLK_OK;
    const unsigned blocks = (batch + 63) / 64;
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: k_affine_to_projective<TweedledeeBaseParams><<<blocks, 64, 0, stream>>>((const uint4*)d_xy, (const uint8_t*)d_zero, (uint4*)d_xyz, batch); break;
        case PLK_CURVE_TWEEDLEDUM: k_affine_to_projective<TweedledumBaseParams><<<blocks, 64, 0, stream>>>((const uint4*)d_xy, (const uint8_t*)d_zero, (uint4*)d_xyz, batch); break;
        case PLK_CURVE_PALLAS: k_affine_to_projective<PallasBaseParams><<<blocks, 64, 0, stream>>>((const uint4*)d_xy, (const uint8_t*)d_zero, (uint4*)d_xyz, batch); break;
        case PLK_CURVE_VESTA: k_affine_to_projective<VestaBaseParams><<<blocks, 64, 0, stream>>>((const uint4*)d_xy, (const uint8_t*)d_zero, (uint4*)d_xyz, batch); break;
        case PLK_CURVE_BLS12_377: k_affine_to_projective<Bls12377BaseParams><<<blocks, 64, 0, stream>>>((const uint4*)d_xy, (const uint8_t*)d_zero, (uint4*)d_xyz, batch); break;
        default: return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}
int msm_ctx_is_comb(const plk_msm_ctx* ctx) { return ctx && ctx->comb ? 1 : 0; }

int msm_set_profiling_impl(plk_msm_ctx* ctx, int enable) {
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->profiling = enable != 0 && !ctx->comb;
    // a comb (few generators, automatic window: comb.hip) has no stages to time - two launches - and says so instead of staying silently off
    if (enable && ctx->comb) return set_error(PLK_ERR_INVALID_ARG, "per-stage timings: this context over %zu generators is a comb (no bucket stages)", ctx->n);
    return PLK_OK;
}
// sum_ms[7]: digits, partition (counts), bucket scan + final scatter, accumulate, bucket sums, planes, final -- summed over `calls` executions since the last read
int msm_get_timings_impl(plk_msm_ctx* ctx, double* sum_ms, unsigned* calls) {
    if (!ctx || !sum_ms) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (int k = 0; k < plk_msm_ctx::N_STAGES; ++k) sum_ms[k] = 0;
    unsigned cnt = 0;
    for (auto& set : ctx->prof_sets) {
        if (hipEventSynchronize(set.back()) != hipSuccess) continue;
        for (int k = 0; k < plk_msm_ctx::N_STAGES; ++k) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, set[k], set[k + 1]);
            sum_ms[k] += ms;
        }
        ++cnt;
        ctx->prof_free.push_back(set);
    }
    ctx->prof_sets.clear();
    if (calls) *calls = cnt;
    return PLK_OK;
}

size_t msm_ctx_len(const plk_msm_ctx* ctx) { return ctx->n; }
unsigned msm_ctx_window(const plk_msm_ctx* ctx) { return (unsigned)ctx->c; }
int msm_ctx_curve(const plk_msm_ctx* ctx) { return ctx->curve; }
int msm_ctx_table_free(const plk_msm_ctx* ctx) { return ctx->table_free ? 1 : 0; }
void msm_ctx_delete(plk_msm_ctx* ctx) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    delete ctx;  // frees on the devices its parts live on
    if (cur >= 0) (void)hipSetDevice(cur);
}
int msm_ctx_device(const plk_msm_ctx* ctx) { return ctx->device; }
std::vector<plk_msm_ctx*>& msm_ctx_peers(plk_msm_ctx* ctx) { return ctx->peers; }
std::vector<plk_msm_ctx*>& msm_ctx_shards(plk_msm_ctx* ctx) { return ctx->shards; }

// msm_precompute with the reference's output (curve_msm.rs:27-52): powers_per_generator[i][j] = [2^(w j)] G_i, j < ceil(BITS / w)
int msm_table_digits(int curve, unsigned w) { return w ? (scalar_bits(curve) + (int)w - 1) / (int)w : -1; }

template <class C>
static int msm_reference_table_t(size_t n, const void* d_bases, const void* d_zero, int w, int digits, void* d_out_xy, void* d_out_zero,
                                 hipStream_t stream) {
    using FP = typename C::FP;
    const size_t pt_bytes = (size_t)2 * FP::NL * 4;
    void* tab = scratch_acquire(n * digits * pt_bytes + 16, stream);
    if (!tab) return PLK_ERR_OOM;
    k_msm_table<C><<<(unsigned)((n + 127) / 128), 128, 0, stream>>>((const uint4*)d_bases, (const uint8_t*)d_zero, (uint4*)tab, n, w, digits, 0, n, nullptr);
    const size_t total = n * (size_t)digits;
    k_msm_table_export<C><<<(unsigned)((total + 255) / 256), 256, 0, stream>>>((const uint4*)tab, n, digits, (uint4*)d_out_xy, (uint8_t*)d_out_zero);
    hipError_t e = hipGetLastError();
    scratch_release(tab, stream);
    if (e != hipSuccess) return set_error(PLK_ERR_HIP, "table launch failed: %s", hipGetErrorString(e));
    return PLK_OK;
}

int msm_reference_table_dev_impl(int curve, size_t n, const void* d_bases, const void* d_zero, unsigned w, void* d_out_xy, void* d_out_zero,
                                 hipStream_t stream) {
    if (curve_limbs(curve) < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (w < 1 || w > 64) return set_error(PLK_ERR_INVALID_ARG, "window size %u outside [1, 64]", w);
    if (n == 0) return PLK_OK;
    if (!d_bases || !d_out_xy || !d_out_zero) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    const int digits = msm_table_digits(curve, w);
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: return msm_reference_table_t<TweedledeeCurve>(n, d_bases, d_zero, (int)w, digits, d_out_xy, d_out_zero, stream);
        case PLK_CURVE_TWEEDLEDUM: return msm_reference_table_t<TweedledumCurve>(n, d_bases, d_zero, (int)w, digits, d_out_xy, d_out_zero, stream);
        case PLK_CURVE_PALLAS: return msm_reference_table_t<PallasCurve>(n, d_bases, d_zero, (int)w, digits, d_out_xy, d_out_zero, stream);
        case PLK_CURVE_VESTA: return msm_reference_table_t<VestaCurve>(n, d_bases, d_zero, (int)w, digits, d_out_xy, d_out_zero, stream);
        default: return msm_reference_table_t<Bls12377Curve>(n, d_bases, d_zero, (int)w, digits, d_out_xy, d_out_zero, stream);
    }
}

int curve_sum_affine_dev_impl(int curve, size_t k, const void* d_pts, const void* d_zero, void* d_out_xy, void* d_out_zero, hipStream_t stream) {
    switch (curve) {
#define CASE(ID, C)                                                                                                                     \
    case ID:                                                                                                                            \
        k_sum_affine<C><<<1, 64, 64 * 4 * C::FP::NL * 4, stream>>>((const uint4*)d_pts, (const uint8_t*)d_zero, k, (uint4*)d_out_xy,     \
                                                                    (uint8_t*)d_out_zero);                                              \
        break;
        CASE(PLK_CURVE_TWEEDLEDEE, TweedledeeCurve)
        CASE(PLK_CURVE_TWEEDLEDUM, TweedledumCurve)
        CASE(PLK_CURVE_BLS12_377, Bls12377Curve)
        CASE(PLK_CURVE_PALLAS, PallasCurve)
        CASE(PLK_CURVE_VESTA, VestaCurve)
#undef CASE
        default: return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

size_t msm_partials_bytes(int curve, unsigned slots) {
    const int L = curve_limbs(curve);
    if (L < 0) return 0;
    return ((size_t)slots * 2 * L * 8 + slots + 15) & ~(size_t)15;
}
int msm_combine_partials_dev_impl(int curve, unsigned world, unsigned batch, unsigned whole_per_rank, const void* d_gathered, void* d_out_xy, void* d_out_zero,
                                  hipStream_t stream) {
    if (curve_limbs(curve) < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (batch == 0) return PLK_OK;
    if (world == 0 || !d_gathered || !d_out_xy || !d_out_zero) return set_error(PLK_ERR_INVALID_ARG, "null pointer or world = 0");
    if ((size_t)whole_per_rank * world > batch) return set_error(PLK_ERR_INVALID_ARG, "whole_per_rank %u x world %u exceeds the batch %u", whole_per_rank, world, batch);
    PLK_TRY(ensure_device());
    const unsigned slots = whole_per_rank + (batch - whole_per_rank * world);
    const size_t rec = msm_partials_bytes(curve, slots);
    switch (curve) {
#define CASE(ID, C)                                                                                                                                     \
    case ID:                                                                                                                                            \
        k_combine_partials<C><<<batch, 64, 64 * 4 * C::FP::NL * 4, stream>>>((const uint8_t*)d_gathered, rec, world, slots, whole_per_rank,            \
                                                                             (uint4*)d_out_xy, (uint8_t*)d_out_zero);                                   \
        break;
        CASE(PLK_CURVE_TWEEDLEDEE, TweedledeeCurve)
        CASE(PLK_CURVE_TWEEDLEDUM, TweedledumCurve)
        CASE(PLK_CURVE_BLS12_377, Bls12377Curve)
        CASE(PLK_CURVE_PALLAS, PallasCurve)
        CASE(PLK_CURVE_VESTA, VestaCurve)
#undef CASE
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

// the device's digit recoding on its own (plk_msm_debug_digits): d_digits = n * windows int32, windows = ceil((BITS + 1) / window_bits)
int msm_debug_digits_impl(int curve, unsigned window_bits, size_t n, const void* d_scalars, void* d_digits, hipStream_t stream) {
    if (window_bits < 2 || window_bits > (unsigned)MSM_MAX_WINDOW) return set_error(PLK_ERR_INVALID_ARG, "window of %u bits (2..%d)", window_bits, MSM_MAX_WINDOW);
    if (curve < 0 || curve > PLK_CURVE_VESTA) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (!n) return PLK_OK;
    if (!d_scalars || !d_digits) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    OrdCfg o{};
    o.c = (int)window_bits;
    o.windows = (scalar_bits(curve) + 1 + o.c - 1) / o.c;
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: return msm_launch_digits<TweedledeeCurve>(d_scalars, n, o, d_digits, stream);
        case PLK_CURVE_TWEEDLEDUM: return msm_launch_digits<TweedledumCurve>(d_scalars, n, o, d_digits, stream);
        case PLK_CURVE_PALLAS: return msm_launch_digits<PallasCurve>(d_scalars, n, o, d_digits, stream);
        case PLK_CURVE_VESTA: return msm_launch_digits<VestaCurve>(d_scalars, n, o, d_digits, stream);
        default: return msm_launch_digits<Bls12377Curve>(d_scalars, n, o, d_digits, stream);
    }
}
int msm_debug_digit_count(int curve, unsigned window_bits) {
    if (curve < 0 || curve > PLK_CURVE_VESTA || window_bits < 2 || window_bits > (unsigned)MSM_MAX_WINDOW) return -1;
    return (scalar_bits(curve) + 1 + (int)window_bits - 1) / (int)window_bits;
}

// workspaces for batches of up to `count` vectors, allocated ahead of the first batched execution
int msm_reserve_workspaces_impl(plk_msm_ctx* ctx, unsigned count, hipStream_t stream) {
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    PLK_TRY(ensure_device());
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->comb) return PLK_OK;  // nothing to reserve
    if (count > (unsigned)TAIL_MAX) count = TAIL_MAX;
    while (ctx->ws.size() < count) {
        ctx->ws.emplace_back();
        int rc;
        switch (ctx->curve) {
            case PLK_CURVE_TWEEDLEDEE: rc = msm_alloc_work<TweedledeeCurve>(ctx, ctx->ws.back(), stream); break;
            case PLK_CURVE_TWEEDLEDUM: rc = msm_alloc_work<TweedledumCurve>(ctx, ctx->ws.back(), stream); break;
            case PLK_CURVE_PALLAS: rc = msm_alloc_work<PallasCurve>(ctx, ctx->ws.back(), stream); break;
            case PLK_CURVE_VESTA: rc = msm_alloc_work<VestaCurve>(ctx, ctx->ws.back(), stream); break;
            default: rc = msm_alloc_work<Bls12377Curve>(ctx, ctx->ws.back(), stream); break;
        }
        if (rc != PLK_OK) {
            ctx->ws.back().release();
            ctx->ws.pop_back();
            return rc;
        }
    }
    return PLK_OK;
}

// counts[8]: mismatches per case of k_selftest_quad over `quads` quads on the n points d_pts
int selftest_quad_dev_impl(int curve, const void* d_pts, uint32_t n, uint32_t quads, uint32_t* counts) {
    if (!d_pts || !counts || n == 0 || quads == 0) return set_error(PLK_ERR_INVALID_ARG, "bad argument");
    PLK_TRY(ensure_device());
    uint32_t* d_cnt = (uint32_t*)scratch_acquire(32, nullptr);
    if (!d_cnt) return PLK_ERR_OOM;
    (void)hipMemsetAsync(d_cnt, 0, 32, nullptr);
    const unsigned blocks = (quads * 4 + 255) / 256;
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: k_selftest_quad<TweedledeeCurve><<<blocks, 256>>>((const uint4*)d_pts, n, d_cnt); break;
        case PLK_CURVE_TWEEDLEDUM: k_selftest_quad<TweedledumCurve><<<blocks, 256>>>((const uint4*)d_pts, n, d_cnt); break;
        case PLK_CURVE_BLS12_377: k_selftest_quad<Bls12377Curve><<<blocks, 256>>>((const uint4*)d_pts, n, d_cnt); break;
        case PLK_CURVE_PALLAS: k_selftest_quad<PallasCurve><<<blocks, 256>>>((const uint4*)d_pts, n, d_cnt); break;
        case PLK_CURVE_VESTA: k_selftest_quad<VestaCurve><<<blocks, 256>>>((const uint4*)d_pts, n, d_cnt); break;
        default: scratch_release(d_cnt, nullptr); return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(counts, d_cnt, 32, hipMemcpyDeviceToHost);
    scratch_release(d_cnt, nullptr);
    if (e != hipSuccess) return set_error(PLK_ERR_HIP, "selftest failed: %s", hipGetErrorString(e));
    return PLK_OK;
}

int curve_gen_bases_dev_impl(int curve, size_t n, uint64_t first, const void* d_g0d, void* d_out, hipStream_t stream) {
    if (n == 0) return PLK_OK;
    switch (curve) {
#define CASE(ID, C)                                                                                                          \
    case ID: k_gen_bases<C><<<(unsigned)((n + 127) / 128), 128, 0, stream>>>((const uint4*)d_g0d, (uint4*)d_out, n, first); break;
        CASE(PLK_CURVE_TWEEDLEDEE, TweedledeeCurve)
        CASE(PLK_CURVE_TWEEDLEDUM, TweedledumCurve)
        CASE(PLK_CURVE_BLS12_377, Bls12377Curve)
        CASE(PLK_CURVE_PALLAS, PallasCurve)
        CASE(PLK_CURVE_VESTA, VestaCurve)
#undef CASE
        default: return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

}  // namespace plk
