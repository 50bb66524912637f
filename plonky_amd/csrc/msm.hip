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

#include "common.h"
#include "ec.cuh"
#include "ecz.cuh"
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
constexpr size_t COMB_MAX_N = (size_t)1 << 15;  // generators up to which a tabled context with an automatic window is a comb (1 GB of table at 2^15)
constexpr int COMB_WINDOW = 4, COMB_WINDOWS = 64;

// -DPLK_CHECKED (make checked -> libplonky_hip_checked.so; SURVEY.md section 5: the reference's debug assertions and overflow
// checks have no equivalent in a release kernel): every index the ordering and accumulation kernels compute into sorted[],
// tmp[], the tables and the bucket arrays is compared with its bound; a violation is counted per site and the access is
// skipped.  plk_checked_failures() reads the counters.  In the normal build the guards compile to nothing.
#ifdef PLK_CHECKED
__device__ unsigned g_plk_chk[8];
#define PLK_CHK(cond, site) (!(cond) ? (atomicAdd(&g_plk_chk[site], 1u), false) : true)
#else
#define PLK_CHK(cond, site) (true)
#endif
enum { CHK_TMP_INDEX = 0, CHK_TILE_STAGE = 1, CHK_SORTED_INDEX = 2, CHK_SEG_STAGE = 3, CHK_TABLE_INDEX = 4, CHK_BUCKET = 5, CHK_ENTRY_RANGE = 6 };

constexpr int MSM_MAX_PLANE_PARTS = 16;  // blocks per bit-plane in the reduction (planes * parts quads must fit the final block)
constexpr int MSM_TF_MAX_WINDOW = 16;  // table-free mode: every window has its own 2^(c-1) buckets
constexpr int MSM_MAX_WINDOW = 21;   // c - 1 <= 10 coarse + 11 fine bits in the partition (ORD_MAX_BINS, ORD_MAX_FINE)
constexpr uint32_t CODE_INVALID = 0xFFFFFFFFu;

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

// ---------------------------------------------------------------------------------------------
// scalars -> signed window digits  (curve_msm.rs:159-180), computed where they are consumed
// ---------------------------------------------------------------------------------------------
// Digits are never stored: the two kernels of the first partition level recompute them from the scalars
// (one Montgomery -> canonical conversion per scalar and kernel, ~10 instructions per digit), which replaces
// a 4-byte write and two 4-byte reads per (scalar, window) by two extra reads of the 32-byte scalar.
// The canonical limbs are parked in LDS (limb-major: conflict-free) so that the window loop can index them.
constexpr int ORD_THREADS = 256;
constexpr int ORD_TILE = 4096;      // entries staged per tile of the level-1 scatter
constexpr int ORD_MAX_BINS = 1024;  // coarse bins
constexpr int ORD_MAX_FINE = 11;    // fine bits: buckets per coarse bin <= 2048
constexpr int ORD_BIN_THREADS = 512;
constexpr uint32_t ORD_SEG = 8192;  // entries per level-2 workgroup
// k_ord_bin_scatter stages a whole segment in LDS (3 fine-bit tables + the staged entries): ~74 KB, above the 64 KB a workgroup
// gets on gfx90a / gfx942 - this library is built for gfx950 (160 KB of LDS per CU) only, plk_init refuses other devices
static_assert(3 * (4u << ORD_MAX_FINE) + 4 * ORD_BIN_THREADS + 6 * ORD_SEG <= 160 * 1024, "k_ord_bin_scatter's LDS tile must fit a gfx950 CU");

struct OrdCfg {
    int c;                   // window bits
    int windows;             // digits per scalar
    uint32_t window_buckets; // table-free mode: 2^(c-1) (every window has its own bucket range), else 0
    int fine_bits;           // bucket id = [coarse bin | fine]
    int nbins;               // coarse bins in use
    uint32_t spt;            // scalars per sub-tile (<= ORD_THREADS, spt * windows <= ORD_TILE)
    uint32_t sub;            // sub-tiles per tile (one block walks them in turn)
    uint32_t nt1;            // tiles
    int raw_signed;          // 1: the "scalars" are half scalars of a GLV split: canonical magnitude, sign in bit 255 (glv.cuh)
    uint32_t entries_cap;    // n_eff * windows: size of tmp[] / sorted[] and of the table (checked build)
    uint32_t ent_stride;     // entry id of (window j, scalar i) = j * ent_stride + ent_first + i: the table index.  ent_stride = n_eff of the
    uint32_t ent_first;      // context; ent_first > 0 when the scalars belong to generators first .. first + n - 1 only (plk_msm_execute_parts_dev)
};

template <class SP> PLK_DI void ord_park_scalar(const uint4* __restrict__ scalars, size_t i, uint32_t* s_lim, int tid, bool raw_signed) {
    static_assert(SP::NL == 8, "scalar fields are 256-bit");
    const uint4 lo = scalars[i * 2], hi = scalars[i * 2 + 1];
    Fe<SP> s;
    s.v[0] = lo.x; s.v[1] = lo.y; s.v[2] = lo.z; s.v[3] = lo.w;
    s.v[4] = hi.x; s.v[5] = hi.y; s.v[6] = hi.z; s.v[7] = hi.w;
    // Montgomery -> canonical in the SCALAR field (to_canonical_u64_vec, curve_msm.rs:164); half scalars are canonical already
    if (!raw_signed) s = fe_to_canonical<SP>(s);
#pragma unroll
    for (int k = 0; k < 8; ++k) s_lim[k * ORD_THREADS + tid] = s.v[k];
}
// digit j of the parked scalar: signed c-bit window by carry-based integer recoding (never s -> r - s, so it is valid on
// BLS12-377 G1 whose cofactor is even).  Returns (bucket << 1) | negative, or CODE_INVALID for a zero digit.
PLK_DI uint32_t ord_digit(const uint32_t* s_lim, int tid, int j, const OrdCfg& cfg, uint32_t& carry) {
    const int c = cfg.c;
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    const int bp = j * c, li = bp >> 5, sh = bp & 31;
    uint64_t two = li < 8 ? s_lim[li * ORD_THREADS + tid] : 0u;
    if (li + 1 < 8) two |= (uint64_t)s_lim[(li + 1) * ORD_THREADS + tid] << 32;
    const uint32_t v = ((uint32_t)(two >> sh) & mask) + carry;
    // v in [0, 2^c]; v > 2^(c-1) becomes v - 2^c with a carry into the next window
    const uint32_t neg = v > half ? 1u : 0u;
    const uint32_t mag = neg ? (1u << c) - v : v;
    carry = neg;
    if (mag == 0) return CODE_INVALID;
    // a negative half scalar (sign parked in bit 255, far above its windows) flips every digit
    const uint32_t flip = cfg.raw_signed ? s_lim[7 * ORD_THREADS + tid] >> 31 : 0u;
    return ((mag - 1u + (uint32_t)j * cfg.window_buckets) << 1) | (neg ^ flip);
}

// exclusive prefix of `v` over the threads of the block (blockDim.x a multiple of 64, <= 1024); *total (optional) = the block sum.
// Shuffles inside a wave, one LDS word per wave across: two barriers instead of two per doubling step.  s_tmp: >= 16 words,
// free again when the call returns.
PLK_DI uint32_t block_excl_prefix(uint32_t v, uint32_t* s_tmp, uint32_t* total = nullptr) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t u = __shfl_up(inc, d);
        if (lane >= d) inc += u;
    }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
    for (int w = 0; w < nw; ++w) {
        const uint32_t x = s_tmp[w];
        if (w < wave) base += x;
        all += x;
    }
    if (total) *total = all;
    __syncthreads();
    return base + inc - v;
}
// exclusive scan of s_data[0..count) in place (count <= 4 * blockDim.x); s_tmp: 16 words.  The caller's writes to s_data must be
// visible (a barrier before the call); ends with a barrier.
PLK_DI void block_excl_scan4(uint32_t* s_data, int count, uint32_t* s_tmp) {
    const int t = threadIdx.x;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = t * 4 + k;
        v[k] = idx < count ? s_data[idx] : 0u;
        sum += v[k];
    }
    uint32_t run = block_excl_prefix(sum, s_tmp);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = t * 4 + k;
        if (idx < count) s_data[idx] = run;
        run += v[k];
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// entries -> bucket order  (replaces the reference's serial digit_occurrences scatter, curve_msm.rs:117-126)
// ---------------------------------------------------------------------------------------------
// Every (scalar i, window j) with a non-zero digit is an entry (id j * n + i = its table index) that goes to bucket
// |d| - 1.  A bucket id is [coarse bin | fine].  Level 1 moves the entries to their coarse bin (per-tile LDS histogram ->
// global [bin][tile] counts -> scan -> staged, run-wise writes); level 2 is one workgroup per coarse bin that counts,
// scans and scatters its bin by the fine bits, producing the bucket offsets on the way.  Only LDS atomics; counts,
// not capacities, drive the layout, so any digit distribution works.

// GLV split of the scalars of a table-free MSM (glv.cuh): half[i] = k1_i, half[n + i] = k2_i (magnitude, sign in bit 255);
// the ordering kernels then see 2n "scalars" of GLV_BITS bits over the points [G_0 .. G_(n-1), phi(G_0) .. phi(G_(n-1))].
template <class C>
__global__ void __launch_bounds__(256) k_glv_split(const uint4* __restrict__ scalars, size_t n, uint4* __restrict__ half) {
    using SP = typename C::SP;
    if constexpr (C::Glv::ENABLED) {
        const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n) return;
        const uint4 lo = scalars[i * 2], hi = scalars[i * 2 + 1];
        Fe<SP> s;
        s.v[0] = lo.x; s.v[1] = lo.y; s.v[2] = lo.z; s.v[3] = lo.w;
        s.v[4] = hi.x; s.v[5] = hi.y; s.v[6] = hi.z; s.v[7] = hi.w;
        s = fe_to_canonical<SP>(s);
        uint32_t k1[8], k2[8];
        glv_split<typename C::Glv>(s.v, k1, k2);
        half[i * 2] = make_uint4(k1[0], k1[1], k1[2], k1[3]);
        half[i * 2 + 1] = make_uint4(k1[4], k1[5], k1[6], k1[7]);
        half[(n + i) * 2] = make_uint4(k2[0], k2[1], k2[2], k2[3]);
        half[(n + i) * 2 + 1] = make_uint4(k2[4], k2[5], k2[6], k2[7]);
    }
}

// level 1, step 1: cnt1[bin * nt1 + tile].  A tile is `sub` consecutive sub-tiles of spt scalars, walked by one block.
template <class C>
__global__ void __launch_bounds__(ORD_THREADS) k_ord_count(const uint4* __restrict__ scalars, size_t n, OrdCfg cfg, uint32_t* __restrict__ cnt1) {
    using SP = typename C::SP;
    __shared__ uint32_t s_lim[8 * ORD_THREADS];
    __shared__ uint32_t s_hist[ORD_MAX_BINS];
    const int tid = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    for (int k = tid; k < cfg.nbins; k += ORD_THREADS) s_hist[k] = 0;
    for (uint32_t st = 0; st < cfg.sub; ++st) {
        const size_t i = ((size_t)tile * cfg.sub + st) * cfg.spt + tid;
        const bool live = (uint32_t)tid < cfg.spt && i < n;
        __syncthreads();
        if (live) ord_park_scalar<SP>(scalars, i, s_lim, tid, cfg.raw_signed != 0);
        __syncthreads();
        if (live) {
            uint32_t carry = 0;
            for (int j = 0; j < cfg.windows; ++j) {
                const uint32_t code = ord_digit(s_lim, tid, j, cfg, carry);
                if (code != CODE_INVALID) atomicAdd(&s_hist[code >> (cfg.fine_bits + 1)], 1u);
            }
        }
    }
    __syncthreads();
    for (int k = tid; k < cfg.nbins; k += ORD_THREADS) cnt1[(size_t)k * cfg.nt1 + tile] = s_hist[k];
}

// level 1, step 2: one block per bin row: in-place exclusive scan over the tiles; the block that finishes last turns the
// bin totals into the bin offsets bin_base[0..nbins] (exclusive scan; bin_base[nbins] = number of entries) and into the
// level-2 segment table seg_base[0..nbins] (a bin of t entries has ceil(t / ORD_SEG) segments).
// It also fixes the accumulation's chunk length for THIS execution from the number of entries actually present (dyn_chunk[0]):
// the lanes the context was laid out for share them, so a sparse vector (a slice of a sharded commitment, a zero-padded
// quotient chunk, Z = 1) runs short chains on all lanes instead of full-length chains on a few - the accumulation of a
// lane is a dependency chain, its length is the kernel's duration.
__global__ void __launch_bounds__(256) k_ord_scan1(uint32_t* __restrict__ cnt1, uint32_t nt1, int nbins, uint32_t* __restrict__ bin_total,
                                                   uint32_t* __restrict__ bin_base, uint32_t* __restrict__ seg_base, uint32_t* __restrict__ done_counter,
                                                   uint32_t* __restrict__ dyn_chunk, uint32_t chunk_cfg, uint32_t lanes_cfg, uint32_t* __restrict__ off_direct) {
    __shared__ uint32_t s_sum[256];
    __shared__ uint32_t s_bins[ORD_MAX_BINS];
    __shared__ bool s_last;
    uint32_t* row = cnt1 + (size_t)blockIdx.x * nt1;
    const uint32_t per = (nt1 + 255) / 256;
    const uint32_t lo = min(nt1, threadIdx.x * per), hi = min(nt1, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += row[i];
    uint32_t row_total = 0;
    uint32_t run = block_excl_prefix(sum, s_sum, &row_total);
    for (uint32_t i = lo; i < hi; ++i) {
        uint32_t v = row[i];
        row[i] = run;
        run += v;
    }
    if (threadIdx.x == 255) {
        bin_total[blockIdx.x] = row_total;
        __threadfence();
        s_last = atomicAdd(done_counter, 1u) == (uint32_t)nbins - 1u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const volatile uint32_t* vt = bin_total;
    for (int k = threadIdx.x; k < nbins; k += 256) s_bins[k] = vt[k];
    __syncthreads();
    const uint32_t last_total = s_bins[nbins - 1];
    block_excl_scan4(s_bins, nbins, s_sum);
    for (int k = threadIdx.x; k < nbins; k += 256) {
        bin_base[k] = s_bins[k];
        if (off_direct) off_direct[k] = s_bins[k];  // one-level ordering: the bins are the buckets
    }
    if (threadIdx.x == 0) {
        const uint32_t total = s_bins[nbins - 1] + last_total;
        bin_base[nbins] = total;
        if (off_direct) off_direct[nbins] = total;
        uint32_t ch = lanes_cfg ? (total + lanes_cfg - 1) / lanes_cfg : chunk_cfg;
        if (ch < 8u) ch = 8u;
        if (ch > chunk_cfg) ch = chunk_cfg;
        dyn_chunk[0] = ch;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += 256) s_bins[k] = (vt[k] + ORD_SEG - 1) / ORD_SEG;
    __syncthreads();
    const uint32_t last_segs = s_bins[nbins - 1];
    block_excl_scan4(s_bins, nbins, s_sum);
    for (int k = threadIdx.x; k < nbins; k += 256) seg_base[k] = s_bins[k];
    if (threadIdx.x == 0) {
        seg_base[nbins] = s_bins[nbins - 1] + last_segs;
        *done_counter = 0;  // ready for the next execution
    }
}

// level 1, step 3: (code, entry id) to its coarse bin; every sub-tile is ordered by bin inside LDS first, so that consecutive
// lanes store to consecutive slots of the same (tile, bin) run
template <class C>
__global__ void __launch_bounds__(ORD_THREADS) k_ord_scatter(const uint4* __restrict__ scalars, size_t n, OrdCfg cfg, const uint32_t* __restrict__ cnt1,
                                                             const uint32_t* __restrict__ bin_base, uint2* __restrict__ tmp,
                                                             uint32_t* __restrict__ sorted_direct) {
    using SP = typename C::SP;
    __shared__ uint32_t s_lim[8 * ORD_THREADS];
    __shared__ uint32_t s_cnt[ORD_MAX_BINS], s_base[ORD_MAX_BINS], s_gbase[ORD_MAX_BINS];
    __shared__ uint32_t s_tmp[ORD_THREADS];
    __shared__ uint2 s_ent[ORD_TILE];
    __shared__ uint16_t s_rank[ORD_TILE];  // [window][scalar of the sub-tile]: spt * windows <= ORD_TILE
    const int tid = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    for (int k = tid; k < cfg.nbins; k += ORD_THREADS) s_gbase[k] = bin_base[k] + cnt1[(size_t)k * cfg.nt1 + tile];
    for (uint32_t st = 0; st < cfg.sub; ++st) {
        const size_t i = ((size_t)tile * cfg.sub + st) * cfg.spt + tid;
        const bool live = (uint32_t)tid < cfg.spt && i < n;
        __syncthreads();  // the previous sub-tile has been written out
        for (int k = tid; k < cfg.nbins; k += ORD_THREADS) s_cnt[k] = 0;
        if (live) ord_park_scalar<SP>(scalars, i, s_lim, tid, cfg.raw_signed != 0);
        __syncthreads();
        if (live) {
            // one atomic per entry: its return value is the entry's rank inside its bin, kept for the placement below
            uint32_t carry = 0;
            for (int j = 0; j < cfg.windows; ++j) {
                const uint32_t code = ord_digit(s_lim, tid, j, cfg, carry);
                if (code != CODE_INVALID) s_rank[j * cfg.spt + tid] = (uint16_t)atomicAdd(&s_cnt[code >> (cfg.fine_bits + 1)], 1u);
            }
        }
        __syncthreads();
        for (int k = tid; k < cfg.nbins; k += ORD_THREADS) s_base[k] = s_cnt[k];
        __syncthreads();
        block_excl_scan4(s_base, cfg.nbins, s_tmp);
        if (live) {
            uint32_t carry = 0;
            for (int j = 0; j < cfg.windows; ++j) {
                const uint32_t code = ord_digit(s_lim, tid, j, cfg, carry);
                if (code != CODE_INVALID) {
                    const uint32_t slot = s_base[code >> (cfg.fine_bits + 1)] + s_rank[j * cfg.spt + tid];
                    if (PLK_CHK(slot < (uint32_t)ORD_TILE, CHK_TILE_STAGE)) s_ent[slot] = make_uint2(code, (uint32_t)((size_t)j * cfg.ent_stride + cfg.ent_first + i));
                }
            }
        }
        __syncthreads();
        const uint32_t total = s_base[cfg.nbins - 1] + s_cnt[cfg.nbins - 1];
        for (uint32_t sidx = tid; sidx < total; sidx += ORD_THREADS) {
            const uint2 e = s_ent[sidx];
            const uint32_t bin = e.x >> (cfg.fine_bits + 1);
            const uint32_t at = s_gbase[bin] + (sidx - s_base[bin]);
            if (PLK_CHK(at < cfg.entries_cap, CHK_TMP_INDEX)) {
                if (sorted_direct) sorted_direct[at] = (e.y << 1) | (e.x & 1u);  // one-level ordering: this IS the bucket order
                else tmp[at] = e;
            }
        }
        __syncthreads();
        for (int k = tid; k < cfg.nbins; k += ORD_THREADS) s_gbase[k] += s_cnt[k];  // this sub-tile's entries of bin k
    }
}

// level 2: a coarse bin is cut into segments of <= ORD_SEG entries, one workgroup each (a hot bin - short top window, skewed
// witness - is shared by many workgroups).  Block -> (bin, segment) by a search in seg_base.
PLK_DI bool ord_segment(const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ bin_base, int nbins, uint32_t blk, uint32_t& bin,
                        uint32_t& seg, uint32_t& lo, uint32_t& hi) {
    if (blk >= seg_base[nbins]) return false;
    uint32_t a = 0, b = (uint32_t)nbins;  // seg_base[a] <= blk < seg_base[b]
    while (b - a > 1) {
        const uint32_t m = (a + b) >> 1;
        if (seg_base[m] <= blk) a = m; else b = m;
    }
    bin = a;
    seg = blk - seg_base[a];
    lo = bin_base[a] + seg * ORD_SEG;
    hi = min(bin_base[a + 1], lo + ORD_SEG);
    return true;
}
// step 1: cnt2[segment][fine]
__global__ void __launch_bounds__(ORD_BIN_THREADS) k_ord_bin_count(const uint2* __restrict__ tmp, const uint32_t* __restrict__ bin_base,
                                                                   const uint32_t* __restrict__ seg_base, int fine_bits, int nbins,
                                                                   uint32_t* __restrict__ cnt2) {
    __shared__ uint32_t s_hist[1 << ORD_MAX_FINE];
    const int tid = threadIdx.x;
    uint32_t bin, seg, lo, hi;
    if (!ord_segment(seg_base, bin_base, nbins, blockIdx.x, bin, seg, lo, hi)) return;
    const int nf = 1 << fine_bits;
    const uint32_t fmask = (uint32_t)nf - 1u;
    for (int k = tid; k < nf; k += ORD_BIN_THREADS) s_hist[k] = 0;
    __syncthreads();
    for (uint32_t p = lo + tid; p < hi; p += ORD_BIN_THREADS) atomicAdd(&s_hist[(tmp[p].x >> 1) & fmask], 1u);
    __syncthreads();
    for (int k = tid; k < nf; k += ORD_BIN_THREADS) cnt2[((size_t)blockIdx.x << fine_bits) + k] = s_hist[k];
}
// step 2: bucket offsets of the bin (sum over its segments, scanned), this segment's start inside every bucket, scatter.
// The segment is ordered by bucket inside LDS first: its entries of one bucket leave as one run.
__global__ void __launch_bounds__(ORD_BIN_THREADS) k_ord_bin_scatter(const uint2* __restrict__ tmp, const uint32_t* __restrict__ bin_base,
                                                                     const uint32_t* __restrict__ seg_base, int fine_bits, int nbins, uint32_t buckets,
                                                                     const uint32_t* __restrict__ cnt2, uint32_t* __restrict__ off,
                                                                     uint32_t* __restrict__ sorted, uint32_t entries_cap) {
    __shared__ uint32_t s_glob[1 << ORD_MAX_FINE], s_loc[1 << ORD_MAX_FINE], s_cur[1 << ORD_MAX_FINE];
    __shared__ uint32_t s_tmp[ORD_BIN_THREADS];
    __shared__ uint32_t s_out[ORD_SEG];
    __shared__ uint16_t s_fine[ORD_SEG];
    const int tid = threadIdx.x;
    uint32_t bin, seg, lo, hi;
    if (!ord_segment(seg_base, bin_base, nbins, blockIdx.x, bin, seg, lo, hi)) {
        // bins without entries still own bucket offsets: written by the blocks past the last segment, one bin each
        // (the grid has at least nbins blocks past the segments; see the launch)
        const uint32_t extra = blockIdx.x - seg_base[nbins];
        if (extra < (uint32_t)nbins && bin_base[extra + 1] == bin_base[extra]) {
            const int nf = 1 << fine_bits;
            for (int k = tid; k < nf; k += ORD_BIN_THREADS) off[((size_t)extra << fine_bits) + k] = bin_base[extra];
        }
        if (extra == 0 && tid == 0) off[buckets] = bin_base[nbins];
        return;
    }
    const int nf = 1 << fine_bits;
    const uint32_t fmask = (uint32_t)nf - 1u;
    const uint32_t s0 = seg_base[bin], s1 = seg_base[bin + 1];
    for (int k = tid; k < nf; k += ORD_BIN_THREADS) {
        uint32_t tot = 0, before = 0, own = 0;
        for (uint32_t sg = s0; sg < s1; ++sg) {
            const uint32_t v = cnt2[((size_t)sg << fine_bits) + k];
            if (sg - s0 < seg) before += v;
            if (sg - s0 == seg) own = v;
            tot += v;
        }
        s_glob[k] = tot;
        s_cur[k] = before;
        s_loc[k] = own;
    }
    __syncthreads();
    block_excl_scan4(s_glob, nf, s_tmp);
    block_excl_scan4(s_loc, nf, s_tmp);
    const uint32_t bb = bin_base[bin];
    for (int k = tid; k < nf; k += ORD_BIN_THREADS) {
        const uint32_t o = bb + s_glob[k];
        if (seg == 0) off[((size_t)bin << fine_bits) + k] = o;
        s_glob[k] = o + s_cur[k];  // where this segment's entries of bucket k start
        s_cur[k] = s_loc[k];       // LDS cursor
    }
    __syncthreads();
    for (uint32_t p = lo + tid; p < hi; p += ORD_BIN_THREADS) {
        const uint2 e = tmp[p];
        const uint32_t f = (e.x >> 1) & fmask;
        const uint32_t idx = atomicAdd(&s_cur[f], 1u);
        if (PLK_CHK(idx < ORD_SEG, CHK_SEG_STAGE)) {
            s_out[idx] = (e.y << 1) | (e.x & 1u);
            s_fine[idx] = (uint16_t)f;
        }
    }
    __syncthreads();
    const uint32_t count = hi - lo;
    for (uint32_t i = tid; i < count; i += ORD_BIN_THREADS) {
        const uint32_t f = s_fine[i];
        const uint32_t at = s_glob[f] + (i - s_loc[f]);
        if (PLK_CHK(at < entries_cap, CHK_SORTED_INDEX)) sorted[at] = s_out[i];
    }
}

// ---------------------------------------------------------------------------------------------
// bucket accumulation: every lane adds exactly `chunk` consecutive sorted entries
// ---------------------------------------------------------------------------------------------
// The sorted entry list is cut into chunks of `chunk` entries regardless of the bucket boundaries, one lane per chunk:
// perfectly balanced whatever the digit distribution and whatever the bucket sizes (26 entries on average at c = 20).  A
// lane that crosses a bucket boundary stores what it has and starts over: the piece of the bucket that STARTS inside the
// chunk goes to p_start[bucket], the piece of the bucket that was already running at the chunk's first entry goes to
// p_head[lane].  bucket b = p_start[b] + sum of p_head[l] for the lanes l0 < l <= l1, l0 = off[b] / chunk,
// l1 = (off[b+1] - 1) / chunk (k_msm_assemble).  Pieces are stored as they are (lazy 29-bit limbs, accumulator invariant of
// ecz.cuh; the identity is ZZ = 0): a store inside the loop must be cheap, because some lane of the wave has one almost every round.
template <class FP> constexpr int raw_u4() { return FzCfg<FP>::NZ; }  // uint4 per raw point: 4 NZ words

template <class FP> PLK_DI void xyzzz_store_raw(uint4* dst, const XyzzZ<FP>& a) {
    constexpr int NZ = FzCfg<FP>::NZ;
    uint32_t w[4 * NZ];
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        // the identity is ZZ = 0 (xyzzz_load_raw); its other coordinates are never looked at, so only ZZ pays for a select - the
        // store sits on the path that some lane of an accumulation wave takes almost every round
        w[i] = a.x.l[i];
        w[NZ + i] = a.y.l[i];
        w[2 * NZ + i] = a.inf ? 0u : a.zz.l[i];
        w[3 * NZ + i] = a.zzz.l[i];
    }
#pragma unroll
    for (int i = 0; i < NZ; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
template <class FP> PLK_DI XyzzZ<FP> xyzzz_load_raw(const uint4* src) {
    constexpr int NZ = FzCfg<FP>::NZ;
    uint32_t w[4 * NZ];
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        const uint4 v = src[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    XyzzZ<FP> r;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        r.x.l[i] = w[i];
        r.y.l[i] = w[NZ + i];
        r.zz.l[i] = w[2 * NZ + i];
        r.zzz.l[i] = w[3 * NZ + i];
        any |= w[2 * NZ + i];
    }
    r.inf = any == 0;  // a live accumulator never has ZZ = 0 (that case is caught as the identity in ecz.cuh)
    return r;
}

// Short generator lists (the frozen generators of an inner-product argument: 2^14 + 2 points): one lane per generator is a
// dependency chain, and in k_msm_table a quarter of it is the inversion that brings every window back to affine before the next
// c doublings.  Here the chain only doubles - c (windows - 1) doublings without a normalisation in between, every window's
// XYZZ point parked in scratch memory - and a second kernel normalises all (window, generator) pairs side by side:
// 1.75 -> 1.2 ms for 2^14 + 2 generators at c = 13 with a lane per generator, 0.65 ms on quads in one-wave workgroups (a chain of quad
// doublings takes 3.6 us per step in two-wave workgroups at this occupancy, 2.2 us in one-wave ones: tools/mul_latency.hip).  Same points, so the same (unique) affine table entries.
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

// Head pieces are mostly short-lived: the head piece of lane l (closed at l's first bucket boundary) belongs to the last
// bucket of lane l - 1, whose piece is still in registers when the loop ends.  Lanes therefore park a closed head piece in
// LDS and their predecessor in the block adds it to its last piece before storing it: at c = 20 (26 entries per bucket,
// 24 per lane) almost every bucket leaves the kernel whole, and k_msm_assemble only finds the head pieces of the first lane
// of a block and of lanes that lie entirely inside one bucket (head_live[lane] = 1).
constexpr int ACC_THREADS = 128;
// the bucket of sorted position pos, known to lie after bucket b: usually b + 1; a search when empty buckets follow (sparse
// scalar vectors - Z = 1, zero-padded quotient chunks - leave most buckets empty: a linear walk would cost a load per bucket)
PLK_DI uint32_t next_bucket(const uint32_t* __restrict__ off, uint32_t b, uint32_t buckets, uint32_t pos) {
    uint32_t lo = b + 1;
    if (off[lo + 1] > pos) return lo;
    uint32_t hi = buckets;  // off[lo] <= pos < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}
template <class C>
PLK_DI void msm_accumulate_body(const uint4* __restrict__ tab, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ off,
                                uint4* __restrict__ p_start, uint4* __restrict__ p_head, uint8_t* __restrict__ head_live, uint32_t buckets,
                                const uint32_t* __restrict__ dyn_chunk, int wshift, uint32_t n_sub, uint4* s_head, uint8_t* s_parked, uint32_t tab_entries) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    constexpr int RU = raw_u4<FP>();
    const int tid = threadIdx.x;
    const uint32_t lane = blockIdx.x * blockDim.x + tid;
    const uint32_t total = off[buckets];
    const uint32_t chunk = dyn_chunk[0];
    const uint64_t begin64 = (uint64_t)lane * chunk;
    const bool active = begin64 < total;
    XyzzZ<FP> acc = xyzzz_identity<FP>();
    bool head = false, parked = false;
    uint32_t b = 0;
    if (active) {
        const uint32_t begin = (uint32_t)begin64;
        const uint32_t end = (uint32_t)min((uint64_t)total, begin64 + chunk);
        // bucket of the first entry: largest b with off[b] <= begin (empty buckets share an offset with their successor)
        uint32_t lo = 0, hi = buckets;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (off[mid] <= begin) lo = mid; else hi = mid;
        }
        b = lo;
        uint32_t next = off[b + 1];
        head = off[b] < begin;  // the bucket was already running: this lane's first piece is a head piece
        // entry ids are window * n + generator; with tables that is the table index, without (table-free mode:
        // n_sub = n, buckets of window w are [w << wshift, (w + 1) << wshift)) the window part is taken off
        const uint32_t ent_sub = (b >> wshift) * n_sub;
        // Software pipeline: the table gather for entry k+1 (two dependent loads: index, then a random
        // 64/96-byte point) is issued before the ~10^4-cycle addition of entry k.
        (void)PLK_CHK(b < buckets && end <= total, CHK_ENTRY_RANGE);
        uint32_t ent = sorted[begin];
        Fe<FP> x, y;
        bool ident = true;
        if (PLK_CHK((ent >> 1) - ent_sub < tab_entries, CHK_TABLE_INDEX)) ident = affine_load<FP>(tab + (size_t)((ent >> 1) - ent_sub) * 2 * W, x, y);
        for (uint32_t k = begin; k < end; ++k) {
            if (k == next) {  // entry k opens a new bucket: the piece of the old one is closed
                xyzzz_settle<FP>(acc);  // the additions keep Y uncarried (ecz.cuh): move its carries before the piece is stored
                if (head) {
                    xyzzz_store_raw<FP>(s_head + tid * RU, acc);
                    parked = true;
                } else if (PLK_CHK(b < buckets, CHK_BUCKET)) {
                    xyzzz_store_raw<FP>(p_start + (size_t)b * RU, acc);
                }
                head = false;
                acc.inf = true;  // the coordinates stay as they are (36 register clears less): the next addition overwrites them (ecz.cuh)
                b = next_bucket(off, b, buckets, k);
                next = off[b + 1];
            }
            const uint32_t cur = ent;
            const Fe<FP> cx = x, cy = y;
            const bool cident = ident;
            if (k + 1 < end) {
                // the next entry may belong to a later bucket (another window in table-free mode): its bucket is known here
                const uint32_t nb = k + 1 == next ? next_bucket(off, b, buckets, k + 1) : b;
                const uint32_t nsub = (nb >> wshift) * n_sub;
                ent = sorted[k + 1];
                ident = true;
                if (PLK_CHK((ent >> 1) - nsub < tab_entries && nb < buckets, CHK_TABLE_INDEX))
                    ident = affine_load<FP>(tab + (size_t)((ent >> 1) - nsub) * 2 * W, x, y);
            }
            if (cident) continue;
            xyzzz_madd_entry<FP>(acc, cx, cy, (cur & 1u) != 0);
        }
        xyzzz_settle<FP>(acc);
    }
    s_parked[tid] = parked ? 1 : 0;
    __syncthreads();
    if (!active) return;
    // the successor's closed head piece continues this lane's last bucket
    if (tid + 1 < ACC_THREADS && s_parked[tid + 1]) acc = xyzzz_add<FP>(acc, xyzzz_load_raw<FP>(s_head + (tid + 1) * RU));
    if (head) {
        xyzzz_store_raw<FP>(p_head + (size_t)lane * RU, acc);  // the whole chunk lies inside one bucket
    } else {
        xyzzz_store_raw<FP>(p_start + (size_t)b * RU, acc);
        if (parked && tid == 0) {  // no predecessor in this block: the head piece stays a head piece
            const XyzzZ<FP> h = xyzzz_load_raw<FP>(s_head);
            xyzzz_store_raw<FP>(p_head + (size_t)lane * RU, h);
        }
    }
    head_live[lane] = (head || (parked && tid == 0)) ? 1 : 0;
}
#ifndef PLK_ACC_WAVES
#define PLK_ACC_WAVES 1  // waves per SIMD the register allocation of the accumulation is held to (tuning builds: tools/acc_ab.sh)
#endif
#if PLK_ACC_WAVES > 0
#define PLK_ACC_BOUNDS __launch_bounds__(ACC_THREADS, PLK_ACC_WAVES)
#else
#define PLK_ACC_BOUNDS __launch_bounds__(ACC_THREADS)  // round 2's form (A/B builds)
#endif
template <class C>
__global__ void PLK_ACC_BOUNDS k_msm_accumulate(const uint4* __restrict__ tab, const uint32_t* __restrict__ sorted,
                                                                const uint32_t* __restrict__ off, uint4* __restrict__ p_start, uint4* __restrict__ p_head,
                                                                uint8_t* __restrict__ head_live, uint32_t buckets, const uint32_t* __restrict__ dyn_chunk,
                                                                int wshift, uint32_t n_sub, uint32_t tab_entries) {
    __shared__ uint4 s_head[ACC_THREADS * raw_u4<typename C::FP>()];
    __shared__ uint8_t s_parked[ACC_THREADS];
    msm_accumulate_body<C>(tab, sorted, off, p_start, p_head, head_live, buckets, dyn_chunk, wshift, n_sub, s_head, s_parked, tab_entries);
}

// ---------------------------------------------------------------------------------------------
// reduction  sum_d d * bucket_d   (replaces the serial Yao tail of curve_msm.rs:149-154)
// ---------------------------------------------------------------------------------------------
// Steps (all batched over the MSMs of a group: blockIdx.y / a factor of blockIdx.z picks the MSM's slot):
//  * buckets with very many head pieces (a hot digit of a skewed witness) are summed by whole workgroups (k_msm_heavy_*);
//  * k_msm_assemble: bucket = start piece + head pieces;
//  * two-level weighting (tabled mode, many buckets): with b = hi 2^L + lo, sum_b (b + 1) B_b =
//    2^L sum_hi hi R_hi + sum_lo (lo + 1) C_lo, R_hi / C_lo the row / column sums of the 2^H x 2^L bucket grid:
//    2 additions per bucket at one lane each (k_msm_gsum: groups of G serially; k_msm_lsum: the rest by wave shuffles),
//    which leaves two weighted sums over 2^H and 2^L points;
//  * those (or, with few buckets and in table-free mode, the buckets themselves) go through bit-plane tree sums on quads,
//    sum_d d P_d = sum_p 2^p sum_{d: bit p} P_d, are doubled into place and added (k_msm_planes, k_msm_final, k_msm_combine),
//    then normalised (to_affine, curve.rs:206-214).
template <class FP> PLK_DI XyzzZ<FP> wave_sum(XyzzZ<FP> v, int width) {
    for (int m = 1; m < width; m <<= 1) v = xyzzz_add<FP>(v, xyzzz_shfl_xor<FP>(v, m));
    return v;
}

constexpr int TAIL_MAX = 16;
struct TailSlot {
    const uint32_t* off;  // bucket offsets off[buckets + 1]
    uint4* p_start;       // raw, one per bucket (becomes the assembled bucket)
    const uint4* p_head;  // raw, one per accumulation lane
    const uint8_t* head_live;  // 1: p_head[lane] holds a piece that is not part of a start piece yet
    uint4* bucket;        // packed points: the operands of the plane sums
    uint32_t* heavy;
    uint4* heavy_part;    // raw
    uint4* line_part;     // raw: row partials then column partials
    uint4* plane_part;
    uint4* win_pts;
    uint32_t* final_done;  // windows finished by k_msm_final (the last one adds them up); zero between executions
    const uint32_t* dyn_chunk;  // entries per accumulation lane of this execution (k_ord_scan1)
    uint4* out_xy;
    uint8_t* out_zero;
};
struct TailBatch {
    int count;
    TailSlot s[TAIL_MAX];
};

constexpr uint32_t HEAVY_HEADS = 32;   // more head pieces than this PER LANE of k_msm_assemble (2^lpb_log lanes per bucket): the bucket is summed by workgroups
constexpr uint32_t HEAVY_CHUNK = 2048;

// head pieces of bucket b: lanes first .. first + count - 1
PLK_DI bool bucket_heads(const uint32_t* __restrict__ off, uint32_t b, uint32_t chunk, uint32_t& first, uint32_t& count) {
    const uint32_t o0 = off[b], o1 = off[b + 1];
    first = 0;
    count = 0;
    if (o1 == o0) return false;
    const uint32_t l0 = o0 / chunk, l1 = (o1 - 1) / chunk;
    first = l0 + 1;
    count = l1 - l0;
    return true;
}

// heavy[0] = number of work items, heavy[1] = number of heavy buckets;
// items at heavy[2 + 2k] = bucket, heavy[3 + 2k] = chunk index; heavy bucket ids at heavy[2 + 2 cap + k]
__global__ void __launch_bounds__(256) k_msm_heavy_list(TailBatch tb, uint32_t buckets, uint32_t cap, int lpb_log) {
    const uint32_t* __restrict__ off = tb.s[blockIdx.y].off;
    const uint32_t chunk = tb.s[blockIdx.y].dyn_chunk[0];
    uint32_t* __restrict__ heavy = tb.s[blockIdx.y].heavy;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= buckets) return;
    uint32_t first, ns;
    bucket_heads(off, b, chunk, first, ns);
    // (a small MSM - an IPA round over frozen generators: 2^10 buckets of ~190 entries, 8-entry chunks - has 24-48 head pieces in
    // EVERY bucket; k_msm_assemble takes up to 32 per lane, so with 8 lanes per bucket nothing there is heavy)
    if (ns <= (HEAVY_HEADS << lpb_log)) return;
    const uint32_t chunks = (ns + HEAVY_CHUNK - 1) / HEAVY_CHUNK;
    const uint32_t at = atomicAdd(&heavy[0], chunks);
    const uint32_t hb = atomicAdd(&heavy[1], 1u);
    if (hb < cap) heavy[2 + 2 * cap + hb] = b;
    for (uint32_t k = 0; k < chunks; ++k)
        if (at + k < cap) {
            heavy[2 + 2 * (at + k)] = b;
            heavy[3 + 2 * (at + k)] = k;
        }
}

template <class FP> PLK_DI XyzzZ<FP> block256_sum(XyzzZ<FP> acc, uint4* s_pts) {
    constexpr int RU = raw_u4<FP>();
    acc = wave_sum<FP>(acc, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) xyzzz_store_raw<FP>(s_pts + wave * RU, acc);
    __syncthreads();
    acc = (wave == 0 && lane < 4) ? xyzzz_load_raw<FP>(s_pts + lane * RU) : xyzzz_identity<FP>();
    if (wave == 0) acc = wave_sum<FP>(acc, 4);
    __syncthreads();
    return acc;  // valid in thread 0
}

// one workgroup per (bucket, chunk) item: chunk partial -> heavy_part[item]
template <class C>
__global__ void __launch_bounds__(256) k_msm_heavy_chunks(TailBatch tb, uint32_t cap) {
    using FP = typename C::FP;
    constexpr int RU = raw_u4<FP>();
    __shared__ uint4 s_pts[4 * RU];
    const uint32_t chunk = tb.s[blockIdx.y].dyn_chunk[0];
    const uint4* __restrict__ p_head = tb.s[blockIdx.y].p_head;
    const uint32_t* __restrict__ off = tb.s[blockIdx.y].off;
    const uint32_t* __restrict__ heavy = tb.s[blockIdx.y].heavy;
    uint4* __restrict__ heavy_part = tb.s[blockIdx.y].heavy_part;
    const uint32_t items = min(heavy[0], cap);
    for (uint32_t it = blockIdx.x; it < items; it += gridDim.x) {
        const uint32_t b = heavy[2 + 2 * it], k = heavy[3 + 2 * it];
        uint32_t first, ns;
        bucket_heads(off, b, chunk, first, ns);
        const uint32_t s0 = first + k * HEAVY_CHUNK, s1 = min(first + ns, s0 + HEAVY_CHUNK);
        XyzzZ<FP> acc = xyzzz_identity<FP>();
        for (uint32_t s = s0 + threadIdx.x; s < s1; s += 256)
            if (tb.s[blockIdx.y].head_live[s]) acc = xyzzz_add<FP>(acc, xyzzz_load_raw<FP>(p_head + (size_t)s * RU));
        acc = block256_sum<FP>(acc, s_pts);
        if (threadIdx.x == 0) xyzzz_store_raw<FP>(heavy_part + (size_t)it * RU, acc);
    }
}
// one workgroup per heavy bucket: its start piece + the sum of its chunk partials -> p_start[b] (the whole bucket)
template <class C>
__global__ void __launch_bounds__(256) k_msm_heavy_final(TailBatch tb, uint32_t cap) {
    using FP = typename C::FP;
    constexpr int RU = raw_u4<FP>();
    __shared__ uint4 s_pts[4 * RU];
    const uint32_t* __restrict__ heavy = tb.s[blockIdx.y].heavy;
    const uint4* __restrict__ heavy_part = tb.s[blockIdx.y].heavy_part;
    uint4* __restrict__ p_start = tb.s[blockIdx.y].p_start;
    const uint32_t items = min(heavy[0], cap), nb = min(heavy[1], cap);
    for (uint32_t hb = blockIdx.x; hb < nb; hb += gridDim.x) {
        const uint32_t b = heavy[2 + 2 * cap + hb];
        XyzzZ<FP> acc = threadIdx.x == 0 ? xyzzz_load_raw<FP>(p_start + (size_t)b * RU) : xyzzz_identity<FP>();
        for (uint32_t it = threadIdx.x; it < items; it += 256)
            if (heavy[2 + 2 * it] == b) acc = xyzzz_add<FP>(acc, xyzzz_load_raw<FP>(heavy_part + (size_t)it * RU));
        acc = block256_sum<FP>(acc, s_pts);
        if (threadIdx.x == 0) xyzzz_store_raw<FP>(p_start + (size_t)b * RU, acc);
    }
}

// bucket = start piece + the head pieces that are still live, 2^lpb_log adjacent lanes per bucket (each takes every
// 2^lpb_log-th head, shuffles combine).  PACKED: the result goes to bucket[] in the packed exchange format (operand of the
// plane sums); else it stays in p_start[] raw, which is only rewritten when something was added (or the bucket is empty).
template <class C, bool PACKED>
__global__ void __launch_bounds__(256) k_msm_assemble(TailBatch tb, uint32_t buckets, int lpb_log) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    constexpr int RU = raw_u4<FP>();
    const TailSlot& sl = tb.s[blockIdx.y];
    const uint32_t chunk = sl.dyn_chunk[0];
    if (blockIdx.x == 0 && threadIdx.x < 2) sl.heavy[threadIdx.x] = 0;  // the counters of k_msm_heavy_list are free again
    if (blockIdx.x == 0 && threadIdx.x == 2) *sl.final_done = 0;        // and so is k_msm_final's (left at zero by its last block anyway)
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = gid >> lpb_log, part = gid & ((1u << lpb_log) - 1u);
    XyzzZ<FP> acc = xyzzz_identity<FP>();
    bool nonempty = false, touched = false;
    if (b < buckets) {
        uint32_t first, ns;
        nonempty = bucket_heads(sl.off, b, chunk, first, ns);
        uint32_t live = 0;
        if (ns <= (HEAVY_HEADS << lpb_log))  // heavier buckets are already whole (k_msm_heavy_final)
            for (uint32_t h = part; h < ns; h += 1u << lpb_log) live |= sl.head_live[first + h] ? (1u << (h >> lpb_log)) : 0u;  // ns <= 32
        touched = live != 0;
        if (lpb_log > 0 || PACKED || touched) {
            if (nonempty && part == 0) acc = xyzzz_load_raw<FP>(sl.p_start + (size_t)b * RU);
            for (uint32_t h = part, k = 0; h < ns && live; h += 1u << lpb_log, ++k)
                if ((live >> k) & 1u) acc = xyzzz_add<FP>(acc, xyzzz_load_raw<FP>(sl.p_head + (size_t)(first + h) * RU));
        }
    }
    if (lpb_log > 0) acc = wave_sum<FP>(acc, 1 << lpb_log);  // the lanes of a bucket are adjacent; every lane takes part in the shuffles
    if (b < buckets && part == 0) {
        if constexpr (PACKED) xyzzz_store_packed<FP>(sl.bucket + (size_t)b * 4 * W, acc);
        else if (lpb_log > 0 || touched || !nonempty) xyzzz_store_raw<FP>(sl.p_start + (size_t)b * RU, acc);
    }
}

// Two-level weighting, step 1: partial row and column sums over groups of G = 2^g_log buckets, one lane per group.
// lanes [0, NB / G): row hi, group g: buckets (hi << L) + g G + k;   lanes [NB / G, 2 NB / G): column lo, group g:
// buckets ((g G + k) << L) + lo (adjacent lanes = adjacent columns = adjacent addresses).
template <class C>
__global__ void __launch_bounds__(128) k_msm_gsum(TailBatch tb, int L, int H, int g_log) {
    using FP = typename C::FP;
    constexpr int RU = raw_u4<FP>();
    const TailSlot& sl = tb.s[blockIdx.y];
    const uint32_t nbg = (1u << (L + H)) >> g_log;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * nbg) return;
    // table-free mode: blockIdx.z is the window, every window its own 2^H x 2^L grid of buckets and its own partials
    const uint32_t wbase = blockIdx.z << (L + H);
    uint4* part = sl.line_part + (size_t)blockIdx.z * 2 * nbg * RU;
    const uint32_t G = 1u << g_log;
    uint32_t b0, bstep;  // first bucket, distance between consecutive buckets of the group
    uint4* dst;
    if (t < nbg) {
        const uint32_t pr = (1u << L) >> g_log;  // groups per row
        const uint32_t hi = t / pr, g = t % pr;
        b0 = wbase + (hi << L) + (g << g_log);
        bstep = 1;
        dst = part + (size_t)t * RU;
    } else {
        const uint32_t u = t - nbg;
        const uint32_t lo = u & ((1u << L) - 1u), g = u >> L;
        const uint32_t pc = (1u << H) >> g_log;  // groups per column
        b0 = wbase + ((g << g_log) << L) + lo;
        bstep = 1u << L;
        dst = part + ((size_t)nbg + (size_t)lo * pc + g) * RU;
    }
    // the load of element k + 1 is in flight while element k is added
    XyzzZ<FP> acc = xyzzz_identity<FP>();
    XyzzZ<FP> nxt = xyzzz_load_raw<FP>(sl.p_start + (size_t)b0 * RU);
    for (uint32_t k = 0; k < G; ++k) {
        XyzzZ<FP> cur = nxt;
        const uint32_t b = b0 + k * bstep;
        if (k + 1 < G) nxt = xyzzz_load_raw<FP>(sl.p_start + (size_t)(b + bstep) * RU);
        acc = xyzzz_add<FP>(acc, cur);
    }
    xyzzz_store_raw<FP>(dst, acc);
}

// step 2: the lines.  Output slot s of 2 * wb (wb = 2^H): window 0 holds the column sums C_lo at index lo (weight lo + 1),
// window 1 the row sums R_hi at index hi - 1 (weight hi; R_0 has weight 0 and is dropped); the rest is the identity.
// A chain of additions on few points, i.e. latency: it runs on quads (ecz_coop.cuh), 2^qpl_log adjacent quads per line (<= 16):
// each sums its share of the line's partials, quad-wide shuffles combine.
template <class C>
__global__ void __launch_bounds__(256) k_msm_lsum(TailBatch tb, int L, int H, int g_log, int qpl_log) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    constexpr int RU = raw_u4<FP>();
    const TailSlot& sl = tb.s[blockIdx.y];
    const uint32_t wb = 1u << H;
    const uint32_t nbg = (1u << (L + H)) >> g_log;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int ql = threadIdx.x & 3;
    const uint32_t quad = gid >> 2;
    const uint32_t slot = quad >> qpl_log, part = quad & ((1u << qpl_log) - 1u);
    XyzzZ<FP> acc = xyzzz_identity<FP>();
    const uint4* lines = sl.line_part + (size_t)blockIdx.z * 2 * nbg * RU;  // blockIdx.z: the window (table-free mode)
    if (slot < 2 * wb) {
        const uint32_t win = slot >> H, idx = slot & (wb - 1u);
        const uint4* src = nullptr;
        uint32_t cnt = 0;
        if (win == 0 && idx < (1u << L)) {
            cnt = (1u << H) >> g_log;
            src = lines + ((size_t)nbg + (size_t)idx * cnt) * RU;
        } else if (win == 1 && idx + 1 < wb) {
            cnt = (1u << L) >> g_log;
            src = lines + (size_t)(idx + 1) * cnt * RU;
        }
        for (uint32_t k = part; k < cnt; k += 1u << qpl_log) acc = xyzzz_add_q<FP>(acc, xyzzz_load_raw<FP>(src + (size_t)k * RU), ql);
    }
    acc = wave_sum_q<FP>(acc, 1 << qpl_log, ql);
    if (slot < 2 * wb && part == 0 && ql == 0) xyzzz_store_packed<FP>(sl.bucket + ((size_t)blockIdx.z * 2 * wb + slot) * 4 * W, acc);
}

// The planes and the final kernel run on quads (ecz_coop.cuh): four lanes per point, a doubling is 3
// multiplication latencies deep instead of 9, an addition 4 instead of 14.
//
// plane p of window z: tree-sum of { bucket_b : bit p of (b + 1) } over the window's buckets.
// grid = (parts, planes, windows), 128 quads per block.
constexpr int PLANE_THREADS = 512;
template <class C>
__global__ void __launch_bounds__(PLANE_THREADS) k_msm_planes(TailBatch tb, int windows, uint32_t wbuckets) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    __shared__ uint4 s_pts[(PLANE_THREADS / 64) * 4 * W];  // one packed point per wave
    const int ql = threadIdx.x & 3, quad = threadIdx.x >> 2;
    const int plane = blockIdx.y;
    const int slot = blockIdx.z / windows, win = blockIdx.z % windows;
    uint4* __restrict__ plane_part = tb.s[slot].plane_part;
    const uint4* wb = tb.s[slot].bucket + (size_t)win * wbuckets * 4 * W;
    XyzzZ<FP> acc = xyzzz_identity<FP>();
    for (uint32_t b = blockIdx.x * (PLANE_THREADS / 4) + quad; b < wbuckets; b += gridDim.x * (PLANE_THREADS / 4)) {
        if (((b + 1u) >> plane) & 1u) acc = xyzzz_add_q<FP>(acc, xyzzz_load_packed<FP>(wb + (size_t)b * 4 * W), ql);
    }
    acc = wave_sum_q<FP>(acc, 16, ql);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) xyzzz_store_packed<FP>(s_pts + wave * 4 * W, acc);
    __syncthreads();
    if (wave == 0) {
        acc = (lane >> 2) < PLANE_THREADS / 64 ? xyzzz_load_packed<FP>(s_pts + (lane >> 2) * 4 * W) : xyzzz_identity<FP>();
        acc = wave_sum_q<FP>(acc, PLANE_THREADS / 64, ql);
        if (lane == 0)
            xyzzz_store_packed<FP>(plane_part + (((size_t)win * gridDim.y + plane) * gridDim.x + blockIdx.x) * 4 * W, acc);
    }
}

// One block per window: sum_p 2^p (sum of the parts of plane p), one quad per (plane, part); parts a power of
// two <= 16, planes <= 32, planes * parts <= 256.  With one window (tables) the block also normalises the result; with several
// (table-free mode) it doubles its window into place, 2^(c * window), and k_msm_combine adds the windows.
constexpr int FINAL_FUSE_WINDOWS = 4;  // up to this many tail windows are added by the last block of k_msm_final itself
constexpr int FINAL_THREADS = 512;  // <= 8 waves, so the compiler may use 256 VGPRs: the point arithmetic must not spill
// (256-thread workgroups - one wave per SIMD - were measured for the planes and this kernel: the same 66 / 139 us at c = 20, and the
// one-shot MSM with its 2 x 13 tail windows went from 2.8 to 4.4 ms)
template <class C>
__global__ void __launch_bounds__(FINAL_THREADS) k_msm_final(TailBatch tb, int windows, int parts, int planes, int window_bits, int pair_shift) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    __shared__ uint4 s_pts[33 * 4 * W];
    const int tid = threadIdx.x, ql = tid & 3, item = tid >> 2;
    const int slot = blockIdx.x / windows;
    const uint4* __restrict__ plane_part = tb.s[slot].plane_part;
    // a quad takes two parts when there are several (then 4 * planes * parts / 2 <= FINAL_THREADS), else one
    const int ipq = parts > 1 ? 2 : 1, qpp = parts / ipq;  // quads per plane
    const int plane = item / qpp, sub = item % qpp;
    const int win = blockIdx.x % windows;
    const bool live = plane < planes;
    const uint4* src = plane_part + ((size_t)(win * planes + plane) * parts + sub * ipq) * 4 * W;
    XyzzZ<FP> acc = live ? xyzzz_load_packed<FP>(src) : xyzzz_identity<FP>();
    if (ipq == 2) acc = xyzzz_add_q<FP>(acc, live ? xyzzz_load_packed<FP>(src + 4 * W) : xyzzz_identity<FP>(), ql);
    acc = wave_sum_q<FP>(acc, qpp, ql);  // the quads of a plane are adjacent in one wave
    if (live && sub == 0) {
        for (int k = 0; k < plane; ++k) acc = xyzzz_dbl_q<FP>(acc, ql);
        if (ql == 0) xyzzz_store_packed<FP>(s_pts + plane * 4 * W, acc);
    }
    __syncthreads();
    if (tid < 128) {  // the planes: 32 quads, two waves
        acc = item < planes ? xyzzz_load_packed<FP>(s_pts + item * 4 * W) : xyzzz_identity<FP>();
        acc = wave_sum_q<FP>(acc, 16, ql);
        if (tid == 64) xyzzz_store_packed<FP>(s_pts + 32 * 4 * W, acc);
    }
    __syncthreads();
    if (tid < 4) {
        if (planes > 16) acc = xyzzz_add_q<FP>(acc, xyzzz_load_packed<FP>(s_pts + 32 * 4 * W), ql);
        // pair_shift < 0: window `win` weighs 2^(win * window_bits).  Two-level tail: the windows come in pairs (column sums,
        // row sums) of real window win / 2, the row sums shifted by pair_shift = L more
        const int shift = pair_shift < 0 ? win * window_bits : (win >> 1) * window_bits + (win & 1) * pair_shift;
        for (int k = 0; k < shift; ++k) acc = xyzzz_dbl_q<FP>(acc, ql);
        if (windows == 1) {
            if (tid == 0) emit_affine<FP, true>(acc, tb.s[slot].out_xy, tb.s[slot].out_zero);
        } else {
            uint4* win_pts = tb.s[slot].win_pts;
            if (tid == 0) xyzzz_store_packed<FP>(win_pts + (size_t)win * 4 * W, acc);
            if (windows <= FINAL_FUSE_WINDOWS) {
                // few windows (two in the two-level mode): the block that finishes last adds them up - no launch of its own
                uint32_t seen = 0;
                if (tid == 0) {
                    __threadfence();
                    seen = atomicAdd(tb.s[slot].final_done, 1u);
                }
                seen = __shfl(seen, 0, 4);
                if (seen == (uint32_t)windows - 1u) {
                    __threadfence();
                    for (int o = 0; o < windows; ++o)
                        if (o != win) acc = xyzzz_add_q<FP>(acc, xyzzz_load_packed_volatile<FP>(win_pts + (size_t)o * 4 * W), ql);
                    if (tid == 0) {
                        *tb.s[slot].final_done = 0;
                        emit_affine<FP, true>(acc, tb.s[slot].out_xy, tb.s[slot].out_zero);
                    }
                }
            }
        }
    }
}

// table-free mode: the sum of the windows (<= 128 points, already doubled into place), normalised
constexpr int COMBINE_THREADS = 512;
template <class C>
__global__ void __launch_bounds__(COMBINE_THREADS) k_msm_combine(TailBatch tb, int windows) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    __shared__ uint4 s_pts[(COMBINE_THREADS / 64) * 4 * W];
    const uint4* __restrict__ win_pts = tb.s[blockIdx.x].win_pts;
    uint4* __restrict__ out_xy = tb.s[blockIdx.x].out_xy;
    uint8_t* __restrict__ out_zero = tb.s[blockIdx.x].out_zero;
    const int tid = threadIdx.x, ql = tid & 3, item = tid >> 2;
    XyzzZ<FP> acc = item < windows ? xyzzz_load_packed<FP>(win_pts + (size_t)item * 4 * W) : xyzzz_identity<FP>();
    acc = wave_sum_q<FP>(acc, 16, ql);
    if ((tid & 63) == 0) xyzzz_store_packed<FP>(s_pts + (tid >> 6) * 4 * W, acc);
    __syncthreads();
    if (tid < 64) {
        acc = item < COMBINE_THREADS / 64 ? xyzzz_load_packed<FP>(s_pts + item * 4 * W) : xyzzz_identity<FP>();
        acc = wave_sum_q<FP>(acc, COMBINE_THREADS / 64, ql);
        if (tid == 0) emit_affine<FP, true>(acc, out_xy, out_zero);
    }
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
    for (unsigned r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const uint8_t* rec = gathered + (size_t)r * rec_bytes;
        if (rec[(size_t)slots * 2 * W * 16 + slot]) continue;
        const uint4* pt = (const uint4*)(rec + (size_t)slot * 2 * W * 16);
        Fe<FP> x = fe_load<FP>(pt), y = fe_load<FP>(pt + W);
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
    // With PLK_MSM_COMB=1 a tabled context over few generators (<= COMB_MAX_N, automatic window) is a COMB (comb.hip): no window tables,
    // no workspaces, no bucket method - executions are mixed additions of table entries and a tree over the lanes' sums.
    plk::CombPlan* comb = nullptr;
    bool auto_window = false;
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
        // 2^20 and up: 20 (13 additions per scalar, 2^19 buckets: round 2).
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
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_msm_accumulate<C>, ACC_THREADS, 0) != hipSuccess || per_cu <= 0) per_cu = 6;
        s = (size_t)per_cu * ACC_THREADS * (size_t)cus;
    }
    return s;
}

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
        {&w.head_live, ctx->max_lanes + ACC_THREADS},
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
    // PLK_MSM_COMB=1 (read at every precompute): off by default - measured in round 4 (profiles/r04_comb_small_msm.txt): an execution
    // is faster than the bucket method only up to ~2^12 generators (0.19 / 0.25 ms against 0.30 / 0.33 at 2^10 / 2^12; equal at 2^14,
    // slower at 2^15: eight dependent gathers + additions per lane and two trees of full additions are latency too), its table costs
    // 2-4 x the window tables to build, and the frozen generators of an opening argument (2^14 + 2) gain nothing.
    const char* comb_env = getenv("PLK_MSM_COMB");
    const bool want_comb = comb_env && atoi(comb_env) != 0;
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
        PLK_TRY(rc);
        ctx->c = COMB_WINDOW;
        ctx->windows = COMB_WINDOWS;
        PLK_HIP_TRY(hipStreamSynchronize(stream));
        return PLK_OK;
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
    }
    // tail geometry
    ctx->two_level = c - 1 >= 12;
    ctx->L = ctx->H = ctx->g_log = ctx->lpl_log = 0;
    if (ctx->two_level) {
        ctx->L = (c - 1) / 2;
        ctx->H = c - 1 - ctx->L;
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

static TailSlot tail_slot(const plk_msm_ctx* ctx, const MsmWork& w, void* d_out_xy, void* d_out_zero) {
    (void)ctx;
    TailSlot t;
    t.off = (const uint32_t*)w.off;
    t.p_start = (uint4*)w.p_start;
    t.p_head = (const uint4*)w.p_head;
    t.head_live = (const uint8_t*)w.head_live;
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
    return t;
}

// pieces -> buckets -> (row / column sums ->) bit-plane sums -> result, for the tb.count MSMs of a batch at once
template <class C, class Mark>
static int msm_reduce_t(plk_msm_ctx* ctx, TailBatch tb, hipStream_t stream, Mark&& mark) {
    const uint32_t buckets = ctx->buckets;
    const unsigned cnt = (unsigned)tb.count;
    // hot buckets of a skewed scalar distribution (none for uniform scalars: the three launches then exit at once)
    k_msm_heavy_list<<<dim3((buckets + 255) / 256, cnt), 256, 0, stream>>>(tb, buckets, ctx->heavy_cap, ctx->lpb_log);
    k_msm_heavy_chunks<C><<<dim3(256, cnt), 256, 0, stream>>>(tb, ctx->heavy_cap);
    k_msm_heavy_final<C><<<dim3(64, cnt), 256, 0, stream>>>(tb, ctx->heavy_cap);
    const unsigned ab = (unsigned)((((size_t)buckets << ctx->lpb_log) + 255) / 256);
    if (ctx->two_level) {
        const uint32_t nbg = (1u << (ctx->L + ctx->H)) >> ctx->g_log;  // groups per window (rows; as many for the columns)
        const unsigned wins = ctx->table_free ? (unsigned)ctx->windows : 1u;
        // (Reading the pieces directly in the row / column sums - no k_msm_assemble pass - was built and measured in round 3: the merge of
        // the rare live head pieces, inlined or out of line, takes k_msm_gsum from ~100 to 226-232 registers, and the tail went from
        // 0.228 to 0.259 ms.  The separate 44 us pass stays.)
        k_msm_assemble<C, false><<<dim3(ab, cnt), 256, 0, stream>>>(tb, buckets, ctx->lpb_log);
        k_msm_gsum<C><<<dim3((2 * nbg + 127) / 128, cnt, wins), 128, 0, stream>>>(tb, ctx->L, ctx->H, ctx->g_log);
        const size_t lanes = ((size_t)2 << ctx->H) << (ctx->lpl_log + 2);
        k_msm_lsum<C><<<dim3((unsigned)((lanes + 255) / 256), cnt, wins), 256, 0, stream>>>(tb, ctx->L, ctx->H, ctx->g_log, ctx->lpl_log);
    } else {
        k_msm_assemble<C, true><<<dim3(ab, cnt), 256, 0, stream>>>(tb, buckets, ctx->lpb_log);
    }
    PLK_HIP_TRY(hipGetLastError());
    mark();
    const int tw = ctx->tail_windows;
    dim3 pg(ctx->plane_blocks, ctx->planes, tw * cnt);
    k_msm_planes<C><<<pg, PLANE_THREADS, 0, stream>>>(tb, tw, ctx->tail_wbuckets);
    PLK_HIP_TRY(hipGetLastError());
    mark();
    k_msm_final<C><<<tw * cnt, FINAL_THREADS, 0, stream>>>(tb, tw, ctx->plane_blocks, ctx->planes, ctx->tail_shift, ctx->two_level ? ctx->L : -1);
    if (tw > FINAL_FUSE_WINDOWS) k_msm_combine<C><<<cnt, COMBINE_THREADS, 0, stream>>>(tb, tw);
    PLK_HIP_TRY(hipGetLastError());
    mark();
    return PLK_OK;
}

// per-vector generator ranges of plk_msm_execute_parts_dev (host arrays of `batch` entries; scalars[b]: a device pointer)
struct MsmParts {
    const uint64_t* first;
    const uint64_t* count;
    const void* const* scalars;
};

// phases: 1 = bucket ordering, 2 = accumulation, 4 = reduction; 7 = the whole MSM on one stream
constexpr int PH_ORDER = 1, PH_ACC = 2, PH_REDUCE = 4, PH_ALL = 7;
template <class C>
static int msm_execute_t(plk_msm_ctx* ctx, MsmWork& w, const void* d_scalars, void* d_out_xy, void* d_out_zero, hipStream_t stream,
                         int phases = PH_ALL, size_t first = 0, size_t count = (size_t)-1) {
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
            k_glv_split<C><<<(unsigned)((ctx->n + 255) / 256), 256, 0, stream>>>((const uint4*)d_scalars, ctx->n, (uint4*)halves);
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
        k_ord_count<C><<<o.nt1, ORD_THREADS, 0, stream>>>((const uint4*)d_scalars, n, o, (uint32_t*)w.cnt1);
        const bool one_level = o.fine_bits == 0;
        k_ord_scan1<<<o.nbins, 256, 0, stream>>>((uint32_t*)w.cnt1, o.nt1, o.nbins, bin_total, bin_base, seg_base, done_counter, done_counter + 2,
                                                 ctx->chunk, (uint32_t)(ctx->max_lanes > 2 ? ctx->max_lanes - 2 : 1), one_level ? off : nullptr);
        PLK_HIP_TRY(hipGetLastError());
        mark();
        k_ord_scatter<C><<<o.nt1, ORD_THREADS, 0, stream>>>((const uint4*)d_scalars, n, o, (const uint32_t*)w.cnt1, bin_base, (uint2*)w.tmp,
                                                            one_level ? (uint32_t*)w.sorted : nullptr);
        PLK_HIP_TRY(hipGetLastError());
        mark();
        if (!one_level) {
            // the number of segments is only known on the device: launch for the upper bound (+ nbins blocks that write the
            // offsets of the empty bins), blocks past the end exit
            const unsigned segs = (unsigned)(n * (size_t)o.windows / ORD_SEG + o.nbins + 1);
            k_ord_bin_count<<<segs, ORD_BIN_THREADS, 0, stream>>>((const uint2*)w.tmp, bin_base, seg_base, o.fine_bits, o.nbins, (uint32_t*)w.cnt2);
            k_ord_bin_scatter<<<segs + o.nbins, ORD_BIN_THREADS, 0, stream>>>((const uint2*)w.tmp, bin_base, seg_base, o.fine_bits, o.nbins, buckets,
                                                                              (const uint32_t*)w.cnt2, off, (uint32_t*)w.sorted, o.entries_cap);
        }
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
        k_msm_accumulate<C><<<ablocks, ACC_THREADS, 0, stream>>>((const uint4*)ctx->tab, (const uint32_t*)w.sorted, off, (uint4*)w.p_start, (uint4*)w.p_head,
                                                                 (uint8_t*)w.head_live, buckets, done_counter + 2, ctx->table_free ? ctx->c - 1 : 31,
                                                                 ctx->table_free ? (uint32_t)n : 0u,
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
                         hipEvent_t* ready, const MsmParts* parts) {
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    if (parts) {
        // vector b: parts->count[b] scalars at parts->scalars[b] for the generators parts->first[b] .. (plk_msm_execute_parts_dev)
        if (ctx->table_free) return set_error(PLK_ERR_INVALID_ARG, "a sub-range of the generators needs a tabled context");
        if (!parts->first || !parts->count || !parts->scalars) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
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
    if (ctx->comb) {
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
        uint8_t* oxy = (uint8_t*)d_out_xy + (size_t)b * 2 * L * 8;
        uint8_t* oz = (uint8_t*)d_out_zero + b;
        switch (ctx->curve) {
            case PLK_CURVE_TWEEDLEDEE: return msm_execute_t<TweedledeeCurve>(ctx, w, sc, oxy, oz, st, phases, first, count);
            case PLK_CURVE_TWEEDLEDUM: return msm_execute_t<TweedledumCurve>(ctx, w, sc, oxy, oz, st, phases, first, count);
            case PLK_CURVE_PALLAS: return msm_execute_t<PallasCurve>(ctx, w, sc, oxy, oz, st, phases, first, count);
            case PLK_CURVE_VESTA: return msm_execute_t<VestaCurve>(ctx, w, sc, oxy, oz, st, phases, first, count);
            default: return msm_execute_t<Bls12377Curve>(ctx, w, sc, oxy, oz, st, phases, first, count);
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
                tb.s[0] = tail_slot(ctx, ctx->ws[k], (uint8_t*)d_out_xy + (size_t)b * 2 * L * 8, (uint8_t*)d_out_zero + b);
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
            tb.s[k] = tail_slot(ctx, ctx->ws[k], (uint8_t*)d_out_xy + (size_t)b * 2 * L * 8, (uint8_t*)d_out_zero + b);
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

int msm_set_profiling_impl(plk_msm_ctx* ctx, int enable) {
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->profiling = enable != 0 && !ctx->comb;  // a comb has no stages to time: two launches
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

int checked_build_impl() {
#ifdef PLK_CHECKED
    return 1;
#else
    return 0;
#endif
}
// counts[8]: violations per guarded site since the library was loaded (all zero in the normal build, which has no guards)
int checked_failures_impl(unsigned* counts) {
    if (!counts) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    for (int k = 0; k < 8; ++k) counts[k] = 0;
#ifdef PLK_CHECKED
    PLK_TRY(ensure_device());
    PLK_HIP_TRY(hipDeviceSynchronize());
    PLK_HIP_TRY(hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_plk_chk), 8 * sizeof(unsigned)));
#endif
    return PLK_OK;
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
