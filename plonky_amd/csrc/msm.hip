// msm.hip -- multi-scalar multiplication sum_i s_i * G_i on gfx950 (bucket method on window tables).
//
// Replaces the reference's src/curve/curve_msm.rs (+ curve_summations.rs, curve_adds.rs):
//   msm_precompute / precompute_single_generator  curve_msm.rs:27-52  -> k_msm_table (device tables)
//   to_digits                                     curve_msm.rs:159-180 -> k_msm_digits (signed, carry based)
//   digit_occurrences scatter (serial in the ref) curve_msm.rs:117-126 -> two-level LDS partition (k_part*)
//   per-digit affine multi-summation              curve_msm.rs:131-145 -> k_msm_accumulate (XYZZ mixed adds)
//   serial Yao tail  u += acc[d]; y += u          curve_msm.rs:149-154 -> k_msm_bucket_sum + k_msm_planes + k_msm_final
//   msm_execute / msm_execute_parallel            curve_msm.rs:63-157  -> msm_execute_dev_impl
//   msm_parallel (precompute + execute, one use)  curve_msm.rs:54-61   -> the table-free mode (PLK_MSM_TABLE_FREE)
// Same mathematical structure as the reference (Yao's method over per-generator power tables
// [2^(c j)] G_i, one bucket per digit value, result = sum_d d * bucket_d) with two MI355X-first
// changes: (1) digits are signed (carry-based integer recoding, never s -> r - s, so it is valid
// on BLS12-377 G1 whose cofactor is even): half the buckets for the same window; (2) the serial
// running sum is replaced by bit-plane tree sums (sum_d d B_d = sum_p 2^p sum_{d: bit p} B_d), so
// the tail is O(log) deep instead of 2 * 2^w sequential additions.  The result is returned as
// the unique affine point (to_affine, curve.rs:206-214), on which parity is defined.
//
// Work decomposition: every (scalar i, window j) with a non-zero digit is one *entry* that adds
// +-table[j*n + i] into bucket |d|-1.  Entries are counting-sorted by bucket; every bucket is
// cut into slices of <= SLICE entries and one lane accumulates one slice, so the load per lane
// is bounded whatever the digit distribution (a skewed witness cannot serialise the kernel).
//
// Table-free mode (generators used once): only the generators themselves are stored; window j gets its own
// bucket range [j 2^(c-1), (j+1) 2^(c-1)), the same kernels run over all windows at once, every window's
// plane sum is doubled into place (2^(c j)) by a quad and k_msm_combine adds the windows.
// Batches: every MSM of a group has its own workspace and the group shares one reduction (msm_execute_dev_impl).
// The reduction kernels run on quads of lanes (ecz_coop.cuh): they are chains of point operations, i.e. latency.
#include <mutex>
#include <vector>

#include "common.h"
#include "ec.cuh"
#include "ecz.cuh"
#include "ecz_coop.cuh"
#include "tables.cuh"

namespace plk {

constexpr int MSM_SLICE_DEFAULT = 24;  // entries per accumulation slice (PLK_MSM_SLICE overrides)
constexpr int MSM_MAX_PLANE_PARTS = 16;  // blocks per bit-plane in the reduction (planes * parts quads must fit the final block)
constexpr int MSM_MAX_WINDOW = 17;   // c - 1 <= 8 fine + 8 coarse bits in the partition
constexpr uint32_t CODE_INVALID = 0xFFFFFFFFu;

// ---------------------------------------------------------------------------------------------
// table construction: tab[j*n + i] = [2^(c j)] G_i, affine  (curve_msm.rs:40-52)
// ---------------------------------------------------------------------------------------------
// The table is what the accumulation kernel multiplies with, so it is stored in the working form of
// that kernel (ecz.cuh / fz.cuh): coordinates in R'-form (x 2^(29 NZ)), canonical, packed in the
// same 32-bit words.  The generators arrive in the reference's R-form.
// One lane per generator, on the lazy arithmetic of the accumulation kernel (ecz.cuh): c doublings per window,
// then back to affine with the division-step inversion (about a quarter of a window's work).
template <class C>
__global__ void __launch_bounds__(128) k_msm_table(const uint4* __restrict__ bases, const uint8_t* __restrict__ base_zero, uint4* __restrict__ tab,
                                                   size_t n, int c, int windows) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // R-form (the reference's) -> canonical R'-form, the form the table is stored and consumed in
    Fe<FP> xr = to_rprime<FP>(fe_load<FP>(bases + i * 2 * W)), yr = to_rprime<FP>(fe_load<FP>(bases + i * 2 * W + W));
    bool ident = base_zero ? base_zero[i] != 0 : false;
    affine_store<FP>(tab + i * 2 * W, xr, yr, ident);
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
// scalars -> signed window digits  (curve_msm.rs:159-180)
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(256) k_msm_digits(const uint4* __restrict__ scalars, uint32_t* __restrict__ codes, size_t n, int c, int windows,
                                                    uint32_t window_buckets) {
    using SP = typename C::SP;
    static_assert(SP::NL == 8, "scalar fields are 256-bit");
    // Scalars are staged through LDS: the block reads its 256 * 32 B with fully coalesced 16-byte
    // loads, each lane then picks up its own scalar, converts it and parks the canonical limbs
    // back in LDS so the window loop can index them dynamically.
    __shared__ uint4 s_sc[512];
    const size_t base = (size_t)blockIdx.x * 256;
    for (int k = threadIdx.x; k < 512; k += 256) {
        size_t g = base * 2 + k;
        s_sc[k] = g < n * 2 ? scalars[g] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const size_t i = base + threadIdx.x;
    Fe<SP> s;
    {
        uint4 lo = s_sc[2 * threadIdx.x], hi = s_sc[2 * threadIdx.x + 1];
        s.v[0] = lo.x; s.v[1] = lo.y; s.v[2] = lo.z; s.v[3] = lo.w;
        s.v[4] = hi.x; s.v[5] = hi.y; s.v[6] = hi.z; s.v[7] = hi.w;
    }
    // Montgomery -> canonical in the SCALAR field (to_canonical_u64_vec, curve_msm.rs:164)
    s = fe_to_canonical<SP>(s);
    uint32_t* lim = reinterpret_cast<uint32_t*>(s_sc) + threadIdx.x * 8;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) lim[k] = s.v[k];
    if (i >= n) return;
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    uint32_t carry = 0;
    for (int j = 0; j < windows; ++j) {
        const int bp = j * c, li = bp >> 5, sh = bp & 31;
        uint64_t two = li < 8 ? lim[li] : 0u;
        if (li + 1 < 8) two |= (uint64_t)lim[li + 1] << 32;
        uint32_t v = ((uint32_t)(two >> sh) & mask) + carry;
        // signed recoding: v in [0, 2^c]; v > 2^(c-1) becomes v - 2^c with a carry into the next window
        uint32_t neg = v > half ? 1u : 0u;
        uint32_t mag = neg ? (1u << c) - v : v;
        carry = neg;
        uint32_t code = CODE_INVALID;
        // bucket id: |d| - 1, plus the window's own bucket range in table-free mode (window_buckets = 2^(c-1), else 0)
        if (mag != 0) code = ((mag - 1u + (uint32_t)j * window_buckets) << 1) | neg;
        codes[(size_t)j * n + i] = code;
    }
}

// exclusive scans of the bucket sizes (off) and of the per-bucket slice counts (slice_off), two steps:
// every block scans 1024 buckets and publishes its totals, then every block adds the totals of the
// blocks before it (<= 64 of them: a serial walk by one lane is cheaper than another launch).
__global__ void __launch_bounds__(1024) k_msm_scan_local(const uint32_t* __restrict__ hist, uint32_t* __restrict__ off, uint32_t* __restrict__ slice_off,
                                                         uint32_t* __restrict__ block_tot, uint32_t buckets, uint32_t slice) {
    __shared__ uint32_t s_a[1024], s_b[1024];
    const uint32_t b = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t h = b < buckets ? hist[b] : 0u;
    const uint32_t sl = (h + slice - 1) / slice;
    s_a[threadIdx.x] = h;
    s_b[threadIdx.x] = sl;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        uint32_t va = 0, vb = 0;
        if ((int)threadIdx.x >= d) {
            va = s_a[threadIdx.x - d];
            vb = s_b[threadIdx.x - d];
        }
        __syncthreads();
        s_a[threadIdx.x] += va;
        s_b[threadIdx.x] += vb;
        __syncthreads();
    }
    if (b < buckets) {
        off[b] = s_a[threadIdx.x] - h;
        slice_off[b] = s_b[threadIdx.x] - sl;
    }
    if (threadIdx.x == 1023) {
        block_tot[2 * blockIdx.x] = s_a[1023];
        block_tot[2 * blockIdx.x + 1] = s_b[1023];
    }
}
__global__ void __launch_bounds__(1024) k_msm_scan_add(uint32_t* __restrict__ off, uint32_t* __restrict__ slice_off, const uint32_t* __restrict__ block_tot,
                                                       uint32_t buckets) {
    __shared__ uint32_t s_add[2];
    if (threadIdx.x == 0) {
        uint32_t a = 0, c = 0;
        for (uint32_t k = 0; k < blockIdx.x; ++k) {
            a += block_tot[2 * k];
            c += block_tot[2 * k + 1];
        }
        s_add[0] = a;
        s_add[1] = c;
        if (blockIdx.x == gridDim.x - 1) {  // grand totals close both arrays
            off[buckets] = a + block_tot[2 * blockIdx.x];
            slice_off[buckets] = c + block_tot[2 * blockIdx.x + 1];
        }
    }
    __syncthreads();
    const uint32_t b = blockIdx.x * 1024 + threadIdx.x;
    if (b < buckets) {
        off[b] += s_add[0];
        slice_off[b] += s_add[1];
    }
}

// ---------------------------------------------------------------------------------------------
// entries -> bucket order: two-level MSD partition in LDS  (replaces the reference's serial
// digit_occurrences scatter, curve_msm.rs:117-126)
// ---------------------------------------------------------------------------------------------
// A bucket id has c-1 bits = [coarse | fine], fine = min(8, c-1) bits.  Level 1 splits the entry
// stream into <= 256 coarse bins, level 2 splits every coarse bin into its <= 256 buckets.  Both
// levels are the same three steps on tiles of PART_TILE entries: per-tile LDS histogram ->
// global [bin][tile] counts -> scan -> per-tile LDS cursors, so every global write is a run of
// consecutive slots and the only atomics are LDS atomics.  Counts, not capacities, drive the
// layout: any digit distribution works (hot buckets just make long runs).
constexpr int PART_TILE_LOG = 12;
constexpr int PART_TILE = 1 << PART_TILE_LOG;   // entries per tile
constexpr int PART_THREADS = 256;
constexpr int PART_PER_THREAD = PART_TILE / PART_THREADS;

// level 1, step 1: cnt1[bin * nt1 + tile] = number of entries of `tile` falling in coarse bin `bin`
__global__ void __launch_bounds__(PART_THREADS) k_part1_count(const uint32_t* __restrict__ codes, size_t entries, uint32_t* __restrict__ cnt1, uint32_t nt1,
                                                              int fine_bits, int nbins) {
    __shared__ uint32_t s_hist[256];
    const uint32_t tile = blockIdx.x;
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)tile << PART_TILE_LOG;
#pragma unroll
    for (int k = 0; k < PART_PER_THREAD; ++k) {
        const size_t e = base + k * PART_THREADS + threadIdx.x;
        if (e < entries) {
            const uint32_t code = codes[e];
            if (code != CODE_INVALID) atomicAdd(&s_hist[code >> (fine_bits + 1)], 1u);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nbins) cnt1[(size_t)threadIdx.x * nt1 + tile] = s_hist[threadIdx.x];
}

// one block per row: in-place exclusive scan of `len` counters, total to totals[row]
__global__ void __launch_bounds__(256) k_part_rowscan(uint32_t* __restrict__ cnt, uint32_t len, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_sum[256];
    uint32_t* row = cnt + (size_t)blockIdx.x * len;
    const uint32_t per = (len + 255) / 256;
    const uint32_t lo = threadIdx.x * per, hi = min(len, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += row[i];
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        uint32_t v = (int)threadIdx.x >= d ? s_sum[threadIdx.x - d] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_sum[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        uint32_t v = row[i];
        row[i] = run;
        run += v;
    }
    if (threadIdx.x == 255) totals[blockIdx.x] = s_sum[255];
}

// layout of the intermediate array: every coarse bin starts on a tile boundary (so level-2 tiles
// never straddle bins).  meta[0] = number of level-2 tiles.
__global__ void __launch_bounds__(256) k_part_bases(const uint32_t* __restrict__ bin_total, int nbins, uint32_t* __restrict__ bin_base_pad,
                                                    uint32_t* __restrict__ tile2bin, uint32_t* __restrict__ meta) {
    __shared__ uint32_t s_pad[257];
    __shared__ uint32_t s_scan[256];
    // exclusive scan of the bins' sizes rounded up to whole tiles (nbins <= 256 = blockDim.x)
    const int t = threadIdx.x;
    const uint32_t mine = t < nbins ? (bin_total[t] + PART_TILE - 1) >> PART_TILE_LOG << PART_TILE_LOG : 0u;
    s_scan[t] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t u = t >= d ? s_scan[t - d] : 0u;
        __syncthreads();
        s_scan[t] += u;
        __syncthreads();
    }
    if (t < nbins) s_pad[t] = s_scan[t] - mine;
    if (t == nbins - 1) {
        s_pad[nbins] = s_scan[t];
        meta[0] = s_scan[t] >> PART_TILE_LOG;
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= nbins; b += blockDim.x) bin_base_pad[b] = s_pad[b];
    // every thread fills the tiles of its own bin
    if (t < nbins) {
        const uint32_t t0 = s_pad[t] >> PART_TILE_LOG, t1 = s_pad[t + 1] >> PART_TILE_LOG;
        for (uint32_t k = t0; k < t1; ++k) tile2bin[k] = (uint32_t)t;
    }
}

// Shared by both scatter steps: the tile is first ordered by bin inside LDS (returning LDS atomics give
// the rank inside the (tile, bin) run, a 256-wide scan gives the run starts), then written out in
// that order, so consecutive lanes store to consecutive addresses of the same run.
PLK_DI void part_scan256(uint32_t* s_cnt, uint32_t* s_base) {
    // exclusive scan of s_cnt[0..255] into s_base (Hillis-Steele, blockDim.x == 256)
    const int t = threadIdx.x;
    uint32_t v = s_cnt[t];
    s_base[t] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t u = t >= d ? s_base[t - d] : 0u;
        __syncthreads();
        s_base[t] += u;
        __syncthreads();
    }
    const uint32_t incl = s_base[t];
    __syncthreads();
    s_base[t] = incl - v;
    __syncthreads();
}

// level 1, step 3: move (code, entry id) to its coarse bin
__global__ void __launch_bounds__(PART_THREADS) k_part1_scatter(const uint32_t* __restrict__ codes, size_t entries, const uint32_t* __restrict__ cnt1,
                                                                uint32_t nt1, const uint32_t* __restrict__ bin_base_pad, int fine_bits, int nbins,
                                                                uint32_t* __restrict__ tmp_code, uint32_t* __restrict__ tmp_val) {
    __shared__ uint32_t s_cnt[256], s_base[256], s_gbase[256];
    __shared__ uint32_t s_code[PART_TILE], s_val[PART_TILE];
    const uint32_t tile = blockIdx.x;
    s_cnt[threadIdx.x] = 0;
    s_gbase[threadIdx.x] = (int)threadIdx.x < nbins ? bin_base_pad[threadIdx.x] + cnt1[(size_t)threadIdx.x * nt1 + tile] : 0u;
    __syncthreads();
    const size_t base = (size_t)tile << PART_TILE_LOG;
    uint32_t code[PART_PER_THREAD], rank[PART_PER_THREAD];
#pragma unroll
    for (int k = 0; k < PART_PER_THREAD; ++k) {
        const size_t e = base + k * PART_THREADS + threadIdx.x;
        code[k] = e < entries ? codes[e] : CODE_INVALID;
        rank[k] = code[k] != CODE_INVALID ? atomicAdd(&s_cnt[code[k] >> (fine_bits + 1)], 1u) : 0u;
    }
    __syncthreads();
    part_scan256(s_cnt, s_base);
#pragma unroll
    for (int k = 0; k < PART_PER_THREAD; ++k) {
        if (code[k] != CODE_INVALID) {
            const uint32_t sidx = s_base[code[k] >> (fine_bits + 1)] + rank[k];
            s_code[sidx] = code[k];
            s_val[sidx] = (uint32_t)(base + k * PART_THREADS + threadIdx.x);
        }
    }
    __syncthreads();
    const uint32_t total = s_base[255] + s_cnt[255];
#pragma unroll
    for (int k = 0; k < PART_PER_THREAD; ++k) {
        const uint32_t sidx = k * PART_THREADS + threadIdx.x;
        if (sidx < total) {
            const uint32_t c = s_code[sidx], bin = c >> (fine_bits + 1);
            const uint32_t pos = s_gbase[bin] + (sidx - s_base[bin]);
            tmp_code[pos] = c;
            tmp_val[pos] = s_val[sidx];
        }
    }
}

// level 2, step 1: cnt2[fine * nt2max + tile2]  (row = bucket-within-bin, so a row scan over the
// tiles of one bin gives the within-bucket offsets)
__global__ void __launch_bounds__(PART_THREADS) k_part2_count(const uint32_t* __restrict__ tmp_code, const uint32_t* __restrict__ bin_total,
                                                              const uint32_t* __restrict__ bin_base_pad, const uint32_t* __restrict__ tile2bin,
                                                              const uint32_t* __restrict__ meta, uint32_t* __restrict__ cnt2, uint32_t nt2max, int fine_bits) {
    __shared__ uint32_t s_hist[256];
    const uint32_t tile = blockIdx.x;
    if (tile >= meta[0]) return;
    const uint32_t bin = tile2bin[tile];
    const uint32_t valid_end = bin_base_pad[bin] + bin_total[bin];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = tile << PART_TILE_LOG, fmask = (1u << fine_bits) - 1u;
#pragma unroll
    for (int k = 0; k < PART_PER_THREAD; ++k) {
        const uint32_t p = base + k * PART_THREADS + threadIdx.x;
        if (p < valid_end) atomicAdd(&s_hist[(tmp_code[p] >> 1) & fmask], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= fmask) cnt2[(size_t)threadIdx.x * nt2max + tile] = s_hist[threadIdx.x];
}

// level 2, step 2: one block per coarse bin, lane = fine bucket: walk the bin's tiles, turn the counts
// into within-bucket offsets and publish the bucket sizes (hist) for the global bucket scan
__global__ void __launch_bounds__(256) k_part2_scan(uint32_t* __restrict__ cnt2, uint32_t nt2max, const uint32_t* __restrict__ bin_base_pad, int fine_bits,
                                                    uint32_t* __restrict__ hist) {
    const uint32_t bin = blockIdx.x, fine = threadIdx.x;
    if (fine >= (1u << fine_bits)) return;
    const uint32_t t0 = bin_base_pad[bin] >> PART_TILE_LOG, t1 = bin_base_pad[bin + 1] >> PART_TILE_LOG;
    uint32_t run = 0;
    uint32_t* row = cnt2 + (size_t)fine * nt2max;
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t v = row[t];
        row[t] = run;
        run += v;
    }
    hist[(bin << fine_bits) + fine] = run;
}

// level 2, step 3: final position = off[bucket] + within-bucket offset of this tile + rank in the tile
__global__ void __launch_bounds__(PART_THREADS) k_part2_scatter(const uint32_t* __restrict__ tmp_code, const uint32_t* __restrict__ tmp_val,
                                                                const uint32_t* __restrict__ bin_total, const uint32_t* __restrict__ bin_base_pad,
                                                                const uint32_t* __restrict__ tile2bin, const uint32_t* __restrict__ meta,
                                                                const uint32_t* __restrict__ cnt2, uint32_t nt2max, int fine_bits,
                                                                const uint32_t* __restrict__ off, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t s_cnt[256], s_base[256], s_gbase[256];
    __shared__ uint32_t s_out[PART_TILE];
    __shared__ uint8_t s_fine[PART_TILE];
    const uint32_t tile = blockIdx.x;
    if (tile >= meta[0]) return;
    const uint32_t bin = tile2bin[tile];
    const uint32_t valid_end = bin_base_pad[bin] + bin_total[bin];
    const uint32_t fmask = (1u << fine_bits) - 1u;
    s_cnt[threadIdx.x] = 0;
    s_gbase[threadIdx.x] = threadIdx.x <= fmask ? off[(bin << fine_bits) + threadIdx.x] + cnt2[(size_t)threadIdx.x * nt2max + tile] : 0u;
    __syncthreads();
    const uint32_t base = tile << PART_TILE_LOG;
    uint32_t word[PART_PER_THREAD], fine[PART_PER_THREAD], rank[PART_PER_THREAD];
#pragma unroll
    for (int k = 0; k < PART_PER_THREAD; ++k) {
        const uint32_t p = base + k * PART_THREADS + threadIdx.x;
        fine[k] = 0xFFFFFFFFu;
        if (p < valid_end) {
            const uint32_t c = tmp_code[p];
            fine[k] = (c >> 1) & fmask;
            word[k] = (tmp_val[p] << 1) | (c & 1u);
            rank[k] = atomicAdd(&s_cnt[fine[k]], 1u);
        }
    }
    __syncthreads();
    part_scan256(s_cnt, s_base);
#pragma unroll
    for (int k = 0; k < PART_PER_THREAD; ++k) {
        if (fine[k] != 0xFFFFFFFFu) {
            const uint32_t sidx = s_base[fine[k]] + rank[k];
            s_out[sidx] = word[k];
            s_fine[sidx] = (uint8_t)fine[k];
        }
    }
    __syncthreads();
    const uint32_t total = s_base[255] + s_cnt[255];
#pragma unroll
    for (int k = 0; k < PART_PER_THREAD; ++k) {
        const uint32_t sidx = k * PART_THREADS + threadIdx.x;
        if (sidx < total) {
            const uint32_t f = s_fine[sidx];
            sorted[s_gbase[f] + (sidx - s_base[f])] = s_out[sidx];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bucket accumulation: one lane per slice of <= `slice` sorted entries
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(128) k_msm_accumulate(const uint4* __restrict__ tab, const uint32_t* __restrict__ sorted,
                                                           const uint32_t* __restrict__ off, const uint32_t* __restrict__ slice_off,
                                                           uint4* __restrict__ partial, uint32_t buckets, uint32_t slice, int wshift,
                                                           uint32_t n_sub) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = slice_off[buckets];
    if (s >= total) return;
    // bucket of slice s: largest b with slice_off[b] <= s
    uint32_t lo = 0, hi = buckets;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (slice_off[mid] <= s) lo = mid; else hi = mid;
    }
    const uint32_t b = lo;
    // entry ids are window * n + generator; with tables that is the table index, without (table-free mode:
    // n_sub = n, buckets of window w are [w << wshift, (w + 1) << wshift)) the window part is taken off
    const uint32_t ent_sub = (b >> wshift) * n_sub;
    const uint32_t begin = off[b] + (s - slice_off[b]) * slice;
    const uint32_t end = min(off[b + 1], begin + slice);
    // lazy 29-bit-limb accumulator (ecz.cuh); table coordinates are R'-form
    XyzzZ<FP> acc;
    acc.inf = true;
    acc.x = acc.y = acc.zz = acc.zzz = fz_zero<FP>();
    // Software pipeline: the table gather for entry k+1 (two dependent loads: index, then a random
    // 64/96-byte point) is issued before the ~10^4-cycle addition of entry k, so the few resident
    // waves (the lazy arithmetic wants ~180 VGPRs) never wait on HBM.
    uint32_t ent = 0;
    Fe<FP> x = fe_zero<FP>(), y = fe_zero<FP>();
    bool ident = true;
    if (begin < end) {
        ent = sorted[begin];
        ident = affine_load<FP>(tab + (size_t)((ent >> 1) - ent_sub) * 2 * W, x, y);
    }
    for (uint32_t k = begin; k < end; ++k) {
        const uint32_t cur = ent;
        const Fe<FP> cx = x, cy = y;
        const bool cident = ident;
        if (k + 1 < end) {
            ent = sorted[k + 1];
            ident = affine_load<FP>(tab + (size_t)((ent >> 1) - ent_sub) * 2 * W, x, y);
        }
        if (cident) continue;
        Fz<FP> xz = fz_from_fe<FP>(cx), yz = fz_from_fe<FP>(cy);
        if (cur & 1u) yz = fz_neg_canonical<FP>(yz);
        xyzzz_madd<FP>(acc, xz, yz);
    }
    xyzzz_store_packed<FP>(partial + (size_t)s * 4 * W, acc);  // R'-form, canonical (exchange format of ecz.cuh)
}

// ---------------------------------------------------------------------------------------------
// reduction  sum_d d * bucket_d   (replaces the serial Yao tail of curve_msm.rs:149-154)
// ---------------------------------------------------------------------------------------------
// Everything here is latency bound (few points, long dependent chains), so the structure is
// chosen for depth, not work, and runs on the lazy arithmetic (ecz.cuh, fully inlined):
//  * bucket_b = sum of its slice partials: 4 lanes per bucket, each sums every 4th slice, two
//    xor-shuffle additions combine them;
//  * sum_b (b+1) bucket_b = sum_p 2^p P_p with P_p the plain sum of the buckets whose weight b+1 has
//    bit p set: c planes, each a tree sum (serial part per lane, 6 shuffle levels per wave, LDS
//    across waves), c * D / 2 additions in total (~1.5 % of the accumulation work);
//  * one block sums the parts of every plane, doubles the planes into place, adds them and
//    normalises the result (to_affine, curve.rs:206-214).
constexpr int BUCKET_LANES = 4;

template <class FP> PLK_DI XyzzZ<FP> wave_sum(XyzzZ<FP> v, int width) {
    for (int m = 1; m < width; m <<= 1) v = xyzzz_add<FP>(v, xyzzz_shfl_xor<FP>(v, m));
    return v;
}

// A bucket with more than HEAVY_SLICES slice partials (a hot digit of a skewed witness) would make its
// 4 lanes walk thousands of additions.  Such buckets are listed as (bucket, chunk) work items of
// HEAVY_CHUNK slices, each summed by a whole workgroup, then their chunk partials are summed.
// The reduction kernels serve several MSMs per launch (the executions of a batch share one tail): every MSM brings
// its own buffers in a slot, blockIdx.y (planes, final: a factor of the grid) picks the slot.
constexpr int TAIL_MAX = 16;
struct TailSlot {
    const uint4* partial;
    const uint32_t* slice_off;
    uint4* bucket;
    uint32_t* heavy;
    uint4* heavy_part;
    uint4* plane_part;
    uint4* win_pts;
    uint4* out_xy;
    uint8_t* out_zero;
};
struct TailBatch {
    int count;
    TailSlot s[TAIL_MAX];
};

constexpr uint32_t HEAVY_SLICES = 256;
constexpr uint32_t HEAVY_CHUNK = 2048;

// heavy[0] = number of work items, heavy[1] = number of heavy buckets;
// items at heavy[2 + 2k] = bucket, heavy[3 + 2k] = chunk index; heavy bucket ids at heavy[2 + 2 cap + k]
__global__ void __launch_bounds__(256) k_msm_heavy_list(TailBatch tb, uint32_t buckets, uint32_t cap) {
    const uint32_t* __restrict__ slice_off = tb.s[blockIdx.y].slice_off;
    uint32_t* __restrict__ heavy = tb.s[blockIdx.y].heavy;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= buckets) return;
    const uint32_t ns = slice_off[b + 1] - slice_off[b];
    if (ns <= HEAVY_SLICES) return;
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
    constexpr int W = FP::NL / 4;
    acc = wave_sum<FP>(acc, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) xyzzz_store_packed<FP>(s_pts + wave * 4 * W, acc);
    __syncthreads();
    acc = (wave == 0 && lane < 4) ? xyzzz_load_packed<FP>(s_pts + lane * 4 * W) : xyzzz_identity<FP>();
    if (wave == 0) acc = wave_sum<FP>(acc, 4);
    __syncthreads();
    return acc;  // valid in thread 0
}

// one workgroup per (bucket, chunk) item: chunk partial -> heavy_part[item]
template <class C>
__global__ void __launch_bounds__(256) k_msm_heavy_chunks(TailBatch tb, uint32_t cap) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    __shared__ uint4 s_pts[4 * 4 * W];
    const uint4* __restrict__ partial = tb.s[blockIdx.y].partial;
    const uint32_t* __restrict__ slice_off = tb.s[blockIdx.y].slice_off;
    const uint32_t* __restrict__ heavy = tb.s[blockIdx.y].heavy;
    uint4* __restrict__ heavy_part = tb.s[blockIdx.y].heavy_part;
    const uint32_t items = min(heavy[0], cap);
    for (uint32_t it = blockIdx.x; it < items; it += gridDim.x) {
        const uint32_t b = heavy[2 + 2 * it], k = heavy[3 + 2 * it];
        const uint32_t s0 = slice_off[b] + k * HEAVY_CHUNK, s1 = min(slice_off[b + 1], s0 + HEAVY_CHUNK);
        XyzzZ<FP> acc = xyzzz_identity<FP>();
        for (uint32_t s = s0 + threadIdx.x; s < s1; s += 256) acc = xyzzz_add<FP>(acc, xyzzz_load_packed<FP>(partial + (size_t)s * 4 * W));
        acc = block256_sum<FP>(acc, s_pts);
        if (threadIdx.x == 0) xyzzz_store_packed<FP>(heavy_part + (size_t)it * 4 * W, acc);
    }
}
// one workgroup per heavy bucket: sum of its chunk partials -> bucket[b].  Items of one bucket are contiguous.
template <class C>
__global__ void __launch_bounds__(256) k_msm_heavy_final(TailBatch tb, uint32_t cap) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    __shared__ uint4 s_pts[4 * 4 * W];
    const uint32_t* __restrict__ heavy = tb.s[blockIdx.y].heavy;
    const uint4* __restrict__ heavy_part = tb.s[blockIdx.y].heavy_part;
    uint4* __restrict__ bucket = tb.s[blockIdx.y].bucket;
    const uint32_t items = min(heavy[0], cap), nb = min(heavy[1], cap);
    for (uint32_t hb = blockIdx.x; hb < nb; hb += gridDim.x) {
        const uint32_t b = heavy[2 + 2 * cap + hb];
        XyzzZ<FP> acc = xyzzz_identity<FP>();
        for (uint32_t it = threadIdx.x; it < items; it += 256)
            if (heavy[2 + 2 * it] == b) acc = xyzzz_add<FP>(acc, xyzzz_load_packed<FP>(heavy_part + (size_t)it * 4 * W));
        acc = block256_sum<FP>(acc, s_pts);
        if (threadIdx.x == 0) xyzzz_store_packed<FP>(bucket + (size_t)b * 4 * W, acc);
    }
}

template <class C>
__global__ void __launch_bounds__(256) k_msm_bucket_sum(TailBatch tb, uint32_t buckets) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    const uint4* __restrict__ partial = tb.s[blockIdx.y].partial;
    const uint32_t* __restrict__ slice_off = tb.s[blockIdx.y].slice_off;
    uint4* __restrict__ bucket = tb.s[blockIdx.y].bucket;
    if (blockIdx.x == 0 && threadIdx.x < 2) tb.s[blockIdx.y].heavy[threadIdx.x] = 0;  // counters of k_msm_heavy_list, which runs next
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = gid / BUCKET_LANES, part = gid % BUCKET_LANES;
    XyzzZ<FP> acc = xyzzz_identity<FP>();
    bool mine = false;
    if (b < buckets) {
        const uint32_t s0 = slice_off[b], s1 = slice_off[b + 1];
        mine = s1 - s0 <= HEAVY_SLICES;  // heavier buckets are summed by k_msm_heavy_*
        if (mine)
            for (uint32_t s = s0 + part; s < s1; s += BUCKET_LANES) acc = xyzzz_add<FP>(acc, xyzzz_load_packed<FP>(partial + (size_t)s * 4 * W));
    }
    acc = wave_sum<FP>(acc, BUCKET_LANES);  // lanes of a bucket are adjacent; every lane takes part in the shuffles
    if (mine && part == 0) xyzzz_store_packed<FP>(bucket + (size_t)b * 4 * W, acc);
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
constexpr int FINAL_THREADS = 512;  // <= 8 waves, so the compiler may use 256 VGPRs: the point arithmetic must not spill
template <class C>
__global__ void __launch_bounds__(FINAL_THREADS) k_msm_final(TailBatch tb, int windows, int parts, int planes, int window_bits) {
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
        for (int k = 0; k < win * window_bits; ++k) acc = xyzzz_dbl_q<FP>(acc, ql);
        if (tid == 0) {
            if (windows > 1) xyzzz_store_packed<FP>(tb.s[slot].win_pts + (size_t)win * 4 * W, acc);
            else emit_affine<FP>(acc, tb.s[slot].out_xy, tb.s[slot].out_zero);
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
        if (tid == 0) emit_affine<FP>(acc, out_xy, out_zero);
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

// per-execution device workspace; a context owns two so that consecutive MSMs of a batch can overlap
struct MsmWork {
    void* codes = nullptr;
    void* sorted = nullptr;
    void* hist = nullptr;      // bucket sizes
    void* cnt1 = nullptr;      // [nbins][nt1]
    void* cnt2 = nullptr;      // [2^fine_bits][nt2max]
    void* tmp_code = nullptr;  // entries + nbins * PART_TILE
    void* tmp_val = nullptr;
    void* part_meta = nullptr; // bin_total[256] | bin_base_pad[257] | meta[1] | tile2bin[nt2max] | scan block totals[128]
    void* off = nullptr;       // off[buckets+1] followed by slice_off[buckets+1]
    void* partial = nullptr;
    void* bucket = nullptr;    // bucket sums (XYZZ)
    void* heavy = nullptr;     // heavy-bucket work list (see k_msm_heavy_list)
    void* heavy_part = nullptr;
    void* plane_part = nullptr;
    void* win_pts = nullptr;   // table-free: the per-window results
    void* slab = nullptr;      // the one allocation all of the above point into
    bool ready = false;
    void release() {
        if (slab) (void)hipFree(slab);
        slab = nullptr;
        for (void** p : {&codes, &sorted, &hist, &cnt1, &cnt2, &tmp_code, &tmp_val, &part_meta, &off, &partial, &bucket, &heavy, &heavy_part, &plane_part, &win_pts})
            *p = nullptr;
        ready = false;
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
    uint32_t slice = 32;   // entries per accumulation slice
    int planes = 0;        // = c: bit-planes of the bucket weights 1 .. 2^(c-1)
    int plane_blocks = 1;  // blocks (parts) per plane
    size_t max_slices = 0;
    // device memory
    void* tab = nullptr;
    // two-level partition workspace
    int fine_bits = 0, nbins = 1;
    uint32_t nt1 = 0, nt2max = 0;
    uint32_t heavy_cap = 0;
    std::vector<MsmWork> ws;   // ws[0] at precompute; a batched execution allocates one per MSM of a group (<= TAIL_MAX)
    size_t ws_bytes = 0;       // size of one workspace slab
    std::mutex mu;             // one execution at a time per context (workspaces are shared)
    // optional per-kernel timing (HIP events on the launch stream) for bench.py's roofline
    bool profiling = false;
    static constexpr int N_STAGES = 7;  // digits, scan, scatter, accumulate, chunks, planes, final
    std::vector<std::vector<hipEvent_t>> prof_sets;  // each N_STAGES + 1 events, recorded
    std::vector<std::vector<hipEvent_t>> prof_free;
    ~plk_msm_ctx() {
        if (tab) (void)hipFree(tab);
        for (MsmWork& w : ws) w.release();
        for (auto* v : {&prof_sets, &prof_free})
            for (auto& set : *v)
                for (hipEvent_t e : set) (void)hipEventDestroy(e);
    }
};

namespace plk {

static int scalar_bits(int curve) { return curve == PLK_CURVE_BLS12_377 ? 253 : 255; }

static int choose_window(size_t n) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) ++lg;
    // ceil(256 / c) windows of work per scalar: 16 and 18..20 are the useful sizes near 2^20; buckets
    // (2^(c-1)) should stay well below the entry count so that slices are long
    int c = lg - 4;
    if (const char* e = getenv("PLK_MSM_WINDOW")) c = atoi(e);
    if (c < 3) c = 3;
    if (c > 16) c = 16;
    return c;
}

// One slab per workspace: a single hipMalloc / hipFree instead of fifteen (they dominate a one-shot msm_parallel).
template <class C> static int msm_alloc_work(plk_msm_ctx* ctx, MsmWork& w) {
    using FP = typename C::FP;
    const size_t xyzz_bytes = (size_t)4 * FP::NL * 4;
    const size_t entries = ctx->n * ctx->windows;
    ctx->nt1 = (uint32_t)((entries + PART_TILE - 1) >> PART_TILE_LOG);
    if (ctx->nt1 == 0) ctx->nt1 = 1;
    ctx->nt2max = ctx->nt1 + ctx->nbins;
    ctx->max_slices = entries / ctx->slice + ctx->buckets + 1;
    // at most max_slices / HEAVY_SLICES heavy buckets, max_slices / HEAVY_CHUNK + that many chunk items
    ctx->heavy_cap = (uint32_t)(ctx->max_slices / HEAVY_SLICES + ctx->max_slices / HEAVY_CHUNK + 2);
    const int bucket_windows = ctx->table_free ? ctx->windows : 1;
    struct Part { void** p; size_t bytes; };
    const Part parts[] = {
        {&w.codes, entries * 4 + 16},
        {&w.sorted, entries * 4 + 16},
        {&w.hist, (size_t)ctx->buckets * 4 + 16},
        {&w.cnt1, (size_t)ctx->nbins * ctx->nt1 * 4},
        {&w.cnt2, ((size_t)1 << ctx->fine_bits) * ctx->nt2max * 4},
        {&w.tmp_code, ((size_t)ctx->nt2max << PART_TILE_LOG) * 4},
        {&w.tmp_val, ((size_t)ctx->nt2max << PART_TILE_LOG) * 4},
        {&w.part_meta, (size_t)(256 + 257 + 1 + ctx->nt2max + 2 * 64 + 2) * 4},  // + block totals of the bucket scan
        {&w.off, ((size_t)ctx->buckets + 1) * 8},
        {&w.partial, ctx->max_slices * xyzz_bytes},
        {&w.bucket, (size_t)ctx->buckets * xyzz_bytes},
        {&w.heavy, (size_t)(2 + 3 * ctx->heavy_cap) * 4},
        {&w.heavy_part, (size_t)ctx->heavy_cap * xyzz_bytes},
        {&w.plane_part, (size_t)bucket_windows * ctx->planes * ctx->plane_blocks * xyzz_bytes},
        {&w.win_pts, ctx->table_free ? (size_t)ctx->windows * xyzz_bytes : 0},
    };
    size_t total = 0;
    for (const Part& pt : parts) total += (pt.bytes + 255) & ~(size_t)255;
    ctx->ws_bytes = total + 256;
    PLK_HIP_TRY(hipMalloc(&w.slab, total + 256));
    uint8_t* cur = (uint8_t*)w.slab;
    for (const Part& pt : parts) {
        *pt.p = pt.bytes ? cur : nullptr;
        cur += (pt.bytes + 255) & ~(size_t)255;
    }
    w.ready = true;
    return PLK_OK;
}

template <class C>
static int msm_precompute_t(plk_msm_ctx* ctx, const void* d_bases, const void* d_zero, hipStream_t stream) {
    using FP = typename C::FP;
    const size_t n = ctx->n;
    const size_t pt_bytes = (size_t)2 * FP::NL * 4;
    const size_t entries = n * ctx->windows;
    PLK_HIP_TRY(hipMalloc(&ctx->tab, (ctx->table_free ? n : entries) * pt_bytes + 16));
    ctx->ws.resize(1);
    PLK_TRY(msm_alloc_work<C>(ctx, ctx->ws[0]));
    if (n) {
        k_msm_table<C><<<(unsigned)((n + 127) / 128), 128, 0, stream>>>((const uint4*)d_bases, (const uint8_t*)d_zero, (uint4*)ctx->tab, n, ctx->c,
                                                                       ctx->table_free ? 1 : ctx->windows);
        PLK_HIP_TRY(hipGetLastError());
    }
    PLK_HIP_TRY(hipStreamSynchronize(stream));
    return PLK_OK;
}

// table-free window: windows * 2^(c-1) bucket slots must fit the 16 bits of the partition, long slices wanted
static int choose_window_table_free(size_t n) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) ++lg;
    int c = lg - 5;
    if (const char* e = getenv("PLK_MSM_WINDOW_TF")) c = atoi(e);
    if (c < 3) c = 3;
    if (c > 12) c = 12;
    return c;
}

int msm_precompute_dev_impl(int curve, size_t n, const void* d_bases, const void* d_zero, unsigned window_bits, unsigned flags, hipStream_t stream,
                            plk_msm_ctx** out_ctx) {
    if (!out_ctx) return set_error(PLK_ERR_INVALID_ARG, "null out_ctx");
    *out_ctx = nullptr;
    if (curve < 0 || curve > 2) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (n && !d_bases) return set_error(PLK_ERR_INVALID_ARG, "null bases");
    PLK_TRY(ensure_device());
    const bool table_free = (flags & PLK_MSM_TABLE_FREE) != 0;
    int c = window_bits ? (int)window_bits : (table_free ? choose_window_table_free(n ? n : 1) : choose_window(n ? n : 1));
    if (c < 2 || c > MSM_MAX_WINDOW) return set_error(PLK_ERR_INVALID_ARG, "window_bits %d outside [2, %d]", c, MSM_MAX_WINDOW);
    if (table_free) {
        const int w = (scalar_bits(curve) + 1 + c - 1) / c;
        if (((size_t)w << (c - 1)) > 65536 || w > COMBINE_THREADS / 4)
            return set_error(PLK_ERR_INVALID_ARG, "table-free mode: window_bits %d gives %d windows x %d buckets (limits: 65536 slots, %d windows)", c, w,
                             1 << (c - 1), COMBINE_THREADS / 4);
    }
    auto* ctx = new plk_msm_ctx();
    ctx->table_free = table_free;
    PLK_HIP_TRY(hipGetDevice(&ctx->device));
    ctx->curve = curve;
    ctx->n = n;
    ctx->c = c;
    ctx->windows = (scalar_bits(curve) + 1 + c - 1) / c;
    ctx->wbuckets = 1u << (c - 1);
    {
        // partition geometry from the number of bucket slots: 8 fine bits, <= 256 coarse bins
        const uint32_t want = ctx->table_free ? ctx->wbuckets * (uint32_t)ctx->windows : ctx->wbuckets;
        int bits = 0;
        while (((uint32_t)1 << bits) < want) ++bits;
        ctx->fine_bits = bits < 8 ? bits : 8;
        ctx->nbins = (int)((want + (1u << ctx->fine_bits) - 1) >> ctx->fine_bits);
        ctx->buckets = (uint32_t)ctx->nbins << ctx->fine_bits;
    }
    ctx->slice = MSM_SLICE_DEFAULT;
    if (const char* e = getenv("PLK_MSM_SLICE")) {
        int v = atoi(e);
        if (v >= 4 && v <= 4096) ctx->slice = (uint32_t)v;
    }
    ctx->planes = c;
    ctx->plane_blocks = 1;
    while (ctx->plane_blocks < MSM_MAX_PLANE_PARTS && (uint32_t)ctx->plane_blocks * 2048u < ctx->wbuckets &&
           ctx->planes * ctx->plane_blocks * 4 <= FINAL_THREADS)  // after doubling: planes * parts / 2 quads in the final block
        ctx->plane_blocks *= 2;
    if (n * (size_t)ctx->windows >= ((size_t)1 << 31)) {
        delete ctx;
        return set_error(PLK_ERR_INVALID_ARG, "n * windows = %zu entries exceeds 2^31", n * (size_t)ctx->windows);
    }
    int rc;
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: rc = msm_precompute_t<TweedledeeCurve>(ctx, d_bases, d_zero, stream); break;
        case PLK_CURVE_TWEEDLEDUM: rc = msm_precompute_t<TweedledumCurve>(ctx, d_bases, d_zero, stream); break;
        default: rc = msm_precompute_t<Bls12377Curve>(ctx, d_bases, d_zero, stream); break;
    }
    if (rc != PLK_OK) {
        delete ctx;
        return rc;
    }
    *out_ctx = ctx;
    return PLK_OK;
}

static TailSlot tail_slot(const plk_msm_ctx* ctx, const MsmWork& w, void* d_out_xy, void* d_out_zero) {
    TailSlot t;
    t.partial = (const uint4*)w.partial;
    t.slice_off = (const uint32_t*)w.off + ctx->buckets + 1;  // off[buckets + 1] is followed by slice_off[buckets + 1]
    t.bucket = (uint4*)w.bucket;
    t.heavy = (uint32_t*)w.heavy;
    t.heavy_part = (uint4*)w.heavy_part;
    t.plane_part = (uint4*)w.plane_part;
    t.win_pts = (uint4*)w.win_pts;
    t.out_xy = (uint4*)d_out_xy;
    t.out_zero = (uint8_t*)d_out_zero;
    return t;
}

// bucket sums -> bit-plane sums -> result, for the tb.count MSMs of a batch at once (their accumulations have run)
template <class C, class Mark>
static int msm_reduce_t(plk_msm_ctx* ctx, TailBatch tb, hipStream_t stream, Mark&& mark) {
    const uint32_t buckets = ctx->buckets;
    const unsigned cnt = (unsigned)tb.count;
    k_msm_bucket_sum<C><<<dim3((buckets * BUCKET_LANES + 255) / 256, cnt), 256, 0, stream>>>(tb, buckets);
    // hot buckets of a skewed scalar distribution (none for uniform scalars: the three launches then exit at once)
    k_msm_heavy_list<<<dim3((buckets + 255) / 256, cnt), 256, 0, stream>>>(tb, buckets, ctx->heavy_cap);
    k_msm_heavy_chunks<C><<<dim3(256, cnt), 256, 0, stream>>>(tb, ctx->heavy_cap);
    k_msm_heavy_final<C><<<dim3(64, cnt), 256, 0, stream>>>(tb, ctx->heavy_cap);
    PLK_HIP_TRY(hipGetLastError());
    mark();
    const int bucket_windows = ctx->table_free ? ctx->windows : 1;
    dim3 pg(ctx->plane_blocks, ctx->planes, bucket_windows * cnt);
    k_msm_planes<C><<<pg, PLANE_THREADS, 0, stream>>>(tb, bucket_windows, ctx->wbuckets);
    PLK_HIP_TRY(hipGetLastError());
    mark();
    k_msm_final<C><<<bucket_windows * cnt, FINAL_THREADS, 0, stream>>>(tb, bucket_windows, ctx->plane_blocks, ctx->planes, ctx->c);
    if (bucket_windows > 1) k_msm_combine<C><<<cnt, COMBINE_THREADS, 0, stream>>>(tb, ctx->windows);
    PLK_HIP_TRY(hipGetLastError());
    mark();
    return PLK_OK;
}

// phases: 1 = digits + bucket ordering, 2 = accumulation, 4 = reduction; 7 = the whole MSM on one stream
constexpr int PH_ORDER = 1, PH_ACC = 2, PH_REDUCE = 4, PH_ALL = 7;
template <class C>
static int msm_execute_t(plk_msm_ctx* ctx, MsmWork& w, const void* d_scalars, void* d_out_xy, void* d_out_zero, hipStream_t stream,
                         int phases = PH_ALL) {
    const size_t n = ctx->n;
    const size_t entries = n * ctx->windows;
    const uint32_t buckets = ctx->buckets;
    uint32_t* hist = (uint32_t*)w.hist;
    uint32_t* off = (uint32_t*)w.off;
    uint32_t* slice_off = off + buckets + 1;
    uint32_t* bin_total = (uint32_t*)w.part_meta;
    uint32_t* bin_base_pad = bin_total + 256;
    uint32_t* meta = bin_base_pad + 257;
    uint32_t* tile2bin = meta + 1;
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
    if (n) {
        k_msm_digits<C><<<(unsigned)((n + 255) / 256), 256, 0, stream>>>((const uint4*)d_scalars, (uint32_t*)w.codes, n, ctx->c, ctx->windows,
                                                                         ctx->table_free ? ctx->wbuckets : 0u);
        PLK_HIP_TRY(hipGetLastError());
    }
    mark();
    // partition level 1 (coarse bins)
    k_part1_count<<<ctx->nt1, PART_THREADS, 0, stream>>>((const uint32_t*)w.codes, entries, (uint32_t*)w.cnt1, ctx->nt1, ctx->fine_bits, ctx->nbins);
    k_part_rowscan<<<ctx->nbins, 256, 0, stream>>>((uint32_t*)w.cnt1, ctx->nt1, bin_total);
    k_part_bases<<<1, 256, 0, stream>>>(bin_total, ctx->nbins, bin_base_pad, tile2bin, meta);
    k_part1_scatter<<<ctx->nt1, PART_THREADS, 0, stream>>>((const uint32_t*)w.codes, entries, (const uint32_t*)w.cnt1, ctx->nt1, bin_base_pad,
                                                           ctx->fine_bits, ctx->nbins, (uint32_t*)w.tmp_code, (uint32_t*)w.tmp_val);
    // partition level 2 (buckets inside each coarse bin) + bucket offsets / slice offsets
    k_part2_count<<<ctx->nt2max, PART_THREADS, 0, stream>>>((const uint32_t*)w.tmp_code, bin_total, bin_base_pad, tile2bin, meta, (uint32_t*)w.cnt2,
                                                            ctx->nt2max, ctx->fine_bits);
    k_part2_scan<<<ctx->nbins, 256, 0, stream>>>((uint32_t*)w.cnt2, ctx->nt2max, bin_base_pad, ctx->fine_bits, hist);
    PLK_HIP_TRY(hipGetLastError());
    mark();
    {
        const unsigned sb = (buckets + 1023) / 1024;
        k_msm_scan_local<<<sb, 1024, 0, stream>>>(hist, off, slice_off, meta + 1 + ctx->nt2max, buckets, ctx->slice);
        k_msm_scan_add<<<sb, 1024, 0, stream>>>(off, slice_off, meta + 1 + ctx->nt2max, buckets);
    }
    k_part2_scatter<<<ctx->nt2max, PART_THREADS, 0, stream>>>((const uint32_t*)w.tmp_code, (const uint32_t*)w.tmp_val, bin_total, bin_base_pad, tile2bin,
                                                              meta, (const uint32_t*)w.cnt2, ctx->nt2max, ctx->fine_bits, off, (uint32_t*)w.sorted);
    PLK_HIP_TRY(hipGetLastError());
    mark();
    } else {
        stage += 3;
    }
    if (phases & PH_ACC) {
    // the slice count is only known on the device: launch for the upper bound, lanes past it exit
    k_msm_accumulate<C><<<(unsigned)((ctx->max_slices + 127) / 128), 128, 0, stream>>>((const uint4*)ctx->tab, (const uint32_t*)w.sorted, off, slice_off,
                                                                                       (uint4*)w.partial, buckets, ctx->slice, ctx->table_free ? ctx->c - 1 : 31,
                                                                                       ctx->table_free ? (uint32_t)n : 0u);
    PLK_HIP_TRY(hipGetLastError());
    }
    mark();
    if (!(phases & PH_REDUCE)) return PLK_OK;
    TailBatch tb;
    tb.count = 1;
    tb.s[0] = tail_slot(ctx, w, d_out_xy, d_out_zero);
    PLK_TRY(msm_reduce_t<C>(ctx, tb, stream, mark));
    if (!ev.empty()) ctx->prof_sets.push_back(ev);
    return PLK_OK;
}

int msm_execute_dev_impl(plk_msm_ctx* ctx, unsigned batch, const void* d_scalars, size_t n_scalars, void* d_out_xy, void* d_out_zero, hipStream_t stream) {
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    if (n_scalars != ctx->n)
        return set_error(PLK_ERR_SIZE_MISMATCH, "scalars.len() = %zu but the precomputation holds %zu generators (curve_msm.rs:67)", n_scalars, ctx->n);
    if (batch == 0) return PLK_OK;
    if ((ctx->n && !d_scalars) || !d_out_xy || !d_out_zero) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t L = (size_t)curve_limbs(ctx->curve);
    auto run_one = [&](unsigned b, MsmWork& w, hipStream_t st, int phases) -> int {
        const uint8_t* sc = (const uint8_t*)d_scalars + (size_t)b * ctx->n * 32;
        uint8_t* oxy = (uint8_t*)d_out_xy + (size_t)b * 2 * L * 8;
        uint8_t* oz = (uint8_t*)d_out_zero + b;
        switch (ctx->curve) {
            case PLK_CURVE_TWEEDLEDEE: return msm_execute_t<TweedledeeCurve>(ctx, w, sc, oxy, oz, st, phases);
            case PLK_CURVE_TWEEDLEDUM: return msm_execute_t<TweedledumCurve>(ctx, w, sc, oxy, oz, st, phases);
            default: return msm_execute_t<Bls12377Curve>(ctx, w, sc, oxy, oz, st, phases);
        }
    };
    static const bool no_batching = getenv("PLK_MSM_NO_OVERLAP") != nullptr;  // every MSM of a batch start to end, one by one
    if (batch == 1 || ctx->profiling || no_batching) {
        for (unsigned b = 0; b < batch; ++b) PLK_TRY(run_one(b, ctx->ws[0], stream, PH_ALL));
        return PLK_OK;
    }
    // Several scalar vectors against the same generators (commit_polynomials, plonk_util.rs:215-231), in groups of up
    // to TAIL_MAX: every MSM of a group has its own workspace, ordering and accumulation run one MSM after the other,
    // and the latency-bound reduction runs ONCE for the whole group - it is a chain of point operations on few points,
    // so nine of them cost little more than one (1.2 ms against 9 x 0.44 ms at 2^20).  Measured alternatives: running
    // consecutive MSMs on two streams so that the ordering of one overlaps the accumulation of its neighbour gains
    // nothing on top of this (the accumulation fills every SIMD's register file, 3 waves x 168 VGPRs); giving the other
    // phases their own CUs (CU masks) or a more urgent stream is slower.
    unsigned group = batch < (unsigned)TAIL_MAX ? batch : (unsigned)TAIL_MAX;
    const size_t budget = (size_t)16 << 30;  // bytes of workspace a batch may hold
    while (group > 2 && ctx->ws_bytes * group > budget) --group;
    while (ctx->ws.size() < group) {
        ctx->ws.emplace_back();
        int rc;
        switch (ctx->curve) {
            case PLK_CURVE_TWEEDLEDEE: rc = msm_alloc_work<TweedledeeCurve>(ctx, ctx->ws.back()); break;
            case PLK_CURVE_TWEEDLEDUM: rc = msm_alloc_work<TweedledumCurve>(ctx, ctx->ws.back()); break;
            default: rc = msm_alloc_work<Bls12377Curve>(ctx, ctx->ws.back()); break;
        }
        if (rc != PLK_OK) {
            ctx->ws.pop_back();
            (void)hipGetLastError();
            group = (unsigned)ctx->ws.size();  // make do with what fits (at least the workspace of the precomputation)
            break;
        }
    }
    for (unsigned g0 = 0; g0 < batch; g0 += group) {
        const unsigned cnt = batch - g0 < group ? batch - g0 : group;
        TailBatch tb;
        tb.count = (int)cnt;
        for (unsigned k = 0; k < cnt; ++k) {
            const unsigned b = g0 + k;
            PLK_TRY(run_one(b, ctx->ws[k], stream, PH_ORDER | PH_ACC));
            tb.s[k] = tail_slot(ctx, ctx->ws[k], (uint8_t*)d_out_xy + (size_t)b * 2 * L * 8, (uint8_t*)d_out_zero + b);
        }
        int rc;
        auto nomark = [] {};
        switch (ctx->curve) {
            case PLK_CURVE_TWEEDLEDEE: rc = msm_reduce_t<TweedledeeCurve>(ctx, tb, stream, nomark); break;
            case PLK_CURVE_TWEEDLEDUM: rc = msm_reduce_t<TweedledumCurve>(ctx, tb, stream, nomark); break;
            default: rc = msm_reduce_t<Bls12377Curve>(ctx, tb, stream, nomark); break;
        }
        PLK_TRY(rc);
    }
    return PLK_OK;
}

int msm_set_profiling_impl(plk_msm_ctx* ctx, int enable) {
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->profiling = enable != 0;
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
void msm_ctx_delete(plk_msm_ctx* ctx) { delete ctx; }

// msm_precompute with the reference's output (curve_msm.rs:27-52): powers_per_generator[i][j] = [2^(w j)] G_i, j < ceil(BITS / w)
int msm_table_digits(int curve, unsigned w) { return w ? (scalar_bits(curve) + (int)w - 1) / (int)w : -1; }

template <class C>
static int msm_reference_table_t(size_t n, const void* d_bases, const void* d_zero, int w, int digits, void* d_out_xy, void* d_out_zero,
                                 hipStream_t stream) {
    using FP = typename C::FP;
    const size_t pt_bytes = (size_t)2 * FP::NL * 4;
    void* tab = scratch_acquire(n * digits * pt_bytes + 16, stream);
    if (!tab) return PLK_ERR_OOM;
    k_msm_table<C><<<(unsigned)((n + 127) / 128), 128, 0, stream>>>((const uint4*)d_bases, (const uint8_t*)d_zero, (uint4*)tab, n, w, digits);
    const size_t total = n * (size_t)digits;
    k_msm_table_export<C><<<(unsigned)((total + 255) / 256), 256, 0, stream>>>((const uint4*)tab, n, digits, (uint4*)d_out_xy, (uint8_t*)d_out_zero);
    hipError_t e = hipGetLastError();
    scratch_release(tab, stream);
    if (e != hipSuccess) return set_error(PLK_ERR_HIP, "table launch failed: %s", hipGetErrorString(e));
    return PLK_OK;
}

int msm_reference_table_dev_impl(int curve, size_t n, const void* d_bases, const void* d_zero, unsigned w, void* d_out_xy, void* d_out_zero,
                                 hipStream_t stream) {
    if (curve < 0 || curve > 2) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (w < 1 || w > 64) return set_error(PLK_ERR_INVALID_ARG, "window size %u outside [1, 64]", w);
    if (n == 0) return PLK_OK;
    if (!d_bases || !d_out_xy || !d_out_zero) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    const int digits = msm_table_digits(curve, w);
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: return msm_reference_table_t<TweedledeeCurve>(n, d_bases, d_zero, (int)w, digits, d_out_xy, d_out_zero, stream);
        case PLK_CURVE_TWEEDLEDUM: return msm_reference_table_t<TweedledumCurve>(n, d_bases, d_zero, (int)w, digits, d_out_xy, d_out_zero, stream);
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
#undef CASE
        default: return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

// counts[8]: mismatches per case of k_selftest_quad over `quads` quads on the n points d_pts
int selftest_quad_dev_impl(int curve, const void* d_pts, uint32_t n, uint32_t quads, uint32_t* counts) {
    if (!d_pts || !counts || n == 0) return set_error(PLK_ERR_INVALID_ARG, "bad argument");
    PLK_TRY(ensure_device());
    uint32_t* d_cnt = (uint32_t*)scratch_acquire(32, nullptr);
    if (!d_cnt) return PLK_ERR_OOM;
    (void)hipMemsetAsync(d_cnt, 0, 32, nullptr);
    const unsigned blocks = (quads * 4 + 255) / 256;
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: k_selftest_quad<TweedledeeCurve><<<blocks, 256>>>((const uint4*)d_pts, n, d_cnt); break;
        case PLK_CURVE_TWEEDLEDUM: k_selftest_quad<TweedledumCurve><<<blocks, 256>>>((const uint4*)d_pts, n, d_cnt); break;
        case PLK_CURVE_BLS12_377: k_selftest_quad<Bls12377Curve><<<blocks, 256>>>((const uint4*)d_pts, n, d_cnt); break;
        default: scratch_release(d_cnt, nullptr); return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    }
    hipError_t e = hipMemcpy(counts, d_cnt, 32, hipMemcpyDeviceToHost);
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
#undef CASE
        default: return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

}  // namespace plk
