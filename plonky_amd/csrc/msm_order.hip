// msm_order.hip -- the bucket ordering of the MSM: digits (to_digits, curve_msm.rs:159-180) and the two-level LDS partition that
// replaces the reference's serial digit_occurrences scatter (curve_msm.rs:117-126).  Split from msm.hip in round 5 (build time).
#include "msm_dev.cuh"
#include "glv.cuh"
#include <cstdlib>

namespace plk {

// ---------------------------------------------------------------------------------------------
// scalars -> signed window digits  (curve_msm.rs:159-180), computed where they are consumed
// ---------------------------------------------------------------------------------------------
// Digits are never stored: the two kernels of the first partition level recompute them from the scalars
// (one Montgomery -> canonical conversion per scalar and kernel, ~10 instructions per digit), which replaces
// a 4-byte write and two 4-byte reads per (scalar, window) by two extra reads of the 32-byte scalar.
// The canonical limbs are parked in LDS (limb-major: conflict-free) so that the window loop can index them.

// the scalar of a lane as loaded (two 16-byte words); requested one sub-tile ahead of its use (k_ord_count, k_ord_scatter)
struct OrdRaw {
    uint4 lo, hi;
};
PLK_DI OrdRaw ord_load_scalar(const uint4* __restrict__ scalars, size_t i, bool live) {
    OrdRaw r;
    r.lo = r.hi = make_uint4(0u, 0u, 0u, 0u);
    if (live) {
        r.lo = scalars[i * 2];
        r.hi = scalars[i * 2 + 1];
    }
    return r;
}
template <class SP> PLK_DI void ord_park_loaded(const OrdRaw& r, uint32_t* s_lim, int tid, bool raw_signed) {
    static_assert(SP::NL == 8, "scalar fields are 256-bit");
    Fe<SP> s;
    s.v[0] = r.lo.x; s.v[1] = r.lo.y; s.v[2] = r.lo.z; s.v[3] = r.lo.w;
    s.v[4] = r.hi.x; s.v[5] = r.hi.y; s.v[6] = r.hi.z; s.v[7] = r.hi.w;
    // Montgomery -> canonical in the SCALAR field (to_canonical_u64_vec, curve_msm.rs:164); half scalars are canonical already
    if (!raw_signed) s = fe_to_canonical<SP>(s);
#pragma unroll
    for (int k = 0; k < 8; ++k) s_lim[k * ORD_THREADS + tid] = s.v[k];
}
template <class SP> PLK_DI void ord_park_scalar(const uint4* __restrict__ scalars, size_t i, uint32_t* s_lim, int tid, bool raw_signed) {
    ord_park_loaded<SP>(ord_load_scalar(scalars, i, true), s_lim, tid, raw_signed);
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

// the bucket range of a sharded execution (OrdCfg::bin_lo / bin_hi): entries of other bins are dropped where the digits are formed
PLK_DI bool ord_bin_kept(uint32_t bin, const OrdCfg& cfg) { return bin - cfg.bin_lo < cfg.bin_hi - cfg.bin_lo; }

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
    // the scalar of sub-tile st + 1 is requested before sub-tile st is worked on: two workgroups per CU do not hide a load each
    auto sub_index = [&](uint32_t st) { return ((size_t)tile * cfg.sub + st) * cfg.spt + tid; };
    auto sub_live = [&](uint32_t st) { return st < cfg.sub && (uint32_t)tid < cfg.spt && sub_index(st) < n; };
    OrdRaw nxt = ord_load_scalar(scalars, sub_index(0), sub_live(0));
    for (uint32_t st = 0; st < cfg.sub; ++st) {
        const bool live = sub_live(st);
        const OrdRaw cur = nxt;
        nxt = ord_load_scalar(scalars, sub_index(st + 1), sub_live(st + 1));
        __syncthreads();
        if (live) ord_park_loaded<SP>(cur, s_lim, tid, cfg.raw_signed != 0);
        __syncthreads();
        if (live) {
            uint32_t carry = 0;
            for (int j = 0; j < cfg.windows; ++j) {
                const uint32_t code = ord_digit(s_lim, tid, j, cfg, carry);
                if (code != CODE_INVALID && ord_bin_kept(code >> (cfg.fine_bits + 1), cfg)) atomicAdd(&s_hist[code >> (cfg.fine_bits + 1)], 1u);
            }
        }
    }
    __syncthreads();
    for (int k = tid; k < cfg.nbins; k += ORD_THREADS) cnt1[(size_t)k * cfg.nt1 + tile] = s_hist[k];
}

// The signed digits exactly as the ordering kernels form them (same ord_park_scalar / ord_digit), written out: digits[i * windows + j]
// in [-2^(c-1), 2^(c-1)], 0 for an absent entry.  Debug / parity only (plk_msm_debug_digits): the device never stores digits, so
// this is what pins its recoding to the reference's to_digits vector (curve_msm.rs:186-216) through the identity
// u_j = d_j - carry_in + 2^c carry_out, carry_out = [d_j - carry_in < 0] (a zero digit with a carry in of 1 may stand for 2^c).
template <class C>
__global__ void __launch_bounds__(ORD_THREADS) k_ord_digits(const uint4* __restrict__ scalars, size_t n, OrdCfg cfg, int32_t* __restrict__ digits) {
    using SP = typename C::SP;
    __shared__ uint32_t s_lim[8 * ORD_THREADS];
    const int tid = threadIdx.x;
    const size_t i = (size_t)blockIdx.x * ORD_THREADS + tid;
    const bool live = i < n;
    if (live) ord_park_scalar<SP>(scalars, i, s_lim, tid, false);
    __syncthreads();
    if (!live) return;
    uint32_t carry = 0;
    for (int j = 0; j < cfg.windows; ++j) {
        const uint32_t code = ord_digit(s_lim, tid, j, cfg, carry);
        const int32_t mag = code == CODE_INVALID ? 0 : (int32_t)(code >> 1) + 1;
        digits[i * (size_t)cfg.windows + j] = (code != CODE_INVALID && (code & 1u)) ? -mag : mag;
    }
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
    auto sub_index = [&](uint32_t st) { return ((size_t)tile * cfg.sub + st) * cfg.spt + tid; };
    auto sub_live = [&](uint32_t st) { return st < cfg.sub && (uint32_t)tid < cfg.spt && sub_index(st) < n; };
    OrdRaw nxt = ord_load_scalar(scalars, sub_index(0), sub_live(0));  // one sub-tile ahead (k_ord_count)
    for (uint32_t st = 0; st < cfg.sub; ++st) {
        const size_t i = sub_index(st);
        const bool live = sub_live(st);
        const OrdRaw cur = nxt;
        nxt = ord_load_scalar(scalars, sub_index(st + 1), sub_live(st + 1));
        __syncthreads();  // the previous sub-tile has been written out
        for (int k = tid; k < cfg.nbins; k += ORD_THREADS) s_cnt[k] = 0;
        if (live) ord_park_loaded<SP>(cur, s_lim, tid, cfg.raw_signed != 0);
        __syncthreads();
        if (live) {
            // one atomic per entry: its return value is the entry's rank inside its bin, kept for the placement below
            uint32_t carry = 0;
            for (int j = 0; j < cfg.windows; ++j) {
                const uint32_t code = ord_digit(s_lim, tid, j, cfg, carry);
                if (code != CODE_INVALID && ord_bin_kept(code >> (cfg.fine_bits + 1), cfg))
                    s_rank[j * cfg.spt + tid] = (uint16_t)atomicAdd(&s_cnt[code >> (cfg.fine_bits + 1)], 1u);
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
                if (code != CODE_INVALID && ord_bin_kept(code >> (cfg.fine_bits + 1), cfg)) {
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
    // a thread's sixteen entries are requested together (one round trip instead of sixteen behind each other's LDS atomics)
    uint32_t code[ORD_SEG_EPT];
#pragma unroll
    for (int k = 0; k < ORD_SEG_EPT; ++k) {
        const uint32_t p = lo + tid + k * ORD_BIN_THREADS;
        code[k] = tmp[p < hi ? p : lo].x;
    }
    for (int k = tid; k < nf; k += ORD_BIN_THREADS) s_hist[k] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ORD_SEG_EPT; ++k)
        if (lo + tid + k * ORD_BIN_THREADS < hi) atomicAdd(&s_hist[(code[k] >> 1) & fmask], 1u);
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
    // this thread's sixteen entries of the segment, requested now: they arrive while the bucket offsets are worked out
    uint2 ent[ORD_SEG_EPT];
#pragma unroll
    for (int k = 0; k < ORD_SEG_EPT; ++k) {
        const uint32_t p = lo + tid + k * ORD_BIN_THREADS;
        ent[k] = tmp[p < hi ? p : lo];
    }
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
#pragma unroll
    for (int k = 0; k < ORD_SEG_EPT; ++k) {
        if (lo + tid + k * ORD_BIN_THREADS >= hi) continue;
        const uint2 e = ent[k];
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
// round 6: TILE-MAJOR level 1 (large MSMs: two-level bucket ids, >= 2^16 scalars, <= 16 windows)
// ---------------------------------------------------------------------------------------------
// Level 1 above is count -> scan -> scatter: the scalars are read and recoded twice, the [bin][tile] counts need a grid-wide scan before
// anything can be written, and what is written are runs of ~6 eight-byte records (one sub-tile's entries of one bin).  Here a tile of
// 1024 scalars is ordered by coarse bin ENTIRELY inside LDS by one workgroup and written once, contiguously, into the tile's own region
// of tmp[] - no global position to wait for - as FOUR-byte records: inside a tile the entry is (window, scalar of the tile), 14 bits,
// beside the fine bucket bits and the sign (the tile is known to whoever reads the region).  What the rest needs of a tile is one word
// per bin - where the bin's run starts in the region and how long it is - and the bins' totals (atomics: 512 per tile); one small
// workgroup turns the totals into bin offsets, the level-2 segment table and this execution's chunk length (k_ord_scan1's last block),
// and level 2 GATHERS a segment from the tiles' runs (a prefix over the tiles' counts of its bin, one search per thread, sixteen
// consecutive positions each).  Three launches + a one-workgroup one instead of five; the scalars are read once; tmp[] holds 4 bytes per
// entry, written in whole lines, read in runs of ~26 records.
constexpr int ORD2_TS = 1024;        // scalars per tile
constexpr int ORD2_THREADS = 512;    // ... two per thread
constexpr int ORD2_SPT = ORD2_TS / ORD2_THREADS;
constexpr int ORD2_MAX_WINDOWS = 16; // 4 bits of a record
constexpr uint32_t ORD2_MAX_TILES = 8192;  // level 2 keeps a prefix over the tiles in LDS
// record: bit 0 sign, bits 1..11 fine bucket bits, 12..15 window, 16..25 scalar of the tile
PLK_DI uint32_t ord2_record(uint32_t sign, uint32_t fine, int j, uint32_t loc) { return sign | (fine << 1) | ((uint32_t)j << 12) | (loc << 16); }
// digit j of the scalar parked in column `slot` of a limb-major LDS array with `stride` columns (ord_digit with the stride as a parameter)
PLK_DI uint32_t ord_digit_at(const uint32_t* s_lim, int stride, int slot, int j, const OrdCfg& cfg, uint32_t& carry) {
    const int c = cfg.c;
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    const int bp = j * c, li = bp >> 5, sh = bp & 31;
    uint64_t two = li < 8 ? s_lim[li * stride + slot] : 0u;
    if (li + 1 < 8) two |= (uint64_t)s_lim[(li + 1) * stride + slot] << 32;
    const uint32_t v = ((uint32_t)(two >> sh) & mask) + carry;
    const uint32_t neg = v > half ? 1u : 0u;
    const uint32_t mag = neg ? (1u << c) - v : v;
    carry = neg;
    if (mag == 0) return CODE_INVALID;
    const uint32_t flip = cfg.raw_signed ? s_lim[7 * stride + slot] >> 31 : 0u;
    return ((mag - 1u + (uint32_t)j * cfg.window_buckets) << 1) | (neg ^ flip);
}
// The same digits from REGISTERS: the canonical scalar is shifted down by c bits per window, so the digit is always the low bits of
// word 0 and nothing is indexed - no LDS copy of the limbs (32 KiB for a 1024-scalar tile: the difference between one and two
// workgroups per CU).  Same recoding as ord_digit (carry-based, zero digits skipped, the sign of a GLV half scalar in bit 255).
struct OrdWalk {
    uint32_t w[8];
    uint32_t carry, flip;
};
template <class SP> PLK_DI void ord_walk_start(OrdWalk& s, const Fe<SP>& canon, const OrdCfg& cfg) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s.w[k] = canon.v[k];
    s.carry = 0;
    s.flip = cfg.raw_signed ? canon.v[7] >> 31 : 0u;
}
PLK_DI uint32_t ord_walk_next(OrdWalk& s, int j, const OrdCfg& cfg) {
    const int c = cfg.c;  // 2 .. 21
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    const uint32_t v = (s.w[0] & mask) + s.carry;
#pragma unroll
    for (int k = 0; k < 7; ++k) s.w[k] = (s.w[k] >> c) | (s.w[k + 1] << (32 - c));
    s.w[7] >>= c;
    const uint32_t neg = v > half ? 1u : 0u;
    const uint32_t mag = neg ? (1u << c) - v : v;
    s.carry = neg;
    if (mag == 0) return CODE_INVALID;
    return ((mag - 1u + (uint32_t)j * cfg.window_buckets) << 1) | (neg ^ s.flip);
}
// cnt1[bin * nt + tile] = (start of the bin's run in the tile's region) << 16 | (its length)   (both <= 16384)
template <class C>
__global__ void __launch_bounds__(ORD2_THREADS) k_ord_tiles(const uint4* __restrict__ scalars, size_t n, OrdCfg cfg, uint32_t* __restrict__ cnt1,
                                                            uint32_t* __restrict__ bin_total, uint32_t* __restrict__ tmp) {
    using SP = typename C::SP;
    static_assert(SP::NL == 8, "scalar fields are 256-bit");
    __shared__ uint32_t s_cnt[ORD_MAX_BINS], s_base[ORD_MAX_BINS];
    __shared__ uint32_t s_tmp[ORD2_THREADS];
    __shared__ uint32_t s_total;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_ent[];  // ORD2_TS * windows records
    const int tid = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    const uint32_t cap = (uint32_t)ORD2_TS * (uint32_t)cfg.windows;
    static_assert(ORD2_SPT == 2, "two scalars per thread, held in named registers (an indexed array of them is promoted to LDS by the compiler: 32 KiB)");
    const size_t i0 = (size_t)tile * ORD2_TS + tid, i1 = i0 + ORD2_THREADS;
    const bool live0 = i0 < n, live1 = i1 < n;
    const OrdRaw raw0 = ord_load_scalar(scalars, i0, live0), raw1 = ord_load_scalar(scalars, i1, live1);
    for (int k = tid; k < cfg.nbins; k += ORD2_THREADS) s_cnt[k] = 0;
    // Montgomery -> canonical in the SCALAR field (to_canonical_u64_vec, curve_msm.rs:164); half scalars are canonical already
    auto canonical = [&](const OrdRaw& r) {
        Fe<SP> sc;
        sc.v[0] = r.lo.x; sc.v[1] = r.lo.y; sc.v[2] = r.lo.z; sc.v[3] = r.lo.w;
        sc.v[4] = r.hi.x; sc.v[5] = r.hi.y; sc.v[6] = r.hi.z; sc.v[7] = r.hi.w;
        if (!cfg.raw_signed) sc = fe_to_canonical<SP>(sc);
        return sc;
    };
    const Fe<SP> canon0 = canonical(raw0), canon1 = canonical(raw1);
    __syncthreads();
    // bucket slot v = [bin | fine]: the bin is the LOW coarse bits of the digit's bucket |d| - 1 and the fine part its high bits (OrdCfg::perm),
    // so that the small digits of a short top window land in every bin.  The digits are formed ONCE and kept (<= 16 per scalar).
    const int cbits = cfg.c - 1 - cfg.fine_bits;
    const uint32_t bmask = (1u << cbits) - 1u;
    uint32_t code0[ORD2_MAX_WINDOWS], code1[ORD2_MAX_WINDOWS];
    {
        OrdWalk w0, w1;
        ord_walk_start<SP>(w0, canon0, cfg);
        ord_walk_start<SP>(w1, canon1, cfg);
#pragma unroll
        for (int j = 0; j < ORD2_MAX_WINDOWS; ++j) {
            code0[j] = (j < cfg.windows && live0) ? ord_walk_next(w0, j, cfg) : CODE_INVALID;
            code1[j] = (j < cfg.windows && live1) ? ord_walk_next(w1, j, cfg) : CODE_INVALID;
            if (code0[j] != CODE_INVALID && !ord_bin_kept((code0[j] >> 1) & bmask, cfg)) code0[j] = CODE_INVALID;  // another rank's buckets
            if (code1[j] != CODE_INVALID && !ord_bin_kept((code1[j] >> 1) & bmask, cfg)) code1[j] = CODE_INVALID;
        }
    }
#pragma unroll
    for (int j = 0; j < ORD2_MAX_WINDOWS; ++j) {
        if (code0[j] != CODE_INVALID) atomicAdd(&s_cnt[(code0[j] >> 1) & bmask], 1u);
        if (code1[j] != CODE_INVALID) atomicAdd(&s_cnt[(code1[j] >> 1) & bmask], 1u);
    }
    __syncthreads();
    for (int k = tid; k < cfg.nbins; k += ORD2_THREADS) s_base[k] = s_cnt[k];
    __syncthreads();
    block_excl_scan4(s_base, cfg.nbins, s_tmp);
    if (tid == 0) s_total = s_base[cfg.nbins - 1] + s_cnt[cfg.nbins - 1];
    for (int k = tid; k < cfg.nbins; k += ORD2_THREADS) {
        const uint32_t cn = s_cnt[k];
        cnt1[(size_t)k * cfg.nt1 + tile] = (s_base[k] << 16) | cn;
        if (cn) atomicAdd(&bin_total[k], cn);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ORD2_MAX_WINDOWS; ++j) {
        if (code0[j] != CODE_INVALID) {
            const uint32_t slot = atomicAdd(&s_base[(code0[j] >> 1) & bmask], 1u);  // the bin's cursor (its start was written out above)
            if (PLK_CHK(slot < cap, CHK_TILE_STAGE)) s_ent[slot] = ord2_record(code0[j] & 1u, code0[j] >> (cbits + 1), j, (uint32_t)tid);
        }
        if (code1[j] != CODE_INVALID) {
            const uint32_t slot = atomicAdd(&s_base[(code1[j] >> 1) & bmask], 1u);
            if (PLK_CHK(slot < cap, CHK_TILE_STAGE)) s_ent[slot] = ord2_record(code1[j] & 1u, code1[j] >> (cbits + 1), j, (uint32_t)(ORD2_THREADS + tid));
        }
    }
    __syncthreads();
    const uint32_t total = s_total;
    uint32_t* __restrict__ region = tmp + (size_t)tile * cap;
    for (uint32_t i = tid; i < total; i += ORD2_THREADS)
        if (PLK_CHK((size_t)tile * cap + i < (size_t)cfg.nt1 * cap, CHK_TMP_INDEX)) region[i] = s_ent[i];
}

// the bins' totals -> bin offsets, level-2 segment table, this execution's chunk length (k_ord_scan1's last block); the totals are
// left at zero for the next execution's atomics
// (segments only for HOT bins - more than `hot` entries: every other bin is ordered by ONE workgroup of k_ord_bin_sort)
__global__ void __launch_bounds__(256) k_ord_scan_bins(uint32_t* __restrict__ bin_total, int nbins, uint32_t* __restrict__ bin_base,
                                                       uint32_t* __restrict__ seg_base, uint32_t* __restrict__ dyn_chunk, uint32_t chunk_cfg,
                                                       uint32_t lanes_cfg, uint32_t hot) {
    __shared__ uint32_t s_sum[256];
    __shared__ uint32_t s_bins[ORD_MAX_BINS], s_tot[ORD_MAX_BINS];
    for (int k = threadIdx.x; k < nbins; k += 256) {
        s_tot[k] = s_bins[k] = bin_total[k];
        bin_total[k] = 0;
    }
    __syncthreads();
    const uint32_t last_total = s_bins[nbins - 1];
    block_excl_scan4(s_bins, nbins, s_sum);
    for (int k = threadIdx.x; k < nbins; k += 256) bin_base[k] = s_bins[k];
    if (threadIdx.x == 0) {
        const uint32_t total = s_bins[nbins - 1] + last_total;
        bin_base[nbins] = total;
        uint32_t ch = lanes_cfg ? (total + lanes_cfg - 1) / lanes_cfg : chunk_cfg;
        if (ch < 8u) ch = 8u;
        if (ch > chunk_cfg) ch = chunk_cfg;
        dyn_chunk[0] = ch;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += 256) s_bins[k] = s_tot[k] > hot ? (s_tot[k] + ORD_SEG - 1) / ORD_SEG : 0u;
    __syncthreads();
    const uint32_t last_segs = s_bins[nbins - 1];
    block_excl_scan4(s_bins, nbins, s_sum);
    for (int k = threadIdx.x; k < nbins; k += 256) seg_base[k] = s_bins[k];
    if (threadIdx.x == 0) seg_base[nbins] = s_bins[nbins - 1] + last_segs;
}

// A level-2 workgroup's view of its segment: positions plo .. phi of the bin's entries, which lie in the tiles' runs one after the
// other.  s_pre[t] = entries of the bin in tiles before t (exclusive prefix of the runs' lengths; s_pre[nt] = the bin's total),
// s_run[t] = the packed word of tile t.
struct Ord2Seg {
    uint32_t bin, seg, plo, phi;
};
PLK_DI bool ord2_segment(const uint32_t* __restrict__ seg_base, const uint32_t* __restrict__ bin_base, int nbins, uint32_t blk, Ord2Seg& sg) {
    if (blk >= seg_base[nbins]) return false;
    uint32_t a = 0, b = (uint32_t)nbins;  // seg_base[a] <= blk < seg_base[b]
    while (b - a > 1) {
        const uint32_t m = (a + b) >> 1;
        if (seg_base[m] <= blk) a = m; else b = m;
    }
    sg.bin = a;
    sg.seg = blk - seg_base[a];
    sg.plo = sg.seg * ORD_SEG;
    sg.phi = min(bin_base[a + 1] - bin_base[a], sg.plo + ORD_SEG);
    return true;
}
// s_run / s_pre: nt + 1 words each.  Ends with a barrier.
PLK_DI void ord2_prefix(const uint32_t* __restrict__ cnt1, uint32_t nt, uint32_t bin, uint32_t* s_run, uint32_t* s_pre, uint32_t* s_tmp) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t per = (nt + nthr - 1) / nthr;
    const uint32_t lo = min(nt, (uint32_t)tid * per), hi = min(nt, lo + per);
    uint32_t sum = 0;
    for (uint32_t t = lo; t < hi; ++t) {
        const uint32_t w = cnt1[(size_t)bin * nt + t];
        s_run[t] = w;
        sum += w & 0xFFFFu;
    }
    uint32_t total = 0;
    uint32_t run = block_excl_prefix(sum, s_tmp, &total);
    for (uint32_t t = lo; t < hi; ++t) {
        s_pre[t] = run;
        run += s_run[t] & 0xFFFFu;
    }
    if (tid == 0) {
        s_pre[nt] = total;
        s_run[nt] = 0;
    }
    __syncthreads();
}
// the records at positions plo + tid + k * ORD_BIN_THREADS (below phi) and the tile each comes from.  Consecutive LANES take consecutive
// positions - a wave's load is 256 contiguous bytes but for the 2-3 run boundaries in it (sixteen consecutive positions per lane, one
// search and a walk, made every lane's load a cache line of its own: 65 + 143 us for the two level-2 kernels against 20 + 68) - and
// every position is found by its own search in the prefix (ten LDS reads, sixteen independent chains per lane).
template <int THREADS = ORD_BIN_THREADS, int EPT = ORD_SEG_EPT>
PLK_DI void ord2_gather(const uint32_t* __restrict__ tmp, const uint32_t* s_run, const uint32_t* s_pre, uint32_t nt, uint32_t cap, uint32_t plo, uint32_t phi,
                        uint32_t (&rec)[EPT], uint32_t (&til)[EPT]) {
    // largest t with s_pre[t] <= p (tiles without entries of the bin share their prefix with the next one: the largest is the one that holds
    // p), for the sixteen positions of the lane AT ONCE: a fixed number of halving steps, every step sixteen independent LDS reads (sixteen
    // searches one after the other - loops of their own - were ten dependent round trips EACH: 5 us per segment and pass)
    uint32_t t[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) t[k] = 0;
    uint32_t top = 1;
    while (top * 2 <= nt) top *= 2;  // the largest power of two <= nt (nt >= 1)
    for (uint32_t step = top; step; step >>= 1) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const uint32_t p = plo + threadIdx.x + (uint32_t)k * THREADS;
            const uint32_t cand = t[k] + step;
            if (cand < nt && s_pre[cand] <= p) t[k] = cand;
        }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const uint32_t p = plo + threadIdx.x + (uint32_t)k * THREADS;
        til[k] = t[k];
        rec[k] = 0;
        if (p < phi) rec[k] = tmp[(size_t)t[k] * cap + (s_run[t[k]] >> 16) + (p - s_pre[t[k]])];
    }
}
// level 2, step 1: cnt2[segment][fine]
__global__ void __launch_bounds__(ORD_BIN_THREADS) k_ord_bin_count2(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ cnt1, uint32_t nt, uint32_t cap,
                                                                    const uint32_t* __restrict__ bin_base, const uint32_t* __restrict__ seg_base, int fine_bits,
                                                                    int nbins, uint32_t* __restrict__ cnt2) {
    __shared__ uint32_t s_hist[1 << ORD_MAX_FINE];
    __shared__ uint32_t s_tmp[ORD_BIN_THREADS];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];  // s_run[nt + 1] | s_pre[nt + 1]
    uint32_t* s_run = s_dyn;
    uint32_t* s_pre = s_dyn + (nt + 1);
    const int tid = threadIdx.x;
    const int nf = 1 << fine_bits;
    const uint32_t fmask = (uint32_t)nf - 1u;
    // a fixed grid that walks the segments: there are none at all unless some bin is hot, and then the grid must not cost a launch of
    // thousands of workgroups that only look at the count and leave
    for (uint32_t blk = blockIdx.x;; blk += gridDim.x) {
        Ord2Seg sg;
        if (!ord2_segment(seg_base, bin_base, nbins, blk, sg)) return;
        for (int k = tid; k < nf; k += ORD_BIN_THREADS) s_hist[k] = 0;
        ord2_prefix(cnt1, nt, sg.bin, s_run, s_pre, s_tmp);
        uint32_t rec[ORD_SEG_EPT], til[ORD_SEG_EPT];
        ord2_gather(tmp, s_run, s_pre, nt, cap, sg.plo, sg.phi, rec, til);
#pragma unroll
        for (int k = 0; k < ORD_SEG_EPT; ++k)
            if (sg.plo + tid + k * ORD_BIN_THREADS < sg.phi) atomicAdd(&s_hist[(rec[k] >> 1) & fmask], 1u);
        __syncthreads();
        for (int k = tid; k < nf; k += ORD_BIN_THREADS) cnt2[((size_t)blk << fine_bits) + k] = s_hist[k];
        __syncthreads();
    }
}
// level 2, step 2: k_ord_bin_scatter over gathered records
__global__ void __launch_bounds__(ORD_BIN_THREADS) k_ord_bin_scatter2(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ cnt1, uint32_t nt, uint32_t cap,
                                                                      const uint32_t* __restrict__ bin_base, const uint32_t* __restrict__ seg_base, int fine_bits,
                                                                      int nbins, uint32_t buckets, const uint32_t* __restrict__ cnt2, uint32_t* __restrict__ off,
                                                                      uint32_t* __restrict__ sorted, uint32_t entries_cap, uint32_t ent_stride, uint32_t ent_first) {
    __shared__ uint32_t s_tmp[ORD_BIN_THREADS];
    __shared__ uint32_t s_out[ORD_SEG];
    __shared__ uint16_t s_fine[ORD_SEG];
    // the three fine-bit tables are sized for THIS ordering's 2^fine_bits buckets per bin, not for ORD_MAX_FINE: with the tiles' prefix beside
    // them a workgroup holds 70 KiB at c = 20 (74 + 8 with the static tables of k_ord_bin_scatter: ONE workgroup per CU, and a level-2
    // workgroup is latency - 143-185 us against 68)
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];  // s_glob[nf] | s_loc[nf] | s_cur[nf] | s_run[nt + 1] | s_pre[nt + 1]
    uint32_t* s_glob = s_dyn;
    uint32_t* s_loc = s_dyn + ((size_t)1 << fine_bits);
    uint32_t* s_cur = s_dyn + ((size_t)2 << fine_bits);
    uint32_t* s_run = s_dyn + ((size_t)3 << fine_bits);
    uint32_t* s_pre = s_run + (nt + 1);
    const int tid = threadIdx.x;
    for (uint32_t blk = blockIdx.x;; blk += gridDim.x) {  // a fixed grid that walks the segments (k_ord_bin_count2)
    Ord2Seg sg;
    if (!ord2_segment(seg_base, bin_base, nbins, blk, sg)) return;  // (bins without entries get their offsets from k_ord_bin_sort)
    const int nf = 1 << fine_bits;
    const uint32_t fmask = (uint32_t)nf - 1u;
    const uint32_t bin = sg.bin, seg = sg.seg;
    ord2_prefix(cnt1, nt, bin, s_run, s_pre, s_tmp);
    uint32_t rec[ORD_SEG_EPT], til[ORD_SEG_EPT];
    ord2_gather(tmp, s_run, s_pre, nt, cap, sg.plo, sg.phi, rec, til);  // in flight while the bucket offsets are worked out
    const uint32_t s0 = seg_base[bin], s1 = seg_base[bin + 1];
    for (int k = tid; k < nf; k += ORD_BIN_THREADS) {
        uint32_t tot = 0, before = 0, own = 0;
        for (uint32_t sgi = s0; sgi < s1; ++sgi) {
            const uint32_t v = cnt2[((size_t)sgi << fine_bits) + k];
            if (sgi - s0 < seg) before += v;
            if (sgi - s0 == seg) own = v;
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
#pragma unroll
    for (int k = 0; k < ORD_SEG_EPT; ++k) {
        if (sg.plo + tid + k * ORD_BIN_THREADS >= sg.phi) continue;
        const uint32_t r = rec[k];
        const uint32_t f = (r >> 1) & fmask;
        const uint32_t idx = atomicAdd(&s_cur[f], 1u);
        if (PLK_CHK(idx < ORD_SEG, CHK_SEG_STAGE)) {
            // entry id = window * ent_stride + ent_first + scalar index (the table index; k_ord_scatter)
            const uint32_t entry = ((r >> 12) & 15u) * ent_stride + ent_first + til[k] * (uint32_t)ORD2_TS + (r >> 16);
            s_out[idx] = (entry << 1) | (r & 1u);
            s_fine[idx] = (uint16_t)f;
        }
    }
    __syncthreads();
    const uint32_t count = sg.phi - sg.plo;
    for (uint32_t i = tid; i < count; i += ORD_BIN_THREADS) {
        const uint32_t f = s_fine[i];
        const uint32_t at = s_glob[f] + (i - s_loc[f]);
        if (PLK_CHK(at < entries_cap, CHK_SORTED_INDEX)) sorted[at] = s_out[i];
    }
    __syncthreads();  // LDS is reused by the next segment
    }
}

// level 2 for a bin that is not hot: ONE workgroup orders the WHOLE bin inside LDS (round 6).  The segmented pair above stages 8192
// entries at a time, so an entry leaves in a run of ~8 (a bucket's share of a segment: 32 bytes), sets up the tiles' prefix twice per
// segment and keeps the fine-bit counts of every segment in global memory between its two launches.  A bin of the usual size (26.6 k
// entries at c = 20 - and with OrdCfg::perm EVERY bin is of the usual size for uniform scalars, the top window included) fits LDS as
// 4-byte entries: two passes of one workgroup over the bin's records - the fine-bit histogram, whose scan gives the bucket offsets
// and the cursors, then the placement - and the bin leaves as ONE contiguous copy, bucket after bucket.  Hot bins (more than
// ORD2_BIN_CAP entries: a skewed witness - zeros, ones, small values) keep the segmented kernels, many workgroups per bin.
constexpr int ORD2_SORT_THREADS = 1024;
constexpr int ORD2_SORT_UNROLL = 8;  // runs a half-wave has in flight
// A half-wave (32 lanes) takes whole RUNS - the bin's entries of one tile, ~26 at c = 20 - eight at a time: no search for the tile a
// position belongs to (a position-major walk - every lane its own binary search in the prefix over the tiles - was bound by the LDS
// reads of the searches: 137 us for this kernel), a run is one contiguous read, and eight of them are in flight per half-wave.
template <class F>
PLK_DI void ord2_for_runs(const uint32_t* __restrict__ tmp, const uint32_t* s_run, uint32_t nt, uint32_t cap, F&& f) {
    const uint32_t hw = threadIdx.x >> 5, hl = threadIdx.x & 31u;
    constexpr uint32_t HW = ORD2_SORT_THREADS / 32;
    for (uint32_t t0 = hw; t0 < nt; t0 += HW * ORD2_SORT_UNROLL) {
        uint32_t w[ORD2_SORT_UNROLL], r[ORD2_SORT_UNROLL];
#pragma unroll
        for (int u = 0; u < ORD2_SORT_UNROLL; ++u) {
            const uint32_t t = t0 + (uint32_t)u * HW;
            w[u] = t < nt ? s_run[t] : 0u;
            r[u] = 0;
            if (hl < (w[u] & 0xFFFFu)) r[u] = tmp[(size_t)t * cap + (w[u] >> 16) + hl];
        }
#pragma unroll
        for (int u = 0; u < ORD2_SORT_UNROLL; ++u) {
            const uint32_t t = t0 + (uint32_t)u * HW, cnt = w[u] & 0xFFFFu;
            if (hl < cnt) f(r[u], t);
            for (uint32_t i = hl + 32u; i < cnt; i += 32u) f(tmp[(size_t)t * cap + (w[u] >> 16) + i], t);  // a run longer than a half-wave (rare)
        }
    }
}
// CACHE (nt <= 1024, i.e. up to 2^20 scalars): the first 32 records of every run this lane reads in pass 1 stay in registers (4 steps x 8
// runs) and pass 2 does not read the bin again; longer runs' tails are re-read.
template <bool CACHE>
__global__ void __launch_bounds__(ORD2_SORT_THREADS) k_ord_bin_sort(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ cnt1, uint32_t nt, uint32_t cap,
                                                                    const uint32_t* __restrict__ bin_base, int fine_bits, int nbins, uint32_t buckets,
                                                                    uint32_t* __restrict__ off, uint32_t* __restrict__ sorted, uint32_t entries_cap,
                                                                    uint32_t ent_stride, uint32_t ent_first) {
    __shared__ uint32_t s_tmp[ORD2_SORT_THREADS];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];  // s_out[ORD2_BIN_CAP] | s_cur[nf] | s_run[nt]
    const int nf = 1 << fine_bits;
    uint32_t* s_out = s_dyn;
    uint32_t* s_cur = s_dyn + ORD2_BIN_CAP;
    uint32_t* s_run = s_cur + nf;
    const int tid = threadIdx.x;
    const uint32_t bin = blockIdx.x;
    const uint32_t bb = bin_base[bin], total = bin_base[bin + 1] - bb;
    if (bin == 0 && tid == 0) off[buckets] = bin_base[nbins];
    if (total == 0) {  // a bin without entries still owns bucket offsets
        for (int k = tid; k < nf; k += ORD2_SORT_THREADS) off[((size_t)bin << fine_bits) + k] = bb;
        return;
    }
    if (total > ORD2_BIN_CAP) return;  // k_ord_bin_count2 / k_ord_bin_scatter2
    const uint32_t fmask = (uint32_t)nf - 1u;
    for (int k = tid; k < nf; k += ORD2_SORT_THREADS) s_cur[k] = 0;
    for (uint32_t t = tid; t < nt; t += ORD2_SORT_THREADS) s_run[t] = cnt1[(size_t)bin * nt + t];
    __syncthreads();
    auto count = [&](uint32_t r, uint32_t) { atomicAdd(&s_cur[(r >> 1) & fmask], 1u); };
    auto place = [&](uint32_t r, uint32_t t) {
        const uint32_t idx = atomicAdd(&s_cur[(r >> 1) & fmask], 1u);
        // entry id = window * ent_stride + ent_first + scalar index (the table index; k_ord_scatter)
        const uint32_t entry = ((r >> 12) & 15u) * ent_stride + ent_first + t * (uint32_t)ORD2_TS + (r >> 16);
        if (PLK_CHK(idx < ORD2_BIN_CAP, CHK_SEG_STAGE)) s_out[idx] = (entry << 1) | (r & 1u);
    };
    constexpr int STEPS = 4;  // ORD2_SORT_UNROLL runs of each of the 32 half-waves per step: 1024 tiles in four steps
    constexpr uint32_t HW = ORD2_SORT_THREADS / 32;
    const uint32_t hw = (uint32_t)tid >> 5, hl = (uint32_t)tid & 31u;
    uint32_t rc[STEPS * ORD2_SORT_UNROLL];
    // pass 1: the bin's entries per bucket
    if constexpr (CACHE) {
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
#pragma unroll
            for (int u = 0; u < ORD2_SORT_UNROLL; ++u) {
                const uint32_t t = hw + (uint32_t)(st * ORD2_SORT_UNROLL + u) * HW;
                const uint32_t w = t < nt ? s_run[t] : 0u;
                rc[st * ORD2_SORT_UNROLL + u] = 0;
                if (hl < (w & 0xFFFFu)) rc[st * ORD2_SORT_UNROLL + u] = tmp[(size_t)t * cap + (w >> 16) + hl];
            }
#pragma unroll
            for (int u = 0; u < ORD2_SORT_UNROLL; ++u) {
                const uint32_t t = hw + (uint32_t)(st * ORD2_SORT_UNROLL + u) * HW;
                const uint32_t w = t < nt ? s_run[t] : 0u, cnt = w & 0xFFFFu;
                if (hl < cnt) count(rc[st * ORD2_SORT_UNROLL + u], t);
                for (uint32_t i = hl + 32u; i < cnt; i += 32u) count(tmp[(size_t)t * cap + (w >> 16) + i], t);
            }
        }
    } else {
        ord2_for_runs(tmp, s_run, nt, cap, count);
    }
    __syncthreads();
    block_excl_scan4(s_cur, nf, s_tmp);
    for (int k = tid; k < nf; k += ORD2_SORT_THREADS) off[((size_t)bin << fine_bits) + k] = bb + s_cur[k];
    __syncthreads();
    // pass 2: every entry to its bucket's cursor
    if constexpr (CACHE) {
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
#pragma unroll
            for (int u = 0; u < ORD2_SORT_UNROLL; ++u) {
                const uint32_t t = hw + (uint32_t)(st * ORD2_SORT_UNROLL + u) * HW;
                const uint32_t w = t < nt ? s_run[t] : 0u, cnt = w & 0xFFFFu;
                if (hl < cnt) place(rc[st * ORD2_SORT_UNROLL + u], t);
                for (uint32_t i = hl + 32u; i < cnt; i += 32u) place(tmp[(size_t)t * cap + (w >> 16) + i], t);
            }
    } else {
        ord2_for_runs(tmp, s_run, nt, cap, place);
    }
    __syncthreads();
    for (uint32_t i = tid; i < total; i += ORD2_SORT_THREADS)
        if (PLK_CHK(bb + i < entries_cap, CHK_SORTED_INDEX)) sorted[bb + i] = s_out[i];
}

// ---- host side: what msm.hip sees of this file ----------------------------------------------------------------------------
template <class C> int msm_launch_glv_split(const void* d_scalars, size_t n, void* halves, hipStream_t stream) {
    k_glv_split<C><<<(unsigned)((n + 255) / 256), 256, 0, stream>>>((const uint4*)d_scalars, n, (uint4*)halves);
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}
// stage 0: level-1 counts + scan (and this execution's chunk length); 1: level-1 scatter; 2: level 2 (skipped by one-level orderings)
// the tile-major level 1 (round 6): where msm_configure set OrdCfg::perm
static bool order_tiles2(const OrdCfg& o) { return o.perm != 0; }
template <class C> int msm_launch_order_stage(int stage, const OrdCfg& o, const OrdBuffers& b, hipStream_t stream) {
    const bool one_level = o.fine_bits == 0;
    if (order_tiles2(o)) {
        const uint32_t cap = (uint32_t)ORD2_TS * (uint32_t)o.windows;
        if (stage == 0) {
            const size_t lds = (size_t)cap * 4;
            // more than 64 KiB of LDS in all (42 KiB static + the tile's records): asked for per kernel, as the transform's launches do
            (void)hipFuncSetAttribute((const void*)k_ord_tiles<C>, hipFuncAttributeMaxDynamicSharedMemorySize, ORD2_TS * ORD2_MAX_WINDOWS * 4);
            k_ord_tiles<C><<<o.nt1, ORD2_THREADS, lds, stream>>>((const uint4*)b.scalars, b.n, o, (uint32_t*)b.cnt1, b.bin_total, (uint32_t*)b.tmp);
            k_ord_scan_bins<<<1, 256, 0, stream>>>(b.bin_total, o.nbins, b.bin_base, b.seg_base, b.done_counter + 2, b.chunk, b.lanes, ORD2_BIN_CAP);
        } else if (stage == 2) {
            // hot bins hold more than ORD2_BIN_CAP entries each: at most entries / ORD2_BIN_CAP of them, entries / ORD_SEG + that many segments
            const size_t entries = b.n * (size_t)o.windows;
            const size_t hot_bins = entries / ORD2_BIN_CAP < (size_t)o.nbins ? entries / ORD2_BIN_CAP : (size_t)o.nbins;
            const unsigned segs = (unsigned)(entries / ORD_SEG + hot_bins + 1);  // <= the entries / ORD_SEG + nbins + 1 rows cnt2[] holds
            const size_t lds = (size_t)2 * (o.nt1 + 1) * 4, lds_s = lds + ((size_t)3 << o.fine_bits) * 4;
            const size_t lds_b = ((size_t)ORD2_BIN_CAP + ((size_t)1 << o.fine_bits) + o.nt1) * 4;
            // (the records of pass 1 stay in registers when the half-waves cover the tiles in four steps: up to 1024 tiles)
            static const bool no_cache = getenv("PLK_MSM_BINSORT_NOCACHE") != nullptr;
            if (o.nt1 <= 1024u && !no_cache) {
                (void)hipFuncSetAttribute((const void*)k_ord_bin_sort<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
                k_ord_bin_sort<true><<<o.nbins, ORD2_SORT_THREADS, lds_b, stream>>>((const uint32_t*)b.tmp, (const uint32_t*)b.cnt1, o.nt1, cap, b.bin_base, o.fine_bits,
                                                                                   o.nbins, b.buckets, (uint32_t*)b.off, (uint32_t*)b.sorted, o.entries_cap, o.ent_stride,
                                                                                   o.ent_first);
            } else {
                (void)hipFuncSetAttribute((const void*)k_ord_bin_sort<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
                k_ord_bin_sort<false><<<o.nbins, ORD2_SORT_THREADS, lds_b, stream>>>((const uint32_t*)b.tmp, (const uint32_t*)b.cnt1, o.nt1, cap, b.bin_base, o.fine_bits,
                                                                                    o.nbins, b.buckets, (uint32_t*)b.off, (uint32_t*)b.sorted, o.entries_cap, o.ent_stride,
                                                                                    o.ent_first);
            }
            const unsigned hot_grid = segs < 512u ? segs : 512u;
            k_ord_bin_count2<<<hot_grid, ORD_BIN_THREADS, lds, stream>>>((const uint32_t*)b.tmp, (const uint32_t*)b.cnt1, o.nt1, cap, b.bin_base, b.seg_base, o.fine_bits,
                                                                   o.nbins, (uint32_t*)b.cnt2);
            k_ord_bin_scatter2<<<hot_grid, ORD_BIN_THREADS, lds_s, stream>>>((const uint32_t*)b.tmp, (const uint32_t*)b.cnt1, o.nt1, cap, b.bin_base, b.seg_base,
                                                                               o.fine_bits, o.nbins, b.buckets, (const uint32_t*)b.cnt2, (uint32_t*)b.off,
                                                                               (uint32_t*)b.sorted, o.entries_cap, o.ent_stride, o.ent_first);
        }
        PLK_HIP_TRY(hipGetLastError());
        return PLK_OK;
    }
    if (stage == 0) {
        k_ord_count<C><<<o.nt1, ORD_THREADS, 0, stream>>>((const uint4*)b.scalars, b.n, o, (uint32_t*)b.cnt1);
        k_ord_scan1<<<o.nbins, 256, 0, stream>>>((uint32_t*)b.cnt1, o.nt1, o.nbins, b.bin_total, b.bin_base, b.seg_base, b.done_counter, b.done_counter + 2,
                                                 b.chunk, b.lanes, one_level ? (uint32_t*)b.off : nullptr);
    } else if (stage == 1) {
        k_ord_scatter<C><<<o.nt1, ORD_THREADS, 0, stream>>>((const uint4*)b.scalars, b.n, o, (const uint32_t*)b.cnt1, b.bin_base, (uint2*)b.tmp,
                                                            one_level ? (uint32_t*)b.sorted : nullptr);
    } else if (!one_level) {
        // the number of segments is only known on the device: launch for the upper bound (+ nbins blocks that write the
        // offsets of the empty bins), blocks past the end exit
        const unsigned segs = (unsigned)(b.n * (size_t)o.windows / ORD_SEG + o.nbins + 1);
        k_ord_bin_count<<<segs, ORD_BIN_THREADS, 0, stream>>>((const uint2*)b.tmp, b.bin_base, b.seg_base, o.fine_bits, o.nbins, (uint32_t*)b.cnt2);
        k_ord_bin_scatter<<<segs + o.nbins, ORD_BIN_THREADS, 0, stream>>>((const uint2*)b.tmp, b.bin_base, b.seg_base, o.fine_bits, o.nbins, b.buckets,
                                                                          (const uint32_t*)b.cnt2, (uint32_t*)b.off, (uint32_t*)b.sorted, o.entries_cap);
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}
template <class C> int msm_launch_digits(const void* d_scalars, size_t n, const OrdCfg& o, void* d_digits, hipStream_t stream) {
    k_ord_digits<C><<<(unsigned)((n + ORD_THREADS - 1) / ORD_THREADS), ORD_THREADS, 0, stream>>>((const uint4*)d_scalars, n, o, (int32_t*)d_digits);
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}
#define PLK_ORD_INSTANTIATE(C)                                                                     \
    template int msm_launch_glv_split<C>(const void*, size_t, void*, hipStream_t);                \
    template int msm_launch_order_stage<C>(int, const OrdCfg&, const OrdBuffers&, hipStream_t);   \
    template int msm_launch_digits<C>(const void*, size_t, const OrdCfg&, void*, hipStream_t);
PLK_ORD_INSTANTIATE(TweedledeeCurve)
PLK_ORD_INSTANTIATE(TweedledumCurve)
PLK_ORD_INSTANTIATE(Bls12377Curve)
PLK_ORD_INSTANTIATE(PallasCurve)
PLK_ORD_INSTANTIATE(VestaCurve)
#undef PLK_ORD_INSTANTIATE

PLK_CHK_READER(msm_order_checked_failures)

}  // namespace plk
