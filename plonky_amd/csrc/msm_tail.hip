// msm_tail.hip -- the reduction of the MSM, sum_d d * bucket_d (replaces the serial Yao tail of curve_msm.rs:149-154): heavy buckets,
// assembly of the pieces, row / column sums of the bucket grid, bit-plane sums on quads, the final doublings and the normalisation.
// Split from msm.hip in round 5 (build time).
#include "msm_dev.cuh"
#include "ecz_coop.cuh"
#include <cstdlib>

namespace plk {

// the result of an MSM as the slot asks for it: the unique affine point (one inversion on one lane), or the reference's un-normalised
// ProjectivePoint (what msm_execute_parallel returns, curve_msm.rs:102-157)
template <class FP> PLK_DI void emit_result(const XyzzZ<FP>& acc, const TailSlot& sl) {
    if (sl.projective) emit_projective<FP>(acc, sl.out_xy, sl.out_zero);
    else emit_affine<FP, true>(acc, sl.out_xy, sl.out_zero);
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
    if (blockIdx.x == 0 && threadIdx.x == 3) *sl.live_count = 0;        // the list of live lanes k_msm_accumulate keeps for k_msm_heads: unused here
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

// List-driven assembly (round 6; two-level tail with one lane per bucket in k_msm_assemble's terms, i.e. large MSMs).
// k_msm_assemble visits every BUCKET (2^19 at c = 20: three dependent loads each, and a whole wave pays for the one lane in it that has
// an addition to make) to find the few whose head pieces are still pieces of their own - the first lane of every accumulation block,
// lanes that lie inside one bucket: ~1.5 k of 2^19 for uniform scalars - and k_msm_heavy_list visits them once more for the heavy ones.
// k_msm_accumulate now LISTS those lanes (live_list / live_count) and notes the bucket of every lane's head piece (head_bucket), so
// the work is found directly: one QUAD per listed lane; the first live head lane of a bucket adds that bucket's live head pieces to its
// start piece (quad additions, ecz_coop.cuh), the first head lane of a heavy bucket lists it for k_msm_heavy_chunks / _final as
// k_msm_heavy_list did (every head lane of a heavy bucket but the last lies inside the bucket, so its first head lane is live).
// Buckets without a live head piece are not touched; empty buckets are left to k_msm_gsum, which reads their emptiness from off[].
template <class C>
__global__ void __launch_bounds__(256) k_msm_heads(TailBatch tb, uint32_t buckets, uint32_t cap) {
    using FP = typename C::FP;
    constexpr int RU = raw_u4<FP>();
    const TailSlot& sl = tb.s[blockIdx.y];
    if (blockIdx.x == 0 && threadIdx.x == 0) *sl.final_done = 0;  // k_msm_final's counter (left at zero by its last block anyway)
    const int ql = threadIdx.x & 3;
    const uint32_t quads = gridDim.x * (blockDim.x >> 2);
    const uint32_t nlive = *sl.live_count;
    const uint32_t chunk = sl.dyn_chunk[0];
    for (uint32_t item = blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2); item < nlive; item += quads) {
        const uint32_t l = sl.live_list[item];
        const uint32_t b = sl.head_bucket[l];
        if (b >= buckets) continue;  // (never: a live lane has a head piece)
        uint32_t first, ns;
        bucket_heads(sl.off, b, chunk, first, ns);
        if (l < first || l >= first + ns) continue;  // (never)
        if (ns > HEAVY_HEADS) {
            if (l == first && ql == 0) {  // k_msm_heavy_list's entry for this bucket
                uint32_t* __restrict__ heavy = sl.heavy;
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
            continue;
        }
        bool leader = true;  // the first LIVE head lane of the bucket does the bucket's work (ns <= 32: a short walk)
        for (uint32_t h = first; h < l; ++h) leader = leader && !sl.head_live[h];
        if (!leader) continue;
        XyzzZ<FP> acc = xyzzz_load_raw<FP>(sl.p_start + (size_t)b * RU);  // a bucket with a head piece is not empty: its start piece exists
        for (uint32_t h = l; h < first + ns; ++h)
            if (h == l || sl.head_live[h]) acc = xyzzz_add_q<FP>(acc, xyzzz_load_raw<FP>(sl.p_head + (size_t)h * RU), ql);
        if (ql == 0) xyzzz_store_raw<FP>(sl.p_start + (size_t)b * RU, acc);
    }
}

// Two-level weighting, step 1: partial row and column sums over groups of G = 2^g_log buckets, one lane per group.
// lanes [0, NB / G): row hi, group g: buckets (hi << L) + g G + k;   lanes [NB / G, 2 NB / G): column lo, group g:
// buckets ((g G + k) << L) + lo (adjacent lanes = adjacent columns = adjacent addresses).
template <class C>
__global__ void __launch_bounds__(128) k_msm_gsum(TailBatch tb, int L, int H, int g_log, int reset_heavy) {
    using FP = typename C::FP;
    constexpr int RU = raw_u4<FP>();
    const TailSlot& sl = tb.s[blockIdx.y];
    const uint32_t nbg = (1u << (L + H)) >> g_log;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (reset_heavy && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x < 2) sl.heavy[threadIdx.x] = 0;  // k_msm_heads' counters are free again
    if (reset_heavy && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 2) *sl.live_count = 0;        // and so is k_msm_accumulate's list
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
    // the load of element k + 1 is in flight while element k is added.  An EMPTY bucket is the identity whatever its slot holds
    // (off[b + 1] == off[b]): k_msm_assemble writes the identity there, the lane-driven assembly (k_msm_heads) leaves the slot alone.
    const uint32_t* __restrict__ off = sl.off;
    auto load_bucket = [&](uint32_t b) {
        // (an empty bucket is not read at all: a rank that holds a bucket range of a sharded vector sums mostly empty lines)
        if (off[b + 1] == off[b]) return xyzzz_identity<FP>();
        return xyzzz_load_raw<FP>(sl.p_start + (size_t)b * RU);
    };
    XyzzZ<FP> acc = xyzzz_identity<FP>();
    XyzzZ<FP> nxt = load_bucket(b0);
    for (uint32_t k = 0; k < G; ++k) {
        XyzzZ<FP> cur = nxt;
        const uint32_t b = b0 + k * bstep;
        if (k + 1 < G) nxt = load_bucket(b + bstep);
        acc = xyzzz_add<FP>(acc, cur);
    }
    xyzzz_store_raw<FP>(dst, acc);
}

// Step 1 with the first LV levels of every line's tree inside the workgroup (round 6).  k_msm_gsum leaves 2^(L-g) / 2^(H-g) partials per
// row / column (64 / 128 at c = 20) and k_msm_lsum - one wave of 16 quads per line, where every wave pays ~1100 instructions per quad
// addition however few of its quads still hold a partial - is then 8-12 quad additions deep: 79 us for an eighth of k_msm_gsum's
// additions.  Here the 128 lanes of a workgroup own groups of the SAME lines - row lanes as before (64 consecutive groups of a row per
// wave), column lanes as 16 adjacent columns x 8 consecutive groups - so after the G serial additions the partials meet in LDS and
// halve LV times (lane additions, all by the first wave: level k keeps 128 >> k lanes busy, the other wave's SIMD is free for another
// workgroup): 2^LV times fewer partials per line for (8 + LV) / 8 of the first wave's time.  Output layout = k_msm_gsum's with
// g_log + LV in place of g_log, which is what k_msm_lsum is then launched with.
template <class C, int LV>
__global__ void __launch_bounds__(128) k_msm_gsum_tree(TailBatch tb, int L, int H, int g_log) {
    using FP = typename C::FP;
    constexpr int RU = raw_u4<FP>();
    __shared__ uint4 s_a[128 * RU], s_b[64 * RU];
    const TailSlot& sl = tb.s[blockIdx.y];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.z == 0 && tid < 2) sl.heavy[tid] = 0;   // k_msm_heads' counters are free again
    if (blockIdx.x == 0 && blockIdx.z == 0 && tid == 2) *sl.live_count = 0;  // and so is k_msm_accumulate's list
    const uint32_t nbg = (1u << (L + H)) >> g_log;   // row lanes (as many column lanes)
    const uint32_t row_blocks = nbg >> 7;
    const uint32_t wbase = blockIdx.z << (L + H);    // table-free mode: blockIdx.z is the window
    uint4* part = sl.line_part + (size_t)blockIdx.z * 2 * (nbg >> LV) * RU;
    const uint32_t G = 1u << g_log;
    const bool rows = blockIdx.x < row_blocks;
    uint32_t b0, bstep;
    if (rows) {
        const uint32_t t = blockIdx.x * 128u + tid;
        const uint32_t pr = (1u << L) >> g_log;
        b0 = wbase + ((t / pr) << L) + ((t % pr) << g_log);
        bstep = 1;
    } else {
        const uint32_t cb = blockIdx.x - row_blocks, sets = (1u << L) >> 4;
        const uint32_t lo = ((cb % sets) << 4) + (tid & 15), g = ((cb / sets) << 3) + (tid >> 4);
        b0 = wbase + ((g << g_log) << L) + lo;
        bstep = 1u << L;
    }
    const uint32_t* __restrict__ off = sl.off;
    auto load_bucket = [&](uint32_t b) {
        if (off[b + 1] == off[b]) return xyzzz_identity<FP>();  // an empty bucket is the identity whatever its slot holds (k_msm_gsum)
        return xyzzz_load_raw<FP>(sl.p_start + (size_t)b * RU);
    };
    XyzzZ<FP> acc = xyzzz_identity<FP>();
    XyzzZ<FP> nxt = load_bucket(b0);
    for (uint32_t k = 0; k < G; ++k) {
        XyzzZ<FP> cur = nxt;
        if (k + 1 < G) nxt = load_bucket(b0 + (k + 1) * bstep);
        acc = xyzzz_add<FP>(acc, cur);
    }
    xyzzz_store_raw<FP>(s_a + tid * RU, acc);
    __syncthreads();
    // level k: lane i < 128 >> k adds the two partials of level k - 1 that continue each other: rows 2 i and 2 i + 1 (consecutive groups of
    // one row); columns, lane = [pair of groups | column of 16]: the same column 16 lanes apart
    uint4* src = s_a;
    uint4* dst = s_b;
#pragma unroll
    for (int lv = 1; lv <= LV; ++lv) {
        const int act = 128 >> lv;
        if (tid < act) {
            const int i0 = rows ? 2 * tid : (((tid >> 4) << 5) + (tid & 15));
            const int i1 = rows ? i0 + 1 : i0 + 16;
            const XyzzZ<FP> sum = xyzzz_add<FP>(xyzzz_load_raw<FP>(src + i0 * RU), xyzzz_load_raw<FP>(src + i1 * RU));
            if (lv < LV) {
                xyzzz_store_raw<FP>(dst + tid * RU, sum);
            } else if (rows) {
                xyzzz_store_raw<FP>(part + ((size_t)((blockIdx.x * 128u) >> LV) + tid) * RU, sum);
            } else {
                const uint32_t cb = blockIdx.x - row_blocks, sets = (1u << L) >> 4;
                const uint32_t lo = ((cb % sets) << 4) + (tid & 15), gq = ((cb / sets) << (3 - LV)) + (tid >> 4);
                const uint32_t pc = ((1u << H) >> g_log) >> LV;  // partials per column now
                xyzzz_store_raw<FP>(part + ((size_t)(nbg >> LV) + (size_t)lo * pc + gq) * RU, sum);
            }
        }
        if (lv < LV) __syncthreads();
        uint4* t2 = src;
        src = dst;
        dst = t2;
    }
}

// step 2: the lines.  Output slot s of 2 * wb (wb = 2^H): window 0 holds the column sums C_lo at index lo (weight lo + 1),
// window 1 the row sums R_hi at index hi - 1 (weight hi; R_0 has weight 0 and is dropped); the rest is the identity.
// A chain of additions on few points, i.e. latency: it runs on quads (ecz_coop.cuh), 2^qpl_log adjacent quads per line (<= 16):
// each sums its share of the line's partials, quad-wide shuffles combine.
template <class C>
// transposed (TailGeom): the grid in memory is 2^L rows of 2^H slots; its row partials (first region, 2^(H-g) per row) are the column sums
// C_lo of the weighting and its column partials (second region, 2^(L-g) per column) the row sums R_hi.
__global__ void __launch_bounds__(256) k_msm_lsum(TailBatch tb, int L, int H, int g_log, int qpl_log, int transposed) {
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
            src = transposed ? lines + (size_t)idx * cnt * RU : lines + ((size_t)nbg + (size_t)idx * cnt) * RU;
        } else if (win == 1 && idx + 1 < wb) {
            cnt = (1u << L) >> g_log;
            src = transposed ? lines + ((size_t)nbg + (size_t)(idx + 1) * cnt) * RU : lines + (size_t)(idx + 1) * cnt * RU;
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
// (256-thread workgroups - one wave per SIMD - were measured for the planes and this kernel: the same 66 / 139 us at c = 20, and the
// one-shot MSM with its 2 x 13 tail windows went from 2.8 to 4.4 ms)
// (PLK_FT: the shader-clock stamps of tuning builds, ecz.cuh)
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
    PLK_FT(0);
    XyzzZ<FP> acc = live ? xyzzz_load_packed<FP>(src) : xyzzz_identity<FP>();
    if (ipq == 2) acc = xyzzz_add_q<FP>(acc, live ? xyzzz_load_packed<FP>(src + 4 * W) : xyzzz_identity<FP>(), ql);
    acc = wave_sum_q<FP>(acc, qpp, ql);  // the quads of a plane are adjacent in one wave
    PLK_FT(1);
    if (live && sub == 0) {
        for (int k = 0; k < plane; ++k) acc = xyzzz_dbl_q<FP>(acc, ql);
        if (ql == 0) xyzzz_store_packed<FP>(s_pts + plane * 4 * W, acc);
    }
    PLK_FT(2);
    __syncthreads();
    PLK_FT(3);
    if (tid < 128) {  // the planes: 32 quads, two waves
        acc = item < planes ? xyzzz_load_packed<FP>(s_pts + item * 4 * W) : xyzzz_identity<FP>();
        acc = wave_sum_q<FP>(acc, 16, ql);
        if (tid == 64) xyzzz_store_packed<FP>(s_pts + 32 * 4 * W, acc);
    }
    PLK_FT(4);
    __syncthreads();
    PLK_FT(5);
    if (tid < 4) {
        if (planes > 16) acc = xyzzz_add_q<FP>(acc, xyzzz_load_packed<FP>(s_pts + 32 * 4 * W), ql);
        // pair_shift < 0: window `win` weighs 2^(win * window_bits).  Two-level tail: the windows come in pairs (column sums,
        // row sums) of real window win / 2, the row sums shifted by pair_shift = L more
        const int shift = pair_shift < 0 ? win * window_bits : (win >> 1) * window_bits + (win & 1) * pair_shift;
        for (int k = 0; k < shift; ++k) acc = xyzzz_dbl_q<FP>(acc, ql);
        PLK_FT(6);
        if (windows == 1) {
            if (tid == 0) emit_result<FP>(acc, tb.s[slot]);
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
                PLK_FT(7);
                if (seen == (uint32_t)windows - 1u) {
                    __threadfence();
                    for (int o = 0; o < windows; ++o)
                        if (o != win) acc = xyzzz_add_q<FP>(acc, xyzzz_load_packed_volatile<FP>(win_pts + (size_t)o * 4 * W), ql);
                    PLK_FT(8);
                    if (tid == 0) {
                        *tb.s[slot].final_done = 0;
                        emit_result<FP>(acc, tb.s[slot]);
                    }
                    PLK_FT(9);
                }
            }
        }
    }
}

// The two windows of a tabled two-level reduction (column sums, row sums) in ONE workgroup (round 6): its even waves take the column
// window, its odd waves the row window - the same steps as k_msm_final side by side - and the two results meet in LDS instead of through a
// packed point in global memory, a device-scope fence and an atomic counter (3.5 + 9.5 us of k_msm_final's 125, r05_final_kernel_trace).
// Needs planes * parts / 2 <= 64 quads per window.
template <class C>
__global__ void __launch_bounds__(FINAL_THREADS) k_msm_final_pair(TailBatch tb, int parts, int planes, int pair_shift) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    constexpr int RU = raw_u4<FP>();
    static_assert(FINAL_THREADS == 512, "two windows of 256 threads");
    __shared__ uint4 s_pts[2 * 33 * 4 * W];
    __shared__ uint4 s_x[RU];
    // the waves alternate between the windows (even: columns, odd: rows), so that the ONE wave of each window that runs the doubling chain
    // sits on a SIMD of its own (waves go to the SIMDs in turn: with the windows in the two halves of the workgroup both chains shared
    // SIMD 0 and the kernel took 135 us against k_msm_final's 121)
    const int tid = threadIdx.x, ql = tid & 3, half = (tid >> 6) & 1, ltid = ((tid >> 7) << 6) | (tid & 63), item = ltid >> 2;
    const int slot = blockIdx.x;
    const uint4* __restrict__ plane_part = tb.s[slot].plane_part;
    const int ipq = parts > 1 ? 2 : 1, qpp = parts / ipq;  // quads per plane
    const int plane = item / qpp, sub = item % qpp;
    const bool live = plane < planes;
    const uint4* src = plane_part + ((size_t)(half * planes + plane) * parts + sub * ipq) * 4 * W;
    uint4* sp = s_pts + half * 33 * 4 * W;
    XyzzZ<FP> acc = live ? xyzzz_load_packed<FP>(src) : xyzzz_identity<FP>();
    if (ipq == 2) acc = xyzzz_add_q<FP>(acc, live ? xyzzz_load_packed<FP>(src + 4 * W) : xyzzz_identity<FP>(), ql);
    acc = wave_sum_q<FP>(acc, qpp, ql);  // the quads of a plane are adjacent in one wave
    if (live && sub == 0) {
        for (int k = 0; k < plane; ++k) acc = xyzzz_dbl_q<FP>(acc, ql);
        if (ql == 0) xyzzz_store_packed<FP>(sp + plane * 4 * W, acc);
    }
    __syncthreads();
    if (ltid < 128) {  // the planes: 32 quads, two waves per window
        acc = item < planes ? xyzzz_load_packed<FP>(sp + item * 4 * W) : xyzzz_identity<FP>();
        acc = wave_sum_q<FP>(acc, 16, ql);
        if (ltid == 64) xyzzz_store_packed<FP>(sp + 32 * 4 * W, acc);
    }
    __syncthreads();
    if (ltid < 4) {
        if (planes > 16) acc = xyzzz_add_q<FP>(acc, xyzzz_load_packed<FP>(sp + 32 * 4 * W), ql);
        if (half == 1) {  // the row sums weigh 2^L more
            for (int k = 0; k < pair_shift; ++k) acc = xyzzz_dbl_q<FP>(acc, ql);
            if (ql == 0) xyzzz_store_raw<FP>(s_x, acc);
        }
    }
    __syncthreads();
    if (tid < 4) {  // wave 0: the column window's quad
        acc = xyzzz_add_q<FP>(acc, xyzzz_load_raw<FP>(s_x), ql);
        if (tid == 0) emit_result<FP>(acc, tb.s[slot]);
    }
}

// table-free mode: the sum of the windows (<= 128 points, already doubled into place), normalised
template <class C>
__global__ void __launch_bounds__(COMBINE_THREADS) k_msm_combine(TailBatch tb, int windows) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    __shared__ uint4 s_pts[(COMBINE_THREADS / 64) * 4 * W];
    const uint4* __restrict__ win_pts = tb.s[blockIdx.x].win_pts;
    const int tid = threadIdx.x, ql = tid & 3, item = tid >> 2;
    XyzzZ<FP> acc = item < windows ? xyzzz_load_packed<FP>(win_pts + (size_t)item * 4 * W) : xyzzz_identity<FP>();
    acc = wave_sum_q<FP>(acc, 16, ql);
    if ((tid & 63) == 0) xyzzz_store_packed<FP>(s_pts + (tid >> 6) * 4 * W, acc);
    __syncthreads();
    if (tid < 64) {
        acc = item < COMBINE_THREADS / 64 ? xyzzz_load_packed<FP>(s_pts + item * 4 * W) : xyzzz_identity<FP>();
        acc = wave_sum_q<FP>(acc, COMBINE_THREADS / 64, ql);
        if (tid == 0) emit_result<FP>(acc, tb.s[blockIdx.x]);
    }
}

#ifdef PLK_FINAL_TRACE
extern "C" int plk_debug_final_trace(unsigned long long* out) {  // 2 x 16 stamps of the last k_msm_final
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_final_trace), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1;
}
#endif

// ---- host side: what msm.hip sees of this file ----------------------------------------------------------------------------
// pieces -> buckets -> (row / column sums ->) bit-plane sums -> result, for the tb.count MSMs of a batch at once.
// stage 0: heavy buckets + assembly (+ row / column sums); 1: plane sums; 2: the final block(s) (+ the sum of many windows)
template <class C> int msm_launch_reduce_stage(int stage, const TailGeom& g, const TailBatch& tb, hipStream_t stream) {
    const uint32_t buckets = g.buckets;
    const unsigned cnt = (unsigned)tb.count;
    const int tw = g.tail_windows;
    // round 6: the lane-driven assembly and the one-workgroup-per-line sums, where a bucket has one lane in k_msm_assemble's terms
    // (large MSMs).  PLK_MSM_TAIL_V1 keeps the round-5 launches (A/B, tests/test_gpu_knobs.py).
    static const bool tail_v1 = getenv("PLK_MSM_TAIL_V1") != nullptr;
    const bool v2 = g.two_level && g.lpb_log == 0 && !tail_v1;
    if (stage == 0) {
        // hot buckets of a skewed scalar distribution (none for uniform scalars: the heavy launches then exit at once)
        // 32768 quads per MSM, a loop over the list (131072 when a bucket share is in the batch: ~130 k live head pieces, 180 us at 512 workgroups)
        if (v2) k_msm_heads<C><<<dim3(g.many_heads ? 2048 : 512, cnt), 256, 0, stream>>>(tb, buckets, g.heavy_cap);
        else k_msm_heavy_list<<<dim3((buckets + 255) / 256, cnt), 256, 0, stream>>>(tb, buckets, g.heavy_cap, g.lpb_log);
        k_msm_heavy_chunks<C><<<dim3(256, cnt), 256, 0, stream>>>(tb, g.heavy_cap);
        k_msm_heavy_final<C><<<dim3(64, cnt), 256, 0, stream>>>(tb, g.heavy_cap);
        const unsigned ab = (unsigned)((((size_t)buckets << g.lpb_log) + 255) / 256);
        // the grid as it lies in memory: 2^Hm rows of 2^Lm consecutive slots (transposed: the two halves of the bucket number swapped)
        const int Lm = g.transposed ? g.H : g.L, Hm = g.transposed ? g.L : g.H;
        if (v2) {
            const uint32_t nbg = (1u << (g.L + g.H)) >> g.g_log;
            const unsigned wins = g.table_free ? (unsigned)g.windows : 1u;
            // the first levels of the lines' trees inside k_msm_gsum_tree's workgroups (PLK_MSM_TREE: levels, 0 = k_msm_gsum; default 2:
            // same lease, assembly + row / column sums of one 2^20 MSM: 0.187 ms at 2 levels, 0.190-0.193 at 3, 0.189-0.197 at 0, 0.219-0.222 for round 5's launches)
            static const int tree_env = getenv("PLK_MSM_TREE") ? atoi(getenv("PLK_MSM_TREE")) : 2;
            int lv = tree_env;
            if (g.L - g.g_log < 3 || g.H - g.g_log < 3 || Lm < 4 || nbg < 128) lv = 0;
            if (lv >= 3) k_msm_gsum_tree<C, 3><<<dim3(2 * (nbg >> 7), cnt, wins), 128, 0, stream>>>(tb, Lm, Hm, g.g_log);
            else if (lv == 2) k_msm_gsum_tree<C, 2><<<dim3(2 * (nbg >> 7), cnt, wins), 128, 0, stream>>>(tb, Lm, Hm, g.g_log);
            else k_msm_gsum<C><<<dim3((2 * nbg + 127) / 128, cnt, wins), 128, 0, stream>>>(tb, Lm, Hm, g.g_log, 1);
            // (one workgroup of 64 quads per line instead of k_msm_lsum's one wave measured SLOWER, 246 against 215 us for the stage: every
            // wave pays for every level of a quad tree whatever the number of quads still in it)
            const int g2 = g.g_log + (lv >= 3 ? 3 : lv == 2 ? 2 : 0);
            const int longest = (g.H > g.L ? g.H : g.L) - g2, qpl = longest < 4 ? longest : 4;
            const size_t lanes2 = ((size_t)2 << g.H) << (qpl + 2);
            k_msm_lsum<C><<<dim3((unsigned)((lanes2 + 255) / 256), cnt, wins), 256, 0, stream>>>(tb, g.L, g.H, g2, qpl, g.transposed);
        } else if (g.two_level) {
            const uint32_t nbg = (1u << (g.L + g.H)) >> g.g_log;  // groups per window (rows; as many for the columns)
            const unsigned wins = g.table_free ? (unsigned)g.windows : 1u;
            // (Reading the pieces directly in the row / column sums - no k_msm_assemble pass - was built and measured in round 3: the merge of
            // the rare live head pieces, inlined or out of line, takes k_msm_gsum from ~100 to 226-232 registers, and the tail went from
            // 0.228 to 0.259 ms.  The separate 44 us pass stays.)
            k_msm_assemble<C, false><<<dim3(ab, cnt), 256, 0, stream>>>(tb, buckets, g.lpb_log);
            k_msm_gsum<C><<<dim3((2 * nbg + 127) / 128, cnt, wins), 128, 0, stream>>>(tb, Lm, Hm, g.g_log, 0);
            const size_t lanes = ((size_t)2 << g.H) << (g.lpl_log + 2);
            k_msm_lsum<C><<<dim3((unsigned)((lanes + 255) / 256), cnt, wins), 256, 0, stream>>>(tb, g.L, g.H, g.g_log, g.lpl_log, g.transposed);
        } else {
            k_msm_assemble<C, true><<<dim3(ab, cnt), 256, 0, stream>>>(tb, buckets, g.lpb_log);
        }
    } else if (stage == 1) {
        dim3 pg(g.plane_blocks, g.planes, tw * cnt);
        k_msm_planes<C><<<pg, PLANE_THREADS, 0, stream>>>(tb, tw, g.tail_wbuckets);
    } else {
        static const bool final_v1 = getenv("PLK_MSM_FINAL_V1") != nullptr || tail_v1;
        const int qpp = g.plane_blocks > 1 ? g.plane_blocks / 2 : 1;
        if (g.two_level && !g.table_free && tw == 2 && g.planes * qpp <= 64 && g.planes <= 32 && !final_v1) {
            k_msm_final_pair<C><<<cnt, FINAL_THREADS, 0, stream>>>(tb, g.plane_blocks, g.planes, g.L);
        } else {
            k_msm_final<C><<<tw * cnt, FINAL_THREADS, 0, stream>>>(tb, tw, g.plane_blocks, g.planes, g.tail_shift, g.two_level ? g.L : -1);
            if (tw > FINAL_FUSE_WINDOWS) k_msm_combine<C><<<cnt, COMBINE_THREADS, 0, stream>>>(tb, tw);
        }
    }
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}
template int msm_launch_reduce_stage<TweedledeeCurve>(int, const TailGeom&, const TailBatch&, hipStream_t);
template int msm_launch_reduce_stage<TweedledumCurve>(int, const TailGeom&, const TailBatch&, hipStream_t);
template int msm_launch_reduce_stage<Bls12377Curve>(int, const TailGeom&, const TailBatch&, hipStream_t);
template int msm_launch_reduce_stage<PallasCurve>(int, const TailGeom&, const TailBatch&, hipStream_t);
template int msm_launch_reduce_stage<VestaCurve>(int, const TailGeom&, const TailBatch&, hipStream_t);

}  // namespace plk
