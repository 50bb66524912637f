// msm_dev.cuh -- device-side definitions shared by the translation units of the MSM (msm.hip: ordering, reduction, host side;
// msm_acc.hip: the bucket accumulation).  The kernels are split over several files so that a change to one of them rebuilds in
// a minute instead of five; device code is not relocatable here (no -fgpu-rdc), so anything both sides need lives in this header.
#pragma once
#include "common.h"
#include "ec.cuh"
#include "ecz.cuh"

namespace plk {

// -DPLK_CHECKED (make checked -> libplonky_hip_checked.so; SURVEY.md section 5: the reference's debug assertions and overflow
// checks have no equivalent in a release kernel): every index the ordering and accumulation kernels compute into sorted[],
// tmp[], the tables and the bucket arrays is compared with its bound; a violation is counted per site and the access is
// skipped.  plk_checked_failures() reads the counters.  In the normal build the guards compile to nothing.
#ifdef PLK_CHECKED
static __device__ unsigned g_plk_chk[8];  // one copy per translation unit; checked_failures_impl adds them up
#define PLK_CHK(cond, site) (!(cond) ? (atomicAdd(&g_plk_chk[site], 1u), false) : true)
#else
#define PLK_CHK(cond, site) (true)
#endif
enum { CHK_TMP_INDEX = 0, CHK_TILE_STAGE = 1, CHK_SORTED_INDEX = 2, CHK_SEG_STAGE = 3, CHK_TABLE_INDEX = 4, CHK_BUCKET = 5, CHK_ENTRY_RANGE = 6 };

// accumulation pieces as they travel between the kernels (lazy 29-bit limbs, accumulator invariant of ecz.cuh; the identity is ZZ = 0)
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
    // a piece closed inside the accumulation's loop is stored with its Y UNCARRIED (limbs <= 3 * 2^29 - 3, ecz.cuh: the carry pass
    // would sit on the path some lane of a wave takes almost every round); the carries are moved here, where the piece is read
    fz_carry<FP>(r.y);
    return r;
}

constexpr int ACC_THREADS = 128;

// ---- geometry shared by the host side (msm.hip) and the kernels' translation units (msm_order.hip, msm_tail.hip) ----
constexpr int MSM_MAX_PLANE_PARTS = 16;  // blocks per bit-plane in the reduction (planes * parts quads must fit the final block)
constexpr int MSM_TF_MAX_WINDOW = 16;  // table-free mode: every window has its own 2^(c-1) buckets
constexpr int MSM_MAX_WINDOW = 21;   // c - 1 <= 10 coarse + 11 fine bits in the partition (ORD_MAX_BINS, ORD_MAX_FINE)
constexpr uint32_t CODE_INVALID = 0xFFFFFFFFu;
constexpr int ORD_THREADS = 256;
constexpr int ORD_TILE = 4096;      // entries staged per tile of the level-1 scatter
constexpr int ORD_MAX_BINS = 1024;  // coarse bins
constexpr int ORD_MAX_FINE = 11;    // fine bits: buckets per coarse bin <= 2048
constexpr int ORD_BIN_THREADS = 512;
constexpr uint32_t ORD_SEG = 8192;  // entries per level-2 workgroup
constexpr int ORD_SEG_EPT = (int)(ORD_SEG / ORD_BIN_THREADS);  // ... and per thread of it
// k_ord_bin_scatter stages a whole segment in LDS (3 fine-bit tables + the staged entries): ~74 KB, above the 64 KB a workgroup
// gets on gfx90a / gfx942 - this library is built for gfx950 (160 KB of LDS per CU) only, plk_init refuses other devices
static_assert(3 * (4u << ORD_MAX_FINE) + 4 * ORD_BIN_THREADS + 6 * ORD_SEG <= 160 * 1024, "k_ord_bin_scatter's LDS tile must fit a gfx950 CU");
constexpr uint32_t ORD2_BIN_CAP = 32768;  // round 6 (k_ord_bin_sort): entries of a coarse bin that are ordered inside LDS by one workgroup (128 KiB)
constexpr int PLANE_THREADS = 512;
constexpr int FINAL_FUSE_WINDOWS = 4;  // up to this many tail windows are added by the last block of k_msm_final itself
constexpr int FINAL_THREADS = 512;  // <= 8 waves, so the compiler may use 256 VGPRs: the point arithmetic must not spill
constexpr int COMBINE_THREADS = 512;

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
    // round 6: 1 = the tile-major level 1 (k_ord_tiles, msm_order.hip) with the coarse bin taken from the LOW bits of the bucket id: the
    // buckets are ordered (and numbered, for everything downstream) by v = [low coarse bits | high fine bits] of the digit's bucket
    // b = |d| - 1, so that a short top window - whose digits are all small - spreads over every bin instead of filling the first few.
    // The reduction reads the weight of v off its two halves (TailGeom::transposed).
    int perm;
    // round 6: only entries whose coarse bin lies in [bin_lo, bin_hi) are kept (0, nbins: all of them).  A rank of a device group that
    // takes a BUCKET range of a sharded vector - every rank reads the whole vector and keeps its N-th of the bins - orders, accumulates
    // and reduces an N-th of the entries over an N-th of the buckets (plk_msm_execute_parts_buckets_dev).
    uint32_t bin_lo, bin_hi;
};

constexpr int TAIL_MAX = 16;
struct TailSlot {
    const uint32_t* off;  // bucket offsets off[buckets + 1]
    uint4* p_start;       // raw, one per bucket (becomes the assembled bucket)
    const uint4* p_head;  // raw, one per accumulation lane
    const uint8_t* head_live;  // 1: p_head[lane] holds a piece that is not part of a start piece yet
    const uint32_t* head_bucket;  // the bucket lane's head piece belongs to (HEAD_NONE: the lane starts at a bucket boundary)
    const uint32_t* live_list;    // the lanes whose head piece is live, in no particular order (k_msm_accumulate appends)
    uint32_t* live_count;         // how many; zero between executions (reset by the reduction)
    uint4* bucket;        // packed points: the operands of the plane sums
    uint32_t* heavy;
    uint4* heavy_part;    // raw
    uint4* line_part;     // raw: row partials then column partials
    uint4* plane_part;
    uint4* win_pts;
    uint32_t* final_done;  // windows finished by k_msm_final (the last one adds them up); zero between executions
    const uint32_t* dyn_chunk;  // entries per accumulation lane of this execution (k_ord_scan1)
    uint4* out_xy;        // the result: x | y affine, or (projective) x | y | z
    uint8_t* out_zero;
    int projective;       // 1: the reference's ProjectivePoint, not normalised (emit_projective, ecz.cuh)
};
struct TailBatch {
    int count;
    TailSlot s[TAIL_MAX];
};

constexpr uint32_t HEAD_NONE = 0xFFFFFFFFu;
constexpr uint32_t HEAVY_HEADS = 32;   // more head pieces than this PER LANE of k_msm_assemble (2^lpb_log lanes per bucket): the bucket is summed by workgroups
constexpr uint32_t HEAVY_CHUNK = 2048;

// what the reduction's launches need of a context (msm_tail.hip: msm_launch_reduce_stage)
struct TailGeom {
    uint32_t buckets, heavy_cap, tail_wbuckets;
    uint32_t max_lanes;  // upper bound of the accumulation lanes of an execution (k_msm_heads is launched for it)
    int lpb_log, two_level, L, H, g_log, lpl_log, table_free, windows, tail_windows, plane_blocks, planes, tail_shift;
    // 1: bucket slot v = lo * 2^H + hi holds the bucket of weight hi * 2^L + lo + 1 (OrdCfg::perm): the grid in memory is 2^L rows of
    // 2^H slots, its ROW sums are the column sums C_lo of the weighting and its column sums the row sums R_hi
    int transposed;
    // 1: some vector of the batch is a BUCKET share (OrdCfg::bin_lo / bin_hi): its entries are spread over all the accumulation lanes in
    // chains shorter than a bucket, so most lanes end inside a bucket and the list of live head pieces is long - k_msm_heads gets a wide grid
    int many_heads;
};
// the buffers of one ordering (msm_order.hip: msm_launch_order_stage)
struct OrdBuffers {
    const void* scalars;
    size_t n;
    void *cnt1, *tmp, *sorted, *cnt2, *off;
    uint32_t *bin_total, *bin_base, *seg_base, *done_counter;
    uint32_t chunk, lanes, buckets;
};
// the guard counters of a translation unit (-DPLK_CHECKED), read back for plk_checked_failures
#ifdef PLK_CHECKED
#define PLK_CHK_READER(fn)                                                                        \
    int fn(unsigned* counts) {                                                                    \
        PLK_HIP_TRY(hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_plk_chk), 8 * sizeof(unsigned)));    \
        return PLK_OK;                                                                            \
    }
#else
#define PLK_CHK_READER(fn)                       \
    int fn(unsigned* counts) {                   \
        for (int k = 0; k < 8; ++k) counts[k] = 0; \
        return PLK_OK;                           \
    }
#endif

}  // namespace plk
