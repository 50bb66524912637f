// msm_acc.hip -- the bucket accumulation of the MSM (replaces curve_msm.rs:131-145 + affine_multisummation_*, curve_summations.rs:24-158).
// Split from msm.hip in round 5: this kernel is the one the MSM spends 60 % of its time in and the one that is tuned most.
#include "msm_dev.cuh"

namespace plk {

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
// Head pieces are mostly short-lived: the head piece of lane l (closed at l's first bucket boundary) belongs to the last
// bucket of lane l - 1, whose piece is still in registers when the loop ends.  Lanes therefore park a closed head piece in
// LDS and their predecessor in the block adds it to its last piece before storing it: at c = 20 (26 entries per bucket,
// 24 per lane) almost every bucket leaves the kernel whole, and k_msm_assemble only finds the head pieces of the first lane
// of a block and of lanes that lie entirely inside one bucket (head_live[lane] = 1).
// Memory discipline of the loop (round 5).  Until round 4 an iteration that closed a bucket made up to four DEPENDENT round trips
// behind `s_waitcnt vmcnt(0)` - off[b + 2] (is the next bucket empty?) right after the ten stores of the piece, off[b + 1] again,
// off[] once more for the bucket of the prefetched entry, then sorted[k + 1] before the table gather could even be issued - and
// some lane of a wave closes a bucket in 92 % of the iterations (26 entries per bucket, 64 lanes): 27.5 % of the wave cycles
// were parked (profiles/r04_rocprofv3_pmc_sq_wave_cycles.txt).  Now nothing in the common path waits for a load issued in the
// same iteration:
//  * `next2` = off[b + 2] is fetched when bucket b OPENS and first looked at when it closes (~26 iterations later): the new
//    bucket is b + 1 unless next2 says it is empty, and only then a search runs;
//  * the entry ids run two ahead of the additions (ent1 = sorted[k + 1] is in a register when iteration k starts), so the table
//    gather of entry k + 1 is issued with no wait in front of it and has the whole addition of entry k to arrive.
// largest bucket lo' >= lo with off[lo'] <= pos, given off[lo] <= pos (empty buckets share an offset with their successor)
PLK_DI uint32_t bucket_search(const uint32_t* __restrict__ off, uint32_t lo, uint32_t buckets, uint32_t pos) {
    uint32_t hi = buckets;  // off[lo] <= pos < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}
// The mixed addition of the loop in a form whose ONE exit leaves the sum in the accumulator's own registers (round 5).  ecz.cuh's
// xyzzz_madd_lazy has three exits (identity accumulator, equal / opposite points, normal) that meet in ~80 register copies per
// iteration; here the arithmetic always runs to the end and the exceptional case (the operands share x: probability 2^-250 for
// random points, the rule for duplicate generators) comes back as a FLAG - 1: equal points, 2: opposite points - that the caller
// repairs out of line, re-reading the entry instead of keeping it alive across the addition.  Same formulas, same bounds
// (ecz.cuh); precondition: acc is not the identity.
// tuning: -DPLK_ACC_SERIAL=1 puts a scheduling barrier after every product of the addition (less interleaving, fewer live temporaries)
#if defined(PLK_ACC_SERIAL) && PLK_ACC_SERIAL
#define PLK_ACC_SB __builtin_amdgcn_sched_barrier(0)
#else
#define PLK_ACC_SB (void)0
#endif
// Y3 = r t - y1 ppp through ONE Montgomery reduction (fz_mul_add2, fz.cuh; round 5).  PLK_ACC_MERGE_Y=0 builds the two-product form.
// Same lease, alternating (tools/acc_ab.py, profiles/r05_accumulate_merged_y.txt): accumulation 0.8325 -> 0.8095 ms, execution 1.509 -> 1.478 ms.
#ifndef PLK_ACC_MERGE_Y
#define PLK_ACC_MERGE_Y 1
#endif
template <class FP> PLK_DI int acc_madd_flag(XyzzZ<FP>& acc, const Fz<FP>& x2, const Fz<FP>& y2) {
    // ordered so that every coordinate of the old accumulator dies as early as it can (the kernel lives at the edge of its register budget)
    Fz<FP> u2 = fz_mul<FP>(x2, acc.zz);                      // < 2
    PLK_ACC_SB;
    Fz<FP> p = fz_sub<FP, 4>(u2, acc.x);                     // < 18
    Fz<FP> pp = fz_sqr<FP>(p);                               // < 3.6
    PLK_ACC_SB;
    Fz<FP> zz3 = fz_mul<FP>(acc.zz, pp);                     // < 1.1      (ZZ dead)
    PLK_ACC_SB;
    Fz<FP> q = fz_mul<FP>(acc.x, pp);                        // < 1.3      (X dead)
    PLK_ACC_SB;
    Fz<FP> ppp = fz_mul<FP>(p, pp);                          // < 1.6      (p, pp dead)
    PLK_ACC_SB;
    Fz<FP> s2 = fz_mul<FP>(y2, acc.zzz);                     // < 2
    PLK_ACC_SB;
    Fz<FP> r = fz_sub_nc<FP, 2, 31>(s2, acc.y);              // < 6
    fz_carry<FP>(r);
    Fz<FP> zzz3 = fz_mul<FP>(acc.zzz, ppp);                  // < 1.1      (ZZZ dead)
    PLK_ACC_SB;
    constexpr bool MERGE_Y = PLK_ACC_MERGE_Y != 0;
    Fz<FP> yp = fz_zero<FP>();
    if constexpr (!MERGE_Y) yp = fz_mul<FP>(acc.y, ppp);     //            (Y dead)
    PLK_ACC_SB;
    Fz<FP> rr = fz_sqr<FP>(r);                               // < 1.3
    PLK_ACC_SB;
    int special = 0;
    if (zz3.l[0] <= 1u && fz_is_zero_mod_p<FP>(zz3)) special = fz_is_zero_mod_p<FP>(rr) ? 1 : 2;
    Fz<FP> x3 = fz_sub_nc<FP, 2, 30>(fz_sub_nc<FP, 1, 29>(rr, ppp), fz_add_nc<FP>(q, q));  // < 7.3
    fz_carry<FP>(x3);
    Fz<FP> t;
    if constexpr (FzCfg<FP>::NZ <= 10) t = fz_sub_nc<FP, 3, 30>(q, x3);
    else t = fz_sub<FP, 3>(q, x3);
    if constexpr (MERGE_Y) {
        // y1 < 3.6 (limbs <= 3 * 2^29 - 3 on entry, exactly normalised after the first merged addition): 4p - y1, limbs <= 2^31 + 2^29
        // Y is EXACTLY normalised in this loop (a product, or the first entry of a piece normalised where it is taken): 4p - y1 has limbs
        // <= 2^30 and t may keep its carries (limbs <= 2^31, r carried): 9 (2^29 2^31 + 2^30 2^29 + 2^58) = 15.75 * 2^60 < 2^64.
        Fz<FP> yn;
        if constexpr (FzCfg<FP>::NZ <= 9) {
            yn = fz_sub_nc<FP, 2, 29>(fz_zero<FP>(), acc.y);
        } else {  // 14 limbs: t is carried already, and the column sums want yn carried too
            yn = fz_sub_nc<FP, 2, 31>(fz_zero<FP>(), acc.y);
            fz_carry<FP>(yn);
        }
        acc.y = fz_mul_add2<FP>(r, t, yn, ppp);              // (6 * 9.3 + 8 * 1.6) / 128 + 1 < 1.6, exactly normalised
    } else {
        acc.y = fz_sub_nc<FP, 1, 29>(fz_mul<FP>(r, t), yp);  // < 3.5: both products exactly normalised, the difference keeps its carries
    }
    acc.x = x3;
    acc.zz = zz3;
    acc.zzz = zzz3;
    return special;
}
// the table entry in the working form: x, and +-y (2p - y without a carry pass: limbs <= 2^30 - 2, ecz.cuh)
template <class FP> PLK_DI void acc_entry(const Fe<FP>& x, const Fe<FP>& y, bool negate, Fz<FP>& xz, Fz<FP>& yz) {
    xz = fz_from_fe<FP>(x);
    const Fz<FP> yp = fz_from_fe<FP>(y);
    const Fz<FP> yn = fz_sub_nc<FP, 1, 29>(fz_zero<FP>(), yp);
#pragma unroll
    for (int i = 0; i < FzCfg<FP>::NZ; ++i) yz.l[i] = negate ? yn.l[i] : yp.l[i];
}
// the loaded values are materialised HERE (an empty asm the compiler must feed with registers)
template <class FP> PLK_DI void acc_pin(Fe<FP>& x, Fe<FP>& y, uint32_t& a, uint32_t& b) {
#pragma unroll
    for (int i = 0; i < FP::NL; ++i) asm volatile("" : "+v"(x.v[i]), "+v"(y.v[i]));
    asm volatile("" : "+v"(a), "+v"(b) : : "memory");
}
// tuning builds only (tools/gpu: A/B libraries in ab_libs/, never the product): -DPLK_ACC_EXPERIMENT=2 keeps every gather inside the first
// 4 MiB of the table (wrong sums, L2-resident reads) - the time this kernel would take if the random HBM gathers cost nothing
#ifndef PLK_ACC_EXPERIMENT
#define PLK_ACC_EXPERIMENT 0
#endif
PLK_DI size_t acc_table_index(size_t idx) {
#if PLK_ACC_EXPERIMENT == 2
    return idx & 0xFFFFu;
#else
    return idx;
#endif
}
template <class FP> PLK_DI void acc_gather(const uint4* src, Fe<FP>& x, Fe<FP>& y) {
    x = fe_load<FP>(src);
    y = fe_load<FP>(src + FP::NL / 4);
}
template <class C>
PLK_DI void msm_accumulate_body(const uint4* __restrict__ tab, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ off,
                                uint4* __restrict__ p_start, uint4* __restrict__ p_head, uint8_t* __restrict__ head_live, uint32_t* __restrict__ head_bucket,
                                uint32_t* __restrict__ live_list, uint32_t* __restrict__ live_count, uint32_t buckets, const uint32_t* __restrict__ dyn_chunk, int wshift,
                                uint32_t n_sub, uint4* s_head, uint8_t* s_parked, uint32_t tab_entries) {
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
        // bucket of the first entry: largest b with off[b] <= begin
        b = bucket_search(off, 0, buckets, begin);
        uint32_t next = off[b + 1];
        uint32_t next2 = off[min(b + 2, buckets)];  // looked at when bucket b closes
        head = off[b] < begin;  // the bucket was already running: this lane's first piece is a head piece
        // the bucket that piece belongs to, for the lane-driven assembly of the reduction (k_msm_heads, msm_tail.hip): written here,
        // before the loop, so that nothing of it stays live across the additions
        head_bucket[lane] = head ? b : HEAD_NONE;
        // entry ids are window * n + generator; with tables that is the table index, without (table-free mode:
        // n_sub = n, buckets of window w are [w << wshift, (w + 1) << wshift)) the window part is taken off
        const uint32_t ent_sub = (b >> wshift) * n_sub;
        (void)PLK_CHK(b < buckets && end <= total, CHK_ENTRY_RANGE);
        uint32_t ent = sorted[begin];
        uint32_t ent1 = sorted[min(begin + 1, end - 1)];
        // the table point of the entry in flight, as loaded: the identity flag rides in y's top word (affine_store, ec.cuh) and is
        // only looked at when the point is consumed - testing it here would wait for the gather that was just issued
        Fe<FP> x = fe_zero<FP>(), y = fe_zero<FP>();
        y.v[FP::NL - 1] = AFF_IDENTITY_BIT;
        if (PLK_CHK((ent >> 1) - ent_sub < tab_entries, CHK_TABLE_INDEX)) acc_gather<FP>(tab + acc_table_index((ent >> 1) - ent_sub) * 2 * W, x, y);
        for (uint32_t k = begin; k < end; ++k) {
            // (1) everything this iteration reads from memory was requested at least one addition ago: entry k's table point, the id
            // of entry k + 1, off[b + 2].  The ONE wait of the iteration sits here (acc_pin keeps the compiler from sinking it below
            // the stores of a closed piece: the memory counter of gfx950 is in order, a wait behind them would wait for them too).
            const uint32_t cur = ent;
            Fe<FP> cx = x, cy = y;
            uint32_t e1 = ent1, n2 = next2;
            acc_pin<FP>(cx, cy, e1, n2);
            const bool cident = (cy.v[FP::NL - 1] & AFF_IDENTITY_BIT) != 0;
            cy.v[FP::NL - 1] &= ~AFF_IDENTITY_BIT;
            if (k == next) {  // (2) entry k opens a new bucket: the piece of the old one is closed
                // (the additions keep Y uncarried, ecz.cuh: the piece is stored as it is, xyzzz_load_raw moves the carries)
                if (head) {
                    xyzzz_store_raw<FP>(s_head + tid * RU, acc);
                    parked = true;
                } else if (PLK_CHK(b < buckets, CHK_BUCKET)) {
                    xyzzz_store_raw<FP>(p_start + (size_t)b * RU, acc);
                }
                head = false;
                acc.inf = true;  // the coordinates stay as they are (36 register clears less): the next entry overwrites them
                uint32_t nb = b + 1, nn = n2;
                if (n2 <= k) {  // bucket b + 1 is empty (e^-26 of the buckets at c = 20; the rule for sparse vectors)
                    nb = bucket_search(off, b + 1, buckets, k);
                    nn = off[nb + 1];
                }
                b = nb;
                next = nn;
                next2 = off[min(b + 2, buckets)];
            }
            if (k + 1 < end) {  // (3) request entry k + 1's table point and the id of entry k + 2
                uint32_t nsub = 0;
                if (n_sub) {
                    // table-free mode: the next entry may belong to a later bucket in another window; its bucket is known here
                    uint32_t nb = b;
                    if (k + 1 == next) nb = next2 > k + 1 ? b + 1 : bucket_search(off, b + 1, buckets, k + 1);
                    (void)PLK_CHK(nb < buckets, CHK_BUCKET);
                    nsub = (nb >> wshift) * n_sub;
                }
                ent = e1;
                ent1 = sorted[min(k + 2, end - 1)];
#ifdef PLK_CHECKED
                y.v[FP::NL - 1] = AFF_IDENTITY_BIT;  // a guarded (skipped) gather leaves the identity behind
#endif
                if (PLK_CHK((ent >> 1) - nsub < tab_entries, CHK_TABLE_INDEX)) acc_gather<FP>(tab + acc_table_index((ent >> 1) - nsub) * 2 * W, x, y);
            }
            if (cident) continue;
            {  // (4)
                Fz<FP> xz, yz;
                acc_entry<FP>(cx, cy, (cur & 1u) != 0, xz, yz);
                if (acc.inf) {  // first entry of a piece (or after a sum that cancelled)
                    acc.x = xz;
                    acc.y = yz;
                    fz_normalize<FP>(acc.y);  // 2p - y comes without its carries; the additions want Y exactly normalised (acc_madd_flag)
                    acc.zz = fz_one_rprime<FP>();
                    acc.zzz = acc.zz;
                    acc.inf = false;
                } else {
                    const int special = acc_madd_flag<FP>(acc, xz, yz);
                    if (special == 2) {
                        acc.inf = true;  // P + (-P)
                    } else if (special == 1) {
                        // P + P: the entry is read again (rare: nothing of it was kept alive across the addition) and doubled
                        Fe<FP> rx, ry;
                        (void)affine_load<FP>(tab + acc_table_index((cur >> 1) - (b >> wshift) * n_sub) * 2 * W, rx, ry);
                        Fz<FP> dx, dy;
                        acc_entry<FP>(rx, ry, (cur & 1u) != 0, dx, dy);
                        fz_carry<FP>(dy);
                        acc = xyzzz_mdbl<FP>(dx, dy);
                    }
                }
            }
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
    const bool live = head || (parked && tid == 0);
    head_live[lane] = live ? 1 : 0;
    // the lanes whose head piece is still a piece of its own, as a list (order: whoever gets there first; *live_count counts them and is
    // zero between executions): the reduction visits these ~1.5 k lanes of 2^17.6 instead of every bucket (k_msm_heads, msm_tail.hip)
    if (live) live_list[atomicAdd(live_count, 1u)] = lane;
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
                                                                uint8_t* __restrict__ head_live, uint32_t* __restrict__ head_bucket,
                                                                uint32_t* __restrict__ live_list, uint32_t* __restrict__ live_count, uint32_t buckets,
                                                                const uint32_t* __restrict__ dyn_chunk, int wshift, uint32_t n_sub, uint32_t tab_entries) {
    __shared__ uint4 s_head[ACC_THREADS * raw_u4<typename C::FP>()];
    __shared__ uint8_t s_parked[ACC_THREADS];
    msm_accumulate_body<C>(tab, sorted, off, p_start, p_head, head_live, head_bucket, live_list, live_count, buckets, dyn_chunk, wshift, n_sub, s_head, s_parked, tab_entries);
}

// ---- host side: what msm.hip sees of this file ----------------------------------------------------------------------------
// workgroups of the accumulation a CU holds at once (registers and LDS of THIS build of the kernel: the chunk length is sized from it)
template <class C> int msm_accumulate_blocks_per_cu() {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_msm_accumulate<C>, ACC_THREADS, 0) != hipSuccess || per_cu <= 0) per_cu = 6;
    return per_cu;
}
template <class C>
void msm_launch_accumulate(unsigned blocks, hipStream_t stream, const void* tab, const void* sorted, const void* off, void* p_start, void* p_head,
                           void* head_live, void* head_bucket, void* live_list, uint32_t* live_count, uint32_t buckets, const uint32_t* dyn_chunk, int wshift,
                           uint32_t n_sub, uint32_t tab_entries) {
    k_msm_accumulate<C><<<blocks, ACC_THREADS, 0, stream>>>((const uint4*)tab, (const uint32_t*)sorted, (const uint32_t*)off, (uint4*)p_start, (uint4*)p_head,
                                                            (uint8_t*)head_live, (uint32_t*)head_bucket, (uint32_t*)live_list, live_count, buckets, dyn_chunk, wshift, n_sub, tab_entries);
}
#define PLK_ACC_INSTANTIATE(C)                                                                                                          \
    template int msm_accumulate_blocks_per_cu<C>();                                                                                    \
    template void msm_launch_accumulate<C>(unsigned, hipStream_t, const void*, const void*, const void*, void*, void*, void*, void*, void*, uint32_t*, uint32_t, \
                                           const uint32_t*, int, uint32_t, uint32_t);
PLK_ACC_INSTANTIATE(TweedledeeCurve)
PLK_ACC_INSTANTIATE(TweedledumCurve)
PLK_ACC_INSTANTIATE(Bls12377Curve)
PLK_ACC_INSTANTIATE(PallasCurve)
PLK_ACC_INSTANTIATE(VestaCurve)
#undef PLK_ACC_INSTANTIATE

// ---- the checked build (-DPLK_CHECKED: msm_acc_checked.o + msm_order_checked.o; include/plonky_hip.h) ----
int msm_order_checked_failures(unsigned* counts);  // msm_order.hip
PLK_CHK_READER(msm_acc_checked_failures)
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
    unsigned part[8];
    PLK_TRY(msm_acc_checked_failures(part));    // the accumulation kernel's guards
    for (int k = 0; k < 8; ++k) counts[k] += part[k];
    PLK_TRY(msm_order_checked_failures(part));  // the ordering kernels' guards
    for (int k = 0; k < 8; ++k) counts[k] += part[k];
#endif
    return PLK_OK;
}

}  // namespace plk
