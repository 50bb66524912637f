// halo.hip -- the rounds of one inner-product argument (src/halo.rs:63-124) behind the C ABI, device resident.
//
// The reference keeps halo_a, halo_b (scalars) and halo_g (points) on the host and, per round j = log n .. 1:
//   L_j = msm_parallel(a_lo, g_hi) + [l_j] H + [<a_lo, b_hi>] U'        halo.rs:86-89
//   R_j = msm_parallel(a_hi, g_lo) + [r_j] H + [<a_hi, b_lo>] U'        halo.rs:90-93
//   (transcript -> challenge u_j, host side)                             halo.rs:95-114
//   halo_a = u^-1 a_hi + u a_lo,  halo_b = u^-1 b_lo + u b_hi            halo.rs:117-118
//   halo_g_i = msm_parallel([u^-1, u], [g_lo_i, g_hi_i])                 halo.rs:119-123
// Here the three vectors live in HBM inside a context; a round is plk_halo_round_lr (the two points cross PCIe: the
// transcript is on the host) and plk_halo_round_fold (two scalars go in as kernel arguments).  No context is created or
// freed per round, nothing is allocated, the blinding factors and challenges never touch device memory as separate copies.
//
// Two regimes:
//  * long vectors: L_j / R_j are ONE table-free MSM each over [half of G, H, U'] with the scalars [half of a, blinding factor,
//    inner product] (two persistent table-free contexts rebound to the halved generator set every round, on two streams);
//    the generators are folded pair by pair (fold.hip) - SCALED: the context keeps G~ and a scalar c with G^(k) = [c] G~^(k),
//    so that a fold is G~'_i = G~_lo_i + [u^2] G~_hi_i, c' = c u^-1 (one half-length joint sparse form and one addition per pair
//    instead of two joint sparse forms: ~26 % fewer field multiplications); c goes into the scalars of the MSMs
//    (<a_lo, G_hi> = <c a_lo, G~_hi>), into the coefficients when the generators are frozen, and onto the points when they are read.
//  * short vectors (<= 2^freeze_log generators left): a table-free MSM and the fold are then pure latency (a chain of
//    ~120-130 doublings of one point per lane, ~2 ms whatever the size), so the generators are FROZEN: window tables are
//    built once for G^(f) (+ H, U'), the folded generators are never formed again, and every later round uses
//        G^(k)_i = sum_{j = i mod n_k} s_j G^(f)_j,   s_j = prod over the folds since of (u^-1 if j fell in the low half, else u)
//    (the identity behind halo_s, plonk_util.rs:311-339): L_j and R_j are the two scalar vectors of ONE batched tabled MSM
//    over the frozen set with the scalars a_i s_j, and the final generator is one more MSM with the scalars s_j.  Same group
//    elements as the reference's, so the affine results are bit-identical.
//  * VIRTUAL rounds, in stages of r (curves with the endomorphism): the same identity from the other end.  For r rounds the
//    generator set V stays what it is - L_j / R_j run over V with the scalars a_i s_j - and ONE 2^r-to-1 fold (fold.hip) then
//    forms the generators r rounds on: a single doubling chain per output instead of one per output of every round (~40 % of the
//    fold work of two rounds).  The first stage runs over the CALLER's commitment tables when it hands them over
//    (plk_halo_begin_tabled_dev; plonk.rs:65 `pedersen_g_msm_precomputation`): one batched tabled MSM per round, H and U' as two
//    more of its scalars when the tables hold them, else from the pair kernel on side streams.  Later stages (and the first one
//    without tables) run over the explicit, scaled set with the two table-free contexts: every element of V is in exactly one of
//    L_j / R_j, so the second round of a stage costs what its first did (instead of ~60 %) and saves most of a fold.
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.h"
#include "fp.cuh"
#include "field_params.cuh"

struct plk_msm_ctx;

namespace plk {

int msm_precompute_dev_impl(int curve, size_t n, const void* d_bases, const void* d_zero, unsigned window_bits, unsigned flags, hipStream_t stream,
                            plk_msm_ctx** out_ctx, const void* d_extra = nullptr, size_t n_extra = 0, const size_t* also_n = nullptr,
                            int also_count = 0);
int msm_rebind_dev_impl(plk_msm_ctx* ctx, size_t n, const void* d_bases, const void* d_zero, const void* d_extra, size_t n_extra, hipStream_t stream);
int msm_execute_dev_impl(plk_msm_ctx* ctx, unsigned batch, const void* d_scalars, size_t n_scalars, void* d_out_xy, void* d_out_zero, hipStream_t stream,
                         hipEvent_t* ready = nullptr, const MsmParts* parts = nullptr, unsigned out_flags = 0);
int msm_reserve_workspaces_impl(plk_msm_ctx* ctx, unsigned count, hipStream_t stream);
// hostnorm.cpp: ProjectivePoint::to_affine on the host for `count` points (x | y | z -> x | y)
int host_projective_to_affine(int curve, unsigned count, const uint8_t* xyz, const uint8_t* zero, uint8_t* xy);
void msm_ctx_delete(plk_msm_ctx* ctx);
int curve_fold_pairs_dev_impl(int curve, size_t m, const void* d_lo, const void* d_lo_zero, const void* d_hi, const void* d_hi_zero,
                              const uint64_t* a_mont, const uint64_t* b_mont, void* d_out_xy, void* d_out_zero, hipStream_t stream,
                              const void* d_scalars = nullptr, int plus_lo = 0);
int curve_fold_multi_dev_impl(int curve, size_t n_out, int r_bits, const void* d_g, const void* d_gz, const void* d_ratios, void* d_out_xy,
                              void* d_out_zero, hipStream_t stream);
size_t msm_ctx_len(const plk_msm_ctx* ctx);
int msm_ctx_curve(const plk_msm_ctx* ctx);
int msm_ctx_table_free(const plk_msm_ctx* ctx);
int msm_ctx_is_comb(const plk_msm_ctx* ctx);
size_t msm_partials_bytes(int curve, unsigned slots);
int msm_combine_partials_dev_impl(int curve, unsigned world, unsigned batch, unsigned whole_per_rank, const void* d_gathered, void* d_out_xy, void* d_out_zero,
                                  hipStream_t stream);

constexpr int HALO_PART_BLOCKS = 256;
struct HaloScalar {
    uint32_t v[8];
};

// One pass over the halves of a and b: the partial sums of <a_lo, b_hi> and <a_hi, b_lo> (Field::inner_product,
// field.rs:213-221; field addition is exact and associative, any order gives the reference's value) and the scalar vectors
// of the two MSMs.  Not frozen (m0 == 0): sL = a_lo, sR = a_hi (m entries each).  Frozen: entry j < m0 of the frozen
// generator set belongs to folded generator r = j mod n_k; it takes part in L (over G_hi) when r >= m with a_lo[r - m], in R
// (over G_lo) when r < m with a_hi[r] - times its coefficient s_j; the other vector gets 0 (a zero scalar has no digits).
// `scale`: the running scale c of the explicitly folded generators (one element, device memory): their MSM scalars are c a.
template <class P>
__global__ void __launch_bounds__(256) k_halo_prepare(const uint4* __restrict__ a, const uint4* __restrict__ b, size_t m, size_t m0,
                                                      const uint4* __restrict__ coef, const uint4* __restrict__ scale, uint4* __restrict__ sL,
                                                      uint4* __restrict__ sR, uint4* __restrict__ part) {
    constexpr int W = P::NL / 4;
    __shared__ uint4 s_acc[2 * 256 * W];
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fe<P> accL = fe_zero<P>(), accR = fe_zero<P>();
    const Fe<P> c = fe_load<P>(scale);
    for (size_t i = t0; i < m; i += stride) {
        const Fe<P> alo = fe_load<P>(a + i * W), ahi = fe_load<P>(a + (m + i) * W);
        accL = fe_add<P>(accL, fe_mul<P>(alo, fe_load<P>(b + (m + i) * W)));
        accR = fe_add<P>(accR, fe_mul<P>(ahi, fe_load<P>(b + i * W)));
        if (m0 == 0) {
            fe_store<P>(sL + i * W, fe_mul<P>(c, alo));
            fe_store<P>(sR + i * W, fe_mul<P>(c, ahi));
        }
    }
    const size_t mask = 2 * m - 1;
    for (size_t j = t0; j < m0; j += stride) {
        const size_t r = j & mask;
        const Fe<P> v = fe_mul<P>(fe_load<P>(a + (r >= m ? r - m : m + r) * W), fe_load<P>(coef + j * W));
        fe_store<P>(sL + j * W, r >= m ? v : fe_zero<P>());
        fe_store<P>(sR + j * W, r >= m ? fe_zero<P>() : v);
    }
    fe_store<P>(s_acc + threadIdx.x * W, accL);
    fe_store<P>(s_acc + (256 + threadIdx.x) * W, accR);
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            accL = fe_add<P>(accL, fe_load<P>(s_acc + (threadIdx.x + d) * W));
            accR = fe_add<P>(accR, fe_load<P>(s_acc + (256 + threadIdx.x + d) * W));
            fe_store<P>(s_acc + threadIdx.x * W, accL);
            fe_store<P>(s_acc + (256 + threadIdx.x) * W, accR);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        fe_store<P>(part + blockIdx.x * W, accL);
        fe_store<P>(part + (HALO_PART_BLOCKS + blockIdx.x) * W, accR);
    }
}
// the two inner products from their partial sums; the blinding factor and the inner product become the scalars of H and U':
// s[cnt] = blinding factor, s[cnt + 1] = inner product
// ip_scale (lead rounds with H and the fixed generator U inside the caller's tables, U' = [x] U): the blinding factor goes to index
// cnt, the inner product TIMES x to index cnt2
template <class P>
__global__ void __launch_bounds__(64) k_halo_close(const uint4* __restrict__ part, unsigned blocks, HaloScalar l_blind, HaloScalar r_blind, size_t cnt,
                                                   uint4* __restrict__ sL, uint4* __restrict__ sR, size_t cnt2 = 0, const uint4* __restrict__ ip_scale = nullptr) {
    constexpr int W = P::NL / 4;
    __shared__ uint4 s_acc[2 * 64 * W];
    Fe<P> accL = fe_zero<P>(), accR = fe_zero<P>();
    for (unsigned i = threadIdx.x; i < blocks; i += 64) {
        accL = fe_add<P>(accL, fe_load<P>(part + i * W));
        accR = fe_add<P>(accR, fe_load<P>(part + (HALO_PART_BLOCKS + i) * W));
    }
    fe_store<P>(s_acc + threadIdx.x * W, accL);
    fe_store<P>(s_acc + (64 + threadIdx.x) * W, accR);
    __syncthreads();
    for (int d = 32; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            accL = fe_add<P>(accL, fe_load<P>(s_acc + (threadIdx.x + d) * W));
            accR = fe_add<P>(accR, fe_load<P>(s_acc + (64 + threadIdx.x + d) * W));
            fe_store<P>(s_acc + threadIdx.x * W, accL);
            fe_store<P>(s_acc + (64 + threadIdx.x) * W, accR);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        Fe<P> l, r;
#pragma unroll
        for (int k = 0; k < P::NL; ++k) {
            l.v[k] = l_blind.v[k];
            r.v[k] = r_blind.v[k];
        }
        if (ip_scale) {
            const Fe<P> x = fe_load<P>(ip_scale);
            accL = fe_mul<P>(accL, x);
            accR = fe_mul<P>(accR, x);
        } else {
            cnt2 = cnt + 1;
        }
        fe_store<P>(sL + cnt * W, l);
        fe_store<P>(sL + cnt2 * W, accL);
        fe_store<P>(sR + cnt * W, r);
        fe_store<P>(sR + cnt2 * W, accR);
    }
}
// halo_a' = u^-1 a_hi + u a_lo, halo_b' = u^-1 b_lo + u b_hi in place (element i of the low half only depends on elements i and
// m + i); frozen generators: s_j *= u^-1 when j falls in the low half of the current length, else u
// dsc (explicit folds only, else null): [0] unused, [1] = u^2 (the scalar of the scaled generator fold), [2] = c *= u^-1
template <class P>
__global__ void __launch_bounds__(256) k_halo_fold_scalars(uint4* __restrict__ a, uint4* __restrict__ b, size_t m, HaloScalar u_s, HaloScalar uinv_s,
                                                           uint4* __restrict__ coef, size_t m0, uint4* __restrict__ dsc) {
    constexpr int W = P::NL / 4;
    Fe<P> u, uinv;
#pragma unroll
    for (int k = 0; k < P::NL; ++k) {
        u.v[k] = u_s.v[k];
        uinv.v[k] = uinv_s.v[k];
    }
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (dsc && i == 0) {
        fe_store<P>(dsc + 1 * W, fe_sqr<P>(u));
        fe_store<P>(dsc + 2 * W, fe_mul<P>(fe_load<P>(dsc + 2 * W), uinv));
    }
    if (i < m) {
        const Fe<P> alo = fe_load<P>(a + i * W), ahi = fe_load<P>(a + (m + i) * W);
        const Fe<P> blo = fe_load<P>(b + i * W), bhi = fe_load<P>(b + (m + i) * W);
        fe_store<P>(a + i * W, fe_add<P>(fe_mul<P>(uinv, ahi), fe_mul<P>(u, alo)));
        fe_store<P>(b + i * W, fe_add<P>(fe_mul<P>(uinv, blo), fe_mul<P>(u, bhi)));
    }
    if (i < m0) {
        const bool low = (i & (2 * m - 1)) < m;
        fe_store<P>(coef + i * W, fe_mul<P>(fe_load<P>(coef + i * W), low ? uinv : u));
    }
}
// lead rounds: after round k (done = k - 1 rounds before it) the scalar of every index that fell into the upper half this round gains
// u_k^2 (dsc[1], just written by k_halo_fold_scalars): ratios[t + 2^done] = ratios[t] u^2, t < 2^done
template <class P> __global__ void __launch_bounds__(64) k_halo_lead_ratios(uint4* __restrict__ ratios, int done, const uint4* __restrict__ dsc) {
    constexpr int W = P::NL / 4;
    const int t = threadIdx.x;
    if (t < (1 << done)) fe_store<P>(ratios + ((size_t)t + ((size_t)1 << done)) * W, fe_mul<P>(fe_load<P>(ratios + (size_t)t * W), fe_load<P>(dsc + 1 * W)));
}
// coef[i] = *src (the running scale when the generators are frozen), or 1 when src is null
template <class P> __global__ void __launch_bounds__(256) k_halo_fill(uint4* __restrict__ coef, size_t count, const uint4* __restrict__ src) {
    constexpr int W = P::NL / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) fe_store<P>(coef + i * W, src ? fe_load<P>(src) : fe_one<P>());
}

}  // namespace plk

struct plk_halo_ctx {
    int curve = 0, sfield = 0, L = 4;
    size_t n0 = 0, n = 0;       // initial / current length of halo_a, halo_b, halo_g
    unsigned freeze_log = 14;
    bool endo = false;          // the curve has the endomorphism: 2^r-to-1 folds, stages of virtual rounds
    hipStream_t stream = nullptr, side = nullptr, side2 = nullptr;
    hipEvent_t ev_main = nullptr, ev_side = nullptr, ev_side2 = nullptr;
    uint8_t* slab = nullptr;    // one allocation: a | b | g | gz | extra | scal | part | out | coef | dsc | ratios | rec | hu
    uint8_t *a = nullptr, *b = nullptr, *g = nullptr, *gz = nullptr, *extra = nullptr, *scal = nullptr, *part = nullptr, *out = nullptr, *coef = nullptr;
    uint8_t* dsc = nullptr;     // 5 scalars: [0] unused (the fold's first scalar), [1] u^2, [2] the running scale c, [3] zero, [4] x with U' = [x] U
    size_t scal_stride = 0;     // bytes between the L and the R scalar vector
    plk_msm_ctx *mL = nullptr, *mR = nullptr;  // table-free contexts of the long rounds, rebound every round
    bool frozen = false;
    size_t m0 = 0;              // generators with a coefficient: the frozen set, or the set of the current stage of virtual rounds
    plk_msm_ctx* mT = nullptr;  // tables over [G^(f), H, U']
    uint8_t* pin = nullptr;     // pinned staging for the results that cross PCIe
    // a stage of virtual rounds: `virt_left` of `virt_total` still to come over the m0 generators the stage began with
    unsigned virt_total = 0, virt_left = 0, stage_depth = 2, stage_min_log = 16;
    // first stage over the caller's tables (not owned)
    plk_msm_ctx* lead_ctx = nullptr;
    unsigned lead_rounds = 0;
    size_t lead_n = 0;          // generators in the caller's context (>= n0; the scalars of the others stay zero)
    bool lead_inside = false;   // H and U sit in the caller's tables (indices lead_h, lead_u; U' = [x] U, x at dsc[4]): no side work
    size_t lead_h = 0, lead_u = 0;
    uint8_t *ratios = nullptr, *rec = nullptr, *hu = nullptr;  // 2^virt_total fold scalars | two partial-sum records | [l, <a,b>] x 2
    size_t rec_bytes = 0;
    bool lr_done = false;
    // a fold failed after the scalars were folded (scratch, launch): the vectors no longer belong to one round - every later
    // call reports that instead of returning an L / R of a half-folded state
    bool poisoned = false;
    std::mutex mu;
    ~plk_halo_ctx() {
        if (stream) (void)hipStreamSynchronize(stream);
        if (side) (void)hipStreamSynchronize(side);
        if (side2) (void)hipStreamSynchronize(side2);
        if (mL) plk::msm_ctx_delete(mL);
        if (mR) plk::msm_ctx_delete(mR);
        if (mT) plk::msm_ctx_delete(mT);
        if (slab) (void)hipFree(slab);
        if (pin) (void)hipHostFree(pin);
        if (ev_main) (void)hipEventDestroy(ev_main);
        if (ev_side) (void)hipEventDestroy(ev_side);
        if (ev_side2) (void)hipEventDestroy(ev_side2);
        plk::stream_pool_release(side);
        plk::stream_pool_release(side2);
    }
};

namespace plk {

static HaloScalar to_halo_scalar(const uint64_t* s) {
    HaloScalar h;
    for (int k = 0; k < 4; ++k) {
        h.v[2 * k] = (uint32_t)s[k];
        h.v[2 * k + 1] = (uint32_t)(s[k] >> 32);
    }
    return h;
}

#define HALO_FIELD_SWITCH(field, CALL)                                        \
    switch (field) {                                                          \
        case PLK_FIELD_TWEEDLEDEE_BASE: { using P = TweedledeeBaseParams; CALL; } break;   \
        case PLK_FIELD_TWEEDLEDUM_BASE: { using P = TweedledumBaseParams; CALL; } break;   \
        case PLK_FIELD_BLS12_377_SCALAR: { using P = Bls12377ScalarParams; CALL; } break;  \
        case PLK_FIELD_PALLAS_BASE: { using P = PallasBaseParams; CALL; } break;           \
        default: { using P = VestaBaseParams; CALL; } break;                               \
    }

static size_t freeze_len(const plk_halo_ctx* c) { return (size_t)1 << (c->freeze_log > 40 ? 40 : c->freeze_log); }

// builds the tables over the current generators: from here on they are never folded again
static int halo_freeze(plk_halo_ctx* c) {
    c->m0 = c->n;
    PLK_TRY(msm_precompute_dev_impl(c->curve, c->m0 + 2, c->g, c->gz, 0, 0, c->stream, &c->mT, c->extra, 2));
    PLK_TRY(msm_reserve_workspaces_impl(c->mT, 2, c->stream));  // L_j and R_j are one batched call
    const unsigned blocks = (unsigned)((c->m0 + 255) / 256);
    HALO_FIELD_SWITCH(c->sfield, (k_halo_fill<P><<<blocks, 256, 0, c->stream>>>((uint4*)c->coef, c->m0, (const uint4*)(c->dsc + 2 * 32))));
    PLK_HIP_TRY(hipGetLastError());
    c->frozen = true;
    // (Replaying the ~20 launches of a frozen round's batched MSM from a hipGraph was measured: 39.6 against 39.2 ms for the whole
    // argument - the round is bound by the dependency chain on the GPU, not by launch overhead; the plain launches stay.)
    return PLK_OK;
}

// What the rounds from the current length on look like: frozen generators, a stage of virtual rounds (over the caller's tables
// first, if any), or single rounds with pairwise folds.
static int halo_next_stage(plk_halo_ctx* c) {
    c->virt_total = c->virt_left = 0;
    if (c->frozen || c->n < 2) return PLK_OK;
    const size_t fz = freeze_len(c);
    if (c->n <= fz) return halo_freeze(c);
    if (!c->endo) return PLK_OK;
    unsigned d = c->lead_ctx ? c->lead_rounds : c->stage_depth;
    // A stage ends at the freezing length at the latest; over the explicit set it must leave 2^16 outputs: the 2^r-to-1 fold is one
    // lane per output - a chain of ~320 point operations - and 2^16 -> 2^14 in one stage measured slower than the two rounds with
    // pairwise folds it replaces (5.15 against 4.94 ms); 2^18 -> 2^16 pays (4.96 against 5.76 ms, with one-wave workgroups: fold.hip).
    const size_t floor_len = c->lead_ctx ? fz : (fz > ((size_t)1 << c->stage_min_log) ? fz : (size_t)1 << c->stage_min_log);
    while (d > 0 && (c->n >> d) < floor_len) --d;
    if (!c->lead_ctx && d < 2) return PLK_OK;  // a stage of one round is a round with a pairwise fold
    if (d == 0) {
        c->lead_ctx = nullptr;
        return PLK_OK;
    }
    c->virt_total = c->virt_left = d;
    c->m0 = c->n;
    // every generator of the stage's set starts with the coefficient c (the scale of the explicit set; 1 over the caller's tables)
    HALO_FIELD_SWITCH(c->sfield, (k_halo_fill<P><<<(unsigned)((c->m0 + 255) / 256), 256, 0, c->stream>>>((uint4*)c->coef, c->m0, (const uint4*)(c->dsc + 2 * 32)),
                                  k_halo_fill<P><<<1, 256, 0, c->stream>>>((uint4*)c->ratios, 1, nullptr)));
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

int halo_begin_dev_impl(int curve, size_t n, const void* d_a, const void* d_b, const void* d_g, const void* d_gz, const uint64_t* h_xy,
                        const uint64_t* u_xy, unsigned freeze_log, hipStream_t stream, plk_halo_ctx** out, plk_msm_ctx* tables, unsigned lead_rounds,
                        size_t h_index, size_t u_index, const uint64_t* u_prime_scalar) {
    if (!out) return set_error(PLK_ERR_INVALID_ARG, "null out");
    *out = nullptr;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (n == 0 || (n & (n - 1))) return set_error(PLK_ERR_NOT_POW2, "halo vectors of length %zu: not a power of two (util.rs:17)", n);
    if (!d_a || !d_b || !d_g || !h_xy || !u_xy) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    PLK_TRY(ensure_device());
    auto* c = new plk_halo_ctx();
    std::unique_ptr<plk_halo_ctx> guard(c);
    c->curve = curve;
    c->sfield = curve_scalar_field(curve);
    c->L = L;
    c->n0 = c->n = n;
    c->stream = stream;
    c->endo = curve != PLK_CURVE_BLS12_377;  // the 2^r-to-1 fold runs along the endomorphism
    // tuning variables apply where the caller left the choice to the library (argument 0), and only with a sane value
    auto env_uint = [](const char* name, unsigned lo, unsigned hi, unsigned& out) {
        if (const char* e = getenv(name)) {
            const long v = atol(e);
            if (v >= (long)lo && v <= (long)hi) out = (unsigned)v;
        }
    };
    if (freeze_log == 0) env_uint("PLK_HALO_FREEZE_LOG", 1, 40, freeze_log);
    c->freeze_log = freeze_log ? freeze_log : 14u;
    if (const char* e = getenv("PLK_HALO_STAGE")) c->stage_depth = (unsigned)atoi(e);
    if (c->stage_depth > 4) c->stage_depth = 4;
    if (const char* e = getenv("PLK_HALO_STAGE_MIN_LOG")) c->stage_min_log = (unsigned)atoi(e);
    if (c->stage_min_log > 40) c->stage_min_log = 40;
    const size_t pt = (size_t)2 * L * 8;
    const size_t fz = freeze_len(c);
    // a first stage over the caller's tables
    if (tables) {
        if (msm_ctx_curve(tables) != curve || msm_ctx_table_free(tables) || msm_ctx_len(tables) < n)
            return set_error(PLK_ERR_INVALID_ARG, "the tables handed to the argument must be a tabled context of this curve over at least %zu generators", n);
        unsigned lead = lead_rounds ? lead_rounds : 2u;
        if (lead_rounds == 0) env_uint("PLK_HALO_LEAD", 0, 4, lead);
        if (lead > 4) lead = 4;
        while (lead > 0 && (n >> lead) < fz) --lead;
        if (c->endo && lead) {
            c->lead_ctx = tables;
            c->lead_rounds = lead;
            c->lead_n = msm_ctx_len(tables);
            if (u_prime_scalar && h_index != (size_t)-1 && u_index != (size_t)-1) {
                if (h_index < n || u_index < n || h_index >= c->lead_n || u_index >= c->lead_n || h_index == u_index)
                    return set_error(PLK_ERR_INVALID_ARG, "pedersen_h / U must be generators %zu .. %zu of the tables (behind halo_g), got %zu and %zu", n,
                                     c->lead_n - 1, h_index, u_index);
                c->lead_inside = true;
                c->lead_h = h_index;
                c->lead_u = u_index;
            }
        }
    }
    c->rec_bytes = msm_partials_bytes(curve, 2);
    // scalar vectors: up to one scalar per generator of a stage's set (+ H, U'); the caller's tables may hold more generators
    const size_t cnt_max = (c->lead_n > n + 2 ? c->lead_n : n + 2);
    c->scal_stride = cnt_max * 32;
    struct Part { uint8_t** p; size_t bytes; } parts[] = {
        {&c->a, n * 32}, {&c->b, n * 32}, {&c->g, n * pt}, {&c->gz, n}, {&c->extra, 2 * pt}, {&c->scal, 2 * c->scal_stride},
        {&c->part, (size_t)2 * HALO_PART_BLOCKS * 32}, {&c->out, 3 * pt + 16}, {&c->coef, n * 32}, {&c->dsc, 5 * 32},
        {&c->ratios, (size_t)16 * 32}, {&c->rec, 2 * c->rec_bytes}, {&c->hu, 4 * 32},
    };
    size_t total = 0;
    for (auto& p : parts) total += (p.bytes + 255) & ~(size_t)255;
    PLK_HIP_TRY(hipMalloc((void**)&c->slab, total));
    uint8_t* cur = c->slab;
    for (auto& p : parts) {
        *p.p = cur;
        cur += (p.bytes + 255) & ~(size_t)255;
    }
    PLK_HIP_TRY(hipHostMalloc((void**)&c->pin, 4 * pt + 64 + 2 * 32, hipHostMallocDefault));
    if (!(c->side = stream_pool_acquire()) || !(c->side2 = stream_pool_acquire())) return PLK_ERR_HIP;
    PLK_HIP_TRY(hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
    PLK_HIP_TRY(hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
    PLK_HIP_TRY(hipEventCreateWithFlags(&c->ev_side2, hipEventDisableTiming));
    PLK_HIP_TRY(hipMemcpyAsync(c->a, d_a, n * 32, hipMemcpyDeviceToDevice, stream));
    PLK_HIP_TRY(hipMemcpyAsync(c->b, d_b, n * 32, hipMemcpyDeviceToDevice, stream));
    PLK_HIP_TRY(hipMemcpyAsync(c->g, d_g, n * pt, hipMemcpyDeviceToDevice, stream));
    if (d_gz) PLK_HIP_TRY(hipMemcpyAsync(c->gz, d_gz, n, hipMemcpyDeviceToDevice, stream));
    else PLK_HIP_TRY(hipMemsetAsync(c->gz, 0, n, stream));
    memcpy(c->pin, h_xy, pt);
    memcpy(c->pin + pt, u_xy, pt);
    PLK_HIP_TRY(hipMemcpyAsync(c->extra, c->pin, 2 * pt, hipMemcpyHostToDevice, stream));
    PLK_HIP_TRY(hipMemsetAsync(c->dsc, 0, 5 * 32, stream));
    if (c->lead_inside) {
        memcpy(c->pin + 2 * pt, u_prime_scalar, 32);
        PLK_HIP_TRY(hipMemcpyAsync(c->dsc + 4 * 32, c->pin + 2 * pt, 32, hipMemcpyHostToDevice, stream));
    }
    HALO_FIELD_SWITCH(c->sfield, (k_halo_fill<P><<<1, 256, 0, stream>>>((uint4*)(c->dsc + 2 * 32), 1, nullptr)));  // c = 1
    PLK_HIP_TRY(hipGetLastError());
    if (c->lead_ctx) {
        // generators of the caller's context beyond n never get a scalar
        PLK_HIP_TRY(hipMemsetAsync(c->scal, 0, 2 * c->scal_stride, stream));
        PLK_TRY(msm_reserve_workspaces_impl(tables, 2, stream));
    }
    const size_t n1 = c->lead_ctx ? n >> c->lead_rounds : n;  // length when the explicit generators take over
    if (n1 > fz) {
        // the two table-free contexts of the long rounds, sized for every set they will be rebound to: a stage's whole set, or half
        // of the generators of a single round (+ H, U')
        size_t also[64];
        int cnt = 0;
        for (size_t len = n1 / 2; len >= fz / 2 && len >= 1 && cnt < 64; len /= 2) also[cnt++] = len + 2;
        PLK_TRY(msm_precompute_dev_impl(curve, n1 + 2, c->g, c->gz, 0, PLK_MSM_TABLE_FREE, stream, &c->mL, c->extra, 2, also, cnt));
        PLK_TRY(msm_precompute_dev_impl(curve, n1 + 2, c->g, c->gz, 0, PLK_MSM_TABLE_FREE, stream, &c->mR, c->extra, 2, also, cnt));
    }
    PLK_TRY(halo_next_stage(c));
    PLK_HIP_TRY(hipStreamSynchronize(stream));  // the staging copy of H, U' is consumed
    *out = guard.release();
    return PLK_OK;
}

int halo_round_lr_impl(plk_halo_ctx* c, const uint64_t* l_blind, const uint64_t* r_blind, uint64_t* lr_xy, uint8_t* lr_zero) {
    if (!c || !l_blind || !r_blind || !lr_xy || !lr_zero) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    PLK_TRY(ensure_device());
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->poisoned) return set_error(PLK_ERR_INVALID_ARG, "the argument's context is unusable: an earlier fold failed half way");
    if (c->n < 2) return set_error(PLK_ERR_INVALID_ARG, "the argument is finished (length %zu)", c->n);
    const size_t m = c->n / 2, pt = (size_t)2 * c->L * 8;
    const bool virt = c->virt_left > 0, lead = virt && c->lead_ctx;
    const size_t m0 = (c->frozen || virt) ? c->m0 : 0, cnt = (c->frozen || virt) ? c->m0 : m;
    uint8_t* sL = c->scal;
    // frozen generators / the caller's tables: the two vectors back to back, one batched call
    uint8_t* sR = c->scal + (lead ? c->lead_n * 32 : c->frozen ? (cnt + 2) * 32 : c->scal_stride);
    const size_t work = m > m0 ? m : m0;
    unsigned blocks = (unsigned)((work + 255) / 256);
    if (blocks > HALO_PART_BLOCKS) blocks = HALO_PART_BLOCKS;
    const HaloScalar lb = to_halo_scalar(l_blind), rb = to_halo_scalar(r_blind);
    const bool beside = lead && !c->lead_inside;  // H and U' are not in the caller's tables: their scalars go to `hu`
    HALO_FIELD_SWITCH(c->sfield, (k_halo_prepare<P><<<blocks, 256, 0, c->stream>>>((const uint4*)c->a, (const uint4*)c->b, m, m0, (const uint4*)c->coef,
                                                                                     (const uint4*)(c->dsc + 2 * 32), (uint4*)sL, (uint4*)sR, (uint4*)c->part),
                                  k_halo_close<P><<<1, 64, 0, c->stream>>>((const uint4*)c->part, blocks, lb, rb, !lead ? cnt : c->lead_inside ? c->lead_h : 0,
                                                                           (uint4*)(beside ? c->hu : sL), (uint4*)(beside ? c->hu + 64 : sR), c->lead_u,
                                                                           (const uint4*)(lead && c->lead_inside ? c->dsc + 4 * 32 : nullptr))));
    PLK_HIP_TRY(hipGetLastError());
    // L_j and R_j leave the device as msm_execute's ProjectivePoints (halo.rs:93-101) and are normalised where the reference normalises
    // them, on the host: the inversion is the last ~35 us of the round's dependency chain on ONE GPU lane and ~2 us on a core
    // (PLK_HALO_DEVICE_AFFINE=1: the device normalises, as before round 6; the records of the beside-the-tables mode stay affine)
    static const bool device_affine = getenv("PLK_HALO_DEVICE_AFFINE") != nullptr;
    const plk_msm_ctx* used = lead ? c->lead_ctx : c->frozen ? c->mT : c->mL;  // (mR is mL's twin)
    const bool proj = !device_affine && !(lead && !c->lead_inside) && !msm_ctx_is_comb(used);  // a comb context (few generators) returns affine points
    const unsigned of = proj ? 1u : 0u;
    const size_t ps = proj ? pt + pt / 2 : pt;  // bytes of one result
    uint8_t* out_xy = c->out;
    uint8_t* out_z = c->out + 2 * ps;
    if (lead && c->lead_inside) {
        // H and U are generators of the caller's tables: their scalars went into the two vectors, the MSM gives L_j and R_j whole
        PLK_TRY(msm_execute_dev_impl(c->lead_ctx, 2, sL, c->lead_n, out_xy, out_z, c->stream, nullptr, nullptr, of));
    } else if (lead) {
        // record 0: <a s, G> from the caller's tables; record 1: [l] H + [<a, b>] U' (one lane each, ~130 doublings: two side
        // streams, mostly hidden behind the MSM); L_j, R_j = the sums of the two records
        uint8_t* r0 = c->rec;
        uint8_t* r1 = c->rec + c->rec_bytes;
        PLK_HIP_TRY(hipEventRecord(c->ev_main, c->stream));
        PLK_HIP_TRY(hipStreamWaitEvent(c->side, c->ev_main, 0));
        PLK_HIP_TRY(hipStreamWaitEvent(c->side2, c->ev_main, 0));
        PLK_TRY(curve_fold_pairs_dev_impl(c->curve, 1, c->extra, nullptr, c->extra + pt, nullptr, nullptr, nullptr, r1, r1 + 2 * pt, c->side, c->hu, 0));
        PLK_HIP_TRY(hipEventRecord(c->ev_side, c->side));
        PLK_TRY(curve_fold_pairs_dev_impl(c->curve, 1, c->extra, nullptr, c->extra + pt, nullptr, nullptr, nullptr, r1 + pt, r1 + 2 * pt + 1, c->side2, c->hu + 64, 0));
        PLK_HIP_TRY(hipEventRecord(c->ev_side2, c->side2));
        PLK_TRY(msm_execute_dev_impl(c->lead_ctx, 2, sL, c->lead_n, r0, r0 + 2 * pt, c->stream));
        PLK_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_side, 0));
        PLK_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_side2, 0));
        PLK_TRY(msm_combine_partials_dev_impl(c->curve, 2, 2, 0, c->rec, out_xy, out_z, c->stream));
    } else if (c->frozen) {
        PLK_TRY(msm_execute_dev_impl(c->mT, 2, sL, c->m0 + 2, out_xy, out_z, c->stream, nullptr, nullptr, of));
    } else {
        // L on the caller's stream, R on the side stream: below ~2^16 points each is a dependency chain, not throughput.
        // Single round: L over the upper half of the generators, R over the lower.  Stage of virtual rounds: both over the whole set
        // (the scalars of the generators on the other side are zero - a zero scalar has no digits).
        const size_t pts = virt ? c->m0 : m;
        PLK_HIP_TRY(hipEventRecord(c->ev_main, c->stream));
        PLK_HIP_TRY(hipStreamWaitEvent(c->side, c->ev_main, 0));
        PLK_TRY(msm_rebind_dev_impl(c->mR, pts + 2, c->g, c->gz, c->extra, 2, c->side));
        PLK_TRY(msm_execute_dev_impl(c->mR, 1, sR, pts + 2, out_xy + ps, out_z + 1, c->side, nullptr, nullptr, of));
        PLK_HIP_TRY(hipEventRecord(c->ev_side, c->side));
        PLK_TRY(msm_rebind_dev_impl(c->mL, pts + 2, virt ? c->g : c->g + m * pt, virt ? c->gz : c->gz + m, c->extra, 2, c->stream));
        PLK_TRY(msm_execute_dev_impl(c->mL, 1, sL, pts + 2, out_xy, out_z, c->stream, nullptr, nullptr, of));
        PLK_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_side, 0));
    }
    PLK_HIP_TRY(hipMemcpyAsync(c->pin, c->out, 2 * ps + 2, hipMemcpyDeviceToHost, c->stream));
    PLK_HIP_TRY(hipStreamSynchronize(c->stream));
    if (proj) {
        if (host_projective_to_affine(c->curve, 2, c->pin, c->pin + 2 * ps, (uint8_t*)lr_xy) != 0) return set_error(PLK_ERR_INVALID_ARG, "unknown curve %d", c->curve);
    } else {
        memcpy(lr_xy, c->pin, 2 * pt);
    }
    lr_zero[0] = c->pin[2 * ps];
    lr_zero[1] = c->pin[2 * ps + 1];
    c->lr_done = true;
    return PLK_OK;
}

int halo_round_fold_impl(plk_halo_ctx* c, const uint64_t* u_j, const uint64_t* u_j_inv) {
    if (!c || !u_j || !u_j_inv) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    PLK_TRY(ensure_device());
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->poisoned) return set_error(PLK_ERR_INVALID_ARG, "the argument's context is unusable: an earlier fold failed half way");
    if (c->n < 2) return set_error(PLK_ERR_INVALID_ARG, "the argument is finished (length %zu)", c->n);
    const size_t m = c->n / 2, pt = (size_t)2 * c->L * 8;
    const bool virt = c->virt_left > 0;
    struct Poison {  // armed while the scalars are folded and the generators are not
        plk_halo_ctx* c;
        bool armed;
        ~Poison() {
            if (armed) c->poisoned = true;
        }
    } poison{c, false};
    const size_t m0 = (c->frozen || virt) ? c->m0 : 0;
    const size_t work = m > m0 ? m : m0;
    HALO_FIELD_SWITCH(c->sfield, (k_halo_fold_scalars<P><<<(unsigned)((work + 255) / 256), 256, 0, c->stream>>>(
                                     (uint4*)c->a, (uint4*)c->b, m, to_halo_scalar(u_j), to_halo_scalar(u_j_inv), (uint4*)c->coef, m0,
                                     (uint4*)(c->frozen ? nullptr : c->dsc))));
    PLK_HIP_TRY(hipGetLastError());
    c->n = m;
    c->lr_done = false;
    poison.armed = true;
    if (virt) {
        HALO_FIELD_SWITCH(c->sfield, (k_halo_lead_ratios<P><<<1, 64, 0, c->stream>>>((uint4*)c->ratios, (int)(c->virt_total - c->virt_left), (const uint4*)c->dsc)));
        PLK_HIP_TRY(hipGetLastError());
        if (--c->virt_left == 0) {
            // the generators of the stage's rounds at once, scaled like the pairwise folds: G = [c] G~, c = c_0 prod u_k^-1 (dsc[2])
            PLK_TRY(curve_fold_multi_dev_impl(c->curve, m, (int)c->virt_total, c->g, c->gz, c->ratios, c->g, c->gz, c->stream));
            c->m0 = 0;
            c->lead_ctx = nullptr;
            PLK_TRY(halo_next_stage(c));
        }
    } else if (!c->frozen) {
        // G'_i = [u^-1] G_lo_i + [u] G_hi_i = [u^-1] (G_lo_i + [u^2] G_hi_i): the scaled fold G~'_i = G~_lo_i + [u^2] G~_hi_i in place
        // (pair i only touches elements i and m + i); u^2 and the new scale c u^-1 were just written to dsc
        // (long vectors: the 2-to-1 case of the streaming kernel where the curve has the endomorphism - 3 waves per SIMD instead of 1:
        // 6.8 against 7.8 ms for 2^19 pairs, 1.93 against 2.01 for 2^17; below that its three launches are the longer chain: 1.73
        // against 1.07 ms for 2^16 pairs.  Its scalar sits at index bitreverse(1) = 1 of dsc)
        static const bool pair_kernel = getenv("PLK_HALO_PAIR_FOLD") != nullptr;
        if (c->endo && !pair_kernel && m >= ((size_t)1 << 17))
            PLK_TRY(curve_fold_multi_dev_impl(c->curve, m, 1, c->g, c->gz, c->dsc, c->g, c->gz, c->stream));
        else
            PLK_TRY(curve_fold_pairs_dev_impl(c->curve, m, c->g, c->gz, c->g + m * pt, c->gz + m, nullptr, nullptr, c->g, c->gz, c->stream, c->dsc, 1));
        PLK_TRY(halo_next_stage(c));
    }
    poison.armed = false;
    return PLK_OK;
}

size_t halo_len_impl(const plk_halo_ctx* c) { return c ? c->n : 0; }
int halo_frozen_impl(const plk_halo_ctx* c) { return c && (c->frozen || c->virt_left > 0) ? 1 : 0; }

// current halo_a, halo_b (n scalars each) and - while the generators are still folded explicitly - halo_g (n points + flags);
// with n == 1 and frozen generators the single generator is the MSM <s, G^(f)>
int halo_read_impl(plk_halo_ctx* c, uint64_t* a, uint64_t* b, uint64_t* g_xy, uint8_t* g_zero) {
    if (!c) return set_error(PLK_ERR_INVALID_ARG, "null context");
    PLK_TRY(ensure_device());
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->poisoned) return set_error(PLK_ERR_INVALID_ARG, "the argument's context is unusable: an earlier fold failed half way");
    const size_t pt = (size_t)2 * c->L * 8;
    if (a) PLK_HIP_TRY(hipMemcpyAsync(a, c->a, c->n * 32, hipMemcpyDeviceToHost, c->stream));
    if (b) PLK_HIP_TRY(hipMemcpyAsync(b, c->b, c->n * 32, hipMemcpyDeviceToHost, c->stream));
    if (g_xy || g_zero) {
        if (!g_xy || !g_zero) return set_error(PLK_ERR_INVALID_ARG, "g_xy and g_zero go together");
        if (c->virt_left > 0) {
            return set_error(PLK_ERR_INVALID_ARG, "the generators are untouched during a stage of %u virtual rounds (%u to go): halo_g exists again after it",
                             c->virt_total, c->virt_left);
        } else if (!c->frozen) {
            // halo_g_i = [c] G~_i: the pairwise kernel with the scalars (c, 0) over (G~_i, G~_i)
            uint8_t* tmp = (uint8_t*)scratch_acquire(c->n * (pt + 1), c->stream);
            if (!tmp) return PLK_ERR_OOM;
            const int rc = curve_fold_pairs_dev_impl(c->curve, c->n, c->g, c->gz, c->g, c->gz, nullptr, nullptr, tmp, tmp + c->n * pt, c->stream, c->dsc + 2 * 32, 0);
            hipError_t e1 = hipSuccess, e2 = hipSuccess;
            if (rc == PLK_OK) {
                e1 = hipMemcpyAsync(g_xy, tmp, c->n * pt, hipMemcpyDeviceToHost, c->stream);
                e2 = hipMemcpyAsync(g_zero, tmp + c->n * pt, c->n, hipMemcpyDeviceToHost, c->stream);
            }
            (void)hipStreamSynchronize(c->stream);
            scratch_release(tmp, c->stream);
            PLK_TRY(rc);
            PLK_HIP_TRY(e1);
            PLK_HIP_TRY(e2);
        } else if (c->n == 1) {
            // halo_g[0] = sum_j s_j G^(f)_j (scalars of H and U': zero)
            PLK_HIP_TRY(hipMemcpyAsync(c->scal, c->coef, c->m0 * 32, hipMemcpyDeviceToDevice, c->stream));
            PLK_HIP_TRY(hipMemsetAsync(c->scal + c->m0 * 32, 0, 64, c->stream));
            PLK_TRY(msm_execute_dev_impl(c->mT, 1, c->scal, c->m0 + 2, c->out, c->out + 2 * pt, c->stream));
            PLK_HIP_TRY(hipMemcpyAsync(c->pin, c->out, 2 * pt + 2, hipMemcpyDeviceToHost, c->stream));
            PLK_HIP_TRY(hipStreamSynchronize(c->stream));
            memcpy(g_xy, c->pin, pt);
            g_zero[0] = c->pin[2 * pt];
        } else {
            return set_error(PLK_ERR_INVALID_ARG, "the generators are frozen (length %zu <= 2^%u): halo_g exists again when the argument is finished", c->n,
                             c->freeze_log);
        }
    }
    PLK_HIP_TRY(hipStreamSynchronize(c->stream));
    return PLK_OK;
}

void halo_delete(plk_halo_ctx* c) { delete c; }

}  // namespace plk
