// comb.hip -- small fixed-base MSMs WITHOUT buckets (round 4).  AN EXPERIMENT KEPT BEHIND PLK_MSM_COMB=1: it is bit-exact (its own parity
// test, tests/test_gpu_parity.py::test_msm_comb_small_contexts) but only faster than the bucket method up to ~2^12 generators and not
// at the 2^14 + 2 frozen generators it was built for - numbers in profiles/r04_comb_small_msm.txt, discussion in DESIGN.md section 5.8.
//
// The bucket method (msm.hip) pays a fixed ~0.45 ms per execution whatever the size: ordering, then a reduction that is a chain of point
// operations on few points.  For a context over few generators that is used many times - the frozen generators of an inner-product
// argument (halo.hip: fourteen rounds of two 2^14-term MSMs each, 0.48 ms a round), small commitment keys - the table can afford to hold
// every MULTIPLE a digit can ask for: tab[(j n + i) 8 + (k - 1)] = [k 2^(4 j)] G_i, k = 1 .. 8 (signed 4-bit digits by carry recoding,
// curve_msm.rs:159-180 restated with signs; 64 windows; 32 KB per generator, 0.5 GB for 2^14 of the 288 GB).  An execution is then
// NOTHING BUT mixed additions of table entries - sum_i sum_j +-tab[i][j][|d_ij|] - and a tree over the lanes' partial sums:
//   k_comb_accumulate  eight lanes per scalar, eight windows each: <= 8 mixed additions (ecz.cuh, the accumulation's own step), then the
//                      wave's 64 partial sums are added by shuffles: one point per wave;
//   k_comb_reduce      one workgroup per scalar vector adds the waves' points and normalises (to_affine, curve.rs:206-214).
// No ordering, no buckets, no doublings: two launches.  Same group element as the bucket method, hence the same unique affine result.
// Construction (msm_precompute, excluded from every timed region like the window tables): the doubling chain [2^(4 j)] G_i on quads
// (k_msm_table_chain, msm.hip), then one lane per (window, generator): W, 2W .. 8W by four doublings and three additions, ONE inversion
// for the eight (Montgomery's trick over ZZZ), stored affine in R'-form.
#include <cstdlib>

#include "common.h"
#include "ec.cuh"
#include "ecz.cuh"
#include "field_params.cuh"
#include "tables.cuh"

namespace plk {

constexpr int COMB_C = 4;                  // digit bits
constexpr int COMB_MULT = 1 << (COMB_C - 1);  // multiples per window: 1 .. 8
constexpr int COMB_WPL = 8;                // windows per lane
constexpr int COMB_ACC_THREADS = 256;
constexpr int COMB_RED_THREADS = 1024;

struct CombPlan {
    int curve = 0, device = 0;
    size_t n = 0;
    int windows = 0, groups = 0;  // windows = ceil((BITS + 1) / 4); groups = ceil(windows / COMB_WPL) lanes per scalar
    void* tab = nullptr;          // n * windows * 8 affine entries
    size_t tab_bytes = 0;
};

template <class FP> constexpr int comb_raw_u4() { return FzCfg<FP>::NZ; }
template <class FP> PLK_DI XyzzZ<FP> comb_load_raw(const uint4* src) {
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
    r.inf = any == 0;
    return r;
}
template <class FP> PLK_DI void comb_store_raw(uint4* dst, const XyzzZ<FP>& a) {
    constexpr int NZ = FzCfg<FP>::NZ;
    uint32_t w[4 * NZ];
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        w[i] = a.x.l[i];
        w[NZ + i] = a.y.l[i];
        w[2 * NZ + i] = a.inf ? 0u : a.zz.l[i];
        w[3 * NZ + i] = a.zzz.l[i];
    }
#pragma unroll
    for (int i = 0; i < NZ; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// ---- table construction ----
// base0: the n generators, affine, R'-form (window 0); chain: [2^(4 j)] G_i for j >= 1 as raw XYZZ points at (j - 1) n + i (what
// k_msm_table_chain leaves behind); scratch: 8 raw points per lane.  One lane per (window j, generator i).
template <class C>
__global__ void __launch_bounds__(128) k_comb_multiples(const uint4* __restrict__ base0, const uint4* __restrict__ chain, uint4* __restrict__ scratch,
                                                        uint4* __restrict__ tab, size_t n, int windows, size_t lane0, size_t lane1) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    constexpr int RU = comb_raw_u4<FP>();
    const size_t t = lane0 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // this launch covers the lanes lane0 .. lane1 - 1
    if (t >= lane1) return;
    const size_t j = t / n, i = t - j * n;
    XyzzZ<FP> p1;
    if (j == 0) {
        Fe<FP> x, y;
        const bool ident = affine_load<FP>(base0 + i * 2 * W, x, y);
        p1 = xyzzz_identity<FP>();
        if (!ident) {
            p1.x = fz_from_fe<FP>(x);
            p1.y = fz_from_fe<FP>(y);
            p1.zz = fz_one_rprime<FP>();
            p1.zzz = p1.zz;
            p1.inf = false;
        }
    } else {
        p1 = comb_load_raw<FP>(chain + ((j - 1) * n + i) * RU);
    }
    uint4* mine = scratch + (t - lane0) * (size_t)COMB_MULT * RU;
    // W, 2W .. 8W (the group law of ecz.cuh with all its exceptional cases: a 2-torsion generator makes 2W the identity)
    const XyzzZ<FP> p2 = xyzzz_dbl<FP>(p1);
    const XyzzZ<FP> p3 = xyzzz_add<FP>(p2, p1);
    const XyzzZ<FP> p4 = xyzzz_dbl<FP>(p2);
    comb_store_raw<FP>(mine + 0 * RU, p1);
    comb_store_raw<FP>(mine + 1 * RU, p2);
    comb_store_raw<FP>(mine + 2 * RU, p3);
    comb_store_raw<FP>(mine + 3 * RU, p4);
    const XyzzZ<FP> p5 = xyzzz_add<FP>(p4, p1);
    const XyzzZ<FP> p6 = xyzzz_dbl<FP>(p3);
    comb_store_raw<FP>(mine + 4 * RU, p5);
    comb_store_raw<FP>(mine + 5 * RU, p6);
    const XyzzZ<FP> p7 = xyzzz_add<FP>(p6, p1);
    const XyzzZ<FP> p8 = xyzzz_dbl<FP>(p4);
    comb_store_raw<FP>(mine + 6 * RU, p7);
    comb_store_raw<FP>(mine + 7 * RU, p8);
    // Montgomery's trick over the eight ZZZ (an identity counts as 1): prefix products, one inversion, back substitution
    const Fz<FP> one = fz_one_rprime<FP>();
    Fz<FP> pre[COMB_MULT];
    {
        const XyzzZ<FP>* ps[COMB_MULT] = {&p1, &p2, &p3, &p4, &p5, &p6, &p7, &p8};
        Fz<FP> run = one;
#pragma unroll
        for (int k = 0; k < COMB_MULT; ++k) {
            pre[k] = run;  // product of the ZZZ before k
            run = fz_mul<FP>(run, ps[k]->inf ? one : ps[k]->zzz);
        }
        // run = product of all: invert it in the reference's form, bring it back to R'-form limbs (k_msm_table_norm, msm.hip)
        const Fe<FP> all_r = fz_to_fe_canonical<FP>(fz_mul<FP>(run, fz_const_rprime_to_r<FP>()));
        Fz<FP> inv = fz_from_fe<FP>(to_rprime<FP>(fe_inv_safegcd<FP>(all_r)));
        for (int k = COMB_MULT - 1; k >= 0; --k) {
            const XyzzZ<FP> p = comb_load_raw<FP>(mine + k * RU);
            const Fz<FP> i3 = fz_mul<FP>(inv, pre[k]);                 // 1 / ZZZ_k
            inv = fz_mul<FP>(inv, p.inf ? one : p.zzz);               // drop ZZZ_k from the running inverse
            Fe<FP> xr = fe_zero<FP>(), yr = fe_zero<FP>();
            if (!p.inf) {
                const Fz<FP> iz = fz_mul<FP>(p.zz, i3);                 // 1 / Z = ZZ / ZZZ
                const Fz<FP> izz = fz_sqr<FP>(iz);
                xr = fz_to_fe_canonical<FP>(fz_mul<FP>(p.x, izz));
                yr = fz_to_fe_canonical<FP>(fz_mul<FP>(p.y, i3));
            }
            affine_store<FP>(tab + (t * COMB_MULT + (size_t)k) * 2 * W, xr, yr, p.inf);
        }
    }
}

// ---- execution ----
struct CombVec {
    const uint4* scalars;  // count scalars (Montgomery form, scalar field)
    uint64_t first;        // they belong to the generators first .. first + count - 1
    uint64_t count;
    uint4* out_xy;
    uint8_t* out_zero;
};
constexpr int COMB_MAX_BATCH = 16;
struct CombBatch {
    int count;
    CombVec v[COMB_MAX_BATCH];
};

// Eight lanes per scalar.  A block of 256 lanes serves 32 scalars: the first lane of each scalar's eight converts it (Montgomery ->
// canonical in the SCALAR field, to_digits of curve_msm.rs:159-180) and recodes it into signed 4-bit digits (carry based: valid on
// BLS12-377 G1 whose cofactor is even), parked in LDS; every lane then adds the entries of its eight windows.
template <class C>
__global__ void __launch_bounds__(COMB_ACC_THREADS) k_comb_accumulate(const uint4* __restrict__ tab, size_t n, int windows, int groups, CombBatch cb,
                                                                     uint4* __restrict__ part, uint32_t waves_per_vec) {
    using FP = typename C::FP;
    using SP = typename C::SP;
    constexpr int W = FP::NL / 4;
    constexpr int SPB = COMB_ACC_THREADS / 8;  // scalars per block (groups <= 8)
    __shared__ int8_t s_dig[SPB][72];
    const CombVec& cv = cb.v[blockIdx.y];
    const int tid = threadIdx.x, sl = tid >> 3, g = tid & 7;
    const uint64_t i = (uint64_t)blockIdx.x * SPB + sl;
    const bool live = i < cv.count;
    if (live && g == 0) {
        const uint4 lo = cv.scalars[i * 2], hi = cv.scalars[i * 2 + 1];
        Fe<SP> s;
        s.v[0] = lo.x; s.v[1] = lo.y; s.v[2] = lo.z; s.v[3] = lo.w;
        s.v[4] = hi.x; s.v[5] = hi.y; s.v[6] = hi.z; s.v[7] = hi.w;
        s = fe_to_canonical<SP>(s);
        uint32_t carry = 0;
        for (int j = 0; j < windows; ++j) {
            const int bp = j * COMB_C, li = bp >> 5, sh = bp & 31;
            const uint32_t v = (li < 8 ? ((s.v[li] >> sh) & 15u) : 0u) + carry;  // 4 | 32: a digit never straddles two words
            const uint32_t neg = v > 8u ? 1u : 0u;
            s_dig[sl][j] = (int8_t)(neg ? (int)v - 16 : (int)v);
            carry = neg;
        }
    }
    __syncthreads();
    XyzzZ<FP> acc = xyzzz_identity<FP>();
    if (live && g < groups) {
        const size_t gen = (size_t)cv.first + (size_t)i;
        const int j0 = g * COMB_WPL, j1 = j0 + COMB_WPL < windows ? j0 + COMB_WPL : windows;
        for (int j = j0; j < j1; ++j) {
            const int d = s_dig[sl][j];
            if (d == 0) continue;
            const int mag = d < 0 ? -d : d;
            Fe<FP> x, y;
            const bool ident = affine_load<FP>(tab + (((size_t)j * n + gen) * COMB_MULT + (size_t)(mag - 1)) * 2 * W, x, y);
            if (ident) continue;
            xyzzz_madd_entry<FP>(acc, x, y, d < 0);
        }
        xyzzz_settle<FP>(acc);
    }
    // the wave's 64 partial sums -> one point per wave
#pragma unroll 1
    for (int m = 1; m < 64; m <<= 1) acc = xyzzz_add<FP>(acc, xyzzz_shfl_xor<FP>(acc, m));
    if ((tid & 63) == 0) {
        const uint32_t wave = blockIdx.x * (COMB_ACC_THREADS / 64) + (tid >> 6);
        xyzzz_store_packed<FP>(part + ((size_t)blockIdx.y * waves_per_vec + wave) * 4 * W, acc);
    }
}

// one workgroup per scalar vector: the waves' points -> one point -> affine
template <class C>
__global__ void __launch_bounds__(COMB_RED_THREADS) k_comb_reduce(const uint4* __restrict__ part, uint32_t waves_per_vec, CombBatch cb) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    __shared__ uint4 s_pts[(COMB_RED_THREADS / 64) * 4 * (FP::NL / 4)];
    const CombVec& cv = cb.v[blockIdx.x];
    const int tid = threadIdx.x;
    const uint4* src = part + (size_t)blockIdx.x * waves_per_vec * 4 * W;
    XyzzZ<FP> acc = xyzzz_identity<FP>();
    for (uint32_t k = tid; k < waves_per_vec; k += COMB_RED_THREADS) acc = xyzzz_add<FP>(acc, xyzzz_load_packed<FP>(src + (size_t)k * 4 * W));
#pragma unroll 1
    for (int m = 1; m < 64; m <<= 1) acc = xyzzz_add<FP>(acc, xyzzz_shfl_xor<FP>(acc, m));
    if ((tid & 63) == 0) xyzzz_store_packed<FP>(s_pts + (tid >> 6) * 4 * W, acc);
    __syncthreads();
    if (tid < 64) {
        acc = tid < COMB_RED_THREADS / 64 ? xyzzz_load_packed<FP>(s_pts + tid * 4 * W) : xyzzz_identity<FP>();
#pragma unroll 1
        for (int m = 1; m < COMB_RED_THREADS / 64; m <<= 1) acc = xyzzz_add<FP>(acc, xyzzz_shfl_xor<FP>(acc, m));
        if (tid == 0) emit_affine<FP, true>(acc, cv.out_xy, cv.out_zero);
    }
}

// ---- host side ----
static int comb_scalar_bits(int curve) { return curve == PLK_CURVE_BLS12_377 ? 253 : 255; }

size_t comb_table_bytes(int curve, size_t n) {
    const int windows = (comb_scalar_bits(curve) + 1 + COMB_C - 1) / COMB_C;
    return n * (size_t)windows * COMB_MULT * (size_t)2 * curve_limbs(curve) * 8 + 16;
}

template <class C>
static int comb_build_t(CombPlan* p, const void* d_base0, const void* d_chain, hipStream_t stream) {
    using FP = typename C::FP;
    const size_t lanes = p->n * (size_t)p->windows;
    if (!lanes) return PLK_OK;
    // in slices of 2^18 lanes: the eight raw multiples of a lane wait in scratch memory (300 MB a slice) between the additions and
    // the shared inversion
    const size_t slice = lanes < ((size_t)1 << 18) ? lanes : ((size_t)1 << 18);
    const size_t sbytes = slice * COMB_MULT * (size_t)comb_raw_u4<FP>() * 16;
    void* scratch = scratch_acquire(sbytes, stream);
    if (!scratch) return PLK_ERR_OOM;
    for (size_t l0 = 0; l0 < lanes; l0 += slice) {
        const size_t l1 = l0 + slice < lanes ? l0 + slice : lanes;
        k_comb_multiples<C><<<(unsigned)((l1 - l0 + 127) / 128), 128, 0, stream>>>((const uint4*)d_base0, (const uint4*)d_chain, (uint4*)scratch, (uint4*)p->tab,
                                                                                    p->n, p->windows, l0, l1);
    }
    const hipError_t e = hipGetLastError();
    scratch_release(scratch, stream);
    if (e != hipSuccess) return set_error(PLK_ERR_HIP, "comb table launch failed: %s", hipGetErrorString(e));
    return PLK_OK;
}

// d_base0: n affine generators in R'-form (what k_msm_table_chain writes as window 0); d_chain: its raw XYZZ points of the windows 1 ..
int comb_build(int curve, size_t n, const void* d_base0, const void* d_chain, hipStream_t stream, CombPlan** out) {
    *out = nullptr;
    CombPlan* p = new CombPlan();
    p->curve = curve;
    p->n = n;
    p->windows = (comb_scalar_bits(curve) + 1 + COMB_C - 1) / COMB_C;
    p->groups = (p->windows + COMB_WPL - 1) / COMB_WPL;
    (void)hipGetDevice(&p->device);
    p->tab_bytes = comb_table_bytes(curve, n);
    if (hipMalloc(&p->tab, p->tab_bytes) != hipSuccess) {
        delete p;
        return set_error(PLK_ERR_OOM, "comb table of %zu bytes", comb_table_bytes(curve, n));
    }
    int rc;
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: rc = comb_build_t<TweedledeeCurve>(p, d_base0, d_chain, stream); break;
        case PLK_CURVE_TWEEDLEDUM: rc = comb_build_t<TweedledumCurve>(p, d_base0, d_chain, stream); break;
        case PLK_CURVE_PALLAS: rc = comb_build_t<PallasCurve>(p, d_base0, d_chain, stream); break;
        case PLK_CURVE_VESTA: rc = comb_build_t<VestaCurve>(p, d_base0, d_chain, stream); break;
        default: rc = comb_build_t<Bls12377Curve>(p, d_base0, d_chain, stream); break;
    }
    if (rc != PLK_OK) {
        (void)hipFree(p->tab);
        delete p;
        return rc;
    }
    *out = p;
    return PLK_OK;
}
void comb_free(CombPlan* p) {
    if (!p) return;
    if (p->tab) (void)hipFree(p->tab);
    delete p;
}
int comb_windows(const CombPlan* p) { return p->windows; }

template <class C>
static int comb_execute_t(const CombPlan* p, const CombBatch& cb, uint64_t max_count, hipStream_t stream) {
    using FP = typename C::FP;
    constexpr int W = FP::NL / 4;
    constexpr int SPB = COMB_ACC_THREADS / 8;
    const unsigned blocks = max_count ? (unsigned)((max_count + SPB - 1) / SPB) : 1u;
    const uint32_t waves_per_vec = blocks * (COMB_ACC_THREADS / 64);
    uint4* part = (uint4*)scratch_acquire((size_t)cb.count * waves_per_vec * 4 * W * 16 + 16, stream);
    if (!part) return PLK_ERR_OOM;
    k_comb_accumulate<C><<<dim3(blocks, cb.count), COMB_ACC_THREADS, 0, stream>>>((const uint4*)p->tab, p->n, p->windows, p->groups, cb, part, waves_per_vec);
    k_comb_reduce<C><<<cb.count, COMB_RED_THREADS, 0, stream>>>(part, waves_per_vec, cb);
    const hipError_t e = hipGetLastError();
    scratch_release(part, stream);
    if (e != hipSuccess) return set_error(PLK_ERR_HIP, "comb launch failed: %s", hipGetErrorString(e));
    return PLK_OK;
}

// `batch` (<= COMB_MAX_BATCH per launch pair) scalar vectors; vector b: count[b] scalars for the generators first[b] ..
int comb_execute(const CombPlan* p, unsigned batch, const void* const* d_scalars, const uint64_t* first, const uint64_t* count, void* d_out_xy, void* d_out_zero,
                 hipStream_t stream) {
    const size_t L = (size_t)curve_limbs(p->curve);
    for (unsigned b0 = 0; b0 < batch; b0 += COMB_MAX_BATCH) {
        CombBatch cb;
        cb.count = (int)(batch - b0 < (unsigned)COMB_MAX_BATCH ? batch - b0 : (unsigned)COMB_MAX_BATCH);
        uint64_t max_count = 0;
        for (int k = 0; k < cb.count; ++k) {
            const unsigned b = b0 + (unsigned)k;
            cb.v[k].scalars = (const uint4*)d_scalars[b];
            cb.v[k].first = first[b];
            cb.v[k].count = count[b];
            cb.v[k].out_xy = (uint4*)((uint8_t*)d_out_xy + (size_t)b * 2 * L * 8);
            cb.v[k].out_zero = (uint8_t*)d_out_zero + b;
            if (count[b] > max_count) max_count = count[b];
        }
        int rc;
        switch (p->curve) {
            case PLK_CURVE_TWEEDLEDEE: rc = comb_execute_t<TweedledeeCurve>(p, cb, max_count, stream); break;
            case PLK_CURVE_TWEEDLEDUM: rc = comb_execute_t<TweedledumCurve>(p, cb, max_count, stream); break;
            case PLK_CURVE_PALLAS: rc = comb_execute_t<PallasCurve>(p, cb, max_count, stream); break;
            case PLK_CURVE_VESTA: rc = comb_execute_t<VestaCurve>(p, cb, max_count, stream); break;
            default: rc = comb_execute_t<Bls12377Curve>(p, cb, max_count, stream); break;
        }
        PLK_TRY(rc);
    }
    return PLK_OK;
}

}  // namespace plk
