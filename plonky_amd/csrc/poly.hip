// poly.hip -- the polynomial callers either side of the NTT, on the device.
//
// Reference path                                                     here
//   Polynomial::divide_by_z_h            polynomial.rs:330-380  ->  poly_divide_by_z_h_dev_impl
//   Polynomial::mul                      polynomial.rs:208-226  ->  poly_mul_dev_impl
//   polynomials_to_values_padded         plonk_util.rs:179-190  ->  ntt_padded_dev_impl (8x zero padding + NTT, batched)
//   Polynomial::degree / trim            polynomial.rs:99-113,178-180 -> k_poly_degree
//
// The reference makes 5 full sweeps around the two transforms of divide_by_z_h (scale by g^i, the
// denominators, their batch inversion, the pointwise product, scale by g^-i).  Here all of them ride
// on the loads of the first NTT pass and the stores of the last one (NttHooks, ntt.hip): the coset
// factor g^i comes from a two-level geometric table (1024 + size/1024 entries, L2 resident), and the
// inverse denominators 1 / (g^n w^(n i) - 1) are periodic in i with period ord(w^n) = size / gcd(n, size)
// (8 in the Plonk prover, where size = 8n), so they are a tiny table indexed by i mod ord.  The data
// makes exactly the HBM round trips of two plain transforms.
//
// Results are the reference's, bit for bit: every output is a fully reduced field element and the
// reference computes the same field values (its result length, 2^ceil(log2(degree + 1)), included).
#include <map>
#include <memory>
#include <mutex>
#include <tuple>

#include "common.h"
#include "fp.cuh"
#include "fz.cuh"
#include "tables.cuh"

namespace plk {

// ---------------------------------------------------------------------------------------------
// degree + 1 (0 for the zero polynomial): Polynomial::degree_plus_one, polynomial.rs:108-113
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_poly_degree(const uint4* __restrict__ a, size_t len, unsigned long long* __restrict__ out) {
    unsigned long long best = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 lo = a[2 * i], hi = a[2 * i + 1];
        if ((lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) != 0) best = i + 1;  // i grows along the loop
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(best, d);
        best = o > best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(out, best);
}

// ---------------------------------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------------------------------
// two-level geometric table of b = g or g^-1 (MULTIPLICATIVE_SUBGROUP_GENERATOR, field.rs:44): lo[j] = b^j,
// hi[k] = b^(1024 k), plus cst[0] = R'/R (the factor that turns an R-form value into R'-form on a hooked store)
template <class P> __global__ void k_geom_fill(int inverse, uint4* lo, uint4* hi, size_t n_hi, uint4* cst) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n_lo = (size_t)1 << NTT_POW_LO_LOG;
    if (idx >= n_lo + n_hi) return;
    const Fe<P> base = inverse ? fe_const<P>(P::GEN_INV) : fe_const<P>(P::GEN);
    if (idx < n_lo) {
        fe_store<P>(lo + idx * 2, to_rprime<P>(fe_pow_u64<P>(base, idx)));
    } else {
        fe_store<P>(hi + (idx - n_lo) * 2, to_rprime<P>(fe_pow_u64<P>(base, (idx - n_lo) << NTT_POW_LO_LOG)));
    }
    if (idx == 0 && cst) fe_store<P>(cst, to_rprime<P>(fz_to_fe_canonical<P>(fz_one_rprime<P>())));
}

// inv[t] = 1 / (g^n w^(n t) - 1), t < ord, R'-form; w = primitive 2^log_size-th root (polynomial.rs:350-361).
// The reference inverts with Montgomery's trick; an inverse is an inverse, so the values agree.
template <class P> __global__ void __launch_bounds__(64) k_zh_inv_table(const uint4* __restrict__ pw, int log_t, int log_size, uint64_t n,
                                                                        size_t ord, uint4* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ord) return;
    const uint64_t size_mask = ((uint64_t)1 << log_size) - 1;
    const uint64_t e = ((n & size_mask) * (uint64_t)t) & size_mask;  // both factors < 2^30: no overflow
    const Fe<P> wn = pow_from_table<P>(pw, 0, e << (log_t - log_size), log_t);
    const Fe<P> gn = fe_pow_u64<P>(fe_const<P>(P::GEN), n);
    const Fe<P> den = fe_sub<P>(fe_mul<P>(gn, wn), fe_one<P>());
    fe_store<P>(out + t * 2, to_rprime<P>(fe_inv_safegcd<P>(den)));
}

struct GeomTables {
    void* lo[2] = {nullptr, nullptr};  // g, g^-1
    void* hi[2] = {nullptr, nullptr};
    void* cst = nullptr;
    ~GeomTables() {
        for (int d = 0; d < 2; ++d) {
            if (lo[d]) (void)hipFree(lo[d]);
            if (hi[d]) (void)hipFree(hi[d]);
        }
        if (cst) (void)hipFree(cst);
    }
};
struct ZhTable {
    void* inv = nullptr;
    size_t ord = 0;
    ~ZhTable() {
        if (inv) (void)hipFree(inv);
    }
};

static std::mutex g_poly_mu;
static std::map<std::tuple<int, int, int>, std::shared_ptr<GeomTables>> g_geom;          // (device, field, log_size)
static std::map<std::tuple<int, int, int, uint64_t>, std::shared_ptr<ZhTable>> g_zh;     // (device, field, log_size, n)
constexpr size_t ZH_CACHE_MAX_ORD = (size_t)1 << 16;  // larger tables are rebuilt per call in scratch memory
constexpr size_t ZH_CACHE_MAX_ENTRIES = 64;

int poly_clear_cache_impl() {
    std::lock_guard<std::mutex> lk(g_poly_mu);
    g_geom.clear();
    g_zh.clear();
    return PLK_OK;
}

template <class P> static int get_geom_t(int dev, int log_size, hipStream_t stream, std::shared_ptr<GeomTables>& out) {
    std::lock_guard<std::mutex> lk(g_poly_mu);
    const auto key = std::make_tuple(dev, (int)P::FIELD_ID, log_size);
    auto it = g_geom.find(key);
    if (it != g_geom.end()) {
        out = it->second;
        return PLK_OK;
    }
    auto gt = std::make_shared<GeomTables>();
    const size_t n_lo = (size_t)1 << NTT_POW_LO_LOG;
    const size_t n_hi = log_size > NTT_POW_LO_LOG ? (size_t)1 << (log_size - NTT_POW_LO_LOG) : 1;
    PLK_HIP_TRY(hipMalloc(&gt->cst, 32));
    for (int d = 0; d < 2; ++d) {
        PLK_HIP_TRY(hipMalloc(&gt->lo[d], n_lo * 32));
        PLK_HIP_TRY(hipMalloc(&gt->hi[d], n_hi * 32));
        k_geom_fill<P><<<(unsigned)((n_lo + n_hi + 127) / 128), 128, 0, stream>>>(d, (uint4*)gt->lo[d], (uint4*)gt->hi[d], n_hi,
                                                                                  d == 0 ? (uint4*)gt->cst : nullptr);
        PLK_HIP_TRY(hipGetLastError());
    }
    // other streams may pick the tables out of the cache right away
    PLK_HIP_TRY(hipStreamSynchronize(stream));
    g_geom[key] = gt;
    out = gt;
    return PLK_OK;
}

static int read_degrees(const void* const* d_polys, const size_t* lens, int count, size_t* deg_plus_one, hipStream_t stream) {
    unsigned long long* d_deg = (unsigned long long*)scratch_acquire(sizeof(unsigned long long) * count, stream);
    if (!d_deg) return PLK_ERR_OOM;
    int rc = PLK_OK;
    unsigned long long h[4] = {0, 0, 0, 0};
    do {
        hipError_t e = hipMemsetAsync(d_deg, 0, sizeof(unsigned long long) * count, stream);
        for (int k = 0; k < count && e == hipSuccess; ++k) {
            if (lens[k] == 0) continue;
            const size_t blocks = (lens[k] + 255) / 256;
            k_poly_degree<<<(unsigned)(blocks < 2048 ? blocks : 2048), 256, 0, stream>>>((const uint4*)d_polys[k], lens[k], d_deg + k);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h, d_deg, sizeof(unsigned long long) * count, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = set_error(PLK_ERR_HIP, "degree scan failed: %s", hipGetErrorString(e));
    } while (0);
    scratch_release(d_deg, stream);
    for (int k = 0; k < count; ++k) deg_plus_one[k] = (size_t)h[k];
    return rc;
}

static inline int log2_ceil_sz(size_t v) {  // util.rs:2-9
    int l = 0;
    while (((size_t)1 << l) < v) ++l;
    return l;
}

// ---------------------------------------------------------------------------------------------
// Polynomial::divide_by_z_h
// ---------------------------------------------------------------------------------------------
template <class P>
static int divide_by_z_h_t(const void* d_coeffs, size_t len, uint64_t n, void* d_out, size_t out_cap, size_t* out_len, hipStream_t stream) {
    int dev = 0;
    PLK_HIP_TRY(hipGetDevice(&dev));
    size_t dp1 = 0;
    PLK_TRY(read_degrees(&d_coeffs, &len, 1, &dp1, stream));
    if (dp1 == 0) {  // the zero polynomial is returned as it came, untrimmed (polynomial.rs:331-333)
        if (out_cap < len) return set_error(PLK_ERR_INVALID_ARG, "output capacity %zu < %zu", out_cap, len);
        if (len && d_out != d_coeffs) PLK_HIP_TRY(hipMemcpyAsync(d_out, d_coeffs, len * 32, hipMemcpyDeviceToDevice, stream));
        *out_len = len;
        return PLK_OK;
    }
    const int log_size = log2_ceil_sz(dp1);
    const size_t size = (size_t)1 << log_size;
    if (log_size > P::TWO_ADICITY || log_size > 30)
        return set_error(PLK_ERR_TWO_ADICITY, "degree %zu needs a 2^%d domain: beyond the field's 2-adicity", dp1 - 1, log_size);
    if (out_cap < size) return set_error(PLK_ERR_INVALID_ARG, "output capacity %zu < %zu", out_cap, size);

    std::shared_ptr<GeomTables> gt;
    PLK_TRY(get_geom_t<P>(dev, log_size, stream, gt));

    // ord(w^n) = size / gcd(n, size)
    size_t g2 = size;
    while (g2 > 1 && (n & (g2 - 1)) != 0) g2 >>= 1;
    const size_t ord = size / g2;
    const void* pw = nullptr;
    int log_t = 0;
    std::shared_ptr<const void> plan_hold;  // the plan outlives the launches below (hipFree waits for kernels in flight)
    PLK_TRY(ntt_plan_pow_table(P::FIELD_ID, (unsigned)log_size, &pw, &log_t, &plan_hold));
    std::shared_ptr<ZhTable> zh;
    void* zh_scratch = nullptr;
    const void* inv_tab = nullptr;
    if (ord <= ZH_CACHE_MAX_ORD) {
        std::lock_guard<std::mutex> lk(g_poly_mu);
        const auto key = std::make_tuple(dev, (int)P::FIELD_ID, log_size, n);
        auto it = g_zh.find(key);
        if (it != g_zh.end()) {
            zh = it->second;
        } else {
            zh = std::make_shared<ZhTable>();
            zh->ord = ord;
            PLK_HIP_TRY(hipMalloc(&zh->inv, ord * 32));
            k_zh_inv_table<P><<<(unsigned)((ord + 63) / 64), 64, 0, stream>>>((const uint4*)pw, log_t, log_size, n, ord, (uint4*)zh->inv);
            PLK_HIP_TRY(hipGetLastError());
            PLK_HIP_TRY(hipStreamSynchronize(stream));
            if (g_zh.size() >= ZH_CACHE_MAX_ENTRIES) g_zh.erase(g_zh.begin());
            g_zh[key] = zh;
        }
        inv_tab = zh->inv;
    } else {
        zh_scratch = scratch_acquire(ord * 32, stream);
        if (!zh_scratch) return PLK_ERR_OOM;
        k_zh_inv_table<P><<<(unsigned)((ord + 63) / 64), 64, 0, stream>>>((const uint4*)pw, log_t, log_size, n, ord, (uint4*)zh_scratch);
        inv_tab = zh_scratch;
    }

    // a(g w^i) / (g^n w^(n i) - 1): forward transform of a_i g^i, denominators on the way out
    NttHooks fw;
    fw.in_len = dp1;
    fw.in_stride = dp1;
    fw.in_lo = gt->lo[0];
    fw.in_hi = gt->hi[0];
    fw.out_tab = inv_tab;
    fw.out_mask = ord - 1;
    int rc = ntt_dev_hooked_impl(P::FIELD_ID, (unsigned)log_size, 0, 1, d_coeffs, d_out, fw, stream);
    if (rc == PLK_OK) {
        // interpolate on {w^i}, then p_i g^-i
        NttHooks bw;
        bw.in_len = size;
        bw.in_stride = size;
        bw.out_lo = gt->lo[1];
        bw.out_hi = gt->hi[1];
        rc = ntt_dev_hooked_impl(P::FIELD_ID, (unsigned)log_size, 1, 1, d_out, d_out, bw, stream);
    }
    if (zh_scratch) scratch_release(zh_scratch, stream);
    if (rc == PLK_OK) *out_len = size;
    return rc;
}

int poly_divide_by_z_h_dev_impl(int field, const void* d_coeffs, size_t len, size_t n, void* d_out, size_t out_cap, size_t* out_len,
                                hipStream_t stream) {
    if (!out_len) return set_error(PLK_ERR_INVALID_ARG, "null out_len");
    if ((len && !d_coeffs) || (out_cap && !d_out)) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    if (n == 0) return set_error(PLK_ERR_INVALID_ARG, "Z_H = X^0 - 1 is zero");
    PLK_TRY(ensure_device());
    switch (field) {
        case PLK_FIELD_TWEEDLEDEE_BASE: return divide_by_z_h_t<TweedledeeBaseParams>(d_coeffs, len, n, d_out, out_cap, out_len, stream);
        case PLK_FIELD_TWEEDLEDUM_BASE: return divide_by_z_h_t<TweedledumBaseParams>(d_coeffs, len, n, d_out, out_cap, out_len, stream);
        case PLK_FIELD_BLS12_377_SCALAR: return divide_by_z_h_t<Bls12377ScalarParams>(d_coeffs, len, n, d_out, out_cap, out_len, stream);
        case PLK_FIELD_PALLAS_BASE: return divide_by_z_h_t<PallasBaseParams>(d_coeffs, len, n, d_out, out_cap, out_len, stream);
        case PLK_FIELD_VESTA_BASE: return divide_by_z_h_t<VestaBaseParams>(d_coeffs, len, n, d_out, out_cap, out_len, stream);
    }
    return set_error(PLK_ERR_INVALID_ARG, "field %d has no NTT entry point", field);
}

// ---------------------------------------------------------------------------------------------
// Polynomial::mul
// ---------------------------------------------------------------------------------------------
template <class P>
static int poly_mul_t(const void* d_a, size_t la, const void* d_b, size_t lb, void* d_out, size_t out_cap, size_t* out_len, hipStream_t stream) {
    int dev = 0;
    PLK_HIP_TRY(hipGetDevice(&dev));
    const void* polys[2] = {d_a, d_b};
    const size_t lens[2] = {la, lb};
    size_t dp1[2] = {0, 0};
    PLK_TRY(read_degrees(polys, lens, 2, dp1, stream));
    if (dp1[0] == 0 || dp1[1] == 0) {  // Polynomial::zero(1), polynomial.rs:209-211
        if (out_cap < 1) return set_error(PLK_ERR_INVALID_ARG, "output capacity 0");
        PLK_HIP_TRY(hipMemsetAsync(d_out, 0, 32, stream));
        *out_len = 1;
        return PLK_OK;
    }
    const size_t prod_len = dp1[0] + dp1[1] - 1;  // a_deg + b_deg + 1
    const int log_size = log2_ceil_sz(prod_len);
    const size_t size = (size_t)1 << log_size;
    if (log_size > P::TWO_ADICITY || log_size > 30) return set_error(PLK_ERR_TWO_ADICITY, "product needs a 2^%d domain", log_size);
    if (out_cap < size) return set_error(PLK_ERR_INVALID_ARG, "output capacity %zu < %zu", out_cap, size);
    std::shared_ptr<GeomTables> gt;
    PLK_TRY(get_geom_t<P>(dev, log_size, stream, gt));
    void* ev_a = scratch_acquire(size * 32, stream);
    if (!ev_a) return PLK_ERR_OOM;
    // evaluations of a, left in R'-form so that they can be the multiplier table of b's transform
    NttHooks ha;
    ha.in_len = dp1[0];
    ha.in_stride = dp1[0];
    ha.out_tab = gt->cst;
    ha.out_mask = 0;
    int rc = ntt_dev_hooked_impl(P::FIELD_ID, (unsigned)log_size, 0, 1, d_a, ev_a, ha, stream);
    if (rc == PLK_OK) {
        NttHooks hb;
        hb.in_len = dp1[1];
        hb.in_stride = dp1[1];
        hb.out_tab = ev_a;
        hb.out_mask = size - 1;
        rc = ntt_dev_hooked_impl(P::FIELD_ID, (unsigned)log_size, 0, 1, d_b, d_out, hb, stream);
    }
    if (rc == PLK_OK) rc = ntt_dev_impl(P::FIELD_ID, (unsigned)log_size, 1, 1, d_out, d_out, stream);
    scratch_release(ev_a, stream);
    if (rc == PLK_OK) *out_len = size;
    return rc;
}

int poly_mul_dev_impl(int field, const void* d_a, size_t la, const void* d_b, size_t lb, void* d_out, size_t out_cap, size_t* out_len,
                      hipStream_t stream) {
    if (!out_len) return set_error(PLK_ERR_INVALID_ARG, "null out_len");
    if ((la && !d_a) || (lb && !d_b) || !d_out) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    switch (field) {
        case PLK_FIELD_TWEEDLEDEE_BASE: return poly_mul_t<TweedledeeBaseParams>(d_a, la, d_b, lb, d_out, out_cap, out_len, stream);
        case PLK_FIELD_TWEEDLEDUM_BASE: return poly_mul_t<TweedledumBaseParams>(d_a, la, d_b, lb, d_out, out_cap, out_len, stream);
        case PLK_FIELD_BLS12_377_SCALAR: return poly_mul_t<Bls12377ScalarParams>(d_a, la, d_b, lb, d_out, out_cap, out_len, stream);
        case PLK_FIELD_PALLAS_BASE: return poly_mul_t<PallasBaseParams>(d_a, la, d_b, lb, d_out, out_cap, out_len, stream);
        case PLK_FIELD_VESTA_BASE: return poly_mul_t<VestaBaseParams>(d_a, la, d_b, lb, d_out, out_cap, out_len, stream);
    }
    return set_error(PLK_ERR_INVALID_ARG, "field %d has no NTT entry point", field);
}

// ---------------------------------------------------------------------------------------------
// zero-padded forward transforms: eval_domain of padded polynomials (polynomial.rs:135-143, plonk_util.rs:179-190)
// ---------------------------------------------------------------------------------------------
int ntt_padded_dev_impl(int field, unsigned log_n, unsigned batch, const void* d_in, size_t in_len, size_t in_stride, void* d_out,
                        hipStream_t stream) {
    if (log_n > 30) return set_error(PLK_ERR_TWO_ADICITY, "log_n %u too large (max 30)", log_n);
    if (in_len > ((size_t)1 << log_n)) return set_error(PLK_ERR_INVALID_ARG, "in_len %zu exceeds 2^%u", in_len, log_n);
    if (batch > 1 && in_stride < in_len) return set_error(PLK_ERR_INVALID_ARG, "in_stride %zu < in_len %zu", in_stride, in_len);
    if (!d_out || (in_len && !d_in)) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    NttHooks h;
    h.in_len = in_len;
    h.in_stride = in_stride;
    return ntt_dev_hooked_impl(field, log_n, 0, batch, in_len ? d_in : d_out, d_out, h, stream);
}

}  // namespace plk
