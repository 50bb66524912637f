// plonk.hip -- the Plonk quotient numerator on the device.
//
// Reference path                                                              here
//   Prover::vanishing_poly, the 8n-point loop      plonk.rs:392-453      ->  k_vanishing_points
//   evaluate_all_constraints                       gates/mod.rs:46-125   ->  all_constraints()
//   Gate::evaluate_filtered / evaluate_prefix_filter  gates/mod.rs:262-300 -> the shared prefix products below
//   the ten gates' evaluate_unfiltered             gates/*.rs            ->  gate by gate below (file:line at each)
//   eval_l_1                                       plonk_util.rs:14-24   ->  a cached table of L_1 over the 8n domain
//   mds_matrix                                     mds.rs:56-77          ->  seven inverses 1/1 .. 1/7 (Cauchy entries 1/(4 + r - c))
//   reduce_with_powers                             plonk_util.rs:27-33   ->  per gate (ReducedSink), closed in the last launch
// The reference evaluates the 8n points with Rayon, one full field inversion per point for L_1(x) and a freshly
// cloned MDS matrix per gate; here a point is one lane, everything that depends only on the circuit size (the powers
// of the 8n-th root, L_1 over the domain, the MDS entries) is a cached table, and the inversions behind L_1 are
// batched eight to a lane when that table is built.  The closing Polynomial::from_evaluations (plonk.rs:455) is the
// inverse NTT of ntt.hip.  The arithmetic runs on lazily reduced limbs whose value bounds are part of the types (Lz below);
// every operation is exact modulo p and the result is reduced to the unique representative at the end, so it is
// bit-identical to the reference's (field arithmetic is exact and associative; only the grouping differs).
// get_subgroup_shift (partition.rs:140-153) draws its k_i from ChaCha8: they are inputs here.
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>

#include "common.h"
#include "fp.cuh"
#include "tables.cuh"

namespace plk {

constexpr int NUM_WIRES = 9;         // plonk.rs:21
constexpr int NUM_ROUTED_WIRES = 6;  // plonk.rs:22
constexpr int NUM_CONSTANTS = 6;     // plonk.rs:24
constexpr int GRID_WIDTH = 65;       // plonk.rs:25
constexpr int NUM_TERMS = 8;         // the longest gate constraint list (base_4_sum: 1 + 7, rescue_a: 8)
constexpr int XS_LO_LOG = 10;

// Working form of the kernels: lazily reduced 29-bit limbs in R'-form (fz.cuh) with the value bound carried in the TYPE, in
// eighths of p: Lz<P, B> holds a value below (B / 8) p with limbs below 2^29 + 8.  The operators pick the multiple of p a
// subtraction needs, give every result its bound, and refuse at compile time a product whose operand could exceed 15p or a
// sum that could leave the 261 bits of the representation - so the gate formulas below read like the reference's and cannot
// overflow silently.  A product of values below a p and b p comes back below (a b / 127.9 + 1) p (p / R' < 2^-7 (1 + 2^-22)).
constexpr int LZ_MUL_MAX = 120, LZ_VAL_MAX = 1000;
constexpr int lz_sub_k(int b) {
    int k = 0;
    while ((8 << k) - 1 < b) ++k;  // 2^k p must exceed the subtrahend's bound
    return k;
}
constexpr int lz_mul_bound(int a, int b) { return a * b / 1023 + 9; }
template <class P, int B> struct Lz {
    static_assert(FzCfg<P>::NZ == 9 && Mod29<P>::limb(8) <= (1u << 22), "bounds are derived for p < 2^254 (1 + 2^-22), R' = 2^261");
    Fz<P> v;
    template <int B2> PLK_DI Lz<P, B2> widen() const {
        static_assert(B2 >= B, "a bound can only be relaxed");
        return Lz<P, B2>{v};
    }
    PLK_DI Lz<P, lz_mul_bound(B, B)> sq() const {
        static_assert(B <= LZ_MUL_MAX, "operand of a product above 15p");
        return {fz_sqr<P>(v)};
    }
    PLK_DI Lz<P, 2 * B> dbl() const {  // field.rs:181-183 multiplies by TWO: the same value
        static_assert(2 * B <= LZ_VAL_MAX, "value could leave the representation");
        return {fz_add<P>(v, v)};
    }
    PLK_DI auto quad() const { return dbl().dbl(); }  // field.rs:191-193
    PLK_DI Lz<P, 16> rs() const {                      // any value of the representation -> below 2p, no multiplication
        static_assert(B <= 1023, "value could leave the representation");
        return {fz_reduce_small<P>(v)};
    }
    PLK_DI auto pow5() const {  // exp_usize(5), rescue_a.rs:58, rescue_b.rs:44
        const auto x2 = sq();
        return x2.sq() * *this;
    }
};
template <class P, int A, int B> PLK_DI Lz<P, A + B> operator+(const Lz<P, A>& a, const Lz<P, B>& b) {
    static_assert(A + B <= LZ_VAL_MAX, "value could leave the representation");
    return {fz_add<P>(a.v, b.v)};
}
template <class P, int A, int B> PLK_DI Lz<P, A + (8 << lz_sub_k(B))> operator-(const Lz<P, A>& a, const Lz<P, B>& b) {
    static_assert(A + (8 << lz_sub_k(B)) <= LZ_VAL_MAX, "value could leave the representation");
    return {fz_sub<P, lz_sub_k(B)>(a.v, b.v)};
}
template <class P, int A, int B> PLK_DI Lz<P, lz_mul_bound(A, B)> operator*(const Lz<P, A>& a, const Lz<P, B>& b) {
    static_assert(A <= LZ_MUL_MAX && B <= LZ_MUL_MAX, "operand of a product above 15p");
    return {fz_mul<P>(a.v, b.v)};
}
// A sum of products (and plain values) through ONE reduction (fz.cuh: FzWide; round 5).  B: bound of the accumulated value in eighths of p
// (a product of values below a p and b p adds a b / 1023; the reduction's own + p is added when it is taken), U: the column budget used
// (FZ_WIDE_UNITS).  RAW_A / RAW_B: the operand may have limbs up to 2^30 (a row as lz_load returns it) instead of carried / normalised ones.
template <class P, int B, int U> struct LzWide {
    FzWide<P> w;
};
template <class P> PLK_DI LzWide<P, 0, 0> lz_wide() {
    LzWide<P, 0, 0> r;
    fz_wide_clear<P>(r.w);
    return r;
}
template <bool RAW_A = false, bool RAW_B = false, class P, int B, int U, int A1, int A2>
PLK_DI LzWide<P, B + A1 * A2 / 1023 + 1, U + (RAW_A ? 2 : 1) * (RAW_B ? 2 : 1)> lz_mac(const LzWide<P, B, U>& acc, const Lz<P, A1>& a, const Lz<P, A2>& b) {
    static_assert(A1 <= LZ_MUL_MAX && A2 <= LZ_MUL_MAX, "operand of a product above 15p");
    static_assert(U + (RAW_A ? 2 : 1) * (RAW_B ? 2 : 1) <= FZ_WIDE_UNITS, "the column sums have no room for another product");
    static_assert(B + A1 * A2 / 1023 + 1 <= LZ_VAL_MAX, "value could leave the representation");
    LzWide<P, B + A1 * A2 / 1023 + 1, U + (RAW_A ? 2 : 1) * (RAW_B ? 2 : 1)> r{acc.w};
    fz_wide_mac<P>(r.w, a.v, b.v);
    return r;
}
template <class P, int B, int U, int A> PLK_DI LzWide<P, B + A, U> lz_wide_add(const LzWide<P, B, U>& acc, const Lz<P, A>& v) {
    static_assert(B + A <= LZ_VAL_MAX, "value could leave the representation");
    LzWide<P, B + A, U> r{acc.w};
    fz_wide_add<P>(r.w, v.v);
    return r;
}
template <class P, int B, int U> PLK_DI Lz<P, B + 9> lz_reduce(const LzWide<P, B, U>& acc) { return {fz_wide_reduce<P>(acc.w)}; }
// keeps a running value small: past 30p it is brought back below 2p
template <class P, int B> PLK_DI auto lz_tame(const Lz<P, B>& a) {
    if constexpr (B > 240) return a.rs();
    else return a;
}
// The caller's data is in the reference's R-form (x 2^256, canonical).  R' = 2^261 = 32 R: the R'-form of the same element is
// 32 x, i.e. the words re-sliced into 29-bit limbs five bits lower, then reduced below 2p without a multiplication.
template <class P> PLK_DI Lz<P, 16> lz_from_rform(const Fe<P>& x) {
    constexpr int NZ = FzCfg<P>::NZ, S = 29 * NZ - 32 * P::NL;
    static_assert(S > 0 && S < 29, "R' / R must be a shift by less than a limb");
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        const int off = 29 * i - S;
        uint32_t v;
        if (off < 0) v = x.v[0] << (-off);
        else {
            const int w = off >> 5, sh = off & 31;
            if (sh == 0) v = x.v[w];
            else if (sh + 29 <= 32 || w + 1 >= P::NL) v = x.v[w] >> sh;
            else v = (x.v[w] >> sh) | (x.v[w + 1] << (32 - sh));
        }
        r.l[i] = v & FzCfg<P>::M;
    }
    return {fz_reduce_small<P>(r)};
}
// The same conversion for the rows a point reads (~110 loads per point of the quotient numerator), without the reduction: cut the
// word X < p at bit s = floor(log2 p) - 5, X = t 2^s + X_low.  Then 32 X = 32 X_low + t 2^(s + 5) with 32 X_low < 2^(s + 5) <= p, and
// (t 2^(s + 5)) mod p is a table of p / 2^s + 1 <= 38 plain integers in 29-bit limbs (LZ_TOP_ROWS rows, staged in LDS by
// the kernels; built with the circuit-size tables): nine limb additions instead of a quotient estimate and a multiply-subtract per
// limb.  Value below 2p; limbs below 2^30 - 1 (two exact limbs added, no carry pass) - what fz_sub accepts of a subtrahend
// (fz.cuh) and what a product accepts of BOTH operands: 9 (2^30)^2 + 9 2^58 + 2^36 < 2^64.
constexpr int LZ_TOP_ROWS = 64, LZ_TOP_STRIDE = 12;
template <class P> struct LzSplit {
    static constexpr int top_bit() {
        for (int b = 32 * P::NL - 1; b >= 0; --b)
            if ((P::MOD[b >> 5] >> (b & 31)) & 1u) return b;
        return 0;
    }
    // R' / R = 2^SH: the re-slicing of lz_from_rform shifts by SH, so the cut and the table's exponent must use the same SH (5 for the
    // 256-bit fields on nine 29-bit limbs - the only instantiation today; a field with another SH gets the right table, not a wrong residue)
    static constexpr int SH = 29 * FzCfg<P>::NZ - 32 * P::NL;
    static constexpr int S = top_bit() - SH;  // X_low = X mod 2^S
    static constexpr int E = S + SH;          // table row t = (t 2^E) mod p, 2^E <= p
    static_assert(SH > 0 && SH < 29, "R' / R must be a shift by less than a limb");
    static_assert(S >= 29 * (FzCfg<P>::NZ - 1) - SH && S >= 32 * (P::NL - 1), "the cut must lie in the top limb and the top word");
    static_assert((P::MOD[P::NL - 1] >> (S - 32 * (P::NL - 1))) < 64u, "the top part of a canonical word must index the table (LZ_TOP_ROWS)");
};
// LDS copy of the table: a row is read with three 16-byte accesses (a limb-major copy read word by word - no two rows in one bank -
// was measured too: 10.44-10.51 ms against 10.27-10.28 for this layout, profiles/r04_quotient_diet.txt)
template <class P> using LzTop = uint32_t[LZ_TOP_ROWS][LZ_TOP_STRIDE];
template <class P> PLK_DI Lz<P, 16> lz_from_rform(const Fe<P>& x, const LzTop<P>& top) {
    constexpr int NZ = FzCfg<P>::NZ, SH = 29 * NZ - 32 * P::NL, S = LzSplit<P>::S;
    static_assert(SH > 0 && SH < 29 && NZ <= LZ_TOP_STRIDE, "R' / R must be a shift by less than a limb");
    // t <= (p - 1) >> S for a canonical word.  Words >= p violate the boundary's contract (include/plonky_hip.h: elements are fully reduced,
    // as every element of the reference is - monty.rs:41-45,103-106); the clamp only keeps such a word's load inside the table
    const uint32_t t = min(x.v[P::NL - 1] >> (S - 32 * (P::NL - 1)), (uint32_t)LZ_TOP_ROWS - 1u);
    const uint4 t0 = *reinterpret_cast<const uint4*>(&top[t][0]), t1 = *reinterpret_cast<const uint4*>(&top[t][4]),
                t2 = *reinterpret_cast<const uint4*>(&top[t][8]);
    const uint32_t tl[12] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w};
    Fz<P> r;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        const int off = 29 * i - SH;
        uint32_t v;
        if (off < 0) v = x.v[0] << (-off);
        else {
            const int w = off >> 5, sh = off & 31;
            if (sh == 0) v = x.v[w];
            else if (sh + 29 <= 32 || w + 1 >= P::NL) v = x.v[w] >> sh;
            else v = (x.v[w] >> sh) | (x.v[w + 1] << (32 - sh));
        }
        // the top limb ends at the cut: bits 29 (NZ - 1) - SH .. S - 1 of X
        const uint32_t mask = i == NZ - 1 ? (1u << (S - (29 * (NZ - 1) - SH))) - 1u : FzCfg<P>::M;
        r.l[i] = (v & mask) + tl[i];
    }
    return {r};
}
template <class P> PLK_DI Lz<P, 16> lz_load(const uint4* p, size_t i) { return lz_from_rform<P>(fe_load<P>(p + i * 2)); }  // k_all_constraints: no table staged
template <class P> PLK_DI Lz<P, 16> lz_load(const uint4* p, size_t i, const LzTop<P>& top) { return lz_from_rform<P>(fe_load<P>(p + i * 2), top); }
// the table (LZ_TOP_ROWS rows of LZ_TOP_STRIDE words) into LDS; the caller's next barrier publishes it
template <class P> PLK_DI void stage_top_table(const uint32_t* __restrict__ top_global, LzTop<P>& s_top) {
    for (int k = threadIdx.x; k < LZ_TOP_ROWS * LZ_TOP_STRIDE / 4; k += blockDim.x)
        reinterpret_cast<uint4*>(&s_top[0][0])[k] = reinterpret_cast<const uint4*>(top_global)[k];
}
// table entries are stored in R'-form, canonical
template <class P> PLK_DI Lz<P, 8> lz_table(const uint4* p, size_t i) { return {fz_from_fe<P>(fe_load<P>(p + i * 2))}; }
template <class P> PLK_DI Lz<P, 8> lz_one() { return {fz_one_rprime<P>()}; }
// back to the reference's form: x R' * R / R' = x R, below 2p, then the unique representative
template <class P, int B> PLK_DI Fe<P> lz_to_rform(const Lz<P, B>& a) {
    static_assert(B <= LZ_MUL_MAX, "operand of a product above 15p");
    return fz_to_fe_canonical<P>(fz_mul<P>(a.v, fz_const_rprime_to_r<P>()));
}

// circuit-size tables (device, field, log_degree): everything the loop needs that does not depend on the proof
struct PlonkTables {
    void* xs_lo = nullptr;    // g^j, j < 1024 (g = primitive 8n-th root, circuit_builder.rs:1122), R-form: feeds the L_1 table
    void* xs_hi = nullptr;    // g^(1024 j)
    void* xs_lo_z = nullptr;  // the same two tables in R'-form: the point x of a lane is one product of them
    void* xs_hi_z = nullptr;
    void* l1 = nullptr;       // L_1(g^i), i < 8n  (plonk_util.rs:14-24), R'-form
    void* small = nullptr;    // [0..7]: 1/1 .. 1/7 then unused, R'-form; MDS entry (r, c) = 1 / (4 + r - c)  (mds.rs:63-77)
    void* top = nullptr;      // (t 2^(S + 5)) mod p, t < LZ_TOP_ROWS, plain integers in 29-bit limbs (lz_from_rform)
    ~PlonkTables() {
        for (void* p : {xs_lo, xs_hi, xs_lo_z, xs_hi_z, l1, small, top})
            if (p) (void)hipFree(p);
    }
};
static std::mutex g_plonk_mu;
static std::map<std::tuple<int, int, int>, std::shared_ptr<PlonkTables>> g_plonk;

int plonk_clear_cache_impl() {
    std::lock_guard<std::mutex> lk(g_plonk_mu);
    g_plonk.clear();
    return PLK_OK;
}

template <class P> __global__ void k_plonk_xs(const uint4* __restrict__ pw, int log_t, int log_n8, uint4* __restrict__ lo, uint4* __restrict__ hi,
                                              uint4* __restrict__ lo_z, uint4* __restrict__ hi_z, uint4* __restrict__ small, uint32_t* __restrict__ top) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n_lo = (size_t)1 << XS_LO_LOG;
    const size_t n_hi = log_n8 > XS_LO_LOG ? (size_t)1 << (log_n8 - XS_LO_LOG) : 1;
    if (idx < n_lo) {
        const uint64_t e = (idx & (((uint64_t)1 << log_n8) - 1)) << (log_t - log_n8);
        const Fe<P> v = pow_from_table<P>(pw, 0, e, log_t);
        fe_store<P>(lo + idx * 2, v);
        fe_store<P>(lo_z + idx * 2, to_rprime<P>(v));
    } else if (idx < n_lo + n_hi) {
        const uint64_t e = (((idx - n_lo) << XS_LO_LOG) & (((uint64_t)1 << log_n8) - 1)) << (log_t - log_n8);
        const Fe<P> v = pow_from_table<P>(pw, 0, e, log_t);
        fe_store<P>(hi + (idx - n_lo) * 2, v);
        fe_store<P>(hi_z + (idx - n_lo) * 2, to_rprime<P>(v));
    } else if (idx < n_lo + n_hi + 7) {
        // 1 / v for v = 1 .. 7
        const uint32_t v = (uint32_t)(idx - n_lo - n_hi) + 1;
        Fe<P> c = fe_zero<P>();
        c.v[0] = v;
        fe_store<P>(small + (v - 1) * 2, to_rprime<P>(fe_inv_safegcd<P>(fe_from_canonical<P>(c))));
    } else if (idx < n_lo + n_hi + 7 + LZ_TOP_ROWS) {
        // (t 2^(S + 5)) mod p as a plain integer: the product of the two numbers in Montgomery form, brought back
        const uint32_t tt = (uint32_t)(idx - n_lo - n_hi - 7);
        constexpr int E = LzSplit<P>::E;  // 2^E <= p
        Fe<P> a = fe_zero<P>(), b = fe_zero<P>();
        a.v[0] = tt;
        b.v[E >> 5] = 1u << (E & 31);
        const Fe<P> prod = fe_to_canonical<P>(fe_mul<P>(fe_from_canonical<P>(a), fe_from_canonical<P>(b)));
        const Fz<P> z = fz_from_fe<P>(prod);
#pragma unroll
        for (int i = 0; i < LZ_TOP_STRIDE; ++i) top[tt * LZ_TOP_STRIDE + i] = i < FzCfg<P>::NZ ? z.l[i] : 0u;
    }
}
template <class P> PLK_DI Fe<P> plonk_x(const uint4* lo, const uint4* hi, size_t i) {
    const Fe<P> a = fe_load<P>(lo + (i & (((size_t)1 << XS_LO_LOG) - 1)) * 2);
    if ((i >> XS_LO_LOG) == 0) return a;
    return fe_mul<P>(a, fe_load<P>(hi + (i >> XS_LO_LOG) * 2));
}
// L_1(x) = (x^n - 1) / (n (x - 1)), L_1(1) = 1; eight points per lane share one inversion (Montgomery's trick)
constexpr int L1_PER_LANE = 8;
template <class P> __global__ void __launch_bounds__(64) k_plonk_l1(const uint4* __restrict__ lo, const uint4* __restrict__ hi, int log_degree,
                                                                    uint4* __restrict__ l1) {
    const size_t n8 = (size_t)8 << log_degree;
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * L1_PER_LANE;
    if (i0 >= n8) return;
    Fe<P> nf = fe_zero<P>();
    nf.v[log_degree >> 5] = 1u << (log_degree & 31);
    nf = fe_from_canonical<P>(nf);  // from_canonical_usize(n)
    const Fe<P> one = fe_one<P>();
    Fe<P> den[L1_PER_LANE], pre[L1_PER_LANE];
    Fe<P> run = one;
#pragma unroll
    for (int m = 0; m < L1_PER_LANE; ++m) {
        const Fe<P> x = plonk_x<P>(lo, hi, i0 + m);
        den[m] = i0 + m == 0 ? one : fe_mul<P>(nf, fe_sub<P>(x, one));
        pre[m] = run;
        run = fe_mul<P>(run, den[m]);
    }
    Fe<P> inv = fe_inv_safegcd<P>(run);
    // x^n = (g^n)^i with g^n a primitive 8th root of unity: g^(n i) = g^(n (i mod 8))
#pragma unroll
    for (int m = L1_PER_LANE - 1; m >= 0; --m) {
        const Fe<P> dinv = fe_mul<P>(inv, pre[m]);
        inv = fe_mul<P>(inv, den[m]);
        const size_t i = i0 + m;
        const Fe<P> xn = plonk_x<P>(lo, hi, ((i & 7) << log_degree));
        Fe<P> r = fe_mul<P>(fe_sub<P>(xn, one), dinv);
        if (i == 0) r = one;
        fe_store<P>(l1 + i * 2, to_rprime<P>(r));
    }
}

template <class P> static int get_plonk_tables(int log_degree, hipStream_t stream, std::shared_ptr<PlonkTables>& out) {
    int dev = 0;
    PLK_HIP_TRY(hipGetDevice(&dev));
    const int log_n8 = log_degree + 3;
    std::lock_guard<std::mutex> lk(g_plonk_mu);
    const auto key = std::make_tuple(dev, (int)P::FIELD_ID, log_degree);
    auto it = g_plonk.find(key);
    if (it != g_plonk.end()) {
        out = it->second;
        return PLK_OK;
    }
    const void* pw = nullptr;
    int log_t = 0;
    std::shared_ptr<const void> plan_hold;  // keeps the plan (and its power table) alive across the two launches below
    PLK_TRY(ntt_plan_pow_table(P::FIELD_ID, (unsigned)log_n8, &pw, &log_t, &plan_hold));
    auto t = std::make_shared<PlonkTables>();
    const size_t n8 = (size_t)1 << log_n8;
    const size_t n_lo = (size_t)1 << XS_LO_LOG, n_hi = log_n8 > XS_LO_LOG ? (size_t)1 << (log_n8 - XS_LO_LOG) : 1;
    PLK_HIP_TRY(hipMalloc(&t->xs_lo, n_lo * 32));
    PLK_HIP_TRY(hipMalloc(&t->xs_hi, n_hi * 32));
    PLK_HIP_TRY(hipMalloc(&t->xs_lo_z, n_lo * 32));
    PLK_HIP_TRY(hipMalloc(&t->xs_hi_z, n_hi * 32));
    PLK_HIP_TRY(hipMalloc(&t->small, 8 * 32));
    PLK_HIP_TRY(hipMalloc(&t->l1, n8 * 32));
    PLK_HIP_TRY(hipMalloc(&t->top, (size_t)LZ_TOP_ROWS * LZ_TOP_STRIDE * 4));
    const size_t cnt = n_lo + n_hi + 7 + LZ_TOP_ROWS;
    k_plonk_xs<P><<<(unsigned)((cnt + 127) / 128), 128, 0, stream>>>((const uint4*)pw, log_t, log_n8, (uint4*)t->xs_lo, (uint4*)t->xs_hi, (uint4*)t->xs_lo_z,
                                                                     (uint4*)t->xs_hi_z, (uint4*)t->small, (uint32_t*)t->top);
    const size_t lanes = (n8 + L1_PER_LANE - 1) / L1_PER_LANE;
    k_plonk_l1<P><<<(unsigned)((lanes + 63) / 64), 64, 0, stream>>>((const uint4*)t->xs_lo, (const uint4*)t->xs_hi, log_degree, (uint4*)t->l1);
    PLK_HIP_TRY(hipGetLastError());
    PLK_HIP_TRY(hipStreamSynchronize(stream));  // the power table of the plan is only borrowed for these two launches
    g_plonk[key] = t;
    out = t;
    return PLK_OK;
}

// ---------------------------------------------------------------------------------------------
// evaluate_all_constraints (gates/mod.rs:46-125): the unified constraint set at one point
// ---------------------------------------------------------------------------------------------
// The gates hand their filter and their constraint list to a sink.  Two sinks:
//  * TermSink: term t of the unified set += filter * constraint t (what evaluate_all_constraints returns, gates/mod.rs:100-124);
//  * ReducedSink: total += filter * (c_0 + alpha c_1 + alpha^2 c_2 + ..), the gate's share of reduce_with_powers over the unified
//    terms (plonk_util.rs:27-33; the sum over the gates and the powers of alpha commute) - one running value instead of eight,
//    and one product by the filter per gate instead of one per constraint.
// Ten gates, each adds at most once to a term / to the total, every summand a product (below LZ_SUMMAND_MAX / 8 p = 2p - a filter that is
// a difference of two prefix products and a sum of constraints can both be a few p): the sums stay below 20p.
constexpr int LZ_SUMMAND_MAX = 16;
template <class P> using Term = Lz<P, 10 * LZ_SUMMAND_MAX>;
// Between two gates: the instruction scheduler must not interleave them (left alone it overlaps all ten gates for
// instruction-level parallelism and the kernel needs several times the register file).
PLK_DI void gate_fence() {
#ifdef __HIPCC__
    __builtin_amdgcn_sched_barrier(0);
#endif
}
template <int GATE, class P, int B> PLK_DI void add_product(Term<P>& u, const Lz<P, B>& product) {
    static_assert(GATE >= 0 && GATE < 10, "ten gates");
    static_assert(B <= LZ_SUMMAND_MAX, "a summand must be a product");
    u.v = fz_add<P>(u.v, product.v);
}
template <class P> struct TermSink {
    static constexpr bool kReduced = false;
    Term<P> (&u)[NUM_TERMS];
    template <int GATE, int T, class F> PLK_DI void put(const F&) {}
    template <int GATE, int T, class F, class C0, class... Cs> PLK_DI void put(const F& f, const C0& c0, const Cs&... cs) {
        add_product<GATE>(u[T], f * c0);
        put<GATE, T + 1>(f, cs...);
    }
    template <int GATE, class F, class... Cs> PLK_DI void gate(const F& f, const Cs&... cs) {
        gate_fence();
        put<GATE, 0>(f, cs...);
        gate_fence();
    }
};
template <class A, class C0> PLK_DI auto lz_horner(const A&, const C0& c0) { return c0; }
template <class A, class C0, class... Cs> PLK_DI auto lz_horner(const A& alpha, const C0& c0, const Cs&... cs) { return c0 + lz_horner(alpha, cs...) * alpha; }
// The weights of a launch (k_plonk_weights): what depends on alpha / beta / k_is only, so that a point does not recompute it.
//   [0..3]  W_c  = sum_i alpha^(2 i + 1) / (4 + i - c): RescueStepA's share of reduce_with_powers that is linear in its state wire 4 + c
//           (its constraints 2 i + 1 are rows of the MDS matrix, mds.rs:63-77 - sixteen products by matrix entries become four by W_c);
//   [4..7]  W'_c = sum_i alpha^i / (4 + i - c): the same for RescueStepB (constraint i is MDS row i over the fifth powers);
//   [8..13] beta k_is[j] (plonk.rs:431: beta * k_i * x is one product per wire instead of two).
//   [14..21] alpha^1 .. alpha^8: a gate's share of reduce_with_powers is the DOT product of its constraints with them - one reduction for up
//           to six constraints instead of one per Horner step, no sums in between (round 5).
constexpr int WEIGHT_ALPHA = 8 + NUM_ROUTED_WIRES, NUM_ALPHA_POWERS = 8;
constexpr int NUM_WEIGHTS = 8 + NUM_ROUTED_WIRES + NUM_ALPHA_POWERS;
template <class P> struct ReducedSink {
    static constexpr bool kReduced = true;
    Lz<P, 16> alpha;
    Term<P> total;
    const uint32_t (*weights)[FzCfg<P>::NZ];  // staged in LDS, read where a gate uses them
    PLK_DI Lz<P, 16> weight(int w) const {
        Lz<P, 16> r;
#pragma unroll
        for (int i = 0; i < FzCfg<P>::NZ; ++i) r.v.l[i] = weights[w][i];
        return r;
    }
    // c_I alpha^I + ... into the accumulator; when its columns are full the partial sum is reduced and re-enters as a value
    template <int I, class W> PLK_DI auto dot(const W& w) const { return lz_reduce(w); }
    template <int I, int B, int U, class C, class... Cs> PLK_DI auto dot(const LzWide<P, B, U>& w, const C& c, const Cs&... cs) const {
        static_assert(I >= 1 && I <= NUM_ALPHA_POWERS, "alpha^I is not staged");
        if constexpr (U + 1 > FZ_WIDE_UNITS) {
            return dot<I>(lz_wide_add(lz_wide<P>(), lz_reduce(w)), c, cs...);
        } else {
            return dot<I + 1>(lz_mac(w, c, weight(WEIGHT_ALPHA + I - 1)), cs...);
        }
    }
    // value below 15p for the product by the filter
    template <int B> PLK_DI static auto tamed(const Lz<P, B>& h) {
        if constexpr (B > LZ_MUL_MAX) return h.rs();
        else return h;
    }
    template <int GATE, class F, class C0, class... Cs> PLK_DI void gate(const F& f, const C0& c0, const Cs&... cs) {
        gate_fence();
        if constexpr (sizeof...(cs) == 0) add_product<GATE>(total, f * tamed(c0));
        else add_product<GATE>(total, f * tamed(dot<1>(lz_wide_add(lz_wide<P>(), c0), cs...)));
        gate_fence();
    }
    // filter * (c_0 + alpha c_1 + ... + extra): `extra` is the part of the gate's sum that was folded into the launch's weights
    template <int GATE, class F, class E, class C0, class... Cs> PLK_DI void gate_extra(const F& f, const E& extra, const C0& c0, const Cs&... cs) {
        gate_fence();
        add_product<GATE>(total, f * tamed(dot<1>(lz_wide_add(lz_wide_add(lz_wide<P>(), c0), extra), cs...)));
        gate_fence();
    }
};
// Base4SumGate's running sum, computed = 4 computed + limb over the seven limbs (base_4_sum.rs:44-48)
template <int I, class P, int B, class LW> PLK_DI auto base4_chain(const Lz<P, B>& computed, const LW& l) {
    if constexpr (I == NUM_WIRES - 2) return computed;
    else return base4_chain<I + 1>(lz_tame(computed.quad() + l[2 + I]), l);
}

// k: local constants [6]; l: local wires [9]; r: right wires (only indices 0..3 are read by any gate); b2, b3: below wires 2, 3
// (only CurveEndoGate reads below, curve_endo.rs:115-117); small: 1/1 .. 1/7 (R'-form table).
// Prefix filters (gates/mod.rs:289-300) are formed where they are used, along the prefix tree of gates/mod.rs:1-16 (BufferGate,
// PREFIX 101000 in the code - buffer.rs:27; the doc comment of gates/mod.rs says 101010 -, has no constraints: buffer.rs:26-33): a filter kept alive for the whole kernel costs nine registers.
// gate groups (a kernel evaluates a subset: the live set of all ten gates is several register files wide)
constexpr int GATES_RESCUE_A = 1, GATES_ENDO = 2, GATES_BASE4_ARITH = 4, GATES_ADD_PUBLIC = 8, GATES_DBL_CONST = 16, GATES_RESCUE_B = 32, GATES_ALL = 63;
constexpr int GATES_RESCUE = GATES_RESCUE_A | GATES_RESCUE_B;
// The inputs of a point are READ WHERE A GATE USES THEM: k / l / r are rows of the prover's tables (or plain arrays, in
// k_all_constraints) indexed through operator[], every mention a load and a re-slicing into the working form (~80
// instructions against ~220 for a product).  An input held from the top of the kernel costs nine registers for its whole
// length; 21 of them are 189 - with the working set of a gate that is a whole register file and one wave per SIMD.
template <class P> struct LazyRow {
    const uint4* base;
    size_t stride, i;
    const LzTop<P>& top;
    PLK_DI Lz<P, 16> operator[](int j) const { return lz_load<P>(base, (size_t)j * stride + i, top); }
};
template <class P, class D, int MASK, class K, class LW, class RW, class Sink>
PLK_DI void all_constraints(const K& k, const LW& l, const RW& r, const D& b2, const D& b3, const D& zeta, const D& a_coeff,
                            const uint4* __restrict__ small, Sink& sink) {
    const auto one = lz_one<P>();
    if constexpr ((MASK & GATES_RESCUE) != 0) {  // RescueStepAGate 00, rescue_a.rs:38-69, and RescueStepBGate 01, rescue_b.rs:30-58
        const auto nk0 = one - k[0];
        const D k1 = k[1];
        // both steps in one launch: (1 - k0) (1 - k1) = (1 - k0) - (1 - k0) k1, one product for the two filters
        constexpr bool BOTH_STEPS = (MASK & GATES_RESCUE) == GATES_RESCUE;
        const auto f_step_b = nk0 * k1;
        auto filter_a = [&] {
            if constexpr (BOTH_STEPS) return nk0 - f_step_b;
            else return nk0 * (one - k1);
        };
        Lz<P, 8> mds[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) mds[i] = lz_table<P>(small, i);  // mds[v - 1] = 1 / v; entry (r, c) = 1 / (4 + r - c), mds.rs:63-77
        // row i of the MDS matrix times (v0 .. v3), plus the round constant, against the right gate's wire i
#define PLK_MDS_ROW(i, v0, v1, v2, v3) (k[2 + i] + (mds[3 + i] * v0 + mds[2 + i] * v1 + mds[1 + i] * v2 + mds[i] * v3) - r[i])
        if constexpr ((MASK & GATES_RESCUE_A) != 0) {
            const D l4 = l[4], l5 = l[5], l6 = l[6], l7 = l[7];
            if constexpr (Sink::kReduced) {
                // sum_t alpha^t c_t with the matrix part of the odd constraints taken out: sum_c l_(4 + c) W_c (weights 0..3)
                // four products, one reduction; the rows as loaded have limbs up to 2^30 (two column units each: carried copies go in)
                auto carried = [](D v) {
                    fz_carry<P>(v.v);
                    return v;
                };
                const auto folded = lz_reduce(lz_mac(lz_mac(lz_mac(lz_mac(lz_wide<P>(), carried(l4), sink.weight(0)), carried(l5), sink.weight(1)),
                                                            carried(l6), sink.weight(2)), carried(l7), sink.weight(3)));
                sink.template gate_extra<7>(filter_a(), folded,                      //
                                            l4.pow5() - l[0], k[2] - r[0],            //
                                            l5.pow5() - l[1], k[3] - r[1],            //
                                            l6.pow5() - l[2], k[4] - r[2],            //
                                            l7.pow5() - l[3], k[5] - r[3]);
            } else {
                sink.template gate<7>(filter_a(),                                                //
                                      l4.pow5() - l[0], PLK_MDS_ROW(0, l4, l5, l6, l7),           //
                                      l5.pow5() - l[1], PLK_MDS_ROW(1, l4, l5, l6, l7),           //
                                      l6.pow5() - l[2], PLK_MDS_ROW(2, l4, l5, l6, l7),           //
                                      l7.pow5() - l[3], PLK_MDS_ROW(3, l4, l5, l6, l7));
            }
        }
        if constexpr ((MASK & GATES_RESCUE_B) != 0) {
            const auto e0 = l[0].pow5(), e1 = l[1].pow5(), e2 = l[2].pow5(), e3 = l[3].pow5();
            if constexpr (Sink::kReduced) {
                const auto folded = lz_reduce(lz_mac(lz_mac(lz_mac(lz_mac(lz_wide<P>(), e0, sink.weight(4)), e1, sink.weight(5)), e2, sink.weight(6)),
                                                     e3, sink.weight(7)));
                sink.template gate_extra<8>(f_step_b, folded, k[2] - r[0], k[3] - r[1], k[4] - r[2], k[5] - r[3]);
            } else {
                sink.template gate<8>(f_step_b, PLK_MDS_ROW(0, e0, e1, e2, e3), PLK_MDS_ROW(1, e0, e1, e2, e3), PLK_MDS_ROW(2, e0, e1, e2, e3),
                                      PLK_MDS_ROW(3, e0, e1, e2, e3));
            }
        }
#undef PLK_MDS_ROW
    }
    [[maybe_unused]] Lz<P, lz_mul_bound(16, 16)> f_endo{fz_zero<P>()};  // k0 k1; the prefix 10 below is k0 - k0 k1 when both are evaluated here
    if constexpr ((MASK & GATES_ENDO) != 0) {  // CurveEndoGate 11, curve_endo.rs:96-141
        const D &x1 = l[0], &y1 = l[1], &x_in = l[4], &y_in = l[5], &x3 = r[0], &y3 = r[1];
        const D &unsigned_old = l[2], &unsigned_new = b2, &signed_old = l[3], &signed_new = b3, &bit0 = l[6], &bit1 = l[7], &inverse = l[8];
        const auto mult = (zeta - one) * bit1 + one;  // x2's factor and signed_limb_multiplier are the same expression
        const auto x2 = mult * x_in;
        const auto sgn = bit0.dbl() - one;
        const auto y2 = sgn * y_in;
        const auto lambda = (y1 - y2) * inverse;
        const auto computed_x3 = lambda.sq() - x1 - x2;
        const auto computed_y3 = lambda * (x1 - x3) - y1;
        const auto signed_limb = sgn * mult;
        f_endo = k[0] * k[1];
        sink.template gate<2>(f_endo, computed_x3 - x3, computed_y3 - y3, unsigned_new - (unsigned_old.quad() + bit1.dbl() + bit0).rs(),
                              signed_new - (signed_old.dbl() + signed_limb), bit0 * (bit0 - one), bit1 * (bit1 - one), inverse * (x1 - x2) - one);
    }
    if constexpr ((MASK & (GATES_BASE4_ARITH | GATES_ADD_PUBLIC | GATES_DBL_CONST)) == 0) return;
    const auto p10 = [&] {
        if constexpr ((MASK & GATES_ENDO) != 0) return k[0] - f_endo;
        else return k[0] * (one - k[1]);
    }();
    if constexpr ((MASK & GATES_BASE4_ARITH) != 0) {
        const auto p100 = p10 * (one - k[2]);
        // the two filters below p100 share one product: p100 (1 - k3) = p100 - p100 k3
        const auto f_arith = p100 * k[3];
        const auto f_base4 = p100 - f_arith;
        // ArithmeticGate 1001, arithmetic.rs:30-46
        // k4 l0 l1 + k5 l2 - l3: the two outer products through one reduction (rows as loaded: two column units per raw operand)
        sink.template gate<6>(f_arith, lz_reduce(lz_mac<true, true>(lz_mac<false, true>(lz_wide<P>(), k[4] * l[0], l[1]), k[5], l[2])) - l[3]);
        {  // Base4SumGate 1000, base_4_sum.rs:34-63: 7 limbs in wires 2..8
            const auto two = one.dbl();
            // (limb - 0) (limb - 1) (limb - 2) (limb - 3), times ONE in the reference: with t = limb^2 - 3 limb = limb (limb - 3) it is
            // t (t + 2) - a squaring and a product instead of three products
            auto b4 = [&](const D& limb) {
                const auto t = limb.sq() - (limb.dbl() + limb);
                return t * (t + two);
            };
#define PLK_B4(i) b4(l[2 + i])
            sink.template gate<3>(f_base4, (base4_chain<0>(l[0], l) - l[1]).rs(), PLK_B4(0), PLK_B4(1), PLK_B4(2), PLK_B4(3), PLK_B4(4), PLK_B4(5), PLK_B4(6));
#undef PLK_B4
        }
    }
    if constexpr ((MASK & (GATES_ADD_PUBLIC | GATES_DBL_CONST)) == 0) return;
    const auto p101 = p10 * k[2];
    if constexpr ((MASK & GATES_ADD_PUBLIC) != 0) {
        const auto p1010 = p101 * (one - k[3]);
        const auto f_add = p1010 * k[4];        // p1010 (1 - k4) = p1010 - p1010 k4: the two filters below p1010 share one product
        const auto f_not_add = p1010 - f_add;
        {  // CurveAddGate 10101, curve_add.rs:60-101
            const D &x1 = l[0], &y1 = l[1], &x4 = r[0], &y4 = r[1], &acc_old = l[2], &acc_new = l[3], &x2 = l[4], &y2 = l[5], &bit = l[6], &inverse = l[7],
                    &lambda = l[8];
            const auto computed_lambda = (y1 - y2) * inverse;
            const auto x3 = lambda.sq() - x1 - x2;
            const auto y3 = lambda * (x1 - x4) - y1;
            const auto not_bit = one - bit;
            // bit x3 + (1 - bit) x1 = x1 + bit (x3 - x1): one product each instead of two
            const auto computed_x4 = x1 + bit * (x3 - x1);
            const auto computed_y4 = y1 + bit * (y3 - y1);
            sink.template gate<0>(f_add, computed_lambda - lambda, computed_x4 - x4, computed_y4 - y4, acc_new - (acc_old.dbl() + bit), bit * not_bit,
                                  inverse * (x1 - x2) - one);
        }
        // PublicInputGate 101001, public_input.rs:26-37: advice wires 6..8 against the right gate's wires 0..2
        sink.template gate<4>(f_not_add * k[5], l[6] - r[0], l[7] - r[1], l[8] - r[2]);
    }
    if constexpr ((MASK & GATES_DBL_CONST) != 0) {
        const auto p1011 = p101 * k[3];
        const auto f_dbl = p1011 * k[4];
        const auto f_const = p1011 - f_dbl;
        // ConstantGate 10110, constant.rs:28-39
        sink.template gate<5>(f_const, k[5] - l[0]);
        {  // CurveDblGate 10111, curve_dbl.rs:42-68
            const D &x_old = l[0], &y_old = l[1], &x_new = l[2], &y_new = l[3], &inverse = l[4], &lambda = l[5];
            const auto xx = x_old.sq();
            const auto numerator = xx.dbl() + xx + a_coeff;  // square().triple() + A
            const auto computed_lambda = numerator * inverse;
            const auto computed_x_new = lambda.sq() - x_old.dbl();
            const auto computed_y_new = lambda * (x_old - x_new) - y_old;
            sink.template gate<1>(f_dbl, computed_lambda - lambda, computed_x_new - x_new, computed_y_new - y_new, y_old.dbl() * inverse - one);
        }
    }
}

struct PlonkScalars {
    uint32_t k_is[NUM_ROUTED_WIRES][8];
    uint32_t alpha[8], beta[8], gamma[8], zeta[8], a[8];
};
// the scalars of a launch in the working form, converted once per workgroup: [0..5] k_is, 6 alpha, 7 beta, 8 gamma, 9 zeta, 10 a
constexpr int NUM_SCALARS = NUM_ROUTED_WIRES + 5;
// rows NUM_SCALARS .. NUM_SCALARS + NUM_WEIGHTS - 1: the weights of the launch (k_plonk_weights; limb form already), when there are any
template <class P> PLK_DI void stage_scalars(const PlonkScalars& sc, uint32_t (*s_sc)[FzCfg<P>::NZ], const uint32_t* __restrict__ weights = nullptr) {
    const int t = threadIdx.x;
    if (weights && t >= NUM_SCALARS && t < NUM_SCALARS + NUM_WEIGHTS) {
#pragma unroll
        for (int i = 0; i < FzCfg<P>::NZ; ++i) s_sc[t][i] = weights[(t - NUM_SCALARS) * FzCfg<P>::NZ + i];
    }
    if (t < NUM_SCALARS) {
        const uint32_t* w = t < NUM_ROUTED_WIRES ? sc.k_is[t] : t == 6 ? sc.alpha : t == 7 ? sc.beta : t == 8 ? sc.gamma : t == 9 ? sc.zeta : sc.a;
        Fe<P> x;
#pragma unroll
        for (int i = 0; i < 8; ++i) x.v[i] = w[i];
        const Lz<P, 16> z = lz_from_rform<P>(x);
#pragma unroll
        for (int i = 0; i < FzCfg<P>::NZ; ++i) s_sc[t][i] = z.v.l[i];
    }
    __syncthreads();
}
template <class P> PLK_DI Lz<P, 16> scalar_at(const uint32_t (*s_sc)[FzCfg<P>::NZ], int t) {
    Lz<P, 16> r;
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) r.v.l[i] = s_sc[t][i];
    return r;
}

// The weights of a call (ReducedSink), one lane each; alpha^e by repeated products - fourteen lanes, once per call.
template <class P> __global__ void __launch_bounds__(64) k_plonk_weights(const uint4* __restrict__ small, PlonkScalars sc, uint32_t* __restrict__ out) {
    using D = Lz<P, 16>;
    const int t = threadIdx.x;
    if (t >= NUM_WEIGHTS) return;
    auto load = [](const uint32_t (&w)[8]) {
        Fe<P> x;
#pragma unroll
        for (int i = 0; i < 8; ++i) x.v[i] = w[i];
        return lz_from_rform<P>(x);
    };
    D res;
    if (t < 8) {
        const D alpha = load(sc.alpha);
        const int c = t & 3;
        const bool step_b = t >= 4;
        // RescueStepA: exponents 1, 3, 5, 7 (its matrix rows are the odd constraints); RescueStepB: 0, 1, 2, 3
        D pw = step_b ? lz_one<P>().template widen<16>() : alpha;
        const D step = step_b ? alpha : (alpha * alpha).template widen<16>();
        D acc{fz_zero<P>()};
        for (int i = 0; i < 4; ++i) {
            acc = (acc + pw * lz_table<P>(small, 3 + i - c)).rs();  // MDS entry (i, c) = 1 / (4 + i - c), mds.rs:63-77
            pw = (pw * step).template widen<16>();
        }
        res = acc;
    } else if (t < WEIGHT_ALPHA) {
        res = (load(sc.beta) * load(sc.k_is[t - 8])).template widen<16>();
    } else {
        const D alpha = load(sc.alpha);
        res = alpha.rs();  // exactly normalised limbs, like the products below
        for (int e = 1; e < t - WEIGHT_ALPHA + 1; ++e) res = (res * alpha).template widen<16>();
    }
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) out[t * FzCfg<P>::NZ + i] = res.v.l[i];
}

// plonk.rs:392-453, one lane per point of the 8n domain, in FOUR launches.  The ten gates, the permutation argument and their
// inputs (21 + 10 elements per point) are several register files wide; evaluated in one piece the kernel spills to scratch
// and, at one wave per SIMD, waits out every reload.  A launch evaluates a group of gates and hands the running sum on (limb
// form, 48 B per point, in `part`); its inputs are read where a gate uses them (LazyRow), so a launch fits the 256
// registers of two waves per SIMD:
//   PASS 0: RescueStepA + RescueStepB (PASS 1 is RescueStepB alone when PLK_VANISH_MERGE_RESCUE is 0)   PASS 2: CurveEndo, Base4Sum,
//   Arithmetic   PASS 3: CurveAdd, PublicInput, CurveDbl, Constant   PASS 4: the permutation argument, L_1 and reduce_with_powers -> out
// The sum over the gates and the powers of alpha commute (ReducedSink), every value is exact: same result as one loop.
#ifndef PLK_VANISH_MERGE_RESCUE
#define PLK_VANISH_MERGE_RESCUE 1  // 1: the two Rescue gates in one launch (PASS 1 is not launched); 0: a launch each (measured 3 % slower, round 4)
#endif
#ifndef PLK_VANISH_WAVES
#define PLK_VANISH_WAVES 2  // waves per SIMD the register allocation is held to (3 was measured in round 4: tools/gpu/r04_vanish_waves.sh)
#endif
template <class P, int PASS>
__global__ void __launch_bounds__(128, PLK_VANISH_WAVES) k_vanishing_points(const uint4* __restrict__ constants, const uint4* __restrict__ wires, const uint4* __restrict__ s_sigma,
                                                             const uint4* __restrict__ z, const uint4* __restrict__ xs_lo_z, const uint4* __restrict__ xs_hi_z,
                                                             const uint4* __restrict__ l1, const uint4* __restrict__ small, PlonkScalars sc, int log_degree,
                                                             uint32_t* __restrict__ part, uint4* __restrict__ out, size_t first, size_t count,
                                                             const uint32_t* __restrict__ weights, const uint32_t* __restrict__ top_table) {
    static_assert(P::NL == 8, "256-bit scalar fields");
    using D = Lz<P, 16>;
    constexpr int MASK = PASS == 0 ? (PLK_VANISH_MERGE_RESCUE ? GATES_RESCUE : GATES_RESCUE_A) : PASS == 1 ? GATES_RESCUE_B
                         : PASS == 2 ? (GATES_ENDO | GATES_BASE4_ARITH) : PASS == 3 ? (GATES_ADD_PUBLIC | GATES_DBL_CONST) : 0;
    static_assert(NUM_SCALARS + NUM_WEIGHTS <= 128, "one lane of the workgroup stages one row");
    __shared__ uint32_t s_sc[NUM_SCALARS + NUM_WEIGHTS][FzCfg<P>::NZ];
    __shared__ __attribute__((aligned(16))) LzTop<P> s_top;
    stage_top_table<P>(top_table, s_top);
    stage_scalars<P>(sc, s_sc, weights);  // ends with the barrier that publishes both
    const size_t n8 = (size_t)8 << log_degree;
    const size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // this launch covers the points first .. first + count - 1
    if (i >= first + count || i >= n8) return;
    const size_t i_right = (i + 8) & (n8 - 1), i_below = (i + 8 * GRID_WIDTH) & (n8 - 1);
    // the rows a gate reads from: local constants, local wires, the right gate's wires (loaded where they are used)
    const LazyRow<P> k{constants, n8, i, s_top}, l{wires, n8, i, s_top}, r{wires, n8, i_right, s_top};
    const D alpha = scalar_at<P>(s_sc, 6);
    if constexpr (PASS < 4) {
        D b2{fz_zero<P>()}, b3{fz_zero<P>()};
        if constexpr ((MASK & GATES_ENDO) != 0) {
            b2 = lz_load<P>(wires, (size_t)2 * n8 + i_below, s_top);
            b3 = lz_load<P>(wires, (size_t)3 * n8 + i_below, s_top);
        }
        ReducedSink<P> sink{alpha, Term<P>{fz_zero<P>()}, s_sc + NUM_SCALARS};
        all_constraints<P, D, MASK>(k, l, r, b2, b3, scalar_at<P>(s_sc, 9), scalar_at<P>(s_sc, 10), small, sink);
        if constexpr (PASS == 0) {
            limbs_store<P>(part, i, sink.total.v);  // one or two gates, below 2p each
        } else {
            // gates so far: 1 (PASS 1), 2 (PASS 2), 5 (PASS 3), each below 2p
            const Lz<P, LZ_SUMMAND_MAX * (PASS == 1 ? 1 : PASS == 2 ? 2 : 5)> before{limbs_load<P>(part, i)};
            limbs_store<P>(part, i, (before + sink.total).v);
        }
    } else {
        const Lz<P, 9 * LZ_SUMMAND_MAX> total{limbs_load<P>(part, i)};  // nine gates with constraints: < 18p
        const auto one = lz_one<P>();
        const auto x = lz_table<P>(xs_lo_z, i & (((size_t)1 << XS_LO_LOG) - 1)) * lz_table<P>(xs_hi_z, i >> XS_LO_LOG);  // hi[0] = 1
        const D z_x = lz_load<P>(z, i, s_top), z_gz = lz_load<P>(z, i_right, s_top);
        // (z_1_term = L_1(x) (Z(x) - 1), plonk.rs:425, joins the last reduction below)
        const D beta = scalar_at<P>(s_sc, 7), gamma = scalar_at<P>(s_sc, 8);
        Lz<P, 9> f_prime = one.template widen<9>(), g_prime = f_prime;
#pragma unroll
        for (int j = 0; j < NUM_ROUTED_WIRES; ++j) {  // plonk.rs:428-437
            const auto beta_s_id = scalar_at<P>(s_sc, NUM_SCALARS + 8 + j) * x;  // beta * (k_is[j] * x), plonk.rs:430-431: beta k_is[j] is a weight of the call
            const D s_sig = lz_load<P>(s_sigma, (size_t)j * n8 + i, s_top);
            const D lj = l[j];
            f_prime = f_prime * (lj + beta_s_id + gamma);
            g_prime = g_prime * (lj + beta * s_sig + gamma);
        }
        // Z(x) f'(x) - g'(x) Z(g x), plonk.rs:438: two products, one reduction (z_x / z_gz are rows as loaded: two column units each)
        const auto zero = Lz<P, 0>{fz_zero<P>()};
        const auto v_shift_term = lz_reduce(lz_mac<false, true>(lz_mac<false, true>(lz_wide<P>(), f_prime, z_x), zero - g_prime, z_gz));
        // reduce_with_powers over [z_1_term, v_shift_term, constraint terms] (plonk.rs:440-447, plonk_util.rs:27-33):
        // total alpha^2 + v_shift alpha + L_1 (Z - 1), three products through one reduction
        const auto res = lz_reduce(lz_mac(lz_mac(lz_mac(lz_wide<P>(), total.rs(), scalar_at<P>(s_sc, NUM_SCALARS + WEIGHT_ALPHA + 1)), v_shift_term, alpha),
                                          lz_table<P>(l1, i), z_x - one));
        fe_store<P>(out + i * 2, lz_to_rform<P>(res));
    }
}

// evaluate_all_constraints at `count` independent points (constants [count][6], local / right / below [count][9], out [count][8])
template <class P>
__global__ void __launch_bounds__(128) k_all_constraints(const uint4* __restrict__ constants, const uint4* __restrict__ local, const uint4* __restrict__ right,
                                                         const uint4* __restrict__ below, const uint4* __restrict__ small, PlonkScalars sc, size_t count,
                                                         uint4* __restrict__ out) {
    using D = Lz<P, 16>;
    __shared__ uint32_t s_sc[NUM_SCALARS][FzCfg<P>::NZ];
    stage_scalars<P>(sc, s_sc);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    D k[NUM_CONSTANTS], l[NUM_WIRES], r[4];
#pragma unroll
    for (int j = 0; j < NUM_CONSTANTS; ++j) k[j] = lz_load<P>(constants, i * NUM_CONSTANTS + j);
#pragma unroll
    for (int j = 0; j < NUM_WIRES; ++j) l[j] = lz_load<P>(local, i * NUM_WIRES + j);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = lz_load<P>(right, i * NUM_WIRES + j);
    Term<P> u[NUM_TERMS];
#pragma unroll
    for (int t = 0; t < NUM_TERMS; ++t) u[t] = Term<P>{fz_zero<P>()};
    TermSink<P> sink{u};
    all_constraints<P, D, GATES_ALL>(k, l, r, lz_load<P>(below, i * NUM_WIRES + 2), lz_load<P>(below, i * NUM_WIRES + 3), scalar_at<P>(s_sc, 9),
                                     scalar_at<P>(s_sc, 10), small, sink);
#pragma unroll
    for (int t = 0; t < NUM_TERMS; ++t) fe_store<P>(out + (i * NUM_TERMS + t) * 2, lz_to_rform<P>(u[t].rs()));
}

static void put_words(uint32_t (&dst)[8], const uint64_t* src) {
    for (int i = 0; i < 4; ++i) {
        dst[2 * i] = (uint32_t)src[i];
        dst[2 * i + 1] = (uint32_t)(src[i] >> 32);
    }
}

template <class P>
static int vanishing_points_t(unsigned log_degree, const void* d_constants, const void* d_wires, const void* d_s_sigma, const void* d_z, const PlonkScalars& sc,
                              void* d_out, hipStream_t stream) {
    std::shared_ptr<PlonkTables> t;
    PLK_TRY(get_plonk_tables<P>((int)log_degree, stream, t));
    const size_t n8 = (size_t)8 << log_degree;
    void* part = scratch_acquire(limb_bytes(n8, FzCfg<P>::NZ), stream);
    if (!part) return PLK_ERR_OOM;
    uint32_t* weights = (uint32_t*)scratch_acquire((size_t)NUM_WEIGHTS * FzCfg<P>::NZ * 4, stream);
    if (!weights) {
        scratch_release(part, stream);
        return PLK_ERR_OOM;
    }
    k_plonk_weights<P><<<1, 64, 0, stream>>>((const uint4*)t->small, sc, weights);
    // PLK_VANISH_SLAB_LOG=k: the launches walk the domain in slabs of 2^k points, so that the rows a slab reads (29 x 32 B per
    // point) are still in the 256 MiB Infinity Cache when the next launch of the slab re-reads them (round-3 review item 5;
    // measured in DESIGN.md section 4c).  Default: the whole domain per launch.
    static const int slab_log = [] {
        const char* e = getenv("PLK_VANISH_SLAB_LOG");
        const int v = e ? atoi(e) : 0;
        return v >= 10 && v <= 40 ? v : 0;
    }();
    const size_t slab = slab_log ? ((size_t)1 << slab_log) : n8;
#define PLK_VANISH(PASS)                                                                                                                                  \
    k_vanishing_points<P, PASS><<<blocks, 128, 0, stream>>>((const uint4*)d_constants, (const uint4*)d_wires, (const uint4*)d_s_sigma, (const uint4*)d_z, \
                                                            (const uint4*)t->xs_lo_z, (const uint4*)t->xs_hi_z, (const uint4*)t->l1, (const uint4*)t->small, sc, \
                                                            (int)log_degree, (uint32_t*)part, (uint4*)d_out, first, cnt, weights, (const uint32_t*)t->top)
    for (size_t first = 0; first < n8; first += slab) {
        const size_t cnt = n8 - first < slab ? n8 - first : slab;
        const unsigned blocks = (unsigned)((cnt + 127) / 128);
        PLK_VANISH(0);
        if (!PLK_VANISH_MERGE_RESCUE) PLK_VANISH(1);
        PLK_VANISH(2);
        PLK_VANISH(3);
        PLK_VANISH(4);
    }
#undef PLK_VANISH
    const hipError_t e = hipGetLastError();
    scratch_release(weights, stream);
    scratch_release(part, stream);
    if (e != hipSuccess) return set_error(PLK_ERR_HIP, "vanishing points launch failed: %s", hipGetErrorString(e));
    // the tables stay alive in the cache (plk_ntt_clear_cache / plk_shutdown drop them after a device synchronisation)
    return PLK_OK;
}

static int fill_scalars(PlonkScalars& sc, const uint64_t* k_is, const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma, const uint64_t* zeta,
                        const uint64_t* a) {
    if (!zeta || !a) return set_error(PLK_ERR_INVALID_ARG, "null InnerC constant");
    static const uint64_t zero4[4] = {0, 0, 0, 0};
    for (int j = 0; j < NUM_ROUTED_WIRES; ++j) put_words(sc.k_is[j], k_is ? k_is + 4 * j : zero4);
    put_words(sc.alpha, alpha ? alpha : zero4);
    put_words(sc.beta, beta ? beta : zero4);
    put_words(sc.gamma, gamma ? gamma : zero4);
    put_words(sc.zeta, zeta);
    put_words(sc.a, a);
    return PLK_OK;
}

int plonk_vanishing_points_dev_impl(int field, unsigned log_degree, const void* d_constants, const void* d_wires, const void* d_s_sigma, const void* d_z,
                                    const uint64_t* k_is, const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma, const uint64_t* inner_zeta,
                                    const uint64_t* inner_a, void* d_out, hipStream_t stream) {
    if (!d_constants || !d_wires || !d_s_sigma || !d_z || !d_out) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    if (!k_is || !alpha || !beta || !gamma) return set_error(PLK_ERR_INVALID_ARG, "null challenge / shift pointer");
    if (log_degree + 3 > 30) return set_error(PLK_ERR_TWO_ADICITY, "log_degree %u too large", log_degree);
    PLK_TRY(ensure_device());
    PlonkScalars sc;
    PLK_TRY(fill_scalars(sc, k_is, alpha, beta, gamma, inner_zeta, inner_a));
    switch (field) {
        case PLK_FIELD_TWEEDLEDEE_BASE: return vanishing_points_t<TweedledeeBaseParams>(log_degree, d_constants, d_wires, d_s_sigma, d_z, sc, d_out, stream);
        case PLK_FIELD_TWEEDLEDUM_BASE: return vanishing_points_t<TweedledumBaseParams>(log_degree, d_constants, d_wires, d_s_sigma, d_z, sc, d_out, stream);
        case PLK_FIELD_BLS12_377_SCALAR: return vanishing_points_t<Bls12377ScalarParams>(log_degree, d_constants, d_wires, d_s_sigma, d_z, sc, d_out, stream);
        case PLK_FIELD_PALLAS_BASE: return vanishing_points_t<PallasBaseParams>(log_degree, d_constants, d_wires, d_s_sigma, d_z, sc, d_out, stream);
        case PLK_FIELD_VESTA_BASE: return vanishing_points_t<VestaBaseParams>(log_degree, d_constants, d_wires, d_s_sigma, d_z, sc, d_out, stream);
    }
    return set_error(PLK_ERR_INVALID_ARG, "field %d is not a circuit scalar field", field);
}

template <class P>
static int all_constraints_t(size_t count, const void* d_constants, const void* d_local, const void* d_right, const void* d_below, const PlonkScalars& sc,
                             void* d_out, hipStream_t stream) {
    std::shared_ptr<PlonkTables> t;
    PLK_TRY(get_plonk_tables<P>(7, stream, t));  // only the small inverses are used: any circuit size serves
    k_all_constraints<P><<<(unsigned)((count + 127) / 128), 128, 0, stream>>>((const uint4*)d_constants, (const uint4*)d_local, (const uint4*)d_right,
                                                                             (const uint4*)d_below, (const uint4*)t->small, sc, count, (uint4*)d_out);
    PLK_HIP_TRY(hipGetLastError());
    return PLK_OK;
}

int plonk_all_constraints_dev_impl(int field, size_t count, const void* d_constants, const void* d_local, const void* d_right, const void* d_below,
                                   const uint64_t* inner_zeta, const uint64_t* inner_a, void* d_out, hipStream_t stream) {
    if (count == 0) return PLK_OK;
    if (!d_constants || !d_local || !d_right || !d_below || !d_out) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    PLK_TRY(ensure_device());
    PlonkScalars sc;
    PLK_TRY(fill_scalars(sc, nullptr, nullptr, nullptr, nullptr, inner_zeta, inner_a));
    switch (field) {
        case PLK_FIELD_TWEEDLEDEE_BASE: return all_constraints_t<TweedledeeBaseParams>(count, d_constants, d_local, d_right, d_below, sc, d_out, stream);
        case PLK_FIELD_TWEEDLEDUM_BASE: return all_constraints_t<TweedledumBaseParams>(count, d_constants, d_local, d_right, d_below, sc, d_out, stream);
        case PLK_FIELD_BLS12_377_SCALAR: return all_constraints_t<Bls12377ScalarParams>(count, d_constants, d_local, d_right, d_below, sc, d_out, stream);
        case PLK_FIELD_PALLAS_BASE: return all_constraints_t<PallasBaseParams>(count, d_constants, d_local, d_right, d_below, sc, d_out, stream);
        case PLK_FIELD_VESTA_BASE: return all_constraints_t<VestaBaseParams>(count, d_constants, d_local, d_right, d_below, sc, d_out, stream);
    }
    return set_error(PLK_ERR_INVALID_ARG, "field %d is not a circuit scalar field", field);
}

}  // namespace plk
