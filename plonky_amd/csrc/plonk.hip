// plonk.hip -- the Plonk quotient numerator on the device.
//
// Reference path                                                              here
//   Prover::vanishing_poly, the 8n-point loop      plonk.rs:392-453      ->  k_vanishing_points
//   evaluate_all_constraints                       gates/mod.rs:46-125   ->  all_constraints()
//   Gate::evaluate_filtered / evaluate_prefix_filter  gates/mod.rs:262-300 -> the shared prefix products below
//   the ten gates' evaluate_unfiltered             gates/*.rs            ->  gate by gate below (file:line at each)
//   eval_l_1                                       plonk_util.rs:14-24   ->  a cached table of L_1 over the 8n domain
//   mds_matrix                                     mds.rs:56-77          ->  seven inverses 1/1 .. 1/7 (Cauchy entries 1/(4 + r - c))
//   reduce_with_powers                             plonk_util.rs:27-33   ->  Horner at the end of the kernel
// The reference evaluates the 8n points with Rayon, one full field inversion per point for L_1(x) and a freshly
// cloned MDS matrix per gate; here a point is one lane, everything that depends only on the circuit size (the powers
// of the 8n-th root, L_1 over the domain, the MDS entries) is a cached table, and the inversions behind L_1 are
// batched eight to a lane when that table is built.  The closing Polynomial::from_evaluations (plonk.rs:455) is the
// inverse NTT of ntt.hip.  Every value is a fully reduced field element computed by the same field operations as the
// reference, so the result is bit-identical (field arithmetic is exact and associative; only the grouping differs).
// get_subgroup_shift (partition.rs:140-153) draws its k_i from ChaCha8: they are inputs here.
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

// a field element with operators: keeps the gate formulas readable next to the reference's
template <class P> struct El {
    Fe<P> v;
    PLK_DI El operator+(const El& o) const { return El{fe_add<P>(v, o.v)}; }
    PLK_DI El operator-(const El& o) const { return El{fe_sub<P>(v, o.v)}; }
    PLK_DI El operator*(const El& o) const { return El{fe_mul<P>(v, o.v)}; }
    PLK_DI El sq() const { return El{fe_sqr<P>(v)}; }
    PLK_DI El dbl() const { return El{fe_dbl<P>(v)}; }  // field.rs:181-183 multiplies by TWO: the same value
    PLK_DI El quad() const { return dbl().dbl(); }       // field.rs:191-193
    PLK_DI El pow5() const {                             // exp_usize(5), rescue_a.rs:58, rescue_b.rs:44
        const El x2 = sq();
        return x2.sq() * *this;
    }
};
template <class P> PLK_DI El<P> el_load(const uint4* p, size_t i) { return El<P>{fe_load<P>(p + i * 2)}; }
template <class P> PLK_DI El<P> el_one() { return El<P>{fe_one<P>()}; }
template <class P> PLK_DI El<P> el_zero() { return El<P>{fe_zero<P>()}; }

// circuit-size tables (device, field, log_degree): everything the loop needs that does not depend on the proof
struct PlonkTables {
    void* xs_lo = nullptr;  // g^j, j < 1024 (g = primitive 8n-th root, circuit_builder.rs:1122)
    void* xs_hi = nullptr;  // g^(1024 j)
    void* l1 = nullptr;     // L_1(g^i), i < 8n  (plonk_util.rs:14-24)
    void* small = nullptr;  // [0..7]: 1/1 .. 1/7 then unused; MDS entry (r, c) = 1 / (4 + r - c)  (mds.rs:63-77)
    ~PlonkTables() {
        for (void* p : {xs_lo, xs_hi, l1, small})
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
                                              uint4* __restrict__ small) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n_lo = (size_t)1 << XS_LO_LOG;
    const size_t n_hi = log_n8 > XS_LO_LOG ? (size_t)1 << (log_n8 - XS_LO_LOG) : 1;
    if (idx < n_lo) {
        const uint64_t e = (idx & (((uint64_t)1 << log_n8) - 1)) << (log_t - log_n8);
        fe_store<P>(lo + idx * 2, pow_from_table<P>(pw, 0, e, log_t));
    } else if (idx < n_lo + n_hi) {
        const uint64_t e = (((idx - n_lo) << XS_LO_LOG) & (((uint64_t)1 << log_n8) - 1)) << (log_t - log_n8);
        fe_store<P>(hi + (idx - n_lo) * 2, pow_from_table<P>(pw, 0, e, log_t));
    } else if (idx < n_lo + n_hi + 7) {
        // 1 / v for v = 1 .. 7
        const uint32_t v = (uint32_t)(idx - n_lo - n_hi) + 1;
        Fe<P> c = fe_zero<P>();
        c.v[0] = v;
        fe_store<P>(small + (v - 1) * 2, fe_inv_safegcd<P>(fe_from_canonical<P>(c)));
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
        fe_store<P>(l1 + i * 2, r);
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
    PLK_HIP_TRY(hipMalloc(&t->small, 8 * 32));
    PLK_HIP_TRY(hipMalloc(&t->l1, n8 * 32));
    const size_t cnt = n_lo + n_hi + 7;
    k_plonk_xs<P><<<(unsigned)((cnt + 127) / 128), 128, 0, stream>>>((const uint4*)pw, log_t, log_n8, (uint4*)t->xs_lo, (uint4*)t->xs_hi, (uint4*)t->small);
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
// k: local constants [6]; l: local wires [9]; r: right wires (only indices 0..3 are read by any gate); b2, b3: below wires 2, 3
// (only CurveEndoGate reads below, curve_endo.rs:115-117); small: 1/1 .. 1/7.
template <class P>
PLK_DI void all_constraints(const El<P> (&k)[NUM_CONSTANTS], const El<P> (&l)[NUM_WIRES], const El<P> (&r)[4], const El<P>& b2, const El<P>& b3,
                            const El<P>& zeta, const El<P>& a_coeff, const uint4* __restrict__ small, El<P> (&u)[NUM_TERMS]) {
    using E = El<P>;
    const E one = el_one<P>();
#pragma unroll
    for (int i = 0; i < NUM_TERMS; ++i) u[i] = el_zero<P>();
    // prefix filters (gates/mod.rs:289-300); the prefix tree of gates/mod.rs:1-16 shares its products
    E nk[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) nk[i] = one - k[i];
    const E p10 = k[0] * nk[1];          // 10
    const E p101 = p10 * k[2];           // 101
    const E p1010 = p101 * nk[3];        // 1010
    const E p1011 = p101 * k[3];         // 1011
    const E p100 = p10 * nk[2];          // 100
    const E f_curve_add = p1010 * k[4];                 // 10101   curve_add.rs:60
    const E f_curve_dbl = p1011 * k[4];                 // 10111   curve_dbl.rs:42
    const E f_curve_endo = k[0] * k[1];                 // 11      curve_endo.rs:96
    const E f_base4 = p100 * nk[3];                     // 1000    base_4_sum.rs:34
    const E f_public_input = p1010 * nk[4] * k[5];      // 101001  public_input.rs:26
    const E f_constant = p1011 * nk[4];                 // 10110   constant.rs:28
    const E f_arithmetic = p100 * k[3];                 // 1001    arithmetic.rs:30
    const E f_rescue_a = nk[0] * nk[1];                 // 00      rescue_a.rs:38
    const E f_rescue_b = nk[0] * k[1];                  // 01      rescue_b.rs:30
    // (BufferGate, 101010, has no constraints: buffer.rs:26-33)

    {  // CurveAddGate, curve_add.rs:62-101
        const E x1 = l[0], y1 = l[1], x4 = r[0], y4 = r[1], acc_old = l[2], acc_new = l[3], x2 = l[4], y2 = l[5], bit = l[6], inverse = l[7], lambda = l[8];
        const E computed_lambda = (y1 - y2) * inverse;
        const E x3 = lambda.sq() - x1 - x2;
        const E y3 = lambda * (x1 - x4) - y1;
        const E not_bit = one - bit;
        const E computed_x4 = bit * x3 + not_bit * x1;
        const E computed_y4 = bit * y3 + not_bit * y1;
        u[0] = u[0] + f_curve_add * (computed_lambda - lambda);
        u[1] = u[1] + f_curve_add * (computed_x4 - x4);
        u[2] = u[2] + f_curve_add * (computed_y4 - y4);
        u[3] = u[3] + f_curve_add * (acc_new - (acc_old.dbl() + bit));
        u[4] = u[4] + f_curve_add * (bit * not_bit);
        u[5] = u[5] + f_curve_add * (inverse * (x1 - x2) - one);
    }
    {  // CurveDblGate, curve_dbl.rs:44-68
        const E x_old = l[0], y_old = l[1], x_new = l[2], y_new = l[3], inverse = l[4], lambda = l[5];
        const E xx = x_old.sq();
        const E numerator = xx.dbl() + xx + a_coeff;  // square().triple() + A
        const E computed_lambda = numerator * inverse;
        const E computed_x_new = lambda.sq() - x_old.dbl();
        const E computed_y_new = lambda * (x_old - x_new) - y_old;
        u[0] = u[0] + f_curve_dbl * (computed_lambda - lambda);
        u[1] = u[1] + f_curve_dbl * (computed_x_new - x_new);
        u[2] = u[2] + f_curve_dbl * (computed_y_new - y_new);
        u[3] = u[3] + f_curve_dbl * (y_old.dbl() * inverse - one);
    }
    {  // CurveEndoGate, curve_endo.rs:98-141
        const E x1 = l[0], y1 = l[1], x_in = l[4], y_in = l[5], x3 = r[0], y3 = r[1];
        const E unsigned_old = l[2], unsigned_new = b2, signed_old = l[3], signed_new = b3, bit0 = l[6], bit1 = l[7], inverse = l[8];
        const E mult = (zeta - one) * bit1 + one;  // x2's factor and signed_limb_multiplier are the same expression
        const E x2 = mult * x_in;
        const E sgn = bit0.dbl() - one;
        const E y2 = sgn * y_in;
        const E lambda = (y1 - y2) * inverse;
        const E computed_x3 = lambda.sq() - x1 - x2;
        const E computed_y3 = lambda * (x1 - x3) - y1;
        const E signed_limb = sgn * mult;
        u[0] = u[0] + f_curve_endo * (computed_x3 - x3);
        u[1] = u[1] + f_curve_endo * (computed_y3 - y3);
        u[2] = u[2] + f_curve_endo * (unsigned_new - (unsigned_old.quad() + bit1.dbl() + bit0));
        u[3] = u[3] + f_curve_endo * (signed_new - (signed_old.dbl() + signed_limb));
        u[4] = u[4] + f_curve_endo * (bit0 * (bit0 - one));
        u[5] = u[5] + f_curve_endo * (bit1 * (bit1 - one));
        u[6] = u[6] + f_curve_endo * (inverse * (x1 - x2) - one);
    }
    {  // Base4SumGate, base_4_sum.rs:36-63: 7 limbs in wires 2..8
        E computed = l[0];
        const E two = one.dbl(), three = two + one;
#pragma unroll
        for (int i = 0; i < NUM_WIRES - 2; ++i) {
            const E limb = l[2 + i];
            computed = computed.quad() + limb;
            const E product = one * limb * (limb - one) * (limb - two) * (limb - three);  // j = 0: limb - 0
            u[1 + i] = u[1 + i] + f_base4 * product;
        }
        u[0] = u[0] + f_base4 * (computed - l[1]);
    }
    // PublicInputGate, public_input.rs:28-37: advice wires 6..8 against the right gate's wires 0..2
#pragma unroll
    for (int i = 0; i < NUM_WIRES - NUM_ROUTED_WIRES; ++i) u[i] = u[i] + f_public_input * (l[NUM_ROUTED_WIRES + i] - r[i]);
    // ConstantGate, constant.rs:30-39
    u[0] = u[0] + f_constant * (k[5] - l[0]);
    // ArithmeticGate, arithmetic.rs:32-46
    u[0] = u[0] + f_arithmetic * (k[4] * l[0] * l[1] + k[5] * l[2] - l[3]);
    {  // RescueStepAGate, rescue_a.rs:40-69 and RescueStepBGate, rescue_b.rs:32-58
        E mds[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) mds[i] = el_load<P>(small, i);  // mds[v - 1] = 1 / v; entry (r, c) = 1 / (4 + r - c)
        E exps[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) exps[i] = l[i].pow5();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u[2 * i] = u[2 * i] + f_rescue_a * (l[4 + i].pow5() - l[i]);
            E out_a = k[2 + i], out_b = k[2 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const E m = mds[4 + i - j - 1];
                out_a = out_a + m * l[4 + j];
                out_b = out_b + m * exps[j];
            }
            u[2 * i + 1] = u[2 * i + 1] + f_rescue_a * (out_a - r[i]);
            u[i] = u[i] + f_rescue_b * (out_b - r[i]);
        }
    }
}

struct PlonkScalars {
    uint32_t k_is[NUM_ROUTED_WIRES][8];
    uint32_t alpha[8], beta[8], gamma[8], zeta[8], a[8];
};
template <class P> PLK_DI El<P> el_words(const uint32_t (&w)[8]) {
    El<P> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v.v[i] = w[i];
    return r;
}

// plonk.rs:392-453, one lane per point of the 8n domain
template <class P>
__global__ void __launch_bounds__(128) k_vanishing_points(const uint4* __restrict__ constants, const uint4* __restrict__ wires, const uint4* __restrict__ s_sigma,
                                                          const uint4* __restrict__ z, const uint4* __restrict__ xs_lo, const uint4* __restrict__ xs_hi,
                                                          const uint4* __restrict__ l1, const uint4* __restrict__ small, PlonkScalars sc, int log_degree,
                                                          uint4* __restrict__ out) {
    static_assert(P::NL == 8, "256-bit scalar fields");
    using E = El<P>;
    const size_t n8 = (size_t)8 << log_degree;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const size_t i_right = (i + 8) & (n8 - 1), i_below = (i + 8 * GRID_WIDTH) & (n8 - 1);
    E k[NUM_CONSTANTS], l[NUM_WIRES], r[4];
#pragma unroll
    for (int j = 0; j < NUM_CONSTANTS; ++j) k[j] = el_load<P>(constants, (size_t)j * n8 + i);
#pragma unroll
    for (int j = 0; j < NUM_WIRES; ++j) l[j] = el_load<P>(wires, (size_t)j * n8 + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = el_load<P>(wires, (size_t)j * n8 + i_right);
    const E b2 = el_load<P>(wires, (size_t)2 * n8 + i_below), b3 = el_load<P>(wires, (size_t)3 * n8 + i_below);
    E u[NUM_TERMS];
    all_constraints<P>(k, l, r, b2, b3, el_words<P>(sc.zeta), el_words<P>(sc.a), small, u);
    const E one = el_one<P>();
    const E x = E{plonk_x<P>(xs_lo, xs_hi, i)};
    const E z_x = el_load<P>(z, i), z_gz = el_load<P>(z, i_right);
    const E z_1_term = el_load<P>(l1, i) * (z_x - one);  // plonk.rs:425
    const E beta = el_words<P>(sc.beta), gamma = el_words<P>(sc.gamma);
    E f_prime = one, g_prime = one;
#pragma unroll
    for (int j = 0; j < NUM_ROUTED_WIRES; ++j) {  // plonk.rs:428-437
        const E s_id = el_words<P>(sc.k_is[j]) * x;
        const E s_sig = el_load<P>(s_sigma, (size_t)j * n8 + i);
        f_prime = f_prime * (l[j] + beta * s_id + gamma);
        g_prime = g_prime * (l[j] + beta * s_sig + gamma);
    }
    const E v_shift_term = f_prime * z_x - g_prime * z_gz;  // plonk.rs:438
    // reduce_with_powers over [z_1_term, v_shift_term, constraint terms] (plonk.rs:440-447, plonk_util.rs:27-33)
    const E alpha = el_words<P>(sc.alpha);
    E sum = el_zero<P>();
#pragma unroll
    for (int t = NUM_TERMS - 1; t >= 0; --t) sum = sum * alpha + u[t];
    sum = sum * alpha + v_shift_term;
    sum = sum * alpha + z_1_term;
    fe_store<P>(out + i * 2, sum.v);
}

// evaluate_all_constraints at `count` independent points (constants [count][6], local / right / below [count][9], out [count][8])
template <class P>
__global__ void __launch_bounds__(128) k_all_constraints(const uint4* __restrict__ constants, const uint4* __restrict__ local, const uint4* __restrict__ right,
                                                         const uint4* __restrict__ below, const uint4* __restrict__ small, PlonkScalars sc, size_t count,
                                                         uint4* __restrict__ out) {
    using E = El<P>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    E k[NUM_CONSTANTS], l[NUM_WIRES], r[4];
#pragma unroll
    for (int j = 0; j < NUM_CONSTANTS; ++j) k[j] = el_load<P>(constants, i * NUM_CONSTANTS + j);
#pragma unroll
    for (int j = 0; j < NUM_WIRES; ++j) l[j] = el_load<P>(local, i * NUM_WIRES + j);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = el_load<P>(right, i * NUM_WIRES + j);
    E u[NUM_TERMS];
    all_constraints<P>(k, l, r, el_load<P>(below, i * NUM_WIRES + 2), el_load<P>(below, i * NUM_WIRES + 3), el_words<P>(sc.zeta), el_words<P>(sc.a), small, u);
#pragma unroll
    for (int t = 0; t < NUM_TERMS; ++t) fe_store<P>(out + (i * NUM_TERMS + t) * 2, u[t].v);
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
    k_vanishing_points<P><<<(unsigned)((n8 + 127) / 128), 128, 0, stream>>>((const uint4*)d_constants, (const uint4*)d_wires, (const uint4*)d_s_sigma,
                                                                           (const uint4*)d_z, (const uint4*)t->xs_lo, (const uint4*)t->xs_hi,
                                                                           (const uint4*)t->l1, (const uint4*)t->small, sc, (int)log_degree, (uint4*)d_out);
    PLK_HIP_TRY(hipGetLastError());
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
