// tests/capi_host.cpp -- a host program above the C ABI only (no HIP, no Python): what the Rust shim of INTEGRATION.md does.
//
// Nine host threads call the hot path at once, exactly as the reference's Rayon workers do (src/plonk_util.rs:173-189: nine
// wire polynomials are transformed / committed in parallel):
//   * every thread: fft_with_precomputation_power_of_2 and its inverse on its own vector (round trip, bit exact), one
//     msm_execute_parallel against the SHARED MsmPrecomputation, one msm_precompute + msm_execute + free of its own;
//   * the results of the concurrent phase equal those of the same calls made one after the other;
//   * linearity through the ABI: msm(s0 + s1) = msm(s0) + msm(s1)  (plk_field_op add in the scalar field, plk_curve_sum_affine);
//   * one inner-product argument (plk_halo_*): its first L / R against one-shot MSMs, its final vectors against plain sums.
// Generators are real curve points made through the ABI itself: [2^j] G from plk_msm_precompute_table, spread by
// plk_curve_fold_pairs.  Exit code 0 and the line "capi_host: OK" on success.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/plonky_hip.h"

#define CHECK(expr)                                                                            \
    do {                                                                                       \
        int _rc = (expr);                                                                      \
        if (_rc != PLK_OK) {                                                                   \
            std::fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #expr, _rc, plk_last_error()); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

static uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// any 4-limb value below 2^254 is a fully reduced Montgomery representation for the Tweedle fields (p > 2^254)
static void rand_elems(std::vector<uint64_t>& v, size_t n, uint64_t seed) {
    v.resize(n * 4);
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) v[4 * i + k] = splitmix(seed);
        v[4 * i + 3] &= 0x3FFFFFFFFFFFFFFFull;
    }
}

// `capi_host multi`: the same host program over a device group.  Nine 2^16 scalar vectors are committed against 2^16 generators
// (coeffs_vec_to_commitments, poly_commit.rs:52-66) and nine polynomials transformed (polynomials_to_values_padded,
// plonk_util.rs:179-190) first with ONE device (plk_init) and then with the group plk_init_devices(0) gives - PLK_NGPU real
// devices, or PLK_VIRTUAL_DEVICES logical ones on a one-GPU box - through the very same calls; the results must agree bit for bit.
static int run_multi() {
    const int CURVE = PLK_CURVE_TWEEDLEDEE, BASE = PLK_FIELD_TWEEDLEDEE_BASE;
    const unsigned LOG_N = 16, BATCH = 9;
    const size_t n = (size_t)1 << LOG_N;
    CHECK(plk_init(0));
    const uint64_t p_minus_1[4] = {0x842cafd400000000ull, 0x038aa127696286c9ull, 0, 0x4000000000000000ull};
    uint64_t canon[8] = {p_minus_1[0], p_minus_1[1], p_minus_1[2], p_minus_1[3], 2, 0, 0, 0}, g[8];
    CHECK(plk_field_op(BASE, 7, canon, nullptr, g, 2));
    const int digits = plk_msm_table_digits(CURVE, 1);
    std::vector<uint64_t> tab((size_t)digits * 8);
    std::vector<uint8_t> tab_zero(digits);
    CHECK(plk_msm_precompute_table(CURVE, 1, g, nullptr, 1, tab.data(), tab_zero.data()));
    const size_t m = (size_t)digits - 1, rounds = (n + m - 1) / m;
    std::vector<uint64_t> bases(rounds * m * 8), ab;
    std::vector<uint8_t> bz(rounds * m);
    rand_elems(ab, 2 * rounds, 0xB45E5);
    for (size_t r = 0; r < rounds; ++r)
        CHECK(plk_curve_fold_pairs(CURVE, m, tab.data(), nullptr, tab.data() + 8, nullptr, ab.data() + 8 * r, ab.data() + 8 * r + 4, bases.data() + r * m * 8,
                                   bz.data() + r * m));
    std::vector<std::vector<uint64_t>> s(BATCH), x(BATCH);
    std::vector<const uint64_t*> sp(BATCH), xp(BATCH);
    std::vector<size_t> xl(BATCH);
    for (unsigned b = 0; b < BATCH; ++b) {
        rand_elems(s[b], n, 3000 + b);
        rand_elems(x[b], n / 8, 4000 + b);
        sp[b] = s[b].data();
        xp[b] = x[b].data();
        xl[b] = n / 8 - b;  // ragged lengths
    }
    struct Result {
        std::vector<uint64_t> commit, single;
        std::vector<uint8_t> cz;
        std::vector<std::vector<uint64_t>> lde, ntt;
        uint8_t sz = 0;
    } one, grp;
    auto run = [&](Result& r) -> int {
        plk_msm_ctx* ctx = nullptr;
        CHECK(plk_msm_precompute(CURVE, n, bases.data(), nullptr, 0, &ctx));
        r.commit.assign(BATCH * 8, 0);
        r.cz.assign(BATCH, 0);
        CHECK(plk_msm_execute_batch(ctx, BATCH, sp.data(), n, r.commit.data(), r.cz.data()));
        r.single.assign(8, 0);
        CHECK(plk_msm_execute(ctx, sp[4], n, r.single.data(), &r.sz));
        CHECK(plk_msm_free(ctx));
        r.lde.assign(BATCH, std::vector<uint64_t>(n * 4));
        r.ntt.assign(BATCH, std::vector<uint64_t>(n * 4));
        std::vector<uint64_t*> lp(BATCH), np_(BATCH);
        for (unsigned b = 0; b < BATCH; ++b) {
            lp[b] = r.lde[b].data();
            np_[b] = r.ntt[b].data();
        }
        CHECK(plk_ntt_padded_batch(BASE, LOG_N, BATCH, xp.data(), xl.data(), lp.data()));
        CHECK(plk_ntt_batch(BASE, LOG_N, 0, BATCH, sp.data(), np_.data()));  // any 4-limb data below p is a polynomial
        // nine host threads, one transform each - the reference's par_iter
        std::vector<std::vector<uint64_t>> th_out(BATCH, std::vector<uint64_t>(n * 4));
        std::vector<int> rc(BATCH, 0);
        std::vector<std::thread> pool;
        for (unsigned b = 0; b < BATCH; ++b) pool.emplace_back([&, b] { rc[b] = plk_ntt(BASE, LOG_N, 0, sp[b], th_out[b].data()); });
        for (auto& t : pool) t.join();
        for (unsigned b = 0; b < BATCH; ++b)
            if (rc[b] != PLK_OK || th_out[b] != r.ntt[b]) {
                std::fprintf(stderr, "multi: transform %u from its own thread differs (rc %d)\n", b, rc[b]);
                return 1;
            }
        return 0;
    };
    if (run(one)) return 1;
    plk_shutdown();
    CHECK(plk_init_devices(0));
    const int world = plk_device_count();
    if (world < 2) {
        std::fprintf(stderr, "multi: only %d device in the group (set PLK_VIRTUAL_DEVICES=2 on a one-GPU box)\n", world);
        return 1;
    }
    if (run(grp)) return 1;
    if (one.commit != grp.commit || one.cz != grp.cz || one.single != grp.single || one.sz != grp.sz || one.lde != grp.lde || one.ntt != grp.ntt) {
        std::fprintf(stderr, "multi: the device group's results differ from one device's\n");
        return 1;
    }
    if (std::memcmp(grp.single.data(), grp.commit.data() + 4 * 8, 64)) {
        std::fprintf(stderr, "multi: the sharded single MSM differs from its slot of the batch\n");
        return 1;
    }
    plk_shutdown();
    std::printf("capi_host multi: OK (%d devices, nine 2^%u-scalar commitments, nine transforms, bit-identical to one device)\n", world, LOG_N);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !std::strcmp(argv[1], "multi")) return run_multi();
    const int CURVE = PLK_CURVE_TWEEDLEDEE, BASE = PLK_FIELD_TWEEDLEDEE_BASE;
    const int SCALAR = plk_curve_scalar_field(CURVE);
    const unsigned THREADS = 9, LOG_N = 14;
    CHECK(plk_init(0));
    // G = (-1, 2) (tweedledee_curve.rs:14-18) in Montgomery form: from_canonical through the device
    const uint64_t p_minus_1[4] = {0x842cafd400000000ull, 0x038aa127696286c9ull, 0, 0x4000000000000000ull};  // tweedledee_base.rs:22 minus one
    uint64_t canon[8] = {p_minus_1[0], p_minus_1[1], p_minus_1[2], p_minus_1[3], 2, 0, 0, 0}, g[8];
    CHECK(plk_field_op(BASE, 7, canon, nullptr, g, 2));
    // [2^j] G, j < 255, then 16 spreads [a] T_j + [b] T_(j+1): 16 * 254 distinct valid points
    const int digits = plk_msm_table_digits(CURVE, 1);
    std::vector<uint64_t> tab((size_t)digits * 8);
    std::vector<uint8_t> tab_zero(digits);
    CHECK(plk_msm_precompute_table(CURVE, 1, g, nullptr, 1, tab.data(), tab_zero.data()));
    const size_t m = (size_t)digits - 1, n = 16 * m;
    std::vector<uint64_t> bases(n * 8), ab;
    std::vector<uint8_t> bz(n);
    rand_elems(ab, 32, 0xB45E5);
    for (int r = 0; r < 16; ++r)
        CHECK(plk_curve_fold_pairs(CURVE, m, tab.data(), nullptr, tab.data() + 8, nullptr, ab.data() + 8 * r, ab.data() + 8 * r + 4, bases.data() + r * m * 8,
                                   bz.data() + r * m));
    for (uint8_t z : bz)
        if (z) {
            std::fprintf(stderr, "unexpected identity among the generators\n");
            return 1;
        }
    plk_msm_ctx* shared = nullptr;
    CHECK(plk_msm_precompute(CURVE, n, bases.data(), nullptr, 0, &shared));
    CHECK(plk_ntt_precompute(BASE, LOG_N));

    struct Job {
        std::vector<uint64_t> x, y, back, s, own_bases;
        uint64_t r_shared[8], r_own[8];
        uint8_t z_shared = 0, z_own = 0;
        int rc = 0;
        const char* err = "";
    };
    std::vector<Job> jobs(THREADS), seq(THREADS);
    for (unsigned t = 0; t < THREADS; ++t) {
        rand_elems(jobs[t].x, (size_t)1 << LOG_N, 1000 + t);
        rand_elems(jobs[t].s, n, 2000 + t);
        jobs[t].own_bases.assign(bases.begin() + t * 64 * 8, bases.begin() + (t * 64 + 1024) * 8);
        seq[t] = jobs[t];
    }
    auto work = [&](Job& j) {
        const size_t nn = (size_t)1 << LOG_N;
        j.y.resize(nn * 4);
        j.back.resize(nn * 4);
        auto fail = [&](int rc) {
            j.rc = rc;
            j.err = plk_last_error();
        };
        int rc;
        if ((rc = plk_ntt(BASE, LOG_N, 0, j.x.data(), j.y.data())) != PLK_OK) return fail(rc);
        if ((rc = plk_ntt(BASE, LOG_N, 1, j.y.data(), j.back.data())) != PLK_OK) return fail(rc);
        if ((rc = plk_msm_execute(shared, j.s.data(), n, j.r_shared, &j.z_shared)) != PLK_OK) return fail(rc);
        plk_msm_ctx* own = nullptr;
        if ((rc = plk_msm_precompute(CURVE, 1024, j.own_bases.data(), nullptr, 0, &own)) != PLK_OK) return fail(rc);
        rc = plk_msm_execute(own, j.s.data(), 1024, j.r_own, &j.z_own);
        plk_msm_free(own);
        if (rc != PLK_OK) return fail(rc);
    };
    {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < THREADS; ++t) pool.emplace_back(work, std::ref(jobs[t]));
        for (auto& th : pool) th.join();
    }
    for (unsigned t = 0; t < THREADS; ++t) work(seq[t]);
    for (unsigned t = 0; t < THREADS; ++t) {
        const Job &a = jobs[t], &b = seq[t];
        if (a.rc || b.rc) {
            std::fprintf(stderr, "thread %u: rc %d / %d: %s %s\n", t, a.rc, b.rc, a.err, b.err);
            return 1;
        }
        if (a.back != a.x || a.y == a.x) {
            std::fprintf(stderr, "thread %u: NTT round trip failed\n", t);
            return 1;
        }
        if (a.y != b.y || std::memcmp(a.r_shared, b.r_shared, 64) || std::memcmp(a.r_own, b.r_own, 64) || a.z_shared != b.z_shared || a.z_own != b.z_own) {
            std::fprintf(stderr, "thread %u: concurrent result differs from the sequential one\n", t);
            return 1;
        }
        if (a.z_shared || a.z_own) {
            std::fprintf(stderr, "thread %u: unexpected identity result\n", t);
            return 1;
        }
    }
    // linearity: msm(s0 + s1) == msm(s0) + msm(s1)
    std::vector<uint64_t> s01(n * 4);
    CHECK(plk_field_op(SCALAR, 0, jobs[0].s.data(), jobs[1].s.data(), s01.data(), n));
    uint64_t lhs[8], rhs[8], pair[16];
    uint8_t lz = 0, rz = 0;
    CHECK(plk_msm_execute(shared, s01.data(), n, lhs, &lz));
    std::memcpy(pair, jobs[0].r_shared, 64);
    std::memcpy(pair + 8, jobs[1].r_shared, 64);
    CHECK(plk_curve_sum_affine(CURVE, 2, pair, nullptr, rhs, &rz));
    if (lz != rz || std::memcmp(lhs, rhs, 64)) {
        std::fprintf(stderr, "linearity check failed\n");
        return 1;
    }
    // contract violation -> error code, not a crash (curve_msm.rs:67 assert_eq!)
    if (plk_msm_execute(shared, s01.data(), n - 1, lhs, &lz) != PLK_ERR_SIZE_MISMATCH) {
        std::fprintf(stderr, "length mismatch not reported\n");
        return 1;
    }
    // the inner-product argument (halo.rs:63-124) through the ABI, checked through the ABI: halo_b = 0 makes both inner products 0, so
    // L_1 = <a_lo, G_hi> + [l] H and R_1 = <a_hi, G_lo> + [r] H are one-shot plk_msm calls; with the challenges u_j = 1 the folds are sums:
    // final halo_a = sum a_i (plk_field_op add, halving), final halo_g = sum G_i (plk_curve_sum_affine).  2048 generators, frozen below 2^8.
    {
        const size_t nh = 2048, half = nh / 2;
        std::vector<uint64_t> ha, hb(nh * 4, 0), one_c(4, 0), one(4);
        rand_elems(ha, nh, 0x1A10);
        one_c[0] = 1;
        CHECK(plk_field_op(SCALAR, 7, one_c.data(), nullptr, one.data(), 1));  // 1 in Montgomery form
        const uint64_t* H = bases.data() + nh * 8;         // two more valid points of the list
        const uint64_t* U = bases.data() + (nh + 1) * 8;
        plk_halo_ctx* hc = nullptr;
        CHECK(plk_halo_begin(CURVE, nh, ha.data(), hb.data(), bases.data(), nullptr, H, U, 8, &hc));
        uint64_t lr[16], lbl[4], rbl[4];
        uint8_t lrz[2];
        uint64_t seed = 77;
        for (int k = 0; k < 4; ++k) {
            lbl[k] = splitmix(seed);
            rbl[k] = splitmix(seed);
        }
        lbl[3] &= 0x3FFFFFFFFFFFFFFFull;
        rbl[3] &= 0x3FFFFFFFFFFFFFFFull;
        CHECK(plk_halo_round_lr(hc, lbl, rbl, lr, lrz));
        // the same two points by msm_parallel over [half of G, H]
        std::vector<uint64_t> pts((half + 1) * 8), sc((half + 1) * 4);
        uint64_t want[8];
        uint8_t wz = 0;
        for (int side = 0; side < 2; ++side) {
            std::memcpy(pts.data(), bases.data() + (side == 0 ? half : 0) * 8, half * 64);           // L: G_hi, R: G_lo
            std::memcpy(pts.data() + half * 8, H, 64);
            std::memcpy(sc.data(), ha.data() + (side == 0 ? 0 : half) * 4, half * 32);              // L: a_lo, R: a_hi
            std::memcpy(sc.data() + half * 4, side == 0 ? lbl : rbl, 32);
            CHECK(plk_msm(CURVE, half + 1, pts.data(), nullptr, sc.data(), want, &wz));
            if (wz != lrz[side] || std::memcmp(want, lr + 8 * side, 64)) {
                std::fprintf(stderr, "halo: %s_1 differs from the one-shot MSM\n", side == 0 ? "L" : "R");
                return 1;
            }
        }
        size_t len = nh;
        while (len > 1) {
            if (len != nh) CHECK(plk_halo_round_lr(hc, lbl, rbl, lr, lrz));
            CHECK(plk_halo_round_fold(hc, one.data(), one.data()));
            len /= 2;
            if (plk_halo_len(hc) != len) {
                std::fprintf(stderr, "halo: length %zu after a fold, expected %zu\n", plk_halo_len(hc), len);
                return 1;
            }
        }
        uint64_t a0[4], b0[4], g0[8], a_sum[4], g_sum[8];
        uint8_t gz0 = 0, gsz = 0;
        CHECK(plk_halo_read(hc, a0, b0, g0, &gz0));
        CHECK(plk_halo_free(hc));
        std::vector<uint64_t> acc(ha);
        for (size_t l = nh / 2; l >= 1; l /= 2) CHECK(plk_field_op(SCALAR, 0, acc.data(), acc.data() + l * 4, acc.data(), l));  // pairwise sums, in place
        std::memcpy(a_sum, acc.data(), 32);
        CHECK(plk_curve_sum_affine(CURVE, nh, bases.data(), nullptr, g_sum, &gsz));
        const uint64_t zero4[4] = {0, 0, 0, 0};
        if (std::memcmp(a0, a_sum, 32) || std::memcmp(b0, zero4, 32) || gz0 != gsz || std::memcmp(g0, g_sum, 64)) {
            std::fprintf(stderr, "halo: the final vectors differ from the sums\n");
            return 1;
        }
    }
    plk_msm_free(shared);
    plk_shutdown();
    std::printf("capi_host: OK (%u threads, 2^%u-point transforms, %zu-generator MSMs)\n", THREADS, LOG_N, n);
    return 0;
}
