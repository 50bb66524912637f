// oracle/sanitize_main.cpp -- the CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (the
// reference's `cargo test` runs with debug assertions and overflow checks; this is the restatement's equivalent).  Test
// infrastructure: built and run by tests/test_oracle_sanitize.py, never linked into the product.
// A battery over every family of entry points the parity tests lean on: field sweeps, NTT round trips (several threads),
// MSM with tables (serial and parallel, edge scalars, duplicate / identity bases), affine summation edge cases, batch
// inversion, polynomial division, the gates and the byte encodings.  Exits 0 when every self-check holds; the sanitizers
// abort on the first out-of-bounds access, overflow on a signed type, misaligned load or shift past the width.
#include "plk_oracle.cpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "sanitize_main: check failed at line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
    // fields: binary / unary ops on seeded inputs, inverse(x) * x == 1, batch inverse against single inverses
    for (int f : {0, 1, 2, 3, 4, 5}) {
        const int L = orc_field_limbs(f);
        const size_t n = 97;
        std::vector<u64> a(n * L), b(n * L), c(n * L), d(n * L), one(L);
        REQUIRE(orc_rand_field(f, 1000 + f, n, a.data()) == 0);
        REQUIRE(orc_rand_field(f, 2000 + f, n, b.data()) == 0);
        for (int op = 0; op < 3; ++op) REQUIRE(orc_field_binop(f, op, a.data(), b.data(), c.data(), n) == 0);
        REQUIRE(orc_field_unop(f, 2, a.data(), c.data(), n) == 0);  // inverse
        REQUIRE(orc_field_binop(f, 2, a.data(), c.data(), d.data(), n) == 0);
        REQUIRE(orc_field_const(f, 1, one.data()) == 0);
        for (size_t i = 0; i < n; ++i) REQUIRE(memcmp(&d[i * L], one.data(), L * 8) == 0);
        REQUIRE(orc_batch_inverse(f, a.data(), d.data(), n) == 0);
        REQUIRE(memcmp(c.data(), d.data(), n * L * 8) == 0);
    }
    // NTT: forward then inverse, odd thread counts, padded entry point, table layers
    for (int f : {0, 1, 2, 4, 5}) {
        for (size_t n : {(size_t)1, (size_t)2, (size_t)64, (size_t)4096}) {
            void* h = orc_fft_precompute(f, n);
            REQUIRE(h != nullptr);
            std::vector<u64> x(n * 4), y(n * 4), z(n * 4);
            REQUIRE(orc_rand_field(f, 77 + n, n, x.data()) == 0);
            REQUIRE(orc_fft(h, 1, x.data(), n, y.data(), 3) == 0);
            REQUIRE(orc_fft(h, 2, y.data(), n, z.data(), 5) == 0);
            REQUIRE(memcmp(x.data(), z.data(), n * 32) == 0);
            if (n >= 64) {
                std::vector<u64> p(n * 4);
                REQUIRE(orc_fft(h, 0, x.data(), n - 13, p.data(), 2) == 0);
                std::vector<u64> layer(n * 4);
                unsigned top = 0;
                while (((size_t)1 << top) < n) ++top;
                REQUIRE(orc_fft_table_layer(h, top, layer.data()) == 0);
            }
            REQUIRE(orc_fft_free(h) == 0);
        }
    }
    // polynomial division by Z_H of an exact multiple
    {
        const size_t nq = 16, len = 8 * nq;
        std::vector<u64> q(7 * nq * 4), m(len * 4, 0), out(len * 4);
        REQUIRE(orc_rand_field(0, 5, 7 * nq, q.data()) == 0);
        // m = q * (X^nq - 1): m[i] = q[i - nq] - q[i]
        std::vector<u64> hi(len * 4, 0), lo(len * 4, 0);
        memcpy(&hi[nq * 4], q.data(), 7 * nq * 32);
        memcpy(lo.data(), q.data(), 7 * nq * 32);
        REQUIRE(orc_field_binop(0, 1, hi.data(), lo.data(), m.data(), len) == 0);
        size_t out_len = 0;
        REQUIRE(orc_poly_divide_by_z_h(0, m.data(), len, nq, out.data(), &out_len, 3) == 0);
        REQUIRE(out_len >= 7 * nq && memcmp(out.data(), q.data(), 7 * nq * 32) == 0);
    }
    // MSM: tables, serial against parallel, edge scalars (0, 1, r - 1), duplicate and identity bases, length mismatch
    for (int curve : {0, 1, 2, 3, 4}) {
        const int L = curve == 2 ? 6 : 4;
        const size_t n = 131;
        std::vector<u64> g(2 * L), d(2 * L), bases(n * 2 * L), s(n * 4), r1(2 * L), r2(2 * L);
        std::vector<uint8_t> zero(n, 0);
        REQUIRE(orc_curve_generator(curve, g.data()) == 0);
        uint8_t dz = 0;
        std::vector<u64> seven(4, 0);
        {
            const int sf = curve == 0 ? 1 : curve == 1 ? 0 : curve == 2 ? 2 : curve == 3 ? 5 : 4;
            std::vector<u64> rnd(4);
            REQUIRE(orc_rand_field(sf, 99, 1, rnd.data()) == 0);
            REQUIRE(orc_curve_op(curve, 2, g.data(), 0, rnd.data(), 0, d.data(), &dz) == 0);  // scalar multiplication
            REQUIRE(orc_rand_field(sf, 100 + curve, n, s.data()) == 0);
            std::vector<u64> one(4), zero_s(4, 0), neg(4);
            REQUIRE(orc_field_const(sf, 1, one.data()) == 0);
            REQUIRE(orc_field_unop(sf, 0, one.data(), neg.data(), 1) == 0);  // -1 = r - 1
            memcpy(&s[0], zero_s.data(), 32);
            memcpy(&s[4], one.data(), 32);
            memcpy(&s[8], neg.data(), 32);
        }
        REQUIRE(orc_gen_bases(curve, n, g.data(), d.data(), bases.data()) == 0);
        memcpy(&bases[5 * 2 * L], &bases[6 * 2 * L], 2 * L * 8);  // a duplicate base
        zero[9] = 1;                                               // an identity base
        for (unsigned w : {1u, 5u, 11u}) {
            void* h = orc_msm_precompute(curve, n, bases.data(), zero.data(), w, 3);
            REQUIRE(h != nullptr);
            uint8_t z1 = 0, z2 = 0;
            REQUIRE(orc_msm_execute(h, s.data(), n, 0, 1, r1.data(), &z1, nullptr) == 0);
            REQUIRE(orc_msm_execute(h, s.data(), n, 1, 4, r2.data(), &z2, nullptr) == 0);
            REQUIRE(z1 == z2 && memcmp(r1.data(), r2.data(), 2 * L * 8) == 0);
            REQUIRE(orc_msm_execute(h, s.data(), n - 1, 1, 2, r2.data(), &z2, nullptr) == -2);
            REQUIRE(orc_msm_free(h) == 0);
        }
        // summation edge cases {G, G}, {G, 2G}, {G, G, G}, {} (curve_summations.rs:164-184) in every mode
        std::vector<u64> pts(3 * 2 * L);
        for (int k = 0; k < 3; ++k) memcpy(&pts[k * 2 * L], g.data(), 2 * L * 8);
        for (int mode = 0; mode < 3; ++mode)
            for (size_t cnt : {(size_t)0, (size_t)2, (size_t)3}) {
                uint8_t oz = 0;
                REQUIRE(orc_affine_summation(curve, mode, cnt, pts.data(), nullptr, r1.data(), &oz) == 0);
                REQUIRE((cnt == 0) == (oz != 0));
            }
        // byte encodings round trip
        std::vector<uint8_t> rec(n * (1 + L * 8)), st(n), oz(n);
        std::vector<u64> back(n * 2 * L);
        REQUIRE(orc_point_to_bytes(curve, bases.data(), zero.data(), n, rec.data()) == 0);
        REQUIRE(orc_point_from_bytes(curve, rec.data(), n, back.data(), oz.data(), st.data()) == 0);
        for (size_t i = 0; i < n; ++i) REQUIRE(st[i] == 0 && oz[i] == zero[i] && (zero[i] || memcmp(&back[i * 2 * L], &bases[i * 2 * L], 2 * L * 8) == 0));
    }
    // gates: every gate, filtered and unfiltered, on random rows; the vanishing points of a tiny circuit
    for (int f : {0, 1, 2}) {
        std::vector<u64> k(6 * 4), l(9 * 4), r(9 * 4), b(9 * 4), zeta(4), a(4), out(8 * 4);
        REQUIRE(orc_rand_field(f, 1, 6, k.data()) == 0);
        REQUIRE(orc_rand_field(f, 2, 9, l.data()) == 0);
        REQUIRE(orc_rand_field(f, 3, 9, r.data()) == 0);
        REQUIRE(orc_rand_field(f, 4, 9, b.data()) == 0);
        REQUIRE(orc_rand_field(f, 5, 1, zeta.data()) == 0);
        REQUIRE(orc_rand_field(f, 6, 1, a.data()) == 0);
        for (int gate = 0; gate < 10; ++gate)
            for (int unf = 0; unf < 2; ++unf) {
                size_t n_out = 0;
                REQUIRE(orc_gate_constraints(f, gate, unf, k.data(), l.data(), r.data(), b.data(), zeta.data(), a.data(), out.data(), &n_out) == 0 && n_out <= 8);
            }
        const size_t deg = 8, n8 = 64;
        std::vector<u64> cs(6 * n8 * 4), ws(9 * n8 * 4), ss(6 * n8 * 4), z(n8 * 4), kis(6 * 4), sc(3 * 4), vo(n8 * 4);
        REQUIRE(orc_rand_field(f, 11, 6 * n8, cs.data()) == 0);
        REQUIRE(orc_rand_field(f, 12, 9 * n8, ws.data()) == 0);
        REQUIRE(orc_rand_field(f, 13, 6 * n8, ss.data()) == 0);
        REQUIRE(orc_rand_field(f, 14, n8, z.data()) == 0);
        REQUIRE(orc_rand_field(f, 15, 6, kis.data()) == 0);
        REQUIRE(orc_rand_field(f, 16, 3, sc.data()) == 0);
        REQUIRE(orc_vanishing_points(f, deg, cs.data(), ws.data(), ss.data(), z.data(), kis.data(), &sc[0], &sc[4], &sc[8], zeta.data(), a.data(), vo.data(), 3) == 0);
    }
    printf("sanitize_main: ok\n");
    return 0;
}
