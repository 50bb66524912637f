// tests/fp_host_sanitize_main.cpp -- the device arithmetic headers (fp / fp29 / fz / ecz / glv, exactly what the kernels compile)
// under AddressSanitizer + UndefinedBehaviorSanitizer on the host: every operation of the harness on edge values and seeded
// values of all six fields, the lazy point arithmetic incl. P + P and P + (-P), the GLV split.  Values are checked elsewhere
// (tests/test_fp_host.py against Python integers); here two implementations of the same product must agree and nothing may
// shift past a width, overflow a signed type or touch memory out of bounds.  Built and run by tests/test_oracle_sanitize.py.
#include "fp_host_harness.cpp"

#include <cstdio>
#include <cstring>
#include <vector>

static uint64_t sm64(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "fp_host_sanitize: check failed at line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
    const int NL[6] = {8, 8, 8, 12, 8, 8};
    for (int f = 0; f < 6; ++f) {
        const int nl = NL[f];
        const size_t n = 96;
        std::vector<uint32_t> a(n * nl, 0), b(n * nl, 0), o1(n * nl), o2(n * nl);
        uint64_t st = 77 + f;
        for (size_t i = 0; i < n; ++i)
            for (int k = 0; k < nl; ++k) {
                // edge patterns in the first rows (0, 1, all-ones low words), seeded values after; the top word kept small: below every modulus
                uint32_t va = i == 0 ? 0u : i == 1 ? (k == 0) : i == 2 ? 0xffffffffu : (uint32_t)sm64(st);
                uint32_t vb = i == 3 ? 0u : i == 4 ? 0xffffffffu : (uint32_t)sm64(st);
                if (k == nl - 1) { va &= 0x00ffffffu; vb &= 0x00ffffffu; }
                a[i * nl + k] = va;
                b[i * nl + k] = vb;
            }
        for (int op : {0, 1, 2, 3, 4, 6, 10, 11, 12, 13, 14, 15, 16, 17, 19, 20, 21, 22, 23}) REQUIRE(fp_host_op(f, op, a.data(), b.data(), o1.data(), n) == 0);
        REQUIRE(fp_host_op(f, 2, a.data(), b.data(), o1.data(), n) == 0);
        REQUIRE(fp_host_op(f, 3, a.data(), b.data(), o2.data(), n) == 0);
        REQUIRE(memcmp(o1.data(), o2.data(), n * nl * 4) == 0);  // 29-bit product scanning against 32-bit CIOS
        // the three inversions on the non-zero rows agree
        std::vector<uint32_t> nz(a.begin() + nl, a.begin() + 9 * nl), i1(8 * nl), i2(8 * nl), i3(8 * nl);
        REQUIRE(fp_host_op(f, 5, nz.data(), nz.data(), i1.data(), 8) == 0);
        REQUIRE(fp_host_op(f, 8, nz.data(), nz.data(), i2.data(), 8) == 0);
        REQUIRE(fp_host_op(f, 18, nz.data(), nz.data(), i3.data(), 8) == 0);
        REQUIRE(memcmp(i1.data(), i2.data(), 8 * nl * 4) == 0 && memcmp(i1.data(), i3.data(), 8 * nl * 4) == 0);
        if (f != 2) {
            // lazy XYZZ sums over arbitrary coordinate pairs (the formulas do not need curve points to be memory- and overflow-safe),
            // with a repeated operand (the doubling branch) and its negation (the identity branch)
            const size_t m = 40;
            std::vector<uint32_t> xs(a.begin(), a.begin() + m * nl), ys(b.begin(), b.begin() + m * nl), out(4 * nl + 1);
            std::vector<uint8_t> negs(m, 0);
            memcpy(&xs[6 * nl], &xs[5 * nl], nl * 4);
            memcpy(&ys[6 * nl], &ys[5 * nl], nl * 4);
            memcpy(&xs[8 * nl], &xs[7 * nl], nl * 4);
            memcpy(&ys[8 * nl], &ys[7 * nl], nl * 4);
            negs[8] = 1;
            for (size_t cnt : {(size_t)0, (size_t)1, (size_t)2, (size_t)7, (size_t)9, m}) {
                REQUIRE(ecz_host_sum(f, cnt, xs.data(), ys.data(), negs.data(), out.data()) == 0);
                REQUIRE(ecz_host_tree(f, cnt, xs.data(), ys.data(), negs.data(), out.data()) == 0);
            }
        }
    }
    for (int curve : {0, 1, 3, 4}) {
        const size_t n = 64;
        std::vector<uint32_t> k(8 * n), k1(8 * n), k2(8 * n);
        uint64_t st = 5 + curve;
        for (size_t i = 0; i < 8 * n; ++i) k[i] = i < 8 ? 0u : i < 16 ? 0xffffffffu : (uint32_t)sm64(st);
        for (size_t i = 0; i < n; ++i) k[8 * i + 7] &= 0x3fffffffu;  // below the 255-bit group orders
        REQUIRE(glv_host_split(curve, k.data(), k1.data(), k2.data(), n) == 0);
    }
    printf("fp_host_sanitize: ok\n");
    return 0;
}
