// tools/lab/madd_lab.hip -- laboratory for the instruction diet of the MSM accumulation's mixed addition (round-3 review item 2b).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iplonky_amd/csrc tools/lab/madd_lab.hip -o build/madd_lab   (run on the GPU box)
//   hipcc ... --cuda-device-only -S ... -o /tmp/madd_lab.s                                              (instruction counts: tools/isa_count.py)
//   g++ -x c++ -O2 -std=c++17 -DLAB_HOST -Iplonky_amd/csrc tools/lab/madd_lab.hip -o build/madd_lab_host    (the same formulas on the CPU: old == new)
//
// Kernels: V = 0 the product's xyzzz_madd / fz_mul (ecz.cuh / fz.cuh as built into the library); V = 1 the same entry points built with
// -DLAB_NEW... there is only one set of headers, so the "old" forms are kept here verbatim under the names *_r3.
#ifndef LAB_HOST
#include <hip/hip_runtime.h>
#endif
#include <stdio.h>
#include <stdint.h>
#include "fp.cuh"
#include "fz.cuh"
#include "ecz.cuh"
using namespace plk;

// ---- the round-3 forms, verbatim (what the library ran before this round) ----
template <class P> PLK_DI Fz<P> fz_mul_r3(const Fz<P>& a, const Fz<P>& b) {
    constexpr int NZ = FzCfg<P>::NZ;
    constexpr uint32_t M = FzCfg<P>::M;
    uint32_t q[NZ];
    Fz<P> r;
    uint64_t acc = 0;
    const uint32_t p_pow2 = FzPow2Limb<P>::index() >= 0 ? fz_opaque(FzCfg<P>::plimb(FzPow2Limb<P>::index() >= 0 ? FzPow2Limb<P>::index() : 0)) : 0u;
#pragma unroll
    for (int k = 0; k <= 2 * NZ - 2; ++k) {
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (j >= 0 && j < NZ) acc = (uint64_t)a.l[i] * b.l[j] + acc;
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (i < k && j >= 1 && j < NZ && FzCfg<P>::plimb(j) != 0u)
                acc = (uint64_t)q[i] * (j == FzPow2Limb<P>::index() ? p_pow2 : FzCfg<P>::plimb(j)) + acc;
        }
        if (k < NZ) {
            q[k] = (0u - (uint32_t)acc) & M;
            acc += q[k];
        } else {
            r.l[k - NZ] = (uint32_t)acc & M;
        }
        acc = fz_shr29(acc);
    }
    r.l[NZ - 1] = (uint32_t)acc;
    return r;
}
template <class P> PLK_DI Fz<P> fz_sqr_r3(const Fz<P>& a) {
    constexpr int NZ = FzCfg<P>::NZ;
    constexpr uint32_t M = FzCfg<P>::M;
    uint32_t q[NZ], a2[NZ];
#pragma unroll
    for (int i = 0; i < NZ; ++i) a2[i] = a.l[i] << 1;
    Fz<P> r;
    uint64_t acc = 0;
    const uint32_t p_pow2 = FzPow2Limb<P>::index() >= 0 ? fz_opaque(FzCfg<P>::plimb(FzPow2Limb<P>::index() >= 0 ? FzPow2Limb<P>::index() : 0)) : 0u;
#pragma unroll
    for (int k = 0; k <= 2 * NZ - 2; ++k) {
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (j > i && j < NZ) acc = (uint64_t)a.l[i] * a2[j] + acc;
            if (j == i) acc = (uint64_t)a.l[i] * a.l[i] + acc;
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (i < k && j >= 1 && j < NZ && FzCfg<P>::plimb(j) != 0u)
                acc = (uint64_t)q[i] * (j == FzPow2Limb<P>::index() ? p_pow2 : FzCfg<P>::plimb(j)) + acc;
        }
        if (k < NZ) {
            q[k] = (0u - (uint32_t)acc) & M;
            acc += q[k];
        } else {
            r.l[k - NZ] = (uint32_t)acc & M;
        }
        acc = fz_shr29(acc);
    }
    r.l[NZ - 1] = (uint32_t)acc;
    return r;
}
template <class FP> PLK_DI void xyzzz_madd_r3(XyzzZ<FP>& acc, const Fz<FP>& x2, const Fz<FP>& y2) {
    if (acc.inf) {
        acc.x = x2;
        acc.y = y2;
        acc.zz = fz_one_rprime<FP>();
        acc.zzz = acc.zz;
        acc.inf = false;
        return;
    }
    Fz<FP> u2 = fz_mul_r3<FP>(x2, acc.zz);
    Fz<FP> s2 = fz_mul_r3<FP>(y2, acc.zzz);
    Fz<FP> p = fz_sub<FP, 4>(u2, acc.x);
    Fz<FP> r = fz_sub<FP, 2>(s2, acc.y);
    Fz<FP> pp = fz_sqr_r3<FP>(p);
    Fz<FP> ppp = fz_mul_r3<FP>(p, pp);
    Fz<FP> q = fz_mul_r3<FP>(acc.x, pp);
    Fz<FP> rr = fz_sqr_r3<FP>(r);
    Fz<FP> zz3 = fz_mul_r3<FP>(acc.zz, pp);
    if (fz_is_zero_mod_p<FP>(zz3)) {
        if (fz_is_zero_mod_p<FP>(rr)) {
            acc = xyzzz_mdbl<FP>(x2, y2);
        } else {
            acc.inf = true;
        }
        return;
    }
    Fz<FP> x3 = fz_sub_nc<FP, 2, 30>(fz_sub_nc<FP, 1, 29>(rr, ppp), fz_add_nc<FP>(q, q));
    fz_carry<FP>(x3);
    Fz<FP> t;
    if constexpr (FzCfg<FP>::NZ <= 10) t = fz_sub_nc<FP, 3, 30>(q, x3);
    else t = fz_sub<FP, 3>(q, x3);
    acc.y = fz_sub<FP, 1>(fz_mul_r3<FP>(r, t), fz_mul_r3<FP>(acc.y, ppp));
    acc.x = x3;
    acc.zz = zz3;
    acc.zzz = fz_mul_r3<FP>(acc.zzz, ppp);
}

constexpr int ITERS = 512;
#ifndef LAB_HOST
// V: 0 round-3 forms, 1 the library's current headers.  OP: 0 fz_mul chain, 1 fz_sqr chain, 4 the accumulation's step (entry in the
// interface form + sign, conversion, conditional negation, mixed addition: what one iteration of k_msm_accumulate executes)
template <class P, int OP, int V> __global__ void __launch_bounds__(256, 2) k(uint32_t* out, uint32_t seed) {
    Fe<P> x, y;
    for (int i = 0; i < P::NL; ++i) { x.v[i] = seed * (threadIdx.x + i + 1); y.v[i] = seed ^ (0x9e3779b9u * (i + 3 + threadIdx.x)); }
    x.v[P::NL - 1] &= 0x00ffffffu; y.v[P::NL - 1] &= 0x00ffffffu;
    uint32_t r = 0;
    Fz<P> a = fz_from_fe<P>(x), b = fz_from_fe<P>(y);
    if (OP == 0) for (int it = 0; it < ITERS; ++it) a = V ? fz_mul<P>(a, b) : fz_mul_r3<P>(a, b);
    if (OP == 1) for (int it = 0; it < ITERS; ++it) a = V ? fz_sqr<P>(a) : fz_sqr_r3<P>(a);
    if (OP == 4) {
        XyzzZ<P> acc; acc.inf = false; acc.x = a; acc.y = b; acc.zz = a; acc.zzz = b;
        for (int it = 0; it < ITERS / 8; ++it) {
            // a fresh "table entry" per iteration (the real kernel loads it): words derived from the running state so nothing is hoisted
            Fe<P> ex = x, ey = y;
            ex.v[0] ^= acc.x.l[0]; ey.v[1] ^= acc.zz.l[1];
            const uint32_t sign = (acc.x.l[2] ^ acc.zzz.l[3]) & 1u;  // from coordinates that are limb-identical in both forms (Y may be uncarried in the new one)
            if (V) {
                xyzzz_madd_entry<P>(acc, ex, ey, sign != 0);
            } else {
                Fz<P> xz = fz_from_fe<P>(ex), yz = fz_from_fe<P>(ey);
                if (sign) yz = fz_neg_canonical<P>(yz);
                xyzzz_madd_r3<P>(acc, xz, yz);
            }
        }
        xyzzz_settle<P>(acc);
        a = fz_add<P>(fz_add<P>(acc.x, acc.y), fz_add<P>(acc.zz, acc.zzz)); r += acc.inf;
    }
    x = fz_to_fe_canonical<P>(fz_mul<P>(a, fz_one_rprime<P>()));
    for (int i = 0; i < P::NL; ++i) r ^= x.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class P, int OP, int V> double run(uint32_t* d, int waves, uint32_t* host, int nhost) {
    int blocks = 256 * waves;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<P, OP, V><<<blocks, 256>>>(d, 12345u); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); k<P, OP, V><<<blocks, 256>>>(d, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    if (host) hipMemcpy(host, d, nhost * 4, hipMemcpyDeviceToHost);
    return best;
}
template <class P> void field(const char* name, uint32_t* d) {
    static uint32_t h0[4096], h1[4096];
    for (int w : {1, 2, 3, 4}) {
        const double m0 = run<P, 0, 0>(d, w, h0, 4096), m1 = run<P, 0, 1>(d, w, h1, 4096);
        int bad_m = 0, bad_s = 0, bad_a = 0;
        for (int i = 0; i < 4096; ++i) bad_m += h0[i] != h1[i];
        const double s0 = run<P, 1, 0>(d, w, h0, 4096), s1 = run<P, 1, 1>(d, w, h1, 4096);
        for (int i = 0; i < 4096; ++i) bad_s += h0[i] != h1[i];
        const double a0 = run<P, 4, 0>(d, w, h0, 4096), a1 = run<P, 4, 1>(d, w, h1, 4096);
        for (int i = 0; i < 4096; ++i) bad_a += h0[i] != h1[i];
        const bool same = !(bad_m | bad_s | bad_a);
        if (!same) printf("  mismatching lanes: mul %d sqr %d step %d of 4096\n", bad_m, bad_s, bad_a);
        const double ops = 256.0 * w * 4 * 64;
        printf("%s waves/SIMD %d: fz_mul %.1f -> %.1f G/s (%+.1f %%)  fz_sqr %.1f -> %.1f G/s (%+.1f %%)  accumulation step %.2f -> %.2f G/s (%+.1f %%)  results identical: %s\n", name, w,
               ops * ITERS / (m0 * 1e-3) / 1e9, ops * ITERS / (m1 * 1e-3) / 1e9, (m0 / m1 - 1) * 100, ops * ITERS / (s0 * 1e-3) / 1e9, ops * ITERS / (s1 * 1e-3) / 1e9,
               (s0 / s1 - 1) * 100, ops * ITERS / 8 / (a0 * 1e-3) / 1e9, ops * ITERS / 8 / (a1 * 1e-3) / 1e9, (a0 / a1 - 1) * 100, same ? "yes" : "NO");
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    field<TweedledeeBaseParams>("tweedledee", d);
    field<Bls12377BaseParams>("bls12_377", d);
    return 0;
}
#else
// host: the round-3 forms and the current headers on the same pseudo-random chains, compared after every step
#include <random>
template <class P> static int check(const char* name) {
    std::mt19937_64 rng(12345);
    constexpr int NZ = FzCfg<P>::NZ;
    long bad = 0;
    for (int trial = 0; trial < 20000; ++trial) {
        Fe<P> x, y;
        for (int i = 0; i < P::NL; ++i) { x.v[i] = (uint32_t)rng(); y.v[i] = (uint32_t)rng(); }
        x.v[P::NL - 1] &= 0x00ffffffu; y.v[P::NL - 1] &= 0x00ffffffu;
        Fz<P> a = fz_from_fe<P>(x), b = fz_from_fe<P>(y);
        // operands at the limb bound the products accept (2^29 + 2^27) now and then
        if (trial % 7 == 0) for (int i = 0; i < NZ - 1; ++i) a.l[i] = (1u << 29) + (1u << 27) - (trial & 3);
        const Fz<P> m0 = fz_mul_r3<P>(a, b), m1 = fz_mul<P>(a, b), s0 = fz_sqr_r3<P>(a), s1 = fz_sqr<P>(a);
        for (int i = 0; i < NZ; ++i) bad += (m0.l[i] != m1.l[i]) + (s0.l[i] != s1.l[i]);
        XyzzZ<P> p0, p1;
        p0.inf = p1.inf = false;
        p0.x = p1.x = fz_mul<P>(a, b); p0.y = p1.y = fz_sqr<P>(b); p0.zz = p1.zz = fz_sqr<P>(a); p0.zzz = p1.zzz = fz_mul<P>(b, a);
        for (int it = 0; it < 40; ++it) {
            Fe<P> ex = x, ey = y;
            ex.v[0] ^= (uint32_t)rng(); ey.v[1] ^= (uint32_t)rng();
            const bool sign = rng() & 1;
            Fz<P> xz = fz_from_fe<P>(ex), yz = fz_from_fe<P>(ey);
            if (sign) yz = fz_neg_canonical<P>(yz);
            xyzzz_madd_r3<P>(p0, xz, yz);
            xyzzz_madd_entry<P>(p1, ex, ey, sign);
            // the new form may keep Y uncarried inside a chunk: compare the settled copies through canonical words
            XyzzZ<P> c = p1;
            xyzzz_settle<P>(c);
            const Fz<P> one = fz_one_rprime<P>();
            const Fe<P> y0 = fz_to_fe_canonical<P>(fz_mul<P>(p0.y, one)), y1 = fz_to_fe_canonical<P>(fz_mul<P>(c.y, one));
            for (int i = 0; i < P::NL; ++i) bad += y0.v[i] != y1.v[i];
            for (int i = 0; i < NZ; ++i) bad += (p0.x.l[i] != c.x.l[i]) + (p0.zz.l[i] != c.zz.l[i]) + (p0.zzz.l[i] != c.zzz.l[i]);
            bad += p0.inf != c.inf;
            for (int i = 0; i < NZ; ++i) bad += c.y.l[i] >= (1u << 29) + 8u;  // the settled accumulator is back inside the invariant
        }
    }
    printf("%s: %ld mismatches\n", name, bad);
    return bad != 0;
}
int main() { return check<TweedledeeBaseParams>("tweedledee") | check<TweedledumBaseParams>("tweedledum") | check<Bls12377BaseParams>("bls12_377") | check<PallasBaseParams>("pallas"); }
#endif
