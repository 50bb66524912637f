// mul_latency.hip -- what bounds a dependency chain of field / point operations on gfx950 (round 3; results in profiles/r03_mul_latency.txt):
//   LAT   one wave's dependent multiplications: fz_mul (one accumulator through all columns) against a variant with independent column
//         accumulators (below) - no difference: a lone wave issues dependent v_mad_u64_u32 back to back;
//   THR   the same two at 1, 2, 4 waves per SIMD;
//   DBL   a chain of point doublings, one lane (xyzzz_dbl) against a quad (xyzzz_dbl_q);
//   OVER  1024, 1025, ... one-wave workgroups: the 1025th wave shares a SIMD and every step of the chain takes 3.6 instead of 2.2 us;
//   PLACE the same 1024 waves as workgroups of 1 .. 16 waves: with this small kernel two-wave workgroups run 1.6x slower than one- or
//         four-wave ones (their waves share SIMDs); the library's kernels did not change when their workgroups were resized accordingly
//         (fold, table, row / column sums, accumulation: all within noise) - their register use already spreads them.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -I plonky_amd/csrc tools/lab/mul_latency.hip -o build/mul_latency && build/mul_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include "field_params.cuh"
#include "fz.cuh"
#include "ecz_coop.cuh"

// The same Montgomery product with 2 NZ - 1 independent column accumulators (operand scanning) and the serial part - quotient digits and
// carries - afterwards: built to see whether fz_mul's single accumulator (one dependency chain through ~135 v_mad_u64_u32) costs a lone
// wave anything.  It does not: gfx950 issues dependent v_mad_u64_u32 back to back (923 against 913 cycles below), so a wave's
// multiplication is its instruction count x ~4.4 cycles and only fewer instructions - not more parallelism inside a lane - shorten a chain.
namespace plk {
template <class P> PLK_DI Fz<P> fz_mul_lat(const Fz<P>& a, const Fz<P>& b) {
    constexpr int NZ = FzCfg<P>::NZ;
    constexpr uint32_t M = FzCfg<P>::M;
    uint64_t t[2 * NZ - 1];
#pragma unroll
    for (int k = 0; k < 2 * NZ - 1; ++k) t[k] = 0;
#pragma unroll
    for (int i = 0; i < NZ; ++i)
#pragma unroll
        for (int j = 0; j < NZ; ++j) t[i + j] = (uint64_t)a.l[i] * b.l[j] + t[i + j];
    uint32_t q[NZ];
    Fz<P> r;
    uint64_t carry = 0;
    const uint32_t p_pow2 = FzPow2Limb<P>::index() >= 0 ? fz_opaque(FzCfg<P>::plimb(FzPow2Limb<P>::index() >= 0 ? FzPow2Limb<P>::index() : 0)) : 0u;
#pragma unroll
    for (int k = 0; k <= 2 * NZ - 2; ++k) {
        uint64_t acc = t[k];
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int j = k - i;
            if (i < k && j >= 1 && j < NZ && FzCfg<P>::plimb(j) != 0u)
                acc = (uint64_t)q[i] * (j == FzPow2Limb<P>::index() ? p_pow2 : FzCfg<P>::plimb(j)) + acc;
        }
        acc += carry;
        if (k < NZ) {
            q[k] = (0u - (uint32_t)acc) & M;
            acc += q[k];
        } else {
            r.l[k - NZ] = (uint32_t)acc & M;
        }
        carry = fz_shr29(acc);
    }
    r.l[NZ - 1] = (uint32_t)carry;
    return r;
}
template <class P> PLK_DI Fz<P> fz_sqr_lat(const Fz<P>& a) { return fz_mul_lat<P>(a, a); }
}  // namespace plk
using namespace plk;
constexpr int ITERS = 2000;
template <class P, int OP> __global__ void __launch_bounds__(256) k_chain(uint32_t* out, uint32_t seed) {
    Fz<P> x, y;
    for (int i = 0; i < FzCfg<P>::NZ; ++i) { x.l[i] = (seed * (threadIdx.x + 3 + i)) & 0x1FFFFFFFu; y.l[i] = (seed * 7 + i * 11 + threadIdx.x) & 0x1FFFFFFFu; }
    x.l[FzCfg<P>::NZ - 1] &= 0xFFFFF; y.l[FzCfg<P>::NZ - 1] &= 0xFFFFF;
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) x = fz_mul<P>(x, y);
        if (OP == 1) x = fz_mul_lat<P>(x, y);
        if (OP == 2) x = fz_sqr<P>(x);
        if (OP == 3) x = fz_sqr_lat<P>(x);
    }
    uint32_t r = 0;
    for (int i = 0; i < FzCfg<P>::NZ; ++i) r ^= x.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// a chain of doublings of one point: one lane (xyzzz_dbl) against a quad (xyzzz_dbl_q)
template <class P, int OP> __global__ void __launch_bounds__(1024) k_dbl(uint32_t* out, uint32_t seed) {
    XyzzZ<P> p;
    for (int i = 0; i < FzCfg<P>::NZ; ++i) {
        p.x.l[i] = (seed * (3 + i)) & 0x1FFFFFFFu; p.y.l[i] = (seed * 7 + i * 11) & 0x1FFFFFFFu; p.zz.l[i] = i == 0; p.zzz.l[i] = i == 0;
    }
    p.x.l[FzCfg<P>::NZ - 1] &= 0xFFFFF; p.y.l[FzCfg<P>::NZ - 1] &= 0xFFFFF;
    p.inf = false;
    const int ql = threadIdx.x & 3;
    for (int it = 0; it < 200; ++it) {
        if (OP == 0) p = xyzzz_dbl<P>(p);
        if (OP == 1) p = xyzzz_dbl_q<P>(p, ql);
    }
    uint32_t r = 0;
    for (int i = 0; i < FzCfg<P>::NZ; ++i) r ^= p.x.l[i] ^ p.zzz.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class P, int OP> float run_dbl(uint32_t* d, int blocks, int threads) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_dbl<P, OP><<<blocks, threads>>>(d, 12345u); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); k_dbl<P, OP><<<blocks, threads>>>(d, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
template <class P, int OP> float run(uint32_t* d, int blocks, int threads) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_chain<P, OP><<<blocks, threads>>>(d, 12345u); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); k_chain<P, OP><<<blocks, threads>>>(d, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
template <class P> void field(const char* name, uint32_t* d, uint32_t* h) {
    const char* ops[4] = {"fz_mul", "fz_mul_lat", "fz_sqr", "fz_sqr_lat"};
    float lat[4] = {run<P, 0>(d, 1, 64), run<P, 1>(d, 1, 64), run<P, 2>(d, 1, 64), run<P, 3>(d, 1, 64)};
    uint32_t res[4];
    k_chain<P, 0><<<1, 64>>>(d, 12345u); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); res[0] = h[5];
    k_chain<P, 1><<<1, 64>>>(d, 12345u); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); res[1] = h[5];
    k_chain<P, 2><<<1, 64>>>(d, 12345u); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); res[2] = h[5];
    k_chain<P, 3><<<1, 64>>>(d, 12345u); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); res[3] = h[5];
    for (int o = 0; o < 4; ++o) printf("LAT field=%s op=%s one wave, dependent chain: %.1f ns per op (%.0f cycles at 2.4 GHz)  check %08x\n", name, ops[o], lat[o] * 1e6 / ITERS, lat[o] * 1e6 / ITERS * 2.4, res[o]);
    printf("SAME field=%s mul %s sqr %s\n", name, res[0] == res[1] ? "yes" : "NO", res[2] == res[3] ? "yes" : "NO");
    for (int w : {1, 2, 4}) {
        const int blocks = 256 * w;
        const float t0 = run<P, 0>(d, blocks, 256), t1 = run<P, 1>(d, blocks, 256);
        const double ops_total = (double)blocks * 256 * ITERS;
        printf("THR field=%s waves_per_simd=%d fz_mul %.1f Gop/s  fz_mul_lat %.1f Gop/s\n", name, w, ops_total / (t0 * 1e-3) / 1e9, ops_total / (t1 * 1e-3) / 1e9);
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    uint32_t h[64];
    // the same 1024 waves (one per SIMD of the chip) as workgroups of 1, 2, 4, 8 and 16 waves
    for (int threads : {64, 128, 256, 512, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int blocks = 65536 / threads;
        hipLaunchKernelGGL((k_dbl<TweedledeeBaseParams, 1>), dim3(blocks), dim3(threads), 0, 0, d, 12345u); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL((k_dbl<TweedledeeBaseParams, 1>), dim3(blocks), dim3(threads), 0, 0, d, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("PLACE 1024 waves as %d workgroups of %d threads: quad doubling %.2f us\n", blocks, threads, ms * 1e3 / 200);
    }
    // one wave more than there are SIMDs (the frozen tables of an IPA: 2^14 + 2 generators = 1025 waves of quads)
    for (int blocks : {1024, 1025, 1088, 1280, 1536}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k_dbl<TweedledeeBaseParams, 1>), dim3(blocks), dim3(64), 0, 0, d, 12345u); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL((k_dbl<TweedledeeBaseParams, 1>), dim3(blocks), dim3(64), 0, 0, d, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("OVER %d one-wave workgroups: quad doubling %.2f us\n", blocks, ms * 1e3 / 200);
    }
    for (int blocks : {1, 512, 1024, 2048}) {
        const float t0 = run_dbl<TweedledeeBaseParams, 0>(d, blocks, 128), t1 = run_dbl<TweedledeeBaseParams, 1>(d, blocks, 128);
        printf("DBL blocks=%d (x 2 waves): one lane %.2f us per doubling, quad %.2f us per doubling\n", blocks, t0 * 1e3 / 200, t1 * 1e3 / 200);
    }
    field<TweedledeeBaseParams>("tweedledee", d, h);
    return 0;
}
