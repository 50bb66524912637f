// tools/lab/tail_lab.hip -- why does the MSM's reduction tail (k_msm_planes, k_msm_final: a handful of points, long dependency chains of
// quad point operations) run at ~13 cycles per instruction of its critical wave when a lone wave's field product is issue-bound at
// ~4.7?  Hypothesis: the kernels are ~160 KB of straight-line code that runs ONCE (every xyzzz_add_q / xyzzz_dbl_q call site is its
// own inline copy), so the critical wave waits for instruction fetch, not for the ALU.
//
// One workgroup of 512 threads (the shape of k_msm_final).  Chains of N quad doublings / additions, timed with s_memtime on lane 0:
//   "inline": N call sites in a row (every operation executes code that was never fetched before),
//   "loop":   one call site in a loop that is not unrolled (the code of the first iteration is resident for the others).
// Between two measured launches a filler kernel with its own large body runs, so that every launch starts with a cold instruction cache
// the way k_msm_final does after the accumulation.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iplonky_amd/csrc tools/lab/tail_lab.hip -o ab_libs/tail_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "fp.cuh"
#include "fz.cuh"
#include "ecz.cuh"
#include "ecz_coop.cuh"
using namespace plk;

constexpr int N = 8;

template <class P> __device__ XyzzZ<P> lab_point(const uint32_t* seed, int quad) {
    XyzzZ<P> a;
    constexpr int NZ = FzCfg<P>::NZ;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        a.x.l[i] = (seed[i] + 977u * quad) & 0x0fffffffu;
        a.y.l[i] = (seed[NZ + i] + 31u * quad) & 0x0fffffffu;
        a.zz.l[i] = (seed[2 * NZ + i] ^ (uint32_t)quad) & 0x0fffffffu;
        a.zzz.l[i] = (seed[3 * NZ + i] + quad) & 0x0fffffffu;
    }
    a.inf = false;
    return a;
}
template <class P> __device__ uint32_t lab_fold(const XyzzZ<P>& a) {
    uint32_t h = a.inf;
#pragma unroll
    for (int i = 0; i < FzCfg<P>::NZ; ++i) h = h * 31u + a.x.l[i] + a.y.l[i] * 7u + a.zz.l[i] * 13u + a.zzz.l[i] * 17u;
    return h;
}

// MODE 0: doublings inline, 1: doublings in a loop, 2: additions inline, 3: additions in a loop
template <class P, int MODE>
__global__ void __launch_bounds__(512) k_chain(const uint32_t* __restrict__ seed, uint64_t* __restrict__ stamps, uint32_t* __restrict__ sink, int n) {
    const int ql = threadIdx.x & 3, quad = threadIdx.x >> 2;
    XyzzZ<P> acc = lab_point<P>(seed, quad);
    const XyzzZ<P> other = lab_point<P>(seed, quad + 1000);
    uint64_t t[N + 1];
    t[0] = clock64();
    if constexpr (MODE == 0) {
        acc = xyzzz_dbl_q<P>(acc, ql); t[1] = clock64();
        acc = xyzzz_dbl_q<P>(acc, ql); t[2] = clock64();
        acc = xyzzz_dbl_q<P>(acc, ql); t[3] = clock64();
        acc = xyzzz_dbl_q<P>(acc, ql); t[4] = clock64();
        acc = xyzzz_dbl_q<P>(acc, ql); t[5] = clock64();
        acc = xyzzz_dbl_q<P>(acc, ql); t[6] = clock64();
        acc = xyzzz_dbl_q<P>(acc, ql); t[7] = clock64();
        acc = xyzzz_dbl_q<P>(acc, ql); t[8] = clock64();
    } else if constexpr (MODE == 1) {
#pragma nounroll
        for (int k = 0; k < n; ++k) {
            acc = xyzzz_dbl_q<P>(acc, ql);
            const uint64_t now = clock64();
            if (threadIdx.x == 0) stamps[k + 1] = now;
        }
    } else if constexpr (MODE == 2) {
        acc = xyzzz_add_q<P>(acc, other, ql); t[1] = clock64();
        acc = xyzzz_add_q<P>(acc, other, ql); t[2] = clock64();
        acc = xyzzz_add_q<P>(acc, other, ql); t[3] = clock64();
        acc = xyzzz_add_q<P>(acc, other, ql); t[4] = clock64();
        acc = xyzzz_add_q<P>(acc, other, ql); t[5] = clock64();
        acc = xyzzz_add_q<P>(acc, other, ql); t[6] = clock64();
        acc = xyzzz_add_q<P>(acc, other, ql); t[7] = clock64();
        acc = xyzzz_add_q<P>(acc, other, ql); t[8] = clock64();
    } else {
#pragma nounroll
        for (int k = 0; k < n; ++k) {
            acc = xyzzz_add_q<P>(acc, other, ql);
            const uint64_t now = clock64();
            if (threadIdx.x == 0) stamps[k + 1] = now;
        }
    }
    if (threadIdx.x == 0) {
        stamps[0] = t[0];
        if constexpr (MODE == 0 || MODE == 2)
            for (int k = 1; k <= N; ++k) stamps[k] = t[k];
    }
    sink[threadIdx.x] = lab_fold<P>(acc);
}

// a filler with a large body of its own (many inlined products): evicts the instruction cache between measured launches
template <class P> __global__ void __launch_bounds__(256) k_filler(const uint32_t* __restrict__ seed, uint32_t* __restrict__ sink) {
    XyzzZ<P> a = lab_point<P>(seed, threadIdx.x), b = lab_point<P>(seed, threadIdx.x + 7);
#pragma unroll
    for (int k = 0; k < 24; ++k) a = xyzzz_add<P>(a, b);
    sink[blockIdx.x * 256 + threadIdx.x] = lab_fold<P>(a);
}

template <class P, int MODE> static void run(const char* what, const uint32_t* d_seed, uint64_t* d_stamps, uint32_t* d_sink) {
    uint64_t h[N + 1];
    for (int rep = 0; rep < 3; ++rep) {
        k_filler<P><<<2048, 256>>>(d_seed, d_sink + 4096);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k_chain<P, MODE><<<1, 512>>>(d_seed, d_stamps, d_sink, N);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-22s launch %7.1f us | cycles per operation:", what, ms * 1e3);
        for (int k = 1; k <= N; ++k) printf(" %6llu", (unsigned long long)(h[k] - h[k - 1]));
        printf("\n");
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}

int main() {
    using P = TweedledeeBaseParams;
    uint32_t hs[64];
    for (int i = 0; i < 64; ++i) hs[i] = 0x9e3779b9u * (i + 1);
    uint32_t *d_seed, *d_sink; uint64_t* d_stamps;
    hipMalloc(&d_seed, sizeof(hs)); hipMalloc(&d_sink, (4096 + 2048 * 256) * 4); hipMalloc(&d_stamps, 64 * 8);
    hipMemcpy(d_seed, hs, sizeof(hs), hipMemcpyHostToDevice);
    printf("# s_memtime ticks (100 MHz constant clock on gfx950? compare with the launch time): %d operations per chain\n", N);
    run<P, 0>("dbl_q inline sites", d_seed, d_stamps, d_sink);
    run<P, 1>("dbl_q one site, loop", d_seed, d_stamps, d_sink);
    run<P, 2>("add_q inline sites", d_seed, d_stamps, d_sink);
    run<P, 3>("add_q one site, loop", d_seed, d_stamps, d_sink);
    return 0;
}
