// inv_latency.hip -- the division-step inversion of fp.cuh as a dependency chain on ONE lane of one wave (what the normalisation at the
// end of an MSM runs, k_msm_final): shader-clock ticks per inversion of its forms (MODE 0 fixed / 1 data-dependent runs / 2 runs with the
// low words on the scalar unit / 3 fixed with scalar low words), and the same with all 64 lanes of the wave inverting different values.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -I plonky_amd/csrc tools/lab/inv_latency.hip -o build/inv_latency && build/inv_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include "field_params.cuh"
#include "fp.cuh"
using namespace plk;
using P = TweedledeeBaseParams;
constexpr int K = 64;

template <int MODE, bool ONE> __global__ void k_inv(uint32_t seed, unsigned long long* out, uint32_t* sink) {
    if (ONE && threadIdx.x != 0) return;
    Fe<P> x = fe_one<P>();
    x.v[0] += seed + (ONE ? 0u : threadIdx.x * 77u);
    x.v[3] ^= 0x1234567u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < K; ++k) {
        x = fe_inv_safegcd_impl<P, MODE>(x);
        x.v[0] ^= 5u;  // stays below p (top word untouched), never zero
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= x.v[i];
    sink[threadIdx.x] = acc;
}

template <int MODE, bool ONE> static void run(const char* name, unsigned long long* d_out, uint32_t* d_sink) {
    unsigned long long best = ~0ull;
    uint32_t first_sink = 0;
    for (int rep = 0; rep < 5; ++rep) {
        k_inv<MODE, ONE><<<1, 64>>>(17u, d_out, d_sink);
        unsigned long long t = 0;
        (void)hipMemcpy(&t, d_out, sizeof t, hipMemcpyDeviceToHost);
        uint32_t s = 0;
        (void)hipMemcpy(&s, d_sink, sizeof s, hipMemcpyDeviceToHost);
        if (rep == 0) first_sink = s;
        if (t < best) best = t;
    }
    printf("%-58s %8.0f ticks per inversion (check word %08x)\n", name, (double)best / K, first_sink);
}

int main() {
    unsigned long long* d_out;
    uint32_t* d_sink;
    (void)hipMalloc(&d_out, 64);
    (void)hipMalloc(&d_sink, 64 * 4);
    run<0, true>("one lane, fixed form (MODE 0)", d_out, d_sink);
    run<1, true>("one lane, runs of steps (MODE 1)", d_out, d_sink);
    run<2, true>("one lane, runs of steps, scalar low words (MODE 2)", d_out, d_sink);
    run<3, true>("one lane, fixed form, scalar low words (MODE 3)", d_out, d_sink);
    run<0, false>("64 lanes, fixed form (MODE 0)", d_out, d_sink);
    run<1, false>("64 lanes, runs of steps (MODE 1)", d_out, d_sink);
    return 0;
}
