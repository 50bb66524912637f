// tools/lab/field_ceilings.hip -- the integer-ALU ceilings bench.py prices the kernels against, measured on THIS tree's arithmetic
// headers: fz_mul / fz_sqr / fz_add / the lazy mixed addition per field (Gop/s over the whole GPU at 1..4 waves per SIMD) and
// the raw issue rate of v_mad_u64_u32 (the instruction a multiplication is made of), at 8 waves per SIMD.
// tools/measure_ceilings.py builds and runs it and writes profiles/r03_field_op_costs.{txt,json}.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "fp.cuh"
#include "fz.cuh"
#include "ecz.cuh"
#include "ec.cuh"
using namespace plk;
constexpr int ITERS = 512;

template <class P, int OP> __global__ void __launch_bounds__(256, 2) k(uint32_t* out, uint32_t seed) {
    Fe<P> x, y;
    for (int i = 0; i < P::NL; ++i) { x.v[i] = seed * (threadIdx.x + i + 1); y.v[i] = seed ^ (0x9e3779b9u * (i + 3 + threadIdx.x)); }
    x.v[P::NL - 1] &= 0x00ffffffu; y.v[P::NL - 1] &= 0x00ffffffu;
    uint32_t r = 0;
    Fz<P> a = fz_from_fe<P>(x), b = fz_from_fe<P>(y);
    if (OP == 0) for (int it = 0; it < ITERS; ++it) a = fz_mul<P>(a, b);
    if (OP == 1) for (int it = 0; it < ITERS; ++it) a = fz_sqr<P>(a);
    if (OP == 2) for (int it = 0; it < ITERS; ++it) a = fz_sub<P, 2>(a, b);
    if (OP == 3) for (int it = 0; it < ITERS; ++it) a = fz_add<P>(a, b);
    if (OP == 4) {  // the lazy mixed addition of the accumulation kernel (ecz.cuh), 8 M + 2 S
        XyzzZ<P> acc; acc.inf = false; acc.x = a; acc.y = b; acc.zz = a; acc.zzz = b;
        for (int it = 0; it < ITERS / 8; ++it) xyzzz_madd<P>(acc, b, a);
        a = acc.x; r += acc.inf;
    }
    x = fz_to_fe_canonical<P>(fz_mul<P>(a, fz_one_rprime<P>()));
    for (int i = 0; i < P::NL; ++i) r ^= x.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_mad(uint32_t* out, uint32_t seed) {
    constexpr int ILP = 8, IT = 4096;
    uint32_t a[ILP];
    uint64_t acc[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) { a[j] = seed * (threadIdx.x + j + 1) | 1u; acc[j] = ((uint64_t)a[j] << 17) ^ seed; }
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int j = 0; j < ILP; ++j) acc[j] = (uint64_t)a[j] * (uint32_t)acc[j] + acc[j];
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < ILP; ++j) r ^= (uint32_t)acc[j] ^ (uint32_t)(acc[j] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class P, int OP> void run(const char* field, const char* name, double ops, uint32_t* d, int waves) {
    int blocks = 256 * waves;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<P, OP><<<blocks, 256>>>(d, 12345u); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); k<P, OP><<<blocks, 256>>>(d, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double wave_ops = (double)blocks * 4 * ops;
    printf("OP field=%s op=%s waves_per_simd=%d ms=%.4f gops=%.2f\n", field, name, waves, best, wave_ops * 64 / (best * 1e-3) / 1e9);
}
template <class P> void field(const char* name, uint32_t* d) {
    for (int w : {1, 2, 3, 4}) {
        run<P, 0>(name, "fz_mul", ITERS, d, w); run<P, 1>(name, "fz_sqr", ITERS, d, w); run<P, 2>(name, "fz_sub", ITERS, d, w);
        run<P, 3>(name, "fz_add", ITERS, d, w); run<P, 4>(name, "lazy_madd", ITERS / 8, d, w);
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    field<TweedledeeBaseParams>("tweedledee", d);
    field<Bls12377BaseParams>("bls12_377", d);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w : {4, 8}) {
        int blocks = 256 * w;
        k_mad<<<blocks, 256>>>(d, 12345u); hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); k_mad<<<blocks, 256>>>(d, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("MAD waves_per_simd=%d ms=%.4f glaneops=%.1f\n", w, best, (double)blocks * 256 * 4096 * 8 / (best * 1e-3) / 1e9);
    }
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("INFO max_clock_khz=%d\n", clk);
    return 0;
}
