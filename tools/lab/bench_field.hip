// tools/lab/bench_field.hip -- per-operation throughput of the device field arithmetic on the GPU
// (wave-cycles per operation per SIMD at 4 waves/SIMD), feeding DESIGN.md's ALU ceiling.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iplonky_amd/csrc -o build/bench_field tools/lab/bench_field.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "fp.cuh"
#include "fz.cuh"
#include "ecz.cuh"
#include "ec.cuh"
using namespace plk;
using P = TweedledeeBaseParams;
constexpr int ITERS = 512;

template <int OP> __global__ void __launch_bounds__(256, 2) k(uint32_t* out, uint32_t seed) {
    Fe<P> x, y;
    for (int i = 0; i < 8; ++i) { x.v[i] = seed * (threadIdx.x + i + 1); y.v[i] = seed ^ (0x9e3779b9u * (i + 3 + threadIdx.x)); }
    x.v[7] &= 0x3fffffffu; y.v[7] &= 0x3fffffffu;
    uint32_t r = 0;
    if (OP == 0) { for (int it = 0; it < ITERS; ++it) x = fe_mul<P>(x, y); }
    if (OP == 1) { for (int it = 0; it < ITERS; ++it) x = fe_mul_cios<P>(x, y); }
    if (OP == 2) { for (int it = 0; it < ITERS; ++it) x = fe_add<P>(x, y); }
    if (OP == 3) { for (int it = 0; it < ITERS; ++it) x = fe_sub<P>(x, y); }
    if (OP >= 4 && OP <= 8) {
        Fz<P> a = fz_from_fe<P>(x), b = fz_from_fe<P>(y);
        if (OP == 4) for (int it = 0; it < ITERS; ++it) a = fz_mul<P>(a, b);
        if (OP == 5) for (int it = 0; it < ITERS; ++it) a = fz_sqr<P>(a);
        if (OP == 6) for (int it = 0; it < ITERS; ++it) a = fz_sub<P, 2>(a, b);
        if (OP == 7) for (int it = 0; it < ITERS; ++it) a = fz_add<P>(a, b);
        if (OP == 8) for (int it = 0; it < ITERS; ++it) { a = fz_mul<P>(a, b); r += fz_is_zero_mod_p<P>(a); }
        x = fz_to_fe_canonical<P>(fz_mul<P>(a, fz_one_rprime<P>()));
    }
    if (OP == 9) {  // lazy madd
        XyzzZ<P> acc; acc.inf = false; acc.x = fz_from_fe<P>(x); acc.y = fz_from_fe<P>(y); acc.zz = acc.x; acc.zzz = acc.y;
        Fz<P> px = fz_from_fe<P>(y), py = fz_from_fe<P>(x);
        for (int it = 0; it < ITERS / 8; ++it) xyzzz_madd<P>(acc, px, py);
        x = fz_to_fe_canonical<P>(fz_mul<P>(acc.x, fz_one_rprime<P>())); r += acc.inf;
    }
    if (OP == 10) {  // Fe madd
        Xyzz<P> acc; acc.x = x; acc.y = y; acc.zz = x; acc.zzz = y;
        for (int it = 0; it < ITERS / 8; ++it) xyzz_madd<P>(acc, y, x);
        x = acc.x; r += acc.zz.v[0];
    }
    for (int i = 0; i < 8; ++i) r ^= x.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run(const char* name, double ops, uint32_t* d, int waves) {
    int blocks = 256 * waves;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 12345u); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 12345u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_ops = (double)blocks * 4 * ops; double per_simd = wave_ops / 1024.0;
    printf("%-22s waves/SIMD=%d %8.3f ms  %8.1f cycles(@2.4GHz)/op/SIMD  %7.1f Gop/s\n", name, waves, ms, ms * 1e-3 * 2.4e9 / per_simd, wave_ops * 64 / (ms * 1e-3) / 1e9);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("fe_mul (29-bit hybrid)", ITERS, d, w); run<1>("fe_mul_cios", ITERS, d, w); run<2>("fe_add", ITERS, d, w); run<3>("fe_sub", ITERS, d, w);
        run<4>("fz_mul", ITERS, d, w); run<5>("fz_sqr", ITERS, d, w); run<6>("fz_sub", ITERS, d, w); run<7>("fz_add", ITERS, d, w); run<8>("fz_mul+zero test", ITERS, d, w);
        run<9>("lazy madd", ITERS / 8, d, w); run<10>("Fe madd", ITERS / 8, d, w);
        printf("\n");
    }
}
