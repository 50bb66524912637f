// tools/lab/microbench.hip -- issue-rate probes for the integer instructions the field arithmetic
// is built from (v_mad_u64_u32, 64-bit add, add/addc, v_mul_lo/hi, 24-bit mad, f64 fma).
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench tools/lab/microbench.hip ; run on the GPU box.
// Output feeds DESIGN.md "ALU ceiling": cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

template <int OP> __global__ void __launch_bounds__(256) probe(uint32_t* out, uint32_t seed) {
    uint32_t a[ILP], b[ILP];
    uint64_t acc[ILP];
    double d[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        a[k] = seed * (threadIdx.x + k + 1) | 1u;
        b[k] = seed ^ (0x9e3779b9u * (k + 3));
        acc[k] = ((uint64_t)a[k] << 17) ^ b[k];
        d[k] = 1.0 + k * 1e-9 + threadIdx.x * 1e-12;
    }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            if (OP == 0) {  // v_mad_u64_u32
                acc[k] = (uint64_t)a[k] * (uint32_t)acc[k] + acc[k];
            } else if (OP == 1) {  // 64-bit add (v_lshl_add_u64 / add_co+addc)
                acc[k] = acc[k] + ((uint64_t)b[k] << 1) + acc[(k + 1) % ILP];
            } else if (OP == 2) {  // v_mul_lo_u32
                a[k] = a[k] * b[k] + 1u;
            } else if (OP == 3) {  // v_mul_hi_u32
                a[k] = __umulhi(a[k], b[k]) | 0x80000001u;
            } else if (OP == 4) {  // v_mad_u32_u24
                a[k] = __umul24(a[k], b[k]) + a[k];
            } else if (OP == 5) {  // v_fma_f64
                d[k] = __fma_rn(d[k], 1.0000001, 1e-9);
            } else if (OP == 6) {  // v_add_u32 (baseline)
                a[k] = a[k] + b[k];
                b[k] = b[k] ^ a[k];
            } else if (OP == 7) {  // v_add_co / v_addc pair through __int128-free carry idiom
                uint64_t x = (uint64_t)a[k] + b[k];
                a[k] = (uint32_t)x;
                b[k] = b[k] + (uint32_t)(x >> 32) + 1u;
            } else if (OP == 8) {  // v_mul_hi_u32_u24 + v_mul_u32_u24
                a[k] = __umul24(a[k], b[k]) ^ __mul24(a[k] >> 3, b[k] >> 5);
            } else if (OP == 9) {  // v_lshrrev_b64 (+ 64-bit or to keep it alive)
                acc[k] = (acc[k] >> 29) | ((uint64_t)b[k] << 35);
            } else if (OP == 10) {  // v_lshlrev_b64
                acc[k] = (acc[k] << 22) ^ (uint64_t)a[k];
            } else if (OP == 11) {  // v_alignbit_b32
                a[k] = __builtin_amdgcn_alignbit(a[k], b[k], 29) + 1u;
            } else if (OP == 12) {  // mad with inline/literal constant multiplier
                acc[k] = (uint64_t)(uint32_t)acc[k] * 0x1657ea0u + acc[k];
            } else if (OP == 13) {  // column step of the 29-bit multiply: mad, mad, shift, mask
                uint64_t t = (uint64_t)a[k] * b[k] + acc[k];
                t = (uint64_t)b[k] * 0x18a1b261u + t;
                a[k] = (uint32_t)t & 0x1fffffffu;
                acc[k] = t >> 29;
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) r ^= a[k] ^ b[k] ^ (uint32_t)acc[k] ^ (uint32_t)(acc[k] >> 32) ^ (uint32_t)__double_as_longlong(d[k]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP> int run(const char* name, double ops_per_iter_per_k, uint32_t* d_out, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    probe<OP><<<blocks, 256>>>(d_out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    probe<OP><<<blocks, 256>>>(d_out, 12345u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double wave_instr = (double)blocks * 4 * ITERS * ILP * ops_per_iter_per_k;
    double per_simd = wave_instr / (256.0 * 4);
    double cycles = ms * 1e-3 * 2.4e9;  // at the 2.4 GHz max clock; real clock may be lower
    printf("%-28s waves/SIMD=%d  %8.3f ms  %6.2f cycles(@2.4GHz)/wave-instr/SIMD  %8.1f Ginstr-lanes/s\n", name, waves_per_simd, ms,
           cycles / per_simd, wave_instr * 64 / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(uint32_t)));
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_mad_u64_u32", 1, d_out, w);
        run<1>("add64 x2", 2, d_out, w);
        run<2>("v_mul_lo_u32 (+add)", 1, d_out, w);
        run<3>("v_mul_hi_u32 (+or)", 1, d_out, w);
        run<4>("v_mad_u32_u24", 1, d_out, w);
        run<5>("v_fma_f64", 1, d_out, w);
        run<6>("v_add_u32 + v_xor", 2, d_out, w);
        run<7>("add_co + add3", 2, d_out, w);
        run<8>("mul24 x2 + xor", 3, d_out, w);
        run<9>("v_lshrrev_b64 (+or64)", 1, d_out, w);
        run<10>("v_lshlrev_b64 (+xor)", 1, d_out, w);
        run<11>("v_alignbit_b32 (+add)", 1, d_out, w);
        run<12>("v_mad_u64_u32 const", 1, d_out, w);
        run<13>("column: 2 mad + and + lshr64", 4, d_out, w);
        printf("\n");
    }
    return 0;
}
