// ceilings.hip -- the integer-ALU ceilings the rooflines are priced against, measured IN THE CALLING PROCESS on the GPU it runs on
// (plk_bench_ceilings; bench.py calls it next to its timed region, so the ceiling and the kernel time come from the same box, the
// same minute and the same clocks - the round-4 review: a ceiling read from a file belongs to another GPU of a pool that spreads 7 %).
//   [0] v_mad_u64_u32 lane-operations per second at 8 waves per SIMD, eight independent chains per lane: the raw issue rate of the
//       instruction every modular multiplication is made of (126 per 9-limb product, 294 per 14-limb product) - a HARDWARE figure,
//       independent of this library's multiplication routine;
//   [1] [2] fz_mul of THIS build (fz.cuh) at 4 waves per SIMD, 9-limb (Tweedledee base) and 14-limb (BLS12-377 base) fields;
//   [3] [4] the lazy mixed addition of the bucket accumulation (ecz.cuh, 8 M + 2 S) at 3 / 2 waves per SIMD - the occupancies
//       k_msm_accumulate runs at on the two field sizes - with no memory traffic at all.
// Same kernels as tools/lab/field_ceilings.hip (which sweeps more operations and occupancies for profiles/rNN_field_op_costs.txt).
#include "common.h"
#include "fp.cuh"
#include "fz.cuh"
#include "ec.cuh"
#include "ecz.cuh"

namespace plk {

constexpr int CEIL_ITERS = 512;
template <class P, int OP> __global__ void __launch_bounds__(256, 2) k_ceiling(uint32_t* out, uint32_t seed) {
    Fe<P> x, y;
    for (int i = 0; i < P::NL; ++i) {
        x.v[i] = seed * (threadIdx.x + i + 1);
        y.v[i] = seed ^ (0x9e3779b9u * (i + 3 + threadIdx.x));
    }
    x.v[P::NL - 1] &= 0x00ffffffu;
    y.v[P::NL - 1] &= 0x00ffffffu;
    uint32_t r = 0;
    Fz<P> a = fz_from_fe<P>(x), b = fz_from_fe<P>(y);
    if (OP == 0)
        for (int it = 0; it < CEIL_ITERS; ++it) a = fz_mul<P>(a, b);
    if (OP == 1) {
        XyzzZ<P> acc;
        acc.inf = false;
        acc.x = a; acc.y = b; acc.zz = a; acc.zzz = b;
        for (int it = 0; it < CEIL_ITERS / 8; ++it) xyzzz_madd<P>(acc, b, a);
        a = acc.x;
        r += acc.inf;
    }
    x = fz_to_fe_canonical<P>(fz_mul<P>(a, fz_one_rprime<P>()));
    for (int i = 0; i < P::NL; ++i) r ^= x.v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
constexpr int MAD_ILP = 8, MAD_IT = 4096;
__global__ void __launch_bounds__(256) k_ceiling_mad(uint32_t* out, uint32_t seed) {
    uint32_t a[MAD_ILP];
    uint64_t acc[MAD_ILP];
#pragma unroll
    for (int j = 0; j < MAD_ILP; ++j) {
        a[j] = seed * (threadIdx.x + j + 1) | 1u;
        acc[j] = ((uint64_t)a[j] << 17) ^ seed;
    }
    for (int it = 0; it < MAD_IT; ++it)
#pragma unroll
        for (int j = 0; j < MAD_ILP; ++j) acc[j] = (uint64_t)a[j] * (uint32_t)acc[j] + acc[j];
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < MAD_ILP; ++j) r ^= (uint32_t)acc[j] ^ (uint32_t)(acc[j] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <class F> static int best_ms(F&& launch, hipStream_t stream, int reps, float* best) {
    hipEvent_t e0, e1;
    PLK_HIP_TRY(hipEventCreate(&e0));
    PLK_HIP_TRY(hipEventCreate(&e1));
    *best = 1e30f;
    int rc = PLK_OK;
    for (int rep = 0; rep < reps + 2 && rc == PLK_OK; ++rep) {  // two untimed launches first: clocks and instruction cache
        (void)hipEventRecord(e0, stream);
        launch();
        (void)hipEventRecord(e1, stream);
        if (hipEventSynchronize(e1) != hipSuccess) rc = set_error(PLK_ERR_HIP, "ceiling kernel failed: %s", hipGetErrorString(hipGetLastError()));
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < *best) *best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// out[0..n_out): see the head of this file; G operations per second over the whole GPU.  ~60 ms.
int bench_ceilings_impl(double* out, unsigned n_out) {
    if (!out || n_out < 5) return set_error(PLK_ERR_INVALID_ARG, "out must hold 5 values");
    PLK_TRY(ensure_device());
    int dev = 0, cus = 256;
    PLK_HIP_TRY(hipGetDevice(&dev));
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipStream_t stream = stream_pool_acquire();
    uint32_t* d = (uint32_t*)scratch_acquire((size_t)cus * 8 * 256 * 4, stream);
    if (!d) {
        stream_pool_release(stream);
        return PLK_ERR_OOM;
    }
    const int reps = 12;
    float ms = 0;
    int rc = PLK_OK;
    auto blocks = [&](int waves_per_simd) { return (unsigned)(cus * waves_per_simd); };  // 256 threads = 4 waves = one per SIMD of a CU
    // raw multiplier issue rate, 8 waves per SIMD
    rc = best_ms([&] { k_ceiling_mad<<<blocks(8), 256, 0, stream>>>(d, 12345u); }, stream, reps, &ms);
    if (rc == PLK_OK) out[0] = (double)blocks(8) * 256 * MAD_IT * MAD_ILP / (ms * 1e-3) / 1e9;
    auto op = [&](auto launch, int waves, double ops_per_lane, double* dst) {
        if (rc != PLK_OK) return;
        rc = best_ms(launch, stream, reps, &ms);
        if (rc == PLK_OK) *dst = (double)blocks(waves) * 256 * ops_per_lane / (ms * 1e-3) / 1e9;
    };
    op([&] { k_ceiling<TweedledeeBaseParams, 0><<<blocks(4), 256, 0, stream>>>(d, 12345u); }, 4, CEIL_ITERS, &out[1]);
    op([&] { k_ceiling<Bls12377BaseParams, 0><<<blocks(4), 256, 0, stream>>>(d, 12345u); }, 4, CEIL_ITERS, &out[2]);
    op([&] { k_ceiling<TweedledeeBaseParams, 1><<<blocks(3), 256, 0, stream>>>(d, 12345u); }, 3, CEIL_ITERS / 8, &out[3]);
    op([&] { k_ceiling<Bls12377BaseParams, 1><<<blocks(2), 256, 0, stream>>>(d, 12345u); }, 2, CEIL_ITERS / 8, &out[4]);
    scratch_release(d, stream);
    (void)hipStreamSynchronize(stream);
    stream_pool_release(stream);
    return rc;
}

}  // namespace plk
