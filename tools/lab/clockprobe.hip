#include <hip/hip_runtime.h>
#include <stdint.h>
// one wave per XCD-ish (8 blocks): samples shader clock (clock64) against the 100 MHz wall clock for `windows` windows of `us` microseconds
__global__ void k_probe(unsigned long long* out, int windows, int us) {
    if (threadIdx.x != 0) return;
    for (int w = 0; w < windows; ++w) {
        unsigned long long t0 = wall_clock64(), c0 = clock64();
        unsigned long long t1 = t0;
        while (t1 - t0 < (unsigned long long)us * 100) { __builtin_amdgcn_s_sleep(16); t1 = wall_clock64(); }
        unsigned long long c1 = clock64();
        out[(blockIdx.x * windows + w) * 2] = t1 - t0;
        out[(blockIdx.x * windows + w) * 2 + 1] = c1 - c0;
    }
}
extern "C" int probe_launch(void* out, int blocks, int windows, int us, void* stream) {
    k_probe<<<blocks, 64, 0, (hipStream_t)stream>>>((unsigned long long*)out, windows, us);
    return (int)hipGetLastError();
}
