// tools/lab/sync_latency.hip -- what one "result back on the host" costs at the end of a dependency chain (an IPA round):
//   A: kernel -> hipMemcpyAsync(pinned <- device, 200 B) -> hipStreamSynchronize      (halo.hip today)
//   B: kernel writes into mapped pinned host memory -> hipStreamSynchronize
//   C: B, but the host polls a flag the kernel writes last (after __threadfence_system) instead of synchronising
// hipcc --offload-arch=gfx950 -O2 tools/lab/sync_latency.hip -o /tmp/sync_latency && /tmp/sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
__global__ void k_work(unsigned* out, unsigned v, volatile unsigned* flag) {
    // ~20 us of dependent work on one lane, like the end of a reduction
    unsigned x = v;
    for (int i = 0; i < 20000; ++i) x = x * 1664525u + 1013904223u;
    for (int i = 0; i < 48; ++i) out[i] = x + i;
    if (flag) {
        __threadfence_system();
        *flag = v;
    }
}
int main() {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned *dev, *pin, *map, *map_dev;
    hipMalloc(&dev, 256);
    hipHostMalloc((void**)&pin, 256, hipHostMallocDefault);
    hipHostMalloc((void**)&map, 512, hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&map_dev, map, 0);
    const int K = 500;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int mode = -1; mode < 3; ++mode) {
        auto t0 = now();
        for (int i = 1; i <= K; ++i) {
            if (mode <= 0) {  // -1: warm-up of A
                k_work<<<1, 1, 0, st>>>(dev, i, nullptr);
                hipMemcpyAsync(pin, dev, 200, hipMemcpyDeviceToHost, st);
                hipStreamSynchronize(st);
            } else if (mode == 1) {
                k_work<<<1, 1, 0, st>>>(map_dev, i, nullptr);
                hipStreamSynchronize(st);
            } else {
                volatile unsigned* f = map + 64;
                k_work<<<1, 1, 0, st>>>(map_dev, i, map_dev + 64);
                while (*f != (unsigned)i) {}
            }
        }
        hipStreamSynchronize(st);
        if (mode >= 0) printf("%s: %.1f us per round (kernel ~ same in all)\n", mode == 0 ? "A memcpyAsync + synchronize" : mode == 1 ? "B mapped host memory + synchronize" : "C mapped host memory + polled flag", us(t0, now()) / K);
    }
    // the kernel alone (events)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < 50; ++i) k_work<<<1, 1, 0, st>>>(dev, i, nullptr);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("kernel alone: %.1f us\n", ms * 1000 / 50);
    return 0;
}
