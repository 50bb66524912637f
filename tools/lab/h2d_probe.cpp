// tools/h2d_probe.cpp -- how fast can a caller's PAGEABLE buffer reach HBM?  (the host-pointer entry points of the C ABI:
// the reference hands over plain Vecs, src/plonk_util.rs:169-231).  Variants: staging memcpy + DMA (round 2's path), the
// runtime's own pageable copy, hipHostRegister + DMA, chunked staging with the copy of chunk k overlapping the DMA of
// chunk k-1, the same with several staging threads; pinned DMA as the upper bound.
// Build: hipcc -O2 -std=c++17 -pthread -o build/h2d_probe tools/h2d_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <utility>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (size_t mib : {32, 288}) {
        const size_t bytes = mib << 20;
        char* src = (char*)malloc(bytes); memset(src, 1, bytes);
        void* d; CK(hipMalloc(&d, bytes));
        char* pin; CK(hipHostMalloc((void**)&pin, bytes, hipHostMallocDefault)); memset(pin, 2, bytes);
        auto rep = [&](const char* name, auto fn) {
            fn(); double best = 1e9;
            for (int r = 0; r < 3; ++r) { double t0 = now(); fn(); double t = now() - t0; if (t < best) best = t; }
            printf("%4zu MiB  %-44s %8.3f ms  %6.1f GB/s\n", mib, name, best * 1e3, bytes / best / 1e9);
        };
        rep("pinned DMA (upper bound)", [&] { CK(hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); });
        rep("memcpy to pinned, then DMA (round 2)", [&] { memcpy(pin, src, bytes); CK(hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); });
        rep("hipMemcpyAsync from pageable", [&] { CK(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); });
        rep("hipHostRegister + DMA + unregister", [&] { CK(hipHostRegister(src, bytes, hipHostRegisterDefault)); CK(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); CK(hipHostUnregister(src)); });
        for (size_t chunk_mib : {2, 8}) {
            const size_t chunk = chunk_mib << 20;
            char name[64]; snprintf(name, sizeof name, "chunked staging %zu MiB, 1 thread", chunk_mib);
            rep(name, [&] {
                hipEvent_t ev[2]; for (int q = 0; q < 2; ++q) CK(hipEventCreateWithFlags(&ev[q], hipEventDisableTiming));
                for (size_t off = 0, k = 0; off < bytes; off += chunk, ++k) {
                    const size_t len = bytes - off < chunk ? bytes - off : chunk;
                    char* st = pin + (k & 1) * chunk;
                    if (k >= 2) CK(hipEventSynchronize(ev[k & 1]));
                    memcpy(st, src + off, len);
                    CK(hipMemcpyAsync((char*)d + off, st, len, hipMemcpyHostToDevice, s));
                    CK(hipEventRecord(ev[k & 1], s));
                }
                CK(hipStreamSynchronize(s)); for (int q = 0; q < 2; ++q) CK(hipEventDestroy(ev[q]));
            });
        }
        for (int nt : {2, 4, 8}) {
            char name[64]; snprintf(name, sizeof name, "chunked staging 8 MiB, %d copy threads", nt);
            rep(name, [&] {
                const size_t chunk = (size_t)8 << 20;
                hipEvent_t ev[2]; for (int q = 0; q < 2; ++q) CK(hipEventCreateWithFlags(&ev[q], hipEventDisableTiming));
                for (size_t off = 0, k = 0; off < bytes; off += chunk, ++k) {
                    const size_t len = bytes - off < chunk ? bytes - off : chunk;
                    char* st = pin + (k & 1) * chunk;
                    if (k >= 2) CK(hipEventSynchronize(ev[k & 1]));
                    std::vector<std::thread> th;
                    const size_t per = (len + nt - 1) / nt;
                    for (int t = 0; t < nt; ++t) { size_t a = t * per, b = a + per > len ? len : a + per; if (a < b) th.emplace_back([=] { memcpy(st + a, src + off + a, b - a); }); }
                    for (auto& t : th) t.join();
                    CK(hipMemcpyAsync((char*)d + off, st, len, hipMemcpyHostToDevice, s));
                    CK(hipEventRecord(ev[k & 1], s));
                }
                CK(hipStreamSynchronize(s)); for (int q = 0; q < 2; ++q) CK(hipEventDestroy(ev[q]));
            });
        }
        rep("D2H to pinned (upper bound)", [&] { CK(hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
        rep("D2H to pageable (runtime)", [&] { CK(hipMemcpyAsync(src, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
        rep("D2H to pinned, then memcpy (round 2)", [&] { CK(hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); memcpy(src, pin, bytes); });
        free(src); CK(hipFree(d)); CK(hipHostFree(pin));
    }
    // the shape of plk_ntt_batch: nine 32 MiB buffers, each uploaded and downloaded, on three streams round-robin
    {
        const size_t bytes = (size_t)32 << 20; const int nb = 9;
        hipStream_t st[3]; for (int q = 0; q < 3; ++q) CK(hipStreamCreateWithFlags(&st[q], hipStreamNonBlocking));
        void* d; CK(hipMalloc(&d, bytes * nb));
        std::vector<char*> pag(nb), pin(nb);
        for (int b = 0; b < nb; ++b) { pag[b] = (char*)malloc(bytes); memset(pag[b], b, bytes); CK(hipHostMalloc((void**)&pin[b], bytes, hipHostMallocDefault)); memset(pin[b], b, bytes); }
        auto pipe = [&](std::vector<char*>& h, bool reg) {
            double t0 = now();
            if (reg) for (int b = 0; b < nb; ++b) CK(hipHostRegister(h[b], bytes, hipHostRegisterDefault));
            double t_enq0 = now();
            for (int b = 0; b < nb; ++b) {
                CK(hipMemcpyAsync((char*)d + b * bytes, h[b], bytes, hipMemcpyHostToDevice, st[b % 3]));
                CK(hipMemcpyAsync(h[b], (char*)d + b * bytes, bytes, hipMemcpyDeviceToHost, st[b % 3]));
            }
            double t_enq = now() - t_enq0;
            for (int q = 0; q < 3; ++q) CK(hipStreamSynchronize(st[q]));
            if (reg) for (int b = 0; b < nb; ++b) CK(hipHostUnregister(h[b]));
            return std::make_pair(now() - t0, t_enq);
        };
        for (int rep = 0; rep < 2; ++rep) {
            auto a = pipe(pin, false); printf("9 x (H2D + D2H) 32 MiB on 3 streams, pinned:               %8.3f ms (enqueue %.3f ms)\n", a.first * 1e3, a.second * 1e3);
            auto b = pipe(pag, true);  printf("9 x (H2D + D2H) 32 MiB on 3 streams, registered pageable:  %8.3f ms (enqueue %.3f ms)\n", b.first * 1e3, b.second * 1e3);
            auto c = pipe(pag, false); printf("9 x (H2D + D2H) 32 MiB on 3 streams, plain pageable:       %8.3f ms (enqueue %.3f ms)\n", c.first * 1e3, c.second * 1e3);
        }
    }
    return 0;
}
