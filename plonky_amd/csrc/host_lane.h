// host_lane.h -- the host-pointer entry points' plumbing: one lane per (calling thread, device).
// Shared by capi.hip (the extern "C" surface) and multi.hip (the fan-out over the devices of a group).
//
// The reference calls the hot path from Rayon workers (plonk_util.rs:173-189: nine transforms / commitments at once).  Every
// host thread that enters through a host-pointer entry point owns a lane per device it works on: non-blocking streams of its own,
// a small pinned staging buffer, device buffers from the scratch pool.  Nothing goes through the null stream or through hipMalloc /
// hipFree (both synchronise the whole device), so concurrent callers overlap on the GPU instead of queueing behind each other.
//
// The caller's buffers are pageable.  Measured on the MI355X box (profiles/r03_h2d_probe.txt): hipMemcpyAsync straight from /
// to pageable memory runs at the pinned rate (56.5 GB/s both ways) where a staging memcpy + DMA - round 2's path - reaches
// 21.9 GB/s; it blocks the calling thread, though.  Registered with hipHostRegister (57 GB/s INCLUDING registration and
// deregistration) the copies are asynchronous, so ONE caller thread keeps several streams busy: the copy of scalar vector
// k + 1 runs under the reduction of vector k, the upload of transform k + 1 under the download of transform k.  So: large
// buffers are registered for the duration of the call and copied directly; only small pieces (results, flags) are staged.
#pragma once
#include <cstring>
#include <memory>
#include <vector>

#include "common.h"

namespace plk {

constexpr size_t PIN_CAP = (size_t)4 << 20;        // the staging buffer serves pieces up to this size
constexpr size_t DIRECT_MIN = (size_t)64 << 10;    // larger pieces are copied straight from / to the caller's memory
constexpr int LANE_AUX = 2;

struct HostLane {
    hipStream_t stream = nullptr;
    hipStream_t aux[LANE_AUX] = {nullptr, nullptr};  // second / third stream of a call that pipelines copies and kernels
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_ready;                 // per-vector "copy done" events of a batched MSM
    int device = -1;                                  // physical device the streams belong to
    uint8_t* pin = nullptr;
    size_t pin_bytes = 0, pin_used = 0;
    void drop_streams() {
        for (hipEvent_t e : ev_ready) (void)hipEventDestroy(e);
        ev_ready.clear();
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        ev_fork = nullptr;
        for (auto& a : aux) {
            if (a) (void)hipStreamSynchronize(a);
            stream_pool_release(a);
            a = nullptr;
        }
        if (stream) (void)hipStreamSynchronize(stream);
        stream_pool_release(stream);
        stream = nullptr;
    }
    ~HostLane() {
        if (pin) (void)hipHostFree(pin);
        drop_streams();
    }
};

// the calling thread's lane on its current logical device (ensure_device() has selected it)
int lane_get(HostLane*& out);

// the lane's extra streams, ordered after everything enqueued on the main one so far
inline int lane_fork(HostLane& l) {
    if (!l.ev_fork) PLK_HIP_TRY(hipEventCreateWithFlags(&l.ev_fork, hipEventDisableTiming));
    PLK_HIP_TRY(hipEventRecord(l.ev_fork, l.stream));
    for (auto& a : l.aux) {
        if (!a && !(a = stream_pool_acquire())) return PLK_ERR_HIP;
        PLK_HIP_TRY(hipStreamWaitEvent(a, l.ev_fork, 0));
    }
    return PLK_OK;
}
inline int lane_join(HostLane& l) {
    hipError_t first = hipSuccess;
    for (auto& a : l.aux)
        if (a) {
            const hipError_t e = hipStreamSynchronize(a);
            if (first == hipSuccess) first = e;
        }
    const hipError_t e = hipStreamSynchronize(l.stream);
    if (first == hipSuccess) first = e;
    l.pin_used = 0;
    PLK_HIP_TRY(first);
    return PLK_OK;
}

// Caller buffers registered with the HIP runtime so that copies from / to them are asynchronous: a process-wide registry of
// NON-OVERLAPPING intervals, each with a hold count per thread (capi.hip).  Two threads that pass the same read-only buffer at the
// same time (the Rayon-style callers the lanes are built for) share ONE registration, released by whoever finishes last.  A request
// that extends or partly overlaps a registered interval never leaves a half-registered range: it waits for other threads' holds to
// end, or - the calling thread's own holds - re-registers the union after synchronising the devices.  Registration can still fail
// (exotic memory): the copies then block the calling thread - same result.  Registered as portable: every device of the group may
// copy from it.  A hold is released by the thread that took it, with the address it was taken with.
bool pin_registry_acquire(const void* ptr, size_t bytes);  // true: registered (by this call or an earlier one), must be released
void pin_registry_release(const void* ptr);

struct HostPin {
    const void* p = nullptr;
    HostPin() = default;
    HostPin(const HostPin&) = delete;
    HostPin& operator=(const HostPin&) = delete;
    HostPin(HostPin&& o) noexcept : p(o.p) { o.p = nullptr; }
    void pin(const void* ptr, size_t bytes) {
        if (!ptr || bytes < ((size_t)1 << 20) || p) return;
        if (pin_registry_acquire(ptr, bytes)) p = ptr;
    }
    void unpin() {
        if (p) pin_registry_release(p);
        p = nullptr;
    }
    ~HostPin() { unpin(); }
};

// a piece of the lane's pinned buffer, valid until the lane is synchronised; nullptr when it does not fit
inline uint8_t* lane_stage(HostLane& l, size_t bytes) {
    const size_t need = (l.pin_used + bytes + 255) & ~(size_t)255;
    if (need > PIN_CAP) return nullptr;
    if (need > l.pin_bytes) {
        if (l.pin_used) return nullptr;  // pieces handed out earlier in this call are still in flight
        size_t want = l.pin_bytes ? l.pin_bytes : ((size_t)256 << 10);
        while (want < need) want *= 2;
        if (l.pin) (void)hipHostFree(l.pin);
        l.pin = nullptr;
        l.pin_bytes = 0;
        if (hipHostMalloc((void**)&l.pin, want, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        l.pin_bytes = want;
    }
    uint8_t* p = l.pin + l.pin_used;
    l.pin_used = need;
    return p;
}
inline int lane_h2d(HostLane& l, void* d, const void* h, size_t bytes, hipStream_t st = nullptr) {
    if (!bytes) return PLK_OK;
    if (!st) st = l.stream;
    uint8_t* stg = bytes < DIRECT_MIN ? lane_stage(l, bytes) : nullptr;
    if (stg) {
        memcpy(stg, h, bytes);
        PLK_HIP_TRY(hipMemcpyAsync(d, stg, bytes, hipMemcpyHostToDevice, st));
    } else {
        PLK_HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));  // pageable or registered: the pinned rate either way
    }
    return PLK_OK;
}
// device -> caller memory; `flush` pairs (staging piece, destination) are copied out by lane_finish after the synchronisation
struct LaneOut {
    uint8_t* st;
    void* dst;
    size_t bytes;
};
inline int lane_d2h(HostLane& l, std::vector<LaneOut>& outs, void* h, const void* d, size_t bytes, hipStream_t st = nullptr) {
    if (!bytes) return PLK_OK;
    if (!st) st = l.stream;
    uint8_t* stg = bytes < DIRECT_MIN ? lane_stage(l, bytes) : nullptr;
    if (stg) {
        PLK_HIP_TRY(hipMemcpyAsync(stg, d, bytes, hipMemcpyDeviceToHost, st));
        outs.push_back({stg, h, bytes});
    } else {
        PLK_HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
    }
    return PLK_OK;
}
inline int lane_finish(HostLane& l, std::vector<LaneOut>& outs) {
    PLK_HIP_TRY(hipStreamSynchronize(l.stream));
    for (const LaneOut& o : outs) memcpy(o.dst, o.st, o.bytes);
    outs.clear();
    l.pin_used = 0;
    return PLK_OK;
}
// device buffer of a lane, from the scratch pool
struct LaneBuf {
    void* p = nullptr;
    hipStream_t s = nullptr;
    LaneBuf() = default;
    LaneBuf(const LaneBuf&) = delete;
    LaneBuf& operator=(const LaneBuf&) = delete;
    ~LaneBuf() {
        if (p) scratch_release(p, s);
    }
    int alloc(size_t bytes, hipStream_t stream) {
        s = stream;
        p = scratch_acquire(bytes ? bytes : 16, stream);
        return p ? PLK_OK : PLK_ERR_OOM;
    }
};
// One host-pointer call on the calling thread's lane: device buffers, uploads, downloads, one synchronisation at the end.
// Members are destroyed in reverse order of declaration: the stream is synchronised (~LaneCall body) BEFORE the caller-buffer
// registrations (`pins`) and the device buffers go, on every exit path.
struct LaneCall {
    HostLane* l = nullptr;
    std::vector<LaneOut> outs;
    std::vector<std::unique_ptr<LaneBuf>> bufs;
    std::vector<HostPin> pins;  // caller buffers registered for the duration of the call
    int begin() { return lane_get(l); }
    hipStream_t stream() const { return l->stream; }
    void pin(const void* ptr, size_t bytes) {
        pins.emplace_back();
        pins.back().pin(ptr, bytes);
    }
    int tmp(void*& d, size_t bytes) {
        bufs.emplace_back(new LaneBuf());
        PLK_TRY(bufs.back()->alloc(bytes, l->stream));
        d = bufs.back()->p;
        return PLK_OK;
    }
    int in(void*& d, const void* h, size_t bytes) {
        PLK_TRY(tmp(d, bytes));
        return lane_h2d(*l, d, h, bytes);
    }
    int out(void* h, const void* d, size_t bytes) { return lane_d2h(*l, outs, h, d, bytes); }
    int sync() {  // results of the kernels so far are needed on the host before the call goes on
        PLK_TRY(lane_finish(*l, outs));
        return PLK_OK;
    }
    bool done = false;
    int finish() {
        done = true;
        return lane_finish(*l, outs);
    }
    ~LaneCall() {
        if (l && !done) {  // an early return: nothing may stay in flight over the staging buffer or the caller's memory
            for (auto& a : l->aux)
                if (a) (void)hipStreamSynchronize(a);
            (void)hipStreamSynchronize(l->stream);
            l->pin_used = 0;
        }
        pins.clear();  // only now: every copy from / to the registered ranges has completed
    }
};

}  // namespace plk
