// capi.hip -- the extern "C" surface declared in include/plonky_hip.h.
// Thin: argument validation, host<->device staging for the host-pointer variants, dispatch.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"
#include "host_lane.h"

struct plk_msm_ctx;
struct plk_halo_ctx;

namespace plk {

int halo_begin_dev_impl(int curve, size_t n, const void* d_a, const void* d_b, const void* d_g, const void* d_gz, const uint64_t* h_xy,
                        const uint64_t* u_xy, unsigned freeze_log, hipStream_t stream, plk_halo_ctx** out, plk_msm_ctx* tables = nullptr,
                        unsigned lead_rounds = 0, size_t h_index = (size_t)-1, size_t u_index = (size_t)-1, const uint64_t* u_prime_scalar = nullptr);
int curve_fold_multi_dev_impl(int curve, size_t n_out, int r_bits, const void* d_g, const void* d_gz, const void* d_ratios, void* d_out_xy,
                              void* d_out_zero, hipStream_t stream);
int halo_round_lr_impl(plk_halo_ctx* c, const uint64_t* l_blind, const uint64_t* r_blind, uint64_t* lr_xy, uint8_t* lr_zero);
int halo_round_fold_impl(plk_halo_ctx* c, const uint64_t* u_j, const uint64_t* u_j_inv);
size_t halo_len_impl(const plk_halo_ctx* c);
int halo_frozen_impl(const plk_halo_ctx* c);
int halo_read_impl(plk_halo_ctx* c, uint64_t* a, uint64_t* b, uint64_t* g_xy, uint8_t* g_zero);
void halo_delete(plk_halo_ctx* c);

int field_op_impl(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count);
int msm_precompute_dev_impl(int curve, size_t n, const void* d_bases, const void* d_zero, unsigned window_bits, unsigned flags, hipStream_t stream,
                            plk_msm_ctx** out_ctx, const void* d_extra = nullptr, size_t n_extra = 0, const size_t* also_n = nullptr,
                            int also_count = 0);
int msm_execute_dev_impl(plk_msm_ctx* ctx, unsigned batch, const void* d_scalars, size_t n_scalars, void* d_out_xy, void* d_out_zero, hipStream_t stream,
                         hipEvent_t* ready = nullptr, const MsmParts* parts = nullptr, unsigned out_flags = 0);
int msm_reserve_workspaces_impl(plk_msm_ctx* ctx, unsigned count, hipStream_t stream);
int curve_sum_affine_dev_impl(int curve, size_t k, const void* d_pts, const void* d_zero, void* d_out_xy, void* d_out_zero, hipStream_t stream);
int curve_gen_bases_dev_impl(int curve, size_t n, uint64_t first, const void* d_g0d, void* d_out, hipStream_t stream);
size_t msm_partials_bytes(int curve, unsigned batch);
int checked_build_impl();
int checked_failures_impl(unsigned* counts);
int msm_debug_digits_impl(int curve, unsigned window_bits, size_t n, const void* d_scalars, void* d_digits, hipStream_t stream);
int msm_debug_digit_count(int curve, unsigned window_bits);
int bench_ceilings_impl(double* out, unsigned n_out);
int msm_combine_partials_dev_impl(int curve, unsigned world, unsigned batch, unsigned whole_per_rank, const void* d_gathered, void* d_out_xy, void* d_out_zero,
                                  hipStream_t stream);
int curve_fold_pairs_dev_impl(int curve, size_t m, const void* d_lo, const void* d_lo_zero, const void* d_hi, const void* d_hi_zero,
                              const uint64_t* a_mont, const uint64_t* b_mont, void* d_out_xy, void* d_out_zero, hipStream_t stream,
                              const void* d_scalars = nullptr, int plus_lo = 0);
int msm_table_digits(int curve, unsigned w);
int msm_reference_table_dev_impl(int curve, size_t n, const void* d_bases, const void* d_zero, unsigned w, void* d_out_xy, void* d_out_zero,
                                 hipStream_t stream);
int selftest_quad_dev_impl(int curve, const void* d_pts, uint32_t n, uint32_t quads, uint32_t* counts);
int msm_set_profiling_impl(plk_msm_ctx* ctx, int enable);
int msm_get_timings_impl(plk_msm_ctx* ctx, double* sum_ms, unsigned* calls);
int ntt_set_profiling_impl(int enable);
int ntt_get_timings_impl(double* sum_ms, unsigned* launches);
int poly_clear_cache_impl();
int poly_divide_by_z_h_dev_impl(int field, const void* d_coeffs, size_t len, size_t n, void* d_out, size_t out_cap, size_t* out_len,
                                hipStream_t stream);
int poly_mul_dev_impl(int field, const void* d_a, size_t la, const void* d_b, size_t lb, void* d_out, size_t out_cap, size_t* out_len,
                      hipStream_t stream);
int ntt_padded_dev_impl(int field, unsigned log_n, unsigned batch, const void* d_in, size_t in_len, size_t in_stride, void* d_out,
                        hipStream_t stream);
size_t msm_ctx_len(const plk_msm_ctx* ctx);
unsigned msm_ctx_window(const plk_msm_ctx* ctx);
int msm_ctx_curve(const plk_msm_ctx* ctx);
int msm_ctx_is_comb(const plk_msm_ctx* ctx);
int msm_affine_to_projective_impl(int curve, unsigned batch, const void* d_xy, const void* d_zero, void* d_xyz, hipStream_t stream);
void msm_ctx_delete(plk_msm_ctx* ctx);
// multi.hip
void multi_plan_slot(int world, unsigned batch, size_t n, int d, unsigned slot, unsigned* vec, size_t* first, size_t* count);
bool msm_ctx_is_multi(plk_msm_ctx* ctx);
void group_copy_stats(unsigned long long* peer, unsigned long long* staged);
int msm_ctx_device(const plk_msm_ctx* ctx);
int msm_precompute_multi(int curve, size_t n, const void* bases, const void* zero, bool host_src, unsigned window_bits, hipStream_t caller_stream,
                         plk_msm_ctx** out_ctx);
int msm_execute_multi(plk_msm_ctx* ctx, unsigned batch, const void* const* vecs, bool host_src, size_t n, void* d_out_xy, void* d_out_zero,
                      hipStream_t caller_stream);

std::string& last_error_ref() {
    static thread_local std::string s;
    return s;
}
int set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

// ---- scratch pool ----
struct ScratchEntry {
    void* p = nullptr;
    size_t bytes = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    bool in_call = false;   // between acquire and release on some host thread
    bool used = false;      // ev has been recorded at least once
};
static std::mutex g_scratch_mu;
static std::vector<ScratchEntry> g_scratch;

void* scratch_acquire(size_t bytes, hipStream_t stream) {
    if (bytes == 0) bytes = 16;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    ScratchEntry* best = nullptr;
    for (auto& e : g_scratch) {
        if (e.in_call || e.device != dev || e.bytes < bytes) continue;
        // a small request must not take (and keep) a buffer sized for a whole transform: at most 4x the request above 1 MiB
        if (e.bytes > ((size_t)1 << 20) && e.bytes / 4 > bytes) continue;
        const bool ordered = !e.used || e.stream == stream || hipEventQuery(e.ev) == hipSuccess;
        if (!ordered) continue;
        if (!best || e.bytes < best->bytes) best = &e;
    }
    if (best) {
        best->in_call = true;
        return best->p;
    }
    ScratchEntry e;
    e.bytes = bytes;
    e.device = dev;
    hipError_t err = hipMalloc(&e.p, bytes);
    if (err != hipSuccess) {
        set_error(err == hipErrorOutOfMemory ? PLK_ERR_OOM : PLK_ERR_HIP, "hipMalloc(%zu) for scratch failed: %s", bytes, hipGetErrorString(err));
        return nullptr;
    }
    if (hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(e.p);
        set_error(PLK_ERR_HIP, "hipEventCreate failed");
        return nullptr;
    }
    e.in_call = true;
    g_scratch.push_back(e);
    return e.p;
}

void scratch_release(void* p, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (auto& e : g_scratch)
        if (e.p == p) {
            (void)hipEventRecord(e.ev, stream);
            e.stream = stream;
            e.used = true;
            e.in_call = false;
            return;
        }
}

struct PooledStream {
    hipStream_t s;
    int device;
    bool free;
};
static std::mutex g_spool_mu;
static std::vector<PooledStream> g_spool;

hipStream_t stream_pool_acquire() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_spool_mu);
    for (auto& e : g_spool)
        if (e.free && e.device == dev) {
            e.free = false;
            return e.s;
        }
    hipStream_t s = nullptr;
    hipError_t err = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (err != hipSuccess) {
        set_error(PLK_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(err));
        return nullptr;
    }
    g_spool.push_back({s, dev, false});
    return s;
}
void stream_pool_release(hipStream_t s) {
    if (!s) return;
    std::lock_guard<std::mutex> lk(g_spool_mu);
    for (auto& e : g_spool)
        if (e.s == s) e.free = true;
}

void scratch_clear() {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < group_size(); ++d) {  // every device of the group is quiet before its buffers go
        if (d && group_phys(d) == group_phys(d - 1)) continue;
        (void)hipSetDevice(group_phys(d));
        (void)hipDeviceSynchronize();
    }
    (void)hipSetDevice(cur);
    (void)hipDeviceSynchronize();
    for (auto& e : g_scratch) {
        if (e.in_call) continue;
        (void)hipFree(e.p);
        (void)hipEventDestroy(e.ev);
        e.p = nullptr;
    }
    std::vector<ScratchEntry> keep;
    for (auto& e : g_scratch)
        if (e.p) keep.push_back(e);
    g_scratch.swap(keep);
}

// ---- host-pointer entry points: one lane per (calling thread, logical device) - host_lane.h ----
static thread_local HostLane t_lanes[PLK_MAX_DEVICES];

int lane_get(HostLane*& out) {
    PLK_TRY(ensure_device());
    int dev = 0;
    PLK_HIP_TRY(hipGetDevice(&dev));
    HostLane& l = t_lanes[thread_logical_device()];
    if (l.stream && l.device != dev) l.drop_streams();  // the group was re-initialised over other devices
    if (!l.stream) {
        l.stream = stream_pool_acquire();
        if (!l.stream) return PLK_ERR_HIP;
        l.device = dev;
    }
    if (l.pin_used) (void)hipStreamSynchronize(l.stream);  // a call that failed half way may have left copies in flight
    l.pin_used = 0;
    out = &l;
    return PLK_OK;
}

// Registered caller ranges: NON-OVERLAPPING intervals, each with the threads that hold it (host_lane.h).  A request inside a
// registered interval shares it; a request that extends or partly overlaps one never leaves a half-registered range behind (HIP
// resolves a pointer to the registered object it starts in - an asynchronous copy longer than that object is rejected, and the
// first holder's release would unregister pages under the second caller's DMA):
//  * held by OTHER threads: wait until they have released it, then register the union - for at most two seconds in all: holders that
//    in turn wait for this thread (or for a lock its dispatcher holds) would otherwise never release; past that the request fails and
//    the caller's copies go without a registration;
//  * held by the calling thread alone (two overlapping buffers of one batched call, an in-place padded transform): every device
//    of the group is synchronised - the thread's own copies over the old interval are complete - and the interval is re-registered
//    as the union, the thread's holds carried over.
struct PinEntry {
    size_t bytes;
    std::map<std::thread::id, unsigned> holders;
    unsigned refs() const {
        unsigned r = 0;
        for (const auto& h : holders) r += h.second;
        return r;
    }
};
static std::mutex g_pin_mu;
static std::condition_variable g_pin_cv;
static std::map<const uint8_t*, PinEntry> g_pins;  // by start address; intervals never overlap

static void pin_sync_all_devices() {
    int cur = -1;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < group_size(); ++d) {
        if (d && group_phys(d) == group_phys(d - 1)) continue;
        (void)hipSetDevice(group_phys(d));
        (void)hipDeviceSynchronize();
    }
    if (cur >= 0) (void)hipSetDevice(cur);
}
bool pin_registry_acquire(const void* ptr, size_t bytes) {
    if (!ptr || !bytes) return false;
    const uint8_t* lo = (const uint8_t*)ptr;
    const uint8_t* hi = lo + bytes;
    const std::thread::id me = std::this_thread::get_id();
    constexpr long PIN_WAIT_MS = 2000;
    long waited_ms = 0;
    std::unique_lock<std::mutex> lk(g_pin_mu);
    for (;;) {
        // the registered intervals that touch [lo, hi)
        std::vector<std::map<const uint8_t*, PinEntry>::iterator> hit;
        auto it = g_pins.upper_bound(lo);
        if (it != g_pins.begin()) --it;
        for (; it != g_pins.end() && it->first < hi; ++it)
            if (it->first + it->second.bytes > lo) hit.push_back(it);
        if (hit.empty()) {
            if (hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterPortable) != hipSuccess) {
                (void)hipGetLastError();
                return false;  // exotic memory: the copies block the caller instead
            }
            PinEntry e{bytes, {}};
            e.holders[me] = 1;
            g_pins[lo] = e;
            return true;
        }
        if (hit.size() == 1 && hit[0]->first <= lo && hit[0]->first + hit[0]->second.bytes >= hi) {
            ++hit[0]->second.holders[me];
            return true;
        }
        bool others = false;
        for (auto& h : hit)
            for (const auto& who : h->second.holders)
                if (who.first != me && who.second) others = true;
        if (others) {
            // Never wait unboundedly: two threads that each hold a range and ask for one overlapping the other's, or a
            // worker that waits here while its dispatcher holds the fan-out lock another holder is queued for, would wait for ever.  After
            // PIN_WAIT_MS in all the request goes on WITHOUT a registration of its own - the caller's copies then take the runtime's
            // pageable path (they block the calling thread), as they do when a registration fails.
            if (waited_ms >= PIN_WAIT_MS) return false;
            const auto t0 = std::chrono::steady_clock::now();
            (void)g_pin_cv.wait_for(lk, std::chrono::milliseconds(PIN_WAIT_MS - waited_ms));  // a release wakes us; then look again
            waited_ms += (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() + 1;
            continue;
        }
        // every interval in the way is held by this thread alone: replace them by their union with the request
        const uint8_t* ulo = lo;
        const uint8_t* uhi = hi;
        unsigned mine = 1;
        for (auto& h : hit) {
            if (h->first < ulo) ulo = h->first;
            if (h->first + h->second.bytes > uhi) uhi = h->first + h->second.bytes;
            mine += h->second.refs();
        }
        lk.unlock();
        pin_sync_all_devices();  // this thread's copies over the old intervals are complete
        lk.lock();
        // nobody else can have taken a hold meanwhile without waiting for us?  They can (a request INSIDE one of our intervals):
        // look again in that case
        bool changed = false;
        for (auto& h : hit)
            for (const auto& who : h->second.holders)
                if (who.first != me && who.second) changed = true;
        if (changed) continue;
        for (auto& h : hit) {
            (void)hipHostUnregister(const_cast<uint8_t*>(h->first));
            g_pins.erase(h);
        }
        if (hipHostRegister(const_cast<uint8_t*>(ulo), (size_t)(uhi - ulo), hipHostRegisterPortable) != hipSuccess) {
            (void)hipGetLastError();
            // nothing of the union is registered now: this thread's earlier holds are gone with it (its copies become blocking
            // ones, which is correct), and releases of them find no interval and do nothing
            g_pin_cv.notify_all();
            return false;
        }
        PinEntry e{(size_t)(uhi - ulo), {}};
        e.holders[me] = mine;
        g_pins[ulo] = e;
        return true;
    }
}
// ptr: any address a successful acquire was made with (it lies inside exactly one interval)
void pin_registry_release(const void* ptr) {
    const uint8_t* p = (const uint8_t*)ptr;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto it = g_pins.upper_bound(p);
    if (it == g_pins.begin()) return;
    --it;
    if (p >= it->first + it->second.bytes) return;
    auto h = it->second.holders.find(std::this_thread::get_id());
    if (h == it->second.holders.end()) return;  // not ours: a hold is released by the thread that took it (HostPin lives in one call object)
    if (h->second && --h->second == 0) it->second.holders.erase(h);
    if (it->second.refs() == 0) {
        (void)hipHostUnregister(const_cast<uint8_t*>(it->first));
        g_pins.erase(it);
    }
    g_pin_cv.notify_all();
}

int field_limbs(int field) {
    switch (field) {
        case PLK_FIELD_TWEEDLEDEE_BASE:
        case PLK_FIELD_TWEEDLEDUM_BASE:
        case PLK_FIELD_BLS12_377_SCALAR:
        case PLK_FIELD_PALLAS_BASE:
        case PLK_FIELD_VESTA_BASE: return 4;
        case PLK_FIELD_BLS12_377_BASE: return 6;
    }
    return PLK_ERR_INVALID_ARG;
}
int curve_limbs(int curve) {
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE:
        case PLK_CURVE_TWEEDLEDUM:
        case PLK_CURVE_PALLAS:
        case PLK_CURVE_VESTA: return 4;
        case PLK_CURVE_BLS12_377: return 6;
    }
    return PLK_ERR_INVALID_ARG;
}
int curve_scalar_field(int curve) {
    switch (curve) {
        case PLK_CURVE_TWEEDLEDEE: return PLK_FIELD_TWEEDLEDUM_BASE;
        case PLK_CURVE_TWEEDLEDUM: return PLK_FIELD_TWEEDLEDEE_BASE;
        case PLK_CURVE_BLS12_377: return PLK_FIELD_BLS12_377_SCALAR;
        case PLK_CURVE_PALLAS: return PLK_FIELD_VESTA_BASE;
        case PLK_CURVE_VESTA: return PLK_FIELD_PALLAS_BASE;
    }
    return PLK_ERR_INVALID_ARG;
}

}  // namespace plk

using namespace plk;

extern "C" {

int plk_init(int device) { return group_init_single(device); }
int plk_init_devices(int n_devices) { return group_init(n_devices); }
int plk_device_count(void) { return group_size(); }
int plk_set_thread_device(int logical_device) {
    if (logical_device < 0 || logical_device >= group_size())
        return set_error(PLK_ERR_INVALID_ARG, "logical device %d out of range (%d in use)", logical_device, group_size());
    set_thread_logical_device(logical_device);
    return ensure_device();
}

int plk_thread_hip_device(int set_to) {
    if (set_to >= 0) PLK_HIP_TRY(hipSetDevice(set_to));
    int cur = -1;
    PLK_HIP_TRY(hipGetDevice(&cur));
    return cur;
}
int plk_group_copy_stats(unsigned long long* peer_copies, unsigned long long* staged_copies) {
    group_copy_stats(peer_copies, staged_copies);
    return PLK_OK;
}

int plk_multi_plan(unsigned world, unsigned batch, size_t n, unsigned device, unsigned* slots, unsigned* vec, uint64_t* first, uint64_t* count) {
    if (world == 0 || device >= world || !slots) return set_error(PLK_ERR_INVALID_ARG, "bad world / device");
    const unsigned whole = batch / world, total = whole + (batch - whole * world);
    *slots = total;
    for (unsigned s = 0; s < total && vec && first && count; ++s) {
        size_t f = 0, c = 0;
        multi_plan_slot((int)world, batch, n, (int)device, s, &vec[s], &f, &c);
        first[s] = f;
        count[s] = c;
    }
    return PLK_OK;
}

void plk_shutdown(void) {
    (void)plonk_clear_cache_impl();
    (void)poly_clear_cache_impl();
    (void)ntt_clear_cache_impl();
    scratch_clear();
    group_shutdown();
}

const char* plk_last_error(void) { return last_error_ref().c_str(); }
unsigned plk_min_gpu_log_n(void) {
    const char* e = getenv("PLK_MIN_GPU_LOG_N");
    // Default 12 - PROVISIONAL.  The crossover that is measured (profiles/r05_crossover_host_pointer_vs_cpu.txt: through the
    // HOST-pointer entry points, PCIe inside, a 2^10 transform takes 59 us against 148 us, a 2^8 one 34 against 37) is against the C++
    // RESTATEMENT of the reference's algorithm on this host, not against the reference's Rust on Rayon, which cannot be built here and is
    // faster by an unknown small factor; and the gate also covers MSM pairs, where a first call pays the precomputation.  Until the Rust
    // path itself has been timed beside it the gate stays at round 4's conservative 2^12; PLK_MIN_GPU_LOG_N=10 is the measured tie + 2.
    const int v = e ? atoi(e) : 12;
    return v < 0 ? 0u : (unsigned)v;
}
int plk_field_limbs(int field) { return field_limbs(field); }
int plk_curve_limbs(int curve) { return curve_limbs(curve); }
int plk_curve_scalar_field(int curve) { return curve_scalar_field(curve); }

// ---- NTT ----
int plk_ntt_precompute(int field, unsigned log_n) { PLK_API; return ntt_precompute_impl(field, log_n); }
int plk_ntt_clear_cache(void) {
    PLK_API;
    (void)plonk_clear_cache_impl();
    (void)poly_clear_cache_impl();
    const int rc = ntt_clear_cache_impl();
    scratch_clear();  // "memory pressure": the idle scratch buffers go as well
    return rc;
}

int plk_ntt_precompute_table_dev(int field, unsigned log_n, void* d_out, void* stream) {
    PLK_API;
    return ntt_reference_table_dev_impl(field, log_n, d_out, as_stream(stream));
}
int plk_ntt_precompute_table(int field, unsigned log_n, uint64_t* out) {
    PLK_API;
    if (field_limbs(field) < 0) return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    if (log_n > 30) return set_error(PLK_ERR_TWO_ADICITY, "log_n %u too large (max 30)", log_n);
    if (!out) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    HostLane* l = nullptr;
    PLK_TRY(lane_get(l));
    const size_t bytes = (((size_t)2 << log_n) - 1) * (size_t)field_limbs(field) * 8;
    LaneBuf buf;
    PLK_TRY(buf.alloc(bytes, l->stream));
    PLK_TRY(ntt_reference_table_dev_impl(field, log_n, buf.p, l->stream));
    std::vector<LaneOut> outs;
    PLK_TRY(lane_d2h(*l, outs, out, buf.p, bytes));
    return lane_finish(*l, outs);
}

int plk_ntt_dev(int field, unsigned log_n, int inverse, unsigned batch, const void* d_in, void* d_out, void* stream) {
    PLK_API;
    return ntt_dev_impl(field, log_n, inverse, batch, d_in, d_out, as_stream(stream));
}

// `batch` transforms from / to host memory on the calling thread's current device
static int ntt_batch_local(int field, unsigned log_n, int inverse, unsigned batch, const uint64_t* const* in, uint64_t* const* out) {
    if (batch == 0) return PLK_OK;
    const size_t bytes = ((size_t)1 << log_n) * (size_t)field_limbs(field) * 8;  // 32 per element, 48 for Bls12377Base
    LaneCall c;
    PLK_TRY(c.begin());
    HostLane* l = c.l;
    void* buf = nullptr;
    PLK_TRY(c.tmp(buf, bytes * batch));
    if (batch == 1 || bytes < ((size_t)1 << 20)) {
        // one transform, or small ones: upload, one batched launch, download
        for (unsigned b = 0; b < batch; ++b) PLK_TRY(lane_h2d(*l, (uint8_t*)buf + b * bytes, in[b], bytes));
        PLK_TRY(ntt_dev_impl(field, log_n, inverse, batch, buf, buf, l->stream));
        for (unsigned b = 0; b < batch; ++b) PLK_TRY(c.out(out[b], (uint8_t*)buf + b * bytes, bytes));
        return c.finish();
    }
    // several large transforms (the nine wire polynomials, plonk_util.rs:169-190): PCIe is the long pole (2 x 32 MiB per 2^20
    // transform against 0.12 ms of kernels).  Round 5: ONE stream per direction - the uploads follow each other at the full rate of
    // the link on aux[0], the kernels of transform b start on the main stream when its upload has landed (one event), its download
    // goes on aux[1] when they are done (one event): from the second transform on both directions of the link are busy all the time.
    // (Until round 4 transform b ran upload -> kernels -> download on stream b mod 3: three uploads shared the link, nothing ran
    // before all three had landed 1.8 ms in, and the last three downloads had the link to themselves; PLK_NTT_HOST_PIPE=0 is that form.)
    for (unsigned b = 0; b < batch; ++b) {
        c.pin(in[b], bytes);
        if ((const void*)out[b] != (const void*)in[b]) c.pin(out[b], bytes);
    }
    PLK_TRY(lane_fork(*l));
    const char* pe = getenv("PLK_NTT_HOST_PIPE");
    if (pe && atoi(pe) == 0) {
        for (unsigned b = 0; b < batch; ++b) {
            hipStream_t st = b % 3 == 0 ? l->stream : l->aux[b % 3 - 1];
            uint8_t* d = (uint8_t*)buf + b * bytes;
            if (hipMemcpyAsync(d, in[b], bytes, hipMemcpyHostToDevice, st) != hipSuccess) return set_error(PLK_ERR_HIP, "upload of transform %u failed", b);
            PLK_TRY(ntt_dev_impl(field, log_n, inverse, 1, d, d, st));
            if (hipMemcpyAsync(out[b], d, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return set_error(PLK_ERR_HIP, "download of transform %u failed", b);
        }
    } else {
        while (l->ev_ready.size() < 2 * (size_t)batch) {
            hipEvent_t e = nullptr;
            PLK_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            l->ev_ready.push_back(e);
        }
        hipStream_t up = l->aux[0], down = l->aux[1];
        for (unsigned b = 0; b < batch; ++b) {
            uint8_t* d = (uint8_t*)buf + b * bytes;
            if (hipMemcpyAsync(d, in[b], bytes, hipMemcpyHostToDevice, up) != hipSuccess) return set_error(PLK_ERR_HIP, "upload of transform %u failed", b);
            PLK_HIP_TRY(hipEventRecord(l->ev_ready[2 * b], up));
            PLK_HIP_TRY(hipStreamWaitEvent(l->stream, l->ev_ready[2 * b], 0));
            PLK_TRY(ntt_dev_impl(field, log_n, inverse, 1, d, d, l->stream));
            PLK_HIP_TRY(hipEventRecord(l->ev_ready[2 * b + 1], l->stream));
            PLK_HIP_TRY(hipStreamWaitEvent(down, l->ev_ready[2 * b + 1], 0));
            if (hipMemcpyAsync(out[b], d, bytes, hipMemcpyDeviceToHost, down) != hipSuccess) return set_error(PLK_ERR_HIP, "download of transform %u failed", b);
        }
    }
    PLK_TRY(lane_join(*l));  // before the registrations and the device buffer go
    return c.finish();
}

// Units that need no exchange (transforms) over the devices of the group: unit b runs on device b mod N, every device's share in
// one call of `local` on its worker thread.  A single unit - the reference's callers are nine Rayon threads with one transform
// each (plonk_util.rs:173-189) - runs on the calling thread, the calls taking the devices in turn.
static int deal_units(unsigned log_n, unsigned batch, const std::function<int(unsigned, unsigned)>& local) {
    const int world = group_size();
    if (world == 1 || log_n < multi_min_log_n()) return local(0, 1);
    if (batch == 1) {
        DeviceScope on(next_round_robin_device());
        return local(0, 1);
    }
    return run_on_devices(world, [&](int d) -> int { return local((unsigned)d, (unsigned)world); });
}

int plk_ntt_batch(int field, unsigned log_n, int inverse, unsigned batch, const uint64_t* const* in, uint64_t* const* out) {
    PLK_API;
    if (field_limbs(field) < 0) return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    if (log_n > 30) return set_error(PLK_ERR_TWO_ADICITY, "log_n %u too large (max 30)", log_n);
    if (batch == 0) return PLK_OK;
    if (!in || !out) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    for (unsigned b = 0; b < batch; ++b)
        if (!in[b] || !out[b]) return set_error(PLK_ERR_INVALID_ARG, "null pointer in batch slot %u", b);
    return deal_units(log_n, batch, [&](unsigned first, unsigned step) -> int {
        if (step == 1) return ntt_batch_local(field, log_n, inverse, batch, in, out);
        std::vector<const uint64_t*> mi;
        std::vector<uint64_t*> mo;
        for (unsigned b = first; b < batch; b += step) {
            mi.push_back(in[b]);
            mo.push_back(out[b]);
        }
        return ntt_batch_local(field, log_n, inverse, (unsigned)mi.size(), mi.data(), mo.data());
    });
}

int plk_ntt(int field, unsigned log_n, int inverse, const uint64_t* in, uint64_t* out) {
    PLK_API;
    return plk_ntt_batch(field, log_n, inverse, 1, &in, &out);
}

int plk_ntt_padded_dev(int field, unsigned log_n, unsigned batch, const void* d_in, size_t in_len, size_t in_stride, void* d_out,
                       void* stream) {
    PLK_API;
    if (batch == 0) return PLK_OK;
    return ntt_padded_dev_impl(field, log_n, batch, d_in, in_len, in_stride, d_out, as_stream(stream));
}

static int ntt_padded_batch_local(int field, unsigned log_n, unsigned batch, const uint64_t* const* in, const size_t* n_in, uint64_t* const* out) {
    if (batch == 0) return PLK_OK;
    const size_t n = (size_t)1 << log_n;
    size_t max_in = 0;
    for (unsigned b = 0; b < batch; ++b)
        if (n_in[b] > max_in) max_in = n_in[b];
    LaneCall c;
    PLK_TRY(c.begin());
    void *din = nullptr, *dout = nullptr;
    const size_t EB = (size_t)field_limbs(field) * 8;  // bytes per element
    PLK_TRY(c.tmp(din, max_in * EB * batch));
    PLK_TRY(c.tmp(dout, n * EB * batch));
    for (unsigned b = 0; b < batch; ++b) {
        uint8_t* slot = (uint8_t*)din + (size_t)b * max_in * EB;
        // in place (out[b] == in[b]): ONE registration of the larger range, made here
        c.pin(in[b], (const void*)out[b] == (const void*)in[b] && n > n_in[b] ? n * EB : n_in[b] * EB);
        PLK_TRY(lane_h2d(*c.l, slot, in[b], n_in[b] * EB));
        // shorter polynomials of the batch: F::ZERO is all-zero limbs in Montgomery form too
        if (n_in[b] < max_in) PLK_HIP_TRY(hipMemsetAsync(slot + n_in[b] * EB, 0, (max_in - n_in[b]) * EB, c.stream()));
    }
    PLK_TRY(ntt_padded_dev_impl(field, log_n, batch, din, max_in, max_in, dout, c.stream()));
    for (unsigned b = 0; b < batch; ++b) {
        c.pin(out[b], n * EB);  // in place: inside the registration made above
        PLK_TRY(c.out(out[b], (uint8_t*)dout + (size_t)b * n * EB, n * EB));
    }
    return c.finish();
}

int plk_ntt_padded_batch(int field, unsigned log_n, unsigned batch, const uint64_t* const* in, const size_t* n_in, uint64_t* const* out) {
    PLK_API;
    if (field_limbs(field) < 0) return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    if (log_n > 30) return set_error(PLK_ERR_TWO_ADICITY, "log_n %u too large (max 30)", log_n);
    if (batch == 0) return PLK_OK;
    if (!in || !n_in || !out) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    const size_t n = (size_t)1 << log_n;
    for (unsigned b = 0; b < batch; ++b) {
        if (n_in[b] > n) return set_error(PLK_ERR_INVALID_ARG, "n_in[%u] = %zu exceeds 2^%u", b, n_in[b], log_n);
        if ((n_in[b] && !in[b]) || !out[b]) return set_error(PLK_ERR_INVALID_ARG, "null pointer in batch slot %u", b);
    }
    return deal_units(log_n, batch, [&](unsigned first, unsigned step) -> int {
        if (step == 1) return ntt_padded_batch_local(field, log_n, batch, in, n_in, out);
        std::vector<const uint64_t*> mi;
        std::vector<size_t> ml;
        std::vector<uint64_t*> mo;
        for (unsigned b = first; b < batch; b += step) {
            mi.push_back(in[b]);
            ml.push_back(n_in[b]);
            mo.push_back(out[b]);
        }
        return ntt_padded_batch_local(field, log_n, (unsigned)mi.size(), mi.data(), ml.data(), mo.data());
    });
}

int plk_ntt_padded(int field, unsigned log_n, const uint64_t* in, size_t n_in, uint64_t* out) {
    PLK_API;
    return plk_ntt_padded_batch(field, log_n, 1, &in, &n_in, &out);
}

// ---- polynomial callers ----
static size_t pow2_ceil_sz(size_t v) {
    size_t s = 1;
    while (s < v) s <<= 1;
    return s;
}

int plk_poly_divide_by_z_h_dev(int field, const void* d_coeffs, size_t len, size_t n, void* d_out, size_t out_cap, size_t* out_len,
                               void* stream) {
    PLK_API;
    return poly_divide_by_z_h_dev_impl(field, d_coeffs, len, n, d_out, out_cap, out_len, as_stream(stream));
}

int plk_poly_divide_by_z_h(int field, const uint64_t* coeffs, size_t len, size_t n, uint64_t* out, size_t out_cap, size_t* out_len) {
    PLK_API;
    if (field_limbs(field) != 4) return set_error(PLK_ERR_INVALID_ARG, "field %d has no NTT entry point", field);
    if (!out_len || (len && !coeffs)) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    const size_t cap = len > pow2_ceil_sz(len) ? len : pow2_ceil_sz(len);
    LaneCall c;
    PLK_TRY(c.begin());
    void *din = nullptr, *dout = nullptr;
    PLK_TRY(c.in(din, coeffs, len * 32));
    PLK_TRY(c.tmp(dout, cap * 32));
    size_t got = 0;
    PLK_TRY(poly_divide_by_z_h_dev_impl(field, din, len, n, dout, cap, &got, c.stream()));  // synchronises the stream once (the degree)
    if (got > out_cap) return set_error(PLK_ERR_INVALID_ARG, "output capacity %zu < result length %zu", out_cap, got);
    if (got && !out) return set_error(PLK_ERR_INVALID_ARG, "null output");
    PLK_TRY(c.out(out, dout, got * 32));
    *out_len = got;
    return c.finish();
}

int plk_poly_mul_dev(int field, const void* d_a, size_t la, const void* d_b, size_t lb, void* d_out, size_t out_cap, size_t* out_len,
                     void* stream) {
    PLK_API;
    return poly_mul_dev_impl(field, d_a, la, d_b, lb, d_out, out_cap, out_len, as_stream(stream));
}

int plk_poly_mul(int field, const uint64_t* a, size_t la, const uint64_t* b, size_t lb, uint64_t* out, size_t out_cap, size_t* out_len) {
    PLK_API;
    if (field_limbs(field) != 4) return set_error(PLK_ERR_INVALID_ARG, "field %d has no NTT entry point", field);
    if (!out_len || (la && !a) || (lb && !b)) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    const size_t cap = pow2_ceil_sz(la + lb);
    LaneCall c;
    PLK_TRY(c.begin());
    void *da = nullptr, *db = nullptr, *dout = nullptr;
    PLK_TRY(c.in(da, a, la * 32));
    PLK_TRY(c.in(db, b, lb * 32));
    PLK_TRY(c.tmp(dout, cap * 32));
    size_t got = 0;
    PLK_TRY(poly_mul_dev_impl(field, da, la, db, lb, dout, cap, &got, c.stream()));
    if (got > out_cap) return set_error(PLK_ERR_INVALID_ARG, "output capacity %zu < result length %zu", out_cap, got);
    if (got && !out) return set_error(PLK_ERR_INVALID_ARG, "null output");
    PLK_TRY(c.out(out, dout, got * 32));
    *out_len = got;
    return c.finish();
}

// ---- the Plonk quotient numerator ----
int plk_plonk_vanishing_points_dev(int field, unsigned log_degree, const void* d_constants_8n, const void* d_wires_8n, const void* d_s_sigma_8n,
                                   const void* d_plonk_z_8n, const uint64_t* k_is, const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma,
                                   const uint64_t* inner_zeta, const uint64_t* inner_a, void* d_out, void* stream) {
    PLK_API;
    return plonk_vanishing_points_dev_impl(field, log_degree, d_constants_8n, d_wires_8n, d_s_sigma_8n, d_plonk_z_8n, k_is, alpha, beta, gamma, inner_zeta,
                                           inner_a, d_out, as_stream(stream));
}
int plk_plonk_vanishing_points(int field, unsigned log_degree, const uint64_t* constants_8n, const uint64_t* wires_8n, const uint64_t* s_sigma_8n,
                               const uint64_t* plonk_z_8n, const uint64_t* k_is, const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma,
                               const uint64_t* inner_zeta, const uint64_t* inner_a, uint64_t* out) {
    PLK_API;
    if (field_limbs(field) != 4) return set_error(PLK_ERR_INVALID_ARG, "field %d is not a circuit scalar field", field);
    if (log_degree + 3 > 30) return set_error(PLK_ERR_TWO_ADICITY, "log_degree %u too large", log_degree);
    if (!constants_8n || !wires_8n || !s_sigma_8n || !plonk_z_8n || !out) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    const size_t row = ((size_t)8 << log_degree) * 32;
    LaneCall c;
    PLK_TRY(c.begin());
    c.pin(constants_8n, 6 * row);
    c.pin(wires_8n, 9 * row);
    c.pin(s_sigma_8n, 6 * row);
    c.pin(plonk_z_8n, row);
    c.pin(out, row);
    void *dc = nullptr, *dw = nullptr, *ds = nullptr, *dz = nullptr, *dout = nullptr;
    PLK_TRY(c.in(dc, constants_8n, 6 * row));
    PLK_TRY(c.in(dw, wires_8n, 9 * row));
    PLK_TRY(c.in(ds, s_sigma_8n, 6 * row));
    PLK_TRY(c.in(dz, plonk_z_8n, row));
    PLK_TRY(c.tmp(dout, row));
    PLK_TRY(plonk_vanishing_points_dev_impl(field, log_degree, dc, dw, ds, dz, k_is, alpha, beta, gamma, inner_zeta, inner_a, dout, c.stream()));
    PLK_TRY(c.out(out, dout, row));
    return c.finish();
}
int plk_plonk_evaluate_all_constraints(int field, size_t count, const uint64_t* constants, const uint64_t* local_wires, const uint64_t* right_wires,
                                       const uint64_t* below_wires, const uint64_t* inner_zeta, const uint64_t* inner_a, uint64_t* out) {
    PLK_API;
    if (field_limbs(field) != 4) return set_error(PLK_ERR_INVALID_ARG, "field %d is not a circuit scalar field", field);
    if (count == 0) return PLK_OK;
    if (!constants || !local_wires || !right_wires || !below_wires || !out) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    LaneCall c;
    PLK_TRY(c.begin());
    void *dc = nullptr, *dl = nullptr, *dr = nullptr, *db = nullptr, *dout = nullptr;
    PLK_TRY(c.in(dc, constants, count * 6 * 32));
    PLK_TRY(c.in(dl, local_wires, count * 9 * 32));
    PLK_TRY(c.in(dr, right_wires, count * 9 * 32));
    PLK_TRY(c.in(db, below_wires, count * 9 * 32));
    PLK_TRY(c.tmp(dout, count * 8 * 32));
    PLK_TRY(plonk_all_constraints_dev_impl(field, count, dc, dl, dr, db, inner_zeta, inner_a, dout, c.stream()));
    PLK_TRY(c.out(out, dout, count * 8 * 32));
    return c.finish();
}

// ---- MSM ----
// Over a device group (plk_init_devices) a tabled precomputation of at least 2^PLK_MULTI_MIN_LOG_N generators is built on every
// device (multi.hip); smaller ones, and table-free contexts (one-shot MSMs: latency, not throughput), stay on the caller's device.
static bool fan_out_msm(size_t n, unsigned flags) {
    return group_size() > 1 && !(flags & PLK_MSM_TABLE_FREE) && n >= ((size_t)1 << multi_min_log_n()) && n >= (size_t)group_size();
}
int plk_msm_precompute_dev_ex(int curve, size_t n, const void* d_bases_xy, const void* d_base_zero, unsigned window_bits, unsigned flags, void* stream,
                              plk_msm_ctx** out_ctx) {
    PLK_API;
    if (fan_out_msm(n, flags) && d_bases_xy) return msm_precompute_multi(curve, n, d_bases_xy, d_base_zero, false, window_bits, as_stream(stream), out_ctx);
    return msm_precompute_dev_impl(curve, n, d_bases_xy, d_base_zero, window_bits, flags, as_stream(stream), out_ctx);
}
int plk_msm_precompute_dev(int curve, size_t n, const void* d_bases_xy, const void* d_base_zero, unsigned window_bits, void* stream,
                           plk_msm_ctx** out_ctx) {
    PLK_API;
    return plk_msm_precompute_dev_ex(curve, n, d_bases_xy, d_base_zero, window_bits, 0, stream, out_ctx);
}

int plk_msm_precompute_ex(int curve, size_t n, const uint64_t* bases_xy, const uint8_t* base_zero, unsigned window_bits, unsigned flags,
                          plk_msm_ctx** out_ctx) {
    PLK_API;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (n && !bases_xy) return set_error(PLK_ERR_INVALID_ARG, "null bases");
    if (fan_out_msm(n, flags)) {
        DeviceScope primary(0);
        return msm_precompute_multi(curve, n, bases_xy, base_zero, true, window_bits, nullptr, out_ctx);
    }
    HostLane* l = nullptr;
    PLK_TRY(lane_get(l));
    LaneBuf db, dz;
    PLK_TRY(db.alloc(n * 2 * L * 8, l->stream));
    PLK_TRY(lane_h2d(*l, db.p, bases_xy, n * 2 * L * 8));
    if (base_zero) {
        PLK_TRY(dz.alloc(n, l->stream));
        PLK_TRY(lane_h2d(*l, dz.p, base_zero, n));
    }
    const int rc = msm_precompute_dev_impl(curve, n, db.p, base_zero ? dz.p : nullptr, window_bits, flags, l->stream, out_ctx);  // synchronises the stream on success
    if (rc != PLK_OK) (void)hipStreamSynchronize(l->stream);  // an early error return leaves the staged copies in flight
    l->pin_used = 0;
    return rc;
}
int plk_msm_precompute(int curve, size_t n, const uint64_t* bases_xy, const uint8_t* base_zero, unsigned window_bits, plk_msm_ctx** out_ctx) {
    PLK_API;
    return plk_msm_precompute_ex(curve, n, bases_xy, base_zero, window_bits, 0, out_ctx);
}

int plk_msm_free(plk_msm_ctx* ctx) {
    PLK_API;
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    msm_ctx_delete(ctx);
    return PLK_OK;
}
size_t plk_msm_ctx_len(const plk_msm_ctx* ctx) { return ctx ? msm_ctx_len(ctx) : 0; }
unsigned plk_msm_ctx_window(const plk_msm_ctx* ctx) { return ctx ? msm_ctx_window(ctx) : 0; }

int plk_msm_execute_dev(plk_msm_ctx* ctx, unsigned batch, const void* d_scalars, size_t n_scalars, void* d_out_xy, void* d_out_zero, void* stream) {
    PLK_API;
    if (msm_ctx_is_multi(ctx) && batch && d_scalars && n_scalars == msm_ctx_len(ctx)) {
        // a context that lives on every device of the group: the vectors (memory of the caller's device) reach the other devices
        // peer to peer, the results come back to the caller's device; asynchronous on `stream` like the one-device form
        std::vector<const void*> vecs(batch);
        for (unsigned b = 0; b < batch; ++b) vecs[b] = (const uint8_t*)d_scalars + (size_t)b * n_scalars * 32;
        return msm_execute_multi(ctx, batch, vecs.data(), false, n_scalars, d_out_xy, d_out_zero, as_stream(stream));
    }
    return msm_execute_dev_impl(ctx, batch, d_scalars, n_scalars, d_out_xy, d_out_zero, as_stream(stream));
}

// msm_execute_parallel's own return type (curve_msm.rs:102-157: a ProjectivePoint, not normalised): x | y | z per vector.  One-device
// bucket contexts end their reduction with six products instead of an inversion; combs and device-group contexts, which normalise on the
// way anyway, return their affine points with z = 1.
int plk_msm_execute_projective_dev(plk_msm_ctx* ctx, unsigned batch, const void* d_scalars, size_t n_scalars, void* d_out_xyz, void* d_out_zero, void* stream) {
    PLK_API;
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    if (!msm_ctx_is_multi(ctx) && !msm_ctx_is_comb(ctx))
        return msm_execute_dev_impl(ctx, batch, d_scalars, n_scalars, d_out_xyz, d_out_zero, as_stream(stream), nullptr, nullptr, 1u);
    if (batch == 0) return PLK_OK;
    if (!d_out_xyz || !d_out_zero) return set_error(PLK_ERR_INVALID_ARG, "null device pointer");
    const size_t L = (size_t)curve_limbs(msm_ctx_curve(ctx));
    hipStream_t st = as_stream(stream);
    void* tmp = scratch_acquire((size_t)batch * 2 * L * 8, st);
    if (!tmp) return PLK_ERR_OOM;
    int rc = plk_msm_execute_dev(ctx, batch, d_scalars, n_scalars, tmp, d_out_zero, stream);
    if (rc == PLK_OK) rc = msm_affine_to_projective_impl(msm_ctx_curve(ctx), batch, tmp, d_out_zero, d_out_xyz, st);
    scratch_release(tmp, st);
    return rc;
}
int plk_msm_execute_projective(plk_msm_ctx* ctx, const uint64_t* scalars, size_t n_scalars, uint64_t* out_xyz, uint8_t* out_zero) {
    PLK_API;
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    if (n_scalars != msm_ctx_len(ctx))
        return set_error(PLK_ERR_SIZE_MISMATCH, "scalars.len() = %zu but the precomputation holds %zu generators (curve_msm.rs:67)", n_scalars, msm_ctx_len(ctx));
    if ((n_scalars && !scalars) || !out_xyz || !out_zero) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    const size_t L = (size_t)curve_limbs(msm_ctx_curve(ctx));
    DeviceScope primary(msm_ctx_is_multi(ctx) ? 0 : thread_logical_device());
    LaneCall c;
    PLK_TRY(c.begin());
    void *ds = nullptr, *dxyz = nullptr, *dz = nullptr;
    c.pin(scalars, n_scalars * 32);
    PLK_TRY(c.in(ds, scalars, n_scalars * 32));
    PLK_TRY(c.tmp(dxyz, 3 * L * 8));
    PLK_TRY(c.tmp(dz, 1));
    PLK_TRY(plk_msm_execute_projective_dev(ctx, 1, ds, n_scalars, dxyz, dz, (void*)c.stream()));
    PLK_TRY(c.out(out_xyz, dxyz, 3 * L * 8));
    PLK_TRY(c.out(out_zero, dz, 1));
    return c.finish();
}

int plk_msm_execute_parts_dev(plk_msm_ctx* ctx, unsigned batch, const uint64_t* first, const uint64_t* count, const void* const* d_scalars, void* d_out_xy,
                              void* d_out_zero, void* stream) {
    PLK_API;
    MsmParts parts{first, count, d_scalars};
    return msm_execute_dev_impl(ctx, batch, nullptr, 0, d_out_xy, d_out_zero, as_stream(stream), nullptr, &parts);
}

int plk_msm_execute_parts_buckets_dev(plk_msm_ctx* ctx, unsigned batch, const uint64_t* first, const uint64_t* count, const void* const* d_scalars,
                                      const uint32_t* bucket_part, const uint32_t* bucket_parts, void* d_out_xy, void* d_out_zero, void* stream) {
    PLK_API;
    if (!bucket_part || !bucket_parts) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    MsmParts parts{first, count, d_scalars, bucket_part, bucket_parts};
    return msm_execute_dev_impl(ctx, batch, nullptr, 0, d_out_xy, d_out_zero, as_stream(stream), nullptr, &parts);
}

int plk_msm_execute_batch(plk_msm_ctx* ctx, unsigned batch, const uint64_t* const* scalars, size_t n_scalars, uint64_t* out_xy, uint8_t* out_zero) {
    PLK_API;
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    if (n_scalars != msm_ctx_len(ctx))
        return set_error(PLK_ERR_SIZE_MISMATCH, "scalars.len() = %zu but the precomputation holds %zu generators (curve_msm.rs:67)", n_scalars,
                         msm_ctx_len(ctx));
    if (batch == 0) return PLK_OK;
    if (!scalars || !out_xy || !out_zero) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    for (unsigned b = 0; b < batch; ++b)
        if (n_scalars && !scalars[b]) return set_error(PLK_ERR_INVALID_ARG, "null scalars in batch slot %u", b);
    const bool multi = msm_ctx_is_multi(ctx);
    DeviceScope primary(multi ? 0 : thread_logical_device());  // a group context's results are combined on its first device
    const size_t L = (size_t)curve_limbs(msm_ctx_curve(ctx));
    const size_t sb = n_scalars * 32;
    LaneCall c;
    PLK_TRY(c.begin());
    HostLane* l = c.l;
    void *dxy = nullptr, *dz = nullptr;
    PLK_TRY(c.tmp(dxy, (size_t)batch * 2 * L * 8));
    PLK_TRY(c.tmp(dz, batch));
    if (multi) {
        // every device uploads its own share over its own PCIe link (multi.hip); the vectors stay registered until this call's
        // final synchronisation, which - through the events the caller's stream waits for - covers the workers' copies
        for (unsigned b = 0; b < batch; ++b) c.pin(scalars[b], sb);
        PLK_TRY(msm_execute_multi(ctx, batch, (const void* const*)scalars, true, n_scalars, dxy, dz, l->stream));
    } else if (batch == 1 || sb < ((size_t)1 << 20)) {
        void* ds = nullptr;
        PLK_TRY(c.tmp(ds, sb * batch));
        for (unsigned b = 0; b < batch; ++b) PLK_TRY(lane_h2d(*l, (uint8_t*)ds + b * sb, scalars[b], sb));
        PLK_TRY(msm_execute_dev_impl(ctx, batch, ds, n_scalars, dxy, dz, l->stream));
    } else {
        // the scalar vectors of a batch (commit_polynomials, plonk_util.rs:215-231: nine 32 MiB vectors at 2^20) cross PCIe on a
        // second stream, vector b + 1 while vector b is being ordered and accumulated: one event per vector
        void* ds = nullptr;
        PLK_TRY(c.tmp(ds, sb * batch));
        for (unsigned b = 0; b < batch; ++b) c.pin(scalars[b], sb);
        PLK_TRY(lane_fork(*l));
        while (l->ev_ready.size() < batch) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return set_error(PLK_ERR_HIP, "hipEventCreate failed");
            l->ev_ready.push_back(e);
        }
        for (unsigned b = 0; b < batch; ++b) {
            if (hipMemcpyAsync((uint8_t*)ds + b * sb, scalars[b], sb, hipMemcpyHostToDevice, l->aux[0]) != hipSuccess ||
                hipEventRecord(l->ev_ready[b], l->aux[0]) != hipSuccess)
                return set_error(PLK_ERR_HIP, "upload of scalar vector %u failed", b);
        }
        PLK_TRY(msm_execute_dev_impl(ctx, batch, ds, n_scalars, dxy, dz, l->stream, l->ev_ready.data()));
    }
    PLK_TRY(c.out(out_xy, dxy, (size_t)batch * 2 * L * 8));
    PLK_TRY(c.out(out_zero, dz, batch));
    return c.finish();
}

int plk_msm_execute(plk_msm_ctx* ctx, const uint64_t* scalars, size_t n_scalars, uint64_t* out_xy, uint8_t* out_zero) {
    PLK_API;
    return plk_msm_execute_batch(ctx, 1, &scalars, n_scalars, out_xy, out_zero);
}

int plk_msm(int curve, size_t n, const uint64_t* bases_xy, const uint8_t* base_zero, const uint64_t* scalars, uint64_t* out_xy, uint8_t* out_zero) {
    PLK_API;
    plk_msm_ctx* ctx = nullptr;
    // generators used once: no window tables (their construction costs ~30 executions)
    PLK_TRY(plk_msm_precompute_ex(curve, n, bases_xy, base_zero, 0, PLK_MSM_TABLE_FREE, &ctx));
    int rc = plk_msm_execute(ctx, scalars, n, out_xy, out_zero);
    msm_ctx_delete(ctx);
    return rc;
}

int plk_curve_sum_affine(int curve, size_t k, const uint64_t* pts_xy, const uint8_t* pts_zero, uint64_t* out_xy, uint8_t* out_zero) {
    PLK_API;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if ((k && !pts_xy) || !out_xy || !out_zero) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    LaneCall c;
    PLK_TRY(c.begin());
    void *dp = nullptr, *dz = nullptr, *dxy = nullptr, *doz = nullptr;
    PLK_TRY(c.in(dp, pts_xy, k * 2 * L * 8));
    if (pts_zero) PLK_TRY(c.in(dz, pts_zero, k));
    PLK_TRY(c.tmp(dxy, 2 * L * 8));
    PLK_TRY(c.tmp(doz, 1));
    PLK_TRY(curve_sum_affine_dev_impl(curve, k, dp, pts_zero ? dz : nullptr, dxy, doz, c.stream()));
    PLK_TRY(c.out(out_xy, dxy, 2 * L * 8));
    PLK_TRY(c.out(out_zero, doz, 1));
    return c.finish();
}

// ---- multi-GPU exchange ----
size_t plk_msm_partials_bytes(int curve, unsigned slots) { return msm_partials_bytes(curve, slots); }
int plk_msm_combine_partials_dev(int curve, unsigned world, unsigned batch, unsigned whole_per_rank, const void* d_gathered, void* d_out_xy,
                                 void* d_out_zero, void* stream) {
    PLK_API;
    return msm_combine_partials_dev_impl(curve, world, batch, whole_per_rank, d_gathered, d_out_xy, d_out_zero, as_stream(stream));
}

// ---- the reference's own MsmPrecomputation contents ----
int plk_msm_table_digits(int curve, unsigned w) {
    if (curve_limbs(curve) < 0 || w == 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d or window 0", curve);
    return msm_table_digits(curve, w);
}
int plk_msm_precompute_table_dev(int curve, size_t n, const void* d_bases_xy, const void* d_base_zero, unsigned w, void* d_out_xy, void* d_out_zero,
                                 void* stream) {
    PLK_API;
    return msm_reference_table_dev_impl(curve, n, d_bases_xy, d_base_zero, w, d_out_xy, d_out_zero, as_stream(stream));
}
int plk_msm_precompute_table(int curve, size_t n, const uint64_t* bases_xy, const uint8_t* base_zero, unsigned w, uint64_t* out_xy, uint8_t* out_zero) {
    PLK_API;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (w < 1 || w > 64) return set_error(PLK_ERR_INVALID_ARG, "window size %u outside [1, 64]", w);
    if (n && (!bases_xy || !out_xy || !out_zero)) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    if (n == 0) return PLK_OK;
    const size_t digits = (size_t)msm_table_digits(curve, w);
    LaneCall c;
    PLK_TRY(c.begin());
    c.pin(out_xy, n * digits * 2 * L * 8);
    void *db = nullptr, *dz = nullptr, *dout = nullptr, *doz = nullptr;
    PLK_TRY(c.in(db, bases_xy, n * 2 * L * 8));
    if (base_zero) PLK_TRY(c.in(dz, base_zero, n));
    PLK_TRY(c.tmp(dout, n * digits * 2 * L * 8));
    PLK_TRY(c.tmp(doz, n * digits));
    PLK_TRY(msm_reference_table_dev_impl(curve, n, db, base_zero ? dz : nullptr, w, dout, doz, c.stream()));
    PLK_TRY(c.out(out_xy, dout, n * digits * 2 * L * 8));
    PLK_TRY(c.out(out_zero, doz, n * digits));
    return c.finish();
}

// ---- IPA generator fold ----
int plk_curve_fold_pairs_dev(int curve, size_t m, const void* d_lo_xy, const void* d_lo_zero, const void* d_hi_xy, const void* d_hi_zero,
                             const uint64_t* scalar_lo, const uint64_t* scalar_hi, void* d_out_xy, void* d_out_zero, void* stream) {
    PLK_API;
    return curve_fold_pairs_dev_impl(curve, m, d_lo_xy, d_lo_zero, d_hi_xy, d_hi_zero, scalar_lo, scalar_hi, d_out_xy, d_out_zero, as_stream(stream));
}

int plk_curve_fold_pairs(int curve, size_t m, const uint64_t* lo_xy, const uint8_t* lo_zero, const uint64_t* hi_xy, const uint8_t* hi_zero,
                         const uint64_t* scalar_lo, const uint64_t* scalar_hi, uint64_t* out_xy, uint8_t* out_zero) {
    PLK_API;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (m && (!lo_xy || !hi_xy || !out_xy || !out_zero)) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    if (!scalar_lo || !scalar_hi) return set_error(PLK_ERR_INVALID_ARG, "null scalar");
    if (m == 0) return PLK_OK;
    const size_t pb = m * 2 * L * 8;
    LaneCall c;
    PLK_TRY(c.begin());
    void *dlo = nullptr, *dhi = nullptr, *dlz = nullptr, *dhz = nullptr, *dout = nullptr, *doz = nullptr;
    PLK_TRY(c.in(dlo, lo_xy, pb));
    PLK_TRY(c.in(dhi, hi_xy, pb));
    if (lo_zero) PLK_TRY(c.in(dlz, lo_zero, m));
    if (hi_zero) PLK_TRY(c.in(dhz, hi_zero, m));
    PLK_TRY(c.tmp(dout, pb));
    PLK_TRY(c.tmp(doz, m));
    PLK_TRY(curve_fold_pairs_dev_impl(curve, m, dlo, lo_zero ? dlz : nullptr, dhi, hi_zero ? dhz : nullptr, scalar_lo, scalar_hi, dout, doz, c.stream()));
    PLK_TRY(c.out(out_xy, dout, pb));
    PLK_TRY(c.out(out_zero, doz, m));
    return c.finish();
}

// ---- batch inversion ----
int plk_field_batch_inverse_dev(int field, const void* d_x, void* d_out, void* d_is_none, size_t count, void* stream) {
    PLK_API;
    if (field_limbs(field) < 0) return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    return field_batch_inverse_dev_impl(field, d_x, d_out, d_is_none, nullptr, count, as_stream(stream));
}
static int batch_inverse_host(int field, const uint64_t* x, uint64_t* out, uint8_t* is_none, size_t count, bool strict) {
    const int L = field_limbs(field);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    if (count == 0) return PLK_OK;
    if (!x || !out) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    LaneCall c;
    PLK_TRY(c.begin());
    void *dx = nullptr, *dz = nullptr, *dc = nullptr;
    PLK_TRY(c.in(dx, x, count * L * 8));
    PLK_TRY(c.tmp(dz, count));
    PLK_TRY(c.tmp(dc, 4));
    PLK_HIP_TRY(hipMemsetAsync(dc, 0, 4, c.stream()));
    PLK_TRY(field_batch_inverse_dev_impl(field, dx, dx, dz, (unsigned*)dc, count, c.stream()));
    unsigned zeros = 0;
    PLK_TRY(c.out(&zeros, dc, 4));
    PLK_TRY(c.sync());
    if (strict && zeros) {
        c.done = true;
        return set_error(PLK_ERR_INVALID_ARG, "No inverse: %u of the %zu elements are zero (field.rs:266)", zeros, count);
    }
    PLK_TRY(c.out(out, dx, count * L * 8));
    if (is_none) PLK_TRY(c.out(is_none, dz, count));
    return c.finish();
}
int plk_field_batch_inverse(int field, const uint64_t* x, uint64_t* out, size_t count) { PLK_API; return batch_inverse_host(field, x, out, nullptr, count, true); }
int plk_field_batch_inverse_opt(int field, const uint64_t* x, uint64_t* out, uint8_t* is_none, size_t count) {
    PLK_API;
    if (count && !is_none) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    return batch_inverse_host(field, x, out, is_none, count, false);
}
int plk_curve_batch_to_affine_dev(int curve, size_t count, const void* d_proj_xyz, const void* d_proj_zero, void* d_out_xy, void* d_out_zero, void* stream) {
    PLK_API;
    return curve_batch_to_affine_dev_impl(curve, count, d_proj_xyz, d_proj_zero, d_out_xy, d_out_zero, as_stream(stream));
}
int plk_curve_batch_to_affine(int curve, size_t count, const uint64_t* proj_xyz, const uint8_t* proj_zero, uint64_t* out_xy, uint8_t* out_zero) {
    PLK_API;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (count == 0) return PLK_OK;
    if (!proj_xyz || !out_xy || !out_zero) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    LaneCall c;
    PLK_TRY(c.begin());
    void *dp = nullptr, *dz = nullptr, *dxy = nullptr, *doz = nullptr;
    PLK_TRY(c.in(dp, proj_xyz, count * 3 * L * 8));
    if (proj_zero) PLK_TRY(c.in(dz, proj_zero, count));
    PLK_TRY(c.tmp(dxy, count * 2 * L * 8));
    PLK_TRY(c.tmp(doz, count));
    PLK_TRY(curve_batch_to_affine_dev_impl(curve, count, dp, proj_zero ? dz : nullptr, dxy, doz, c.stream()));
    PLK_TRY(c.out(out_xy, dxy, count * 2 * L * 8));
    PLK_TRY(c.out(out_zero, doz, count));
    return c.finish();
}

// ---- canonical byte encodings ----
int plk_field_to_bytes(int field, const uint64_t* x, size_t count, uint8_t* out_bytes) {
    PLK_API;
    const int L = field_limbs(field);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    if (count == 0) return PLK_OK;
    if (!x || !out_bytes) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    LaneCall c;
    PLK_TRY(c.begin());
    void *dx = nullptr, *db = nullptr;
    PLK_TRY(c.in(dx, x, count * L * 8));
    PLK_TRY(c.tmp(db, count * L * 8));
    PLK_TRY(field_bytes_impl(field, 0, dx, count, db, nullptr, c.stream()));
    PLK_TRY(c.out(out_bytes, db, count * L * 8));
    return c.finish();
}
int plk_field_from_bytes(int field, const uint8_t* bytes, size_t count, uint64_t* out) {
    PLK_API;
    const int L = field_limbs(field);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad field id %d", field);
    if (count == 0) return PLK_OK;
    if (!bytes || !out) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    LaneCall c;
    PLK_TRY(c.begin());
    void *dx = nullptr, *db = nullptr, *dc = nullptr;
    PLK_TRY(c.in(db, bytes, count * L * 8));
    PLK_TRY(c.tmp(dx, count * L * 8));
    PLK_TRY(c.tmp(dc, 4));
    PLK_HIP_TRY(hipMemsetAsync(dc, 0, 4, c.stream()));
    PLK_TRY(field_bytes_impl(field, 1, db, count, dx, (unsigned*)dc, c.stream()));
    unsigned bad = 0;
    PLK_TRY(c.out(&bad, dc, 4));
    PLK_TRY(c.out(out, dx, count * L * 8));
    PLK_TRY(c.finish());
    if (bad) return set_error(PLK_ERR_INVALID_ARG, "Out of range: %u of the %zu records are not below the modulus (field.rs:100)", bad, count);
    return PLK_OK;
}
int plk_curve_point_to_bytes(int curve, const uint64_t* xy, const uint8_t* zero, size_t count, uint8_t* out_bytes) {
    PLK_API;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (count == 0) return PLK_OK;
    if (!xy || !out_bytes) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    const size_t rec = 1 + (size_t)L * 8;
    LaneCall c;
    PLK_TRY(c.begin());
    void *dp = nullptr, *dz = nullptr, *db = nullptr;
    PLK_TRY(c.in(dp, xy, count * 2 * L * 8));
    if (zero) PLK_TRY(c.in(dz, zero, count));
    PLK_TRY(c.tmp(db, count * rec));
    PLK_TRY(point_bytes_impl(curve, 0, dp, zero ? dz : nullptr, count, db, nullptr, nullptr, c.stream()));
    PLK_TRY(c.out(out_bytes, db, count * rec));
    return c.finish();
}
int plk_curve_point_from_bytes(int curve, const uint8_t* bytes, size_t count, uint64_t* out_xy, uint8_t* out_zero, uint8_t* status) {
    PLK_API;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (count == 0) return PLK_OK;
    if (!bytes || !out_xy || !out_zero) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    const size_t rec = 1 + (size_t)L * 8;
    LaneCall c;
    PLK_TRY(c.begin());
    void *dp = nullptr, *dz = nullptr, *ds = nullptr, *db = nullptr;
    PLK_TRY(c.in(db, bytes, count * rec));
    PLK_TRY(c.tmp(dp, count * 2 * L * 8));
    PLK_TRY(c.tmp(dz, count));
    PLK_TRY(c.tmp(ds, count));
    PLK_TRY(point_bytes_impl(curve, 1, db, nullptr, count, dp, dz, ds, c.stream()));
    std::vector<uint8_t> st(count);
    PLK_TRY(c.out(st.data(), ds, count));
    PLK_TRY(c.out(out_xy, dp, count * 2 * L * 8));
    PLK_TRY(c.out(out_zero, dz, count));
    PLK_TRY(c.finish());
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < count; ++i) {
        if (status) status[i] = st[i];
        if (st[i] && !bad++) first = i;
    }
    if (bad)
        return set_error(PLK_ERR_INVALID_ARG, "%zu of the %zu points do not decode (first: record %zu, %s)", bad, count, first,
                         st[first] == 1 ? "Out of range" : "Invalid x coordinate");
    return PLK_OK;
}

// ---- scalar side of an IPA round ----
int plk_field_inner_product_dev(int field, const void* d_a, const void* d_b, size_t count, void* d_out, void* stream) {
    PLK_API;
    return field_inner_product_dev_impl(field, d_a, d_b, count, d_out, as_stream(stream));
}
int plk_field_fold_slices_dev(int field, const void* d_lo, const void* d_hi, const uint64_t* scalar_lo, const uint64_t* scalar_hi, size_t count,
                              void* d_out, void* stream) {
    PLK_API;
    return field_fold_slices_dev_impl(field, d_lo, d_hi, scalar_lo, scalar_hi, count, d_out, as_stream(stream));
}

// ---- one inner-product argument ----
int plk_halo_begin_dev(int curve, size_t n, const void* d_halo_a, const void* d_halo_b, const void* d_halo_g_xy, const void* d_halo_g_zero,
                       const uint64_t* pedersen_h_xy, const uint64_t* u_prime_xy, unsigned freeze_log, void* stream, plk_halo_ctx** out_ctx) {
    PLK_API;
    return halo_begin_dev_impl(curve, n, d_halo_a, d_halo_b, d_halo_g_xy, d_halo_g_zero, pedersen_h_xy, u_prime_xy, freeze_log, as_stream(stream), out_ctx);
}
int plk_halo_begin_tabled_dev(int curve, size_t n, const void* d_halo_a, const void* d_halo_b, const void* d_halo_g_xy, const void* d_halo_g_zero,
                              plk_msm_ctx* pedersen_g_tables, const uint64_t* pedersen_h_xy, const uint64_t* u_prime_xy, size_t h_index, size_t u_index,
                              const uint64_t* u_prime_scalar, unsigned freeze_log, unsigned lead_rounds, void* stream, plk_halo_ctx** out_ctx) {
    PLK_API;
    if (!pedersen_g_tables) return set_error(PLK_ERR_INVALID_ARG, "null tables");
    // the argument runs on the calling thread's device, on `stream`; the tables must live there too.  A device-group context
    // (plk_init_devices) is kept on logical device 0 whatever thread built it: open over it from a thread on that device
    PLK_TRY(ensure_device());
    int cur = -1;
    PLK_HIP_TRY(hipGetDevice(&cur));
    if (cur != msm_ctx_device(pedersen_g_tables))
        return set_error(PLK_ERR_INVALID_ARG, "the commitment tables live on HIP device %d but the calling thread works on device %d%s", msm_ctx_device(pedersen_g_tables),
                         cur, msm_ctx_is_multi(pedersen_g_tables) ? " (a device-group context is kept on logical device 0: plk_set_thread_device(0))" : "");
    return halo_begin_dev_impl(curve, n, d_halo_a, d_halo_b, d_halo_g_xy, d_halo_g_zero, pedersen_h_xy, u_prime_xy, freeze_log, as_stream(stream), out_ctx,
                               pedersen_g_tables, lead_rounds, h_index, u_index, u_prime_scalar);
}
int plk_curve_fold_multi_dev(int curve, size_t n_out, unsigned log_inputs, const void* d_g_xy, const void* d_g_zero, const void* d_scalars, void* d_out_xy,
                             void* d_out_zero, void* stream) {
    PLK_API;
    return curve_fold_multi_dev_impl(curve, n_out, (int)log_inputs, d_g_xy, d_g_zero, d_scalars, d_out_xy, d_out_zero, as_stream(stream));
}
int plk_halo_begin(int curve, size_t n, const uint64_t* halo_a, const uint64_t* halo_b, const uint64_t* halo_g_xy, const uint8_t* halo_g_zero,
                   const uint64_t* pedersen_h_xy, const uint64_t* u_prime_xy, unsigned freeze_log, plk_halo_ctx** out_ctx) {
    PLK_API;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (!halo_a || !halo_b || !halo_g_xy) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    HostLane* l = nullptr;
    PLK_TRY(lane_get(l));
    LaneBuf da, db, dg, dz;
    PLK_TRY(da.alloc(n * 32, l->stream));
    PLK_TRY(db.alloc(n * 32, l->stream));
    PLK_TRY(dg.alloc(n * 2 * L * 8, l->stream));
    PLK_TRY(lane_h2d(*l, da.p, halo_a, n * 32));
    PLK_TRY(lane_h2d(*l, db.p, halo_b, n * 32));
    PLK_TRY(lane_h2d(*l, dg.p, halo_g_xy, n * 2 * L * 8));
    if (halo_g_zero) {
        PLK_TRY(dz.alloc(n, l->stream));
        PLK_TRY(lane_h2d(*l, dz.p, halo_g_zero, n));
    }
    // the context works on the lane's stream from here on (one host thread per context)
    const int rc = halo_begin_dev_impl(curve, n, da.p, db.p, dg.p, halo_g_zero ? dz.p : nullptr, pedersen_h_xy, u_prime_xy, freeze_log, l->stream, out_ctx);
    (void)hipStreamSynchronize(l->stream);
    l->pin_used = 0;
    return rc;
}
int plk_halo_round_lr(plk_halo_ctx* ctx, const uint64_t* l_blinding, const uint64_t* r_blinding, uint64_t* lr_xy, uint8_t* lr_zero) {
    PLK_API;
    return halo_round_lr_impl(ctx, l_blinding, r_blinding, lr_xy, lr_zero);
}
int plk_halo_round_fold(plk_halo_ctx* ctx, const uint64_t* u_j, const uint64_t* u_j_inv) { PLK_API; return halo_round_fold_impl(ctx, u_j, u_j_inv); }
size_t plk_halo_len(const plk_halo_ctx* ctx) { return halo_len_impl(ctx); }
int plk_halo_frozen(const plk_halo_ctx* ctx) { return halo_frozen_impl(ctx); }
int plk_halo_read(plk_halo_ctx* ctx, uint64_t* halo_a, uint64_t* halo_b, uint64_t* halo_g_xy, uint8_t* halo_g_zero) {
    PLK_API;
    return halo_read_impl(ctx, halo_a, halo_b, halo_g_xy, halo_g_zero);
}
int plk_halo_free(plk_halo_ctx* ctx) {
    PLK_API;
    if (!ctx) return set_error(PLK_ERR_INVALID_ARG, "null context");
    halo_delete(ctx);
    return PLK_OK;
}

// ---- self-test ----
int plk_selftest_quad(int curve, const uint64_t* pts_xy, size_t n, unsigned quads, unsigned* mismatches) {
    PLK_API;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (!pts_xy || !mismatches || n == 0 || n > 0xffffffffu) return set_error(PLK_ERR_INVALID_ARG, "bad argument");
    PLK_TRY(ensure_device());
    DevBuf dp;
    PLK_TRY(dp.alloc(n * 2 * L * 8));
    PLK_HIP_TRY(hipMemcpy(dp.p, pts_xy, n * 2 * L * 8, hipMemcpyHostToDevice));
    return selftest_quad_dev_impl(curve, dp.p, (uint32_t)n, quads, mismatches);
}

// ---- checked build ----
int plk_checked_build(void) { return checked_build_impl(); }
int plk_checked_failures(unsigned* counts) { PLK_API; return checked_failures_impl(counts); }

// ---- measurement hooks ----
int plk_ntt_set_profiling(int enable) { return ntt_set_profiling_impl(enable); }
int plk_ntt_get_timings(double* sum_ms, unsigned* launches) { PLK_API; return ntt_get_timings_impl(sum_ms, launches); }
// A device-group context runs its executions on its per-device parts (full tables on every device, a share context per device):
// the handle's own stage events would stay empty - refused instead of returning stale or empty numbers.
static int refuse_group_profiling(plk_msm_ctx* ctx) {
    if (msm_ctx_is_multi(ctx))
        return set_error(PLK_ERR_INVALID_ARG, "per-stage timings are collected on one-device contexts only: a device-group context executes on its per-device parts");
    return PLK_OK;
}
int plk_msm_set_profiling(plk_msm_ctx* ctx, int enable) {
    if (enable) PLK_TRY(refuse_group_profiling(ctx));
    return msm_set_profiling_impl(ctx, enable);
}
int plk_msm_get_timings(plk_msm_ctx* ctx, double* sum_ms, unsigned* calls) {
    PLK_API;
    PLK_TRY(refuse_group_profiling(ctx));
    return msm_get_timings_impl(ctx, sum_ms, calls);
}

// ---- utilities ----
int plk_msm_debug_digits(int curve, unsigned window_bits, size_t n, const uint64_t* scalars, int32_t* digits, unsigned* n_digits) {
    PLK_API;
    const int nd = msm_debug_digit_count(curve, window_bits);
    if (nd < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d or window of %u bits", curve, window_bits);
    if (n_digits) *n_digits = (unsigned)nd;
    if (!digits || !n) return PLK_OK;
    if (!scalars) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    PLK_TRY(ensure_device());
    DevBuf ds, dd;
    PLK_TRY(ds.alloc(n * 32));
    PLK_TRY(dd.alloc(n * (size_t)nd * 4));
    PLK_HIP_TRY(hipMemcpy(ds.p, scalars, n * 32, hipMemcpyHostToDevice));
    PLK_TRY(msm_debug_digits_impl(curve, window_bits, n, ds.p, dd.p, nullptr));
    PLK_HIP_TRY(hipDeviceSynchronize());
    PLK_HIP_TRY(hipMemcpy(digits, dd.p, n * (size_t)nd * 4, hipMemcpyDeviceToHost));
    return PLK_OK;
}

int plk_bench_ceilings(double* out, unsigned n_out) { PLK_API; return bench_ceilings_impl(out, n_out); }

int plk_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count) {
    PLK_API;
    return field_op_impl(field, op, a, b, out, count);
}

int plk_curve_gen_bases_dev(int curve, size_t n, uint64_t first, const uint64_t* g0_xy, const uint64_t* d_xy, void* d_out_xy, void* stream) {
    PLK_API;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    if (!g0_xy || !d_xy || (n && !d_out_xy)) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    PLK_TRY(ensure_device());
    DevBuf dg;
    PLK_TRY(dg.alloc(4 * L * 8));
    PLK_HIP_TRY(hipMemcpy(dg.p, g0_xy, 2 * L * 8, hipMemcpyHostToDevice));
    PLK_HIP_TRY(hipMemcpy((uint8_t*)dg.p + 2 * L * 8, d_xy, 2 * L * 8, hipMemcpyHostToDevice));
    PLK_TRY(curve_gen_bases_dev_impl(curve, n, first, dg.p, d_out_xy, as_stream(stream)));
    PLK_HIP_TRY(hipStreamSynchronize(as_stream(stream)));  // dg is freed on return
    return PLK_OK;
}

}  // extern "C"
