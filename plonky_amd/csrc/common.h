// common.h -- host-side plumbing shared by the translation units of libplonky_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <functional>
#include <memory>
#include <string>

#include "../../include/plonky_hip.h"

namespace plk {

// thread-local error text behind plk_last_error()
std::string& last_error_ref();
int set_error(int code, const char* fmt, ...);

#define PLK_HIP_TRY(expr)                                                                              \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return ::plk::set_error(_e == hipErrorOutOfMemory ? PLK_ERR_OOM : PLK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                                    hipGetErrorString(_e), __FILE__, __LINE__);                        \
    } while (0)

#define PLK_TRY(expr)              \
    do {                           \
        int _rc = (expr);          \
        if (_rc != PLK_OK) return _rc; \
    } while (0)

// Makes sure the calling thread's device is selected (plk_init may have been called on another thread: hipSetDevice is per
// thread): the physical device of the thread's logical device in the group (multi.hip).
int ensure_device();

// ---- the device group (multi.hip): logical device d of the process -> physical HIP device ----
constexpr int PLK_MAX_DEVICES = 16;
int group_size();                            // logical devices in use (1 unless plk_init_devices made it more)
int group_phys(int logical);                 // physical HIP device of a logical device
int thread_logical_device();                 // the logical device the calling thread's calls run on
void set_thread_logical_device(int logical);
int next_round_robin_device();               // single-unit calls from many host threads take the devices in turn
unsigned multi_min_log_n();                  // size gate of the fan-out (PLK_MULTI_MIN_LOG_N)
struct DeviceScope {                         // the calling thread works on logical device `logical` for the scope; on exit the
    int prev;                                // thread's logical device AND its current HIP device are what they were
    int prev_hip;
    explicit DeviceScope(int logical);
    ~DeviceScope();
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
// Every public entry point (capi.hip: PLK_API) leaves the calling thread's current HIP device as it found it: the library selects
// devices freely inside a call (the thread's logical device, a context's device, the devices of a group taken in turn), and a host
// shim or torch that allocates on "the current device" afterwards must not land on another GPU.  Nested calls
// (plk_ntt -> plk_ntt_batch) restore once, at the outermost level.
struct ApiGuard {
    int dev = -1;
    bool outer;
    ApiGuard();
    ~ApiGuard();
    ApiGuard(const ApiGuard&) = delete;
    ApiGuard& operator=(const ApiGuard&) = delete;
};
#define PLK_API plk::ApiGuard plk_api_guard_
int group_init_single(int device);           // plk_init
int group_init(int n_devices);               // plk_init_devices
void group_shutdown();
// fn(d) on the worker thread of every logical device d < count, side by side; returns when all have; the first failure wins
int run_on_devices(int count, const std::function<int(int)>& fn);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// RAII device buffer for the host-pointer entry points
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        if (p) (void)hipFree(p);
        p = nullptr;
        if (bytes == 0) bytes = 16;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            p = nullptr;
            return set_error(e == hipErrorOutOfMemory ? PLK_ERR_OOM : PLK_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        }
        return PLK_OK;
    }
};

// Stream-aware scratch buffers owned by the library (no hipMallocAsync: the stream-ordered pool
// of the runtime is left to the host framework).  A buffer is handed out again only to work that
// is ordered after its previous use: same stream, or its last-use event has completed.
void* scratch_acquire(size_t bytes, hipStream_t stream);  // nullptr on allocation failure (error text set)
void scratch_release(void* p, hipStream_t stream);       // call after the last kernel using p is enqueued
void scratch_clear();

// Streams of the library's own (side streams of a context, the lanes of the host-pointer entry points) come from a process-wide
// pool and go back to it instead of being destroyed: events of the scratch pool (and of other contexts) may have been recorded on
// them last, and HIP keeps a pointer to the recording stream in an event - querying or waiting on such an event after
// hipStreamDestroy was seen to fail with "operation not permitted when stream is capturing" (tools/fuzz_gpu.py, round 3).
hipStream_t stream_pool_acquire();         // a non-blocking stream of the current device; nullptr on failure (error text set)
void stream_pool_release(hipStream_t s);   // work still in flight on it simply precedes the next user's

// Optional element-wise work fused into the first pass's loads and the last pass's stores of a transform
// (poly.hip: coset scaling, zero padding, division by the vanishing polynomial).  Tables hold canonical
// R'-form words (the form of the twiddle tables, see ntt.hip).  Geometric tables: b^i = hi[i >> 10] * lo[i & 1023].
constexpr int NTT_POW_LO_LOG = 10;
struct NttHooks {
    size_t in_len = 0;            // input elements per transform actually stored; the rest reads as zero
    size_t in_stride = 0;         // distance between the inputs of a batch, in elements
    const void* in_lo = nullptr;  // x_i *= b^i on load
    const void* in_hi = nullptr;
    const void* out_tab = nullptr;  // y_i *= out_tab[i & out_mask] on store
    size_t out_mask = 0;
    const void* out_lo = nullptr;  // y_i *= b^i on store
    const void* out_hi = nullptr;
};

// ---- entry points implemented in ntt.hip / msm.hip / fieldops.hip, wrapped by capi.hip ----
int ntt_precompute_impl(int field, unsigned log_n);
int ntt_clear_cache_impl();
int ntt_dev_impl(int field, unsigned log_n, int inverse, unsigned batch, const void* d_in, void* d_out, hipStream_t stream);
int ntt_dev_hooked_impl(int field, unsigned log_n, int inverse, unsigned batch, const void* d_in, void* d_out, const NttHooks& hooks,
                        hipStream_t stream);
// device pointer to the power table of the (cached) plan: pw[b] = w^(2^b), b < log_t, w the primitive 2^log_t-th root
// (log_t = max(log_n, 10)), R-form
// *hold (when given) shares ownership of the plan: keep it until the kernels reading pw have been enqueued AND the stream has
// been synchronised or the hold is released after them (a concurrent plk_ntt_clear_cache frees the table otherwise)
int ntt_plan_pow_table(int field, unsigned log_n, const void** pw, int* log_t, std::shared_ptr<const void>* hold = nullptr);
int ntt_reference_table_dev_impl(int field, unsigned log_n, void* d_out, hipStream_t stream);
int plonk_clear_cache_impl();
int field_inner_product_dev_impl(int field, const void* d_a, const void* d_b, size_t count, void* d_out, hipStream_t stream);
int field_fold_slices_dev_impl(int field, const void* d_lo, const void* d_hi, const uint64_t* s_lo, const uint64_t* s_hi, size_t count, void* d_out,
                               hipStream_t stream);
int field_bytes_impl(int field, int from_bytes, const void* d_in, size_t count, void* d_out, unsigned* d_bad, hipStream_t stream);
int point_bytes_impl(int curve, int from_bytes, const void* d_in, const void* d_zero, size_t count, void* d_out, void* d_out_zero, void* d_status,
                     hipStream_t stream);
int field_batch_inverse_dev_impl(int field, const void* d_x, void* d_out, void* d_is_zero, unsigned* d_zero_count, size_t count, hipStream_t stream);
int curve_batch_to_affine_dev_impl(int curve, size_t count, const void* d_xyz, const void* d_zero, void* d_out_xy, void* d_out_zero, hipStream_t stream);
int plonk_vanishing_points_dev_impl(int field, unsigned log_degree, const void* d_constants, const void* d_wires, const void* d_s_sigma, const void* d_z,
                                    const uint64_t* k_is, const uint64_t* alpha, const uint64_t* beta, const uint64_t* gamma, const uint64_t* inner_zeta,
                                    const uint64_t* inner_a, void* d_out, hipStream_t stream);
int plonk_all_constraints_dev_impl(int field, size_t count, const void* d_constants, const void* d_local, const void* d_right, const void* d_below,
                                   const uint64_t* inner_zeta, const uint64_t* inner_a, void* d_out, hipStream_t stream);

// per-vector generator ranges (and, optionally, bucket ranges) of a batched MSM execution (msm.hip: msm_execute_dev_impl; host arrays of
// `batch` entries; scalars[b]: a device pointer)
struct MsmParts {
    const uint64_t* first;
    const uint64_t* count;
    const void* const* scalars;
    const uint32_t* bucket_part = nullptr;   // optional: vector b keeps the bucket_part[b]-th of bucket_parts[b] ranges of the coarse bins
    const uint32_t* bucket_parts = nullptr;  // (absent, 0 or 1: every bucket)
};

int field_limbs(int field);
int curve_limbs(int curve);
int curve_scalar_field(int curve);

}  // namespace plk
