// multi.hip -- several GPUs behind the single-process C ABI (plk_init(n_devices), 8(e); section 5: PLK_NGPU).
//
// The reference's callers are ONE process with Rayon threads: commit_polynomials -> coeffs_vec_to_commitments
// (plonk_util.rs:215-231, poly_commit.rs:52-66) commits the wire polynomials one msm_execute_parallel after the other, and
// polynomials_to_values_padded / values_to_polynomials (plonk_util.rs:169-190) run nine transforms on nine threads.  With those
// files untouched the library is entered from one process - so the split over the GPUs of a node has to happen HERE, below the
// entry points:
//   * a device group (plk_init_devices / PLK_NGPU): logical device d -> physical HIP device; PLK_VIRTUAL_DEVICES=k makes k
//     logical devices out of ONE physical GPU (k contexts, k worker threads, k stream sets) so that the whole path runs in a
//     one-GPU test suite;
//   * one worker thread per logical device (hipSetDevice is per thread; a worker owns the lanes of its device): a fan-out call
//     hands every worker its share, the workers ENQUEUE side by side and return, the caller's stream waits for their events;
//   * msm_precompute builds, on every device, the full window tables (for whole vectors of a batch) and a context over the
//     device's own contiguous share of the generators with the window such a share deserves (for a single sharded MSM);
//   * msm_execute[_batch]: the batch plan of SURVEY 8(e) / DESIGN section 6 - floor(batch / N) WHOLE vectors per device (vector v
//     belongs to device v mod N), the remaining batch mod N vectors SHARDED by contiguous base range, every device's partial
//     results in one record (plk_msm_partials_bytes), the records copied peer to peer (xGMI; a device-to-device copy when the
//     devices are virtual) into one buffer on the caller's device, k_combine_partials there.  Point addition is not a
//     collective's reduction op and the payload is a few hundred bytes: N - 1 small peer writes are the whole exchange;
//   * transforms are independent units: a batch is dealt out round-robin (no exchange), single-transform calls coming from
//     many host threads take the devices in turn.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "host_lane.h"

struct plk_msm_ctx;

namespace plk {

int msm_precompute_dev_impl(int curve, size_t n, const void* d_bases, const void* d_zero, unsigned window_bits, unsigned flags, hipStream_t stream,
                            plk_msm_ctx** out_ctx, const void* d_extra = nullptr, size_t n_extra = 0, const size_t* also_n = nullptr,
                            int also_count = 0);
int msm_execute_dev_impl(plk_msm_ctx* ctx, unsigned batch, const void* d_scalars, size_t n_scalars, void* d_out_xy, void* d_out_zero, hipStream_t stream,
                         hipEvent_t* ready = nullptr, const MsmParts* parts = nullptr, unsigned out_flags = 0);
int msm_combine_partials_dev_impl(int curve, unsigned world, unsigned batch, unsigned whole_per_rank, const void* d_gathered, void* d_out_xy, void* d_out_zero,
                                  hipStream_t stream);
size_t msm_partials_bytes(int curve, unsigned batch);
size_t msm_ctx_len(const plk_msm_ctx* ctx);
int msm_ctx_curve(const plk_msm_ctx* ctx);
int msm_ctx_table_free(const plk_msm_ctx* ctx);
void msm_ctx_delete(plk_msm_ctx* ctx);
std::vector<plk_msm_ctx*>& msm_ctx_peers(plk_msm_ctx* ctx);
std::vector<plk_msm_ctx*>& msm_ctx_shards(plk_msm_ctx* ctx);

// ---- the device group ----
static std::mutex g_group_mu;            // guards changes of the group and of the worker set
static std::atomic<int> g_n{0};          // logical devices; 0: not initialised (the first call adopts the thread's current device)
static int g_phys[PLK_MAX_DEVICES] = {};
static thread_local int t_logical = 0;   // the logical device this thread's calls run on
static std::atomic<unsigned> g_round_robin{0};

int group_size() {
    const int n = g_n.load(std::memory_order_acquire);
    return n > 0 ? n : 1;
}
int group_phys(int logical) { return g_phys[logical >= 0 && logical < group_size() ? logical : 0]; }
int thread_logical_device() { return t_logical < group_size() ? t_logical : 0; }
void set_thread_logical_device(int logical) { t_logical = logical; }
int next_round_robin_device() { return (int)(g_round_robin.fetch_add(1u, std::memory_order_relaxed) % (unsigned)group_size()); }

unsigned multi_min_log_n() {
    // below 2^this many scalars / elements a call stays on one device: a device's share must be worth the fixed latency of an
    // MSM reduction (0.45 ms) resp. of a transform's three launches
    static const unsigned v = [] {
        const char* e = getenv("PLK_MULTI_MIN_LOG_N");
        const int x = e ? atoi(e) : 17;
        return x < 0 ? 0u : (unsigned)x;
    }();
    return v;
}

// ---- copies between two devices of the group -------------------------------------------------------------------------------
// Device-resident inputs (generators, scalar vectors) and the records of partial results cross from one device to another.  With
// peer access (xGMI; enabled pair by pair in group_init, the outcome kept in g_peer) that is ONE hipMemcpyAsync on the worker's
// stream.  Without it the library does not lean on the runtime's silent staging: the copy goes through a pinned host buffer,
// explicitly and synchronously - the source device's null stream after the worker's stream has been synchronised (its inputs are
// complete), then the destination's - and is counted (plk_group_copy_stats), so that a test can tell which path ran.
// PLK_PEER_MODE=host forces that path for every pair, ALSO between the logical devices of PLK_VIRTUAL_DEVICES (which then stop
// sharing their buffers: a one-GPU box executes the no-peer branch line by line).
static bool g_peer[PLK_MAX_DEVICES][PLK_MAX_DEVICES] = {};
static bool g_force_host_copies = false;
static std::atomic<unsigned long long> g_copies_peer{0}, g_copies_staged{0};
bool group_force_host_copies() { return g_force_host_copies; }
void group_copy_stats(unsigned long long* peer, unsigned long long* staged) {
    if (peer) *peer = g_copies_peer.load();
    if (staged) *staged = g_copies_staged.load();
}
// the worker of logical device `me` (current device, stream `st` of it) copies `bytes` between its device and logical device `other`;
// pull: other -> me, else me -> other.  On return the copy is enqueued on `st` (peer) or complete (staged).
static int group_copy(int me, int other, bool pull, void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (!bytes) return PLK_OK;
    const bool direct = me == other || (!g_force_host_copies && (g_phys[me] == g_phys[other] || g_peer[me][other]));
    if (direct) {
        PLK_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st));
        g_copies_peer.fetch_add(1, std::memory_order_relaxed);
        return PLK_OK;
    }
    // ONE bounded pinned buffer per calling thread (a worker, or a host thread that drives a device itself), kept for the life of the
    // thread, and a loop over pieces of it: a copy of n bytes used to take hipHostMalloc + hipHostFree of n bytes - 100 MiB per worker at
    // once for 2^20 generators, both calls synchronising the device
    constexpr size_t BOUNCE_BYTES = (size_t)16 << 20;
    static thread_local void* bounce = nullptr;
    if (!bounce) PLK_HIP_TRY(hipHostMalloc(&bounce, BOUNCE_BYTES, hipHostMallocPortable));
    PLK_HIP_TRY(hipStreamSynchronize(st));  // what the copy reads (pull: the caller's stream, through ev_in; push: this stream's kernels) is complete
    const int from = pull ? other : me, to = pull ? me : other;
    hipError_t e = hipSuccess;
    for (size_t at = 0; at < bytes && e == hipSuccess; at += BOUNCE_BYTES) {
        const size_t piece = bytes - at < BOUNCE_BYTES ? bytes - at : BOUNCE_BYTES;
        (void)hipSetDevice(g_phys[from]);
        e = hipMemcpy(bounce, (const uint8_t*)src + at, piece, hipMemcpyDeviceToHost);
        if (e == hipSuccess) {
            (void)hipSetDevice(g_phys[to]);
            e = hipMemcpy((uint8_t*)dst + at, bounce, piece, hipMemcpyHostToDevice);
        }
    }
    (void)hipSetDevice(g_phys[me]);
    PLK_HIP_TRY(e);
    g_copies_staged.fetch_add(1, std::memory_order_relaxed);
    return PLK_OK;
}
// PLK_TEST_WORKER_JITTER_US=n: every worker sleeps a random 0..n microseconds before and after it enqueues its share - the
// hand-overs of a fan-out call must not depend on the workers arriving in step (tests/test_gpu_multi.py: 200 calls)
static void test_jitter() {
    static const int max_us = [] {
        const char* e = getenv("PLK_TEST_WORKER_JITTER_US");
        return e ? atoi(e) : 0;
    }();
    if (max_us <= 0) return;
    static thread_local unsigned state = (unsigned)std::hash<std::thread::id>()(std::this_thread::get_id()) | 1u;
    state = state * 1664525u + 1013904223u;
    std::this_thread::sleep_for(std::chrono::microseconds((state >> 8) % (unsigned)(max_us + 1)));
}

int ensure_device() {
    int n = g_n.load(std::memory_order_acquire);
    if (n == 0) {
        std::lock_guard<std::mutex> lk(g_group_mu);
        n = g_n.load(std::memory_order_acquire);
        if (n == 0) {
            int count = 0;
            hipError_t e = hipGetDeviceCount(&count);
            if (e != hipSuccess || count <= 0)
                return set_error(PLK_ERR_NO_DEVICE, "no HIP device visible (%s); the HIP path has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
            int cur = 0;
            if (hipGetDevice(&cur) != hipSuccess) cur = 0;
            g_phys[0] = cur;
            g_n.store(1, std::memory_order_release);
            n = 1;
        }
    }
    const int l = t_logical < n ? t_logical : 0;
    PLK_HIP_TRY(hipSetDevice(g_phys[l]));
    return PLK_OK;
}

DeviceScope::DeviceScope(int logical) : prev(t_logical), prev_hip(-1) {
    if (hipGetDevice(&prev_hip) != hipSuccess) prev_hip = -1;
    t_logical = logical;
}
DeviceScope::~DeviceScope() {
    t_logical = prev;
    if (prev_hip >= 0) (void)hipSetDevice(prev_hip);
}

static thread_local int t_api_depth = 0;
ApiGuard::ApiGuard() : outer(t_api_depth++ == 0) {
    if (outer && hipGetDevice(&dev) != hipSuccess) {
        dev = -1;
        (void)hipGetLastError();
    }
}
ApiGuard::~ApiGuard() {
    --t_api_depth;
    if (outer && dev >= 0) (void)hipSetDevice(dev);
}

// ---- worker threads: one per logical device ----
struct Worker {
    int logical = 0;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    const std::function<int(int)>* job = nullptr;
    bool pending = false, finished = false, stop = false;
    std::atomic<int> poke{0};  // bumped with every hand-over in either direction: lets the other side spin briefly instead of sleeping
    int rc = PLK_OK;
    std::string err;
    hipEvent_t ev_done = nullptr;  // on this worker's device: "my share of the current fan-out call is enqueued up to here"
};
static std::vector<Worker*>* g_workers = nullptr;  // never destroyed at exit: detached threads may still wait on their condition variables
static std::mutex g_dispatch_mu;                    // one fan-out call at a time (each uses every device anyway)

// A prover calls back to back: after a hand-over the other side is usually ready within microseconds, and a condition-variable
// wake-up costs tens of them - a noticeable part of a single MSM sharded eight ways (0.5 ms per device).  Both sides therefore spin
// on the hand-over counter for up to ~100 us before they go to sleep.
static void spin_for_poke(const std::atomic<int>& poke, int seen) {
    const auto t0 = std::chrono::steady_clock::now();
    while (poke.load(std::memory_order_acquire) == seen) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(100)) return;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}
static void worker_main(Worker* w) {
    t_logical = w->logical;
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        if (!w->pending && !w->stop) {
            const int seen = w->poke.load(std::memory_order_acquire);
            lk.unlock();
            spin_for_poke(w->poke, seen);
            lk.lock();
        }
        w->cv.wait(lk, [&] { return w->pending || w->stop; });
        if (w->stop) break;
        const std::function<int(int)>* job = w->job;
        w->pending = false;
        lk.unlock();
        int rc = ensure_device();
        if (rc == PLK_OK) rc = (*job)(w->logical);
        lk.lock();
        w->rc = rc;
        w->err = rc != PLK_OK ? last_error_ref() : std::string();
        w->finished = true;
        w->poke.fetch_add(1, std::memory_order_release);
        w->cv.notify_all();
    }
}

static void workers_stop_locked() {
    if (!g_workers) return;
    for (Worker* w : *g_workers) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->stop = true;
        }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        if (w->ev_done) (void)hipEventDestroy(w->ev_done);
        delete w;
    }
    g_workers->clear();
}

static int workers_ensure() {
    std::lock_guard<std::mutex> lk(g_group_mu);
    if (!g_workers) g_workers = new std::vector<Worker*>();
    const int n = group_size();
    if ((int)g_workers->size() == n) return PLK_OK;
    workers_stop_locked();
    for (int d = 0; d < n; ++d) {
        Worker* w = new Worker();
        w->logical = d;
        PLK_HIP_TRY(hipSetDevice(g_phys[d]));
        PLK_HIP_TRY(hipEventCreateWithFlags(&w->ev_done, hipEventDisableTiming));
        w->th = std::thread(worker_main, w);
        g_workers->push_back(w);
    }
    PLK_HIP_TRY(hipSetDevice(g_phys[thread_logical_device()]));
    return PLK_OK;
}

// fn(d) on the worker thread of every logical device d < count, side by side; returns when all have returned.  The first
// failure's code and text become the caller's.  The caller holds g_dispatch_mu.
static int run_on_devices_locked(int count, const std::function<int(int)>& fn) {
    PLK_TRY(workers_ensure());
    std::vector<Worker*>& ws = *g_workers;
    for (int d = 0; d < count; ++d) {
        std::lock_guard<std::mutex> lk(ws[d]->mu);
        ws[d]->job = &fn;
        ws[d]->finished = false;
        ws[d]->pending = true;
        ws[d]->poke.fetch_add(1, std::memory_order_release);
    }
    for (int d = 0; d < count; ++d) ws[d]->cv.notify_all();
    int rc = PLK_OK;
    for (int d = 0; d < count; ++d) {
        std::unique_lock<std::mutex> lk(ws[d]->mu);
        if (!ws[d]->finished) {
            const int seen = ws[d]->poke.load(std::memory_order_acquire);
            lk.unlock();
            spin_for_poke(ws[d]->poke, seen);
            lk.lock();
        }
        ws[d]->cv.wait(lk, [&] { return ws[d]->finished; });
        if (ws[d]->rc != PLK_OK && rc == PLK_OK) {
            rc = ws[d]->rc;
            last_error_ref() = "device " + std::to_string(d) + ": " + ws[d]->err;
        }
    }
    return rc;
}
int run_on_devices(int count, const std::function<int(int)>& fn) {
    std::lock_guard<std::mutex> lk(g_dispatch_mu);
    return run_on_devices_locked(count, fn);
}

static int check_gfx950(int device) {
    hipDeviceProp_t prop;
    PLK_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(PLK_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    return PLK_OK;
}

// plk_init: one device for this process
int group_init_single(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return set_error(PLK_ERR_NO_DEVICE, "no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device == -1) {
        const char* env = getenv("PLK_DEVICE");
        device = env ? atoi(env) : 0;
    }
    if (device < 0 || device >= count) return set_error(PLK_ERR_INVALID_ARG, "device %d out of range (%d visible)", device, count);
    PLK_HIP_TRY(hipSetDevice(device));
    PLK_TRY(check_gfx950(device));
    std::lock_guard<std::mutex> dl(g_dispatch_mu);
    std::lock_guard<std::mutex> lk(g_group_mu);
    if (g_n.load() > 1) workers_stop_locked();
    g_phys[0] = device;
    g_n.store(1, std::memory_order_release);
    return PLK_OK;
}

// plk_init_devices: n_devices > 0: that many; 0: PLK_NGPU, else every visible device.  PLK_VIRTUAL_DEVICES=k (k >= 2): k logical
// devices on the ONE physical device PLK_DEVICE (default 0) - n_devices, when positive, overrides k.
int group_init(int n_devices) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return set_error(PLK_ERR_NO_DEVICE, "no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (n_devices < 0) return set_error(PLK_ERR_INVALID_ARG, "n_devices = %d", n_devices);
    const char* venv = getenv("PLK_VIRTUAL_DEVICES");
    const int virt = venv ? atoi(venv) : 0;
    int phys[PLK_MAX_DEVICES];
    int n = n_devices;
    if (virt >= 2) {
        if (n == 0) n = virt;
        if (n > PLK_MAX_DEVICES) return set_error(PLK_ERR_INVALID_ARG, "%d virtual devices (at most %d)", n, PLK_MAX_DEVICES);
        const char* denv = getenv("PLK_DEVICE");
        const int base = denv ? atoi(denv) : 0;
        if (base < 0 || base >= count) return set_error(PLK_ERR_INVALID_ARG, "PLK_DEVICE %d out of range (%d visible)", base, count);
        for (int d = 0; d < n; ++d) phys[d] = base;
    } else {
        if (n == 0) {
            const char* nenv = getenv("PLK_NGPU");
            n = nenv ? atoi(nenv) : count;
            if (n <= 0) n = count;
        }
        if (n > count) return set_error(PLK_ERR_INVALID_ARG, "%d devices asked for, %d visible", n, count);
        if (n > PLK_MAX_DEVICES) return set_error(PLK_ERR_INVALID_ARG, "%d devices (at most %d)", n, PLK_MAX_DEVICES);
        for (int d = 0; d < n; ++d) phys[d] = d;
    }
    for (int d = 0; d < n; ++d)
        if (d == 0 || phys[d] != phys[d - 1]) PLK_TRY(check_gfx950(phys[d]));
    // peer access both ways between every pair of distinct devices (xGMI): the exchange of partial results and the device-source
    // forms copy straight from one HBM to another.  Where it cannot be enabled the runtime stages such copies through the host.
    bool peer[PLK_MAX_DEVICES][PLK_MAX_DEVICES] = {};
    const char* pm = getenv("PLK_PEER_MODE");
    const bool force_host = pm && strcmp(pm, "host") == 0;
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            if (phys[a] == phys[b] || force_host) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, phys[a], phys[b]) != hipSuccess || !can) {
                (void)hipGetLastError();
                continue;
            }
            PLK_HIP_TRY(hipSetDevice(phys[a]));
            const hipError_t pe = hipDeviceEnablePeerAccess(phys[b], 0);
            if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) peer[a][b] = true;
            if (pe != hipSuccess) (void)hipGetLastError();
        }
    PLK_HIP_TRY(hipSetDevice(phys[0]));
    std::lock_guard<std::mutex> dl(g_dispatch_mu);
    std::lock_guard<std::mutex> lk(g_group_mu);
    bool same = g_n.load() == n;
    for (int d = 0; d < n && same; ++d) same = g_phys[d] == phys[d];
    if (!same) {
        workers_stop_locked();
        for (int d = 0; d < n; ++d) g_phys[d] = phys[d];
        g_n.store(n, std::memory_order_release);
    }
    // a copy between a and b is one peer copy only when BOTH directions are open (the copy engine of either side may run it)
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) g_peer[a][b] = peer[a][b] && peer[b][a];
    g_force_host_copies = force_host;
    if (getenv("PLK_VERBOSE")) {
        int open = 0, pairs = 0;
        for (int a = 0; a < n; ++a)
            for (int b = a + 1; b < n; ++b)
                if (phys[a] != phys[b]) {
                    ++pairs;
                    open += g_peer[a][b] ? 1 : 0;
                }
        fprintf(stderr, "[plonky_hip] device group of %d: %d of %d device pairs with peer access%s\n", n, open, pairs,
                force_host ? " (PLK_PEER_MODE=host: every cross-device copy staged through pinned host memory)" : "");
    }
    return PLK_OK;
}

void group_shutdown() {
    std::lock_guard<std::mutex> dl(g_dispatch_mu);
    std::lock_guard<std::mutex> lk(g_group_mu);
    workers_stop_locked();
    g_n.store(0, std::memory_order_release);
}

// ---- MSM over the group ----
// first generator of device d's contiguous share: balanced, sizes differ by at most one, the larger shares first - the same split
// as parallel.shard_bounds of the one-process-per-GPU form
static inline size_t share_first(size_t n, int d, int world) {
    const size_t base = n / (size_t)world, rem = n % (size_t)world, dd = (size_t)d;
    return dd * base + (dd < rem ? dd : rem);
}

bool msm_ctx_is_multi(plk_msm_ctx* ctx) { return ctx && !msm_ctx_shards(ctx).empty(); }

// The batch plan, in one place (pure arithmetic; also behind plk_multi_plan for hosts and tests): `batch` vectors of n scalars over
// `world` devices.  whole = floor(batch / world) vectors go to every device WHOLE - vector s * world + d is slot s of device d - and
// the remaining batch - whole * world vectors are SHARDED by contiguous base range: vector whole * world + j is slot whole + j of
// every device, which reduces the generators [n d / world, n (d + 1) / world) of it.
void multi_plan_slot(int world, unsigned batch, size_t n, int d, unsigned slot, unsigned* vec, size_t* first, size_t* count) {
    const unsigned whole = batch / (unsigned)world;
    if (slot < whole) {
        *vec = slot * (unsigned)world + (unsigned)d;
        *first = 0;
        *count = n;
    } else {
        *vec = whole * (unsigned)world + (slot - whole);
        *first = share_first(n, d, world);
        *count = share_first(n, d + 1, world) - *first;
    }
}

// msm_precompute on every device of the group.  `bases` / `zero`: host memory (host_src) or memory of the caller's device.
// The caller's stream is synchronised on return (as msm_precompute_dev_impl does).
int msm_precompute_multi(int curve, size_t n, const void* bases, const void* zero, bool host_src, unsigned window_bits, hipStream_t caller_stream,
                         plk_msm_ctx** out_ctx) {
    if (!out_ctx) return set_error(PLK_ERR_INVALID_ARG, "null out_ctx");
    *out_ctx = nullptr;
    const int L = curve_limbs(curve);
    if (L < 0) return set_error(PLK_ERR_INVALID_ARG, "bad curve id %d", curve);
    const int world = group_size();
    const size_t pt = (size_t)2 * L * 8;
    PLK_TRY(ensure_device());
    const int src_logical = thread_logical_device();
    const int src_phys = group_phys(src_logical);
    // registrations BEFORE the fan-out lock (as plk_msm_execute_batch takes them): a thread never waits for another holder's range while it
    // keeps every other fan-out call out; released after the lock, when the workers' copies are complete
    HostPin pin_b, pin_z;
    if (host_src) {
        pin_b.pin(bases, n * pt);
        if (zero) pin_z.pin(zero, n);
    }
    std::lock_guard<std::mutex> dl(g_dispatch_mu);
    PLK_TRY(workers_ensure());
    if (!host_src) PLK_HIP_TRY(hipStreamSynchronize(caller_stream));  // the generators are complete before another device reads them
    std::vector<plk_msm_ctx*> full((size_t)world, nullptr), shard((size_t)world, nullptr);
    auto job = [&](int d) -> int {
        HostLane* l = nullptr;
        PLK_TRY(lane_get(l));
        const void* b = bases;
        const void* z = zero;
        LaneBuf db, dz;
        test_jitter();
        // the caller's buffers serve as they are only on the caller's own device (PLK_PEER_MODE=host: only on its own LOGICAL device)
        const bool in_place = !host_src && group_phys(d) == src_phys && (d == src_logical || !group_force_host_copies());
        if (!in_place) {
            PLK_TRY(db.alloc(n * pt, l->stream));
            if (host_src) PLK_HIP_TRY(hipMemcpyAsync(db.p, bases, n * pt, hipMemcpyHostToDevice, l->stream));  // over this device's own PCIe link
            else PLK_TRY(group_copy(d, src_logical, true, db.p, bases, n * pt, l->stream));
            b = db.p;
            if (zero) {
                PLK_TRY(dz.alloc(n, l->stream));
                if (host_src) PLK_HIP_TRY(hipMemcpyAsync(dz.p, zero, n, hipMemcpyHostToDevice, l->stream));
                else PLK_TRY(group_copy(d, src_logical, true, dz.p, zero, n, l->stream));
                z = dz.p;
            }
        }
        int rc = msm_precompute_dev_impl(curve, n, b, z, window_bits, 0, l->stream, &full[(size_t)d]);
        if (rc == PLK_OK) {
            const size_t f = share_first(n, d, world), e = share_first(n, d + 1, world);
            rc = msm_precompute_dev_impl(curve, e - f, (const uint8_t*)b + f * pt, z ? (const uint8_t*)z + f : nullptr, 0, 0, l->stream, &shard[(size_t)d]);
        }
        const hipError_t se = hipStreamSynchronize(l->stream);
        l->pin_used = 0;
        if (rc == PLK_OK && se != hipSuccess) rc = set_error(PLK_ERR_HIP, "precompute on device %d failed: %s", d, hipGetErrorString(se));
        return rc;
    };
    const int rc = run_on_devices_locked(world, job);
    if (rc != PLK_OK) {
        for (auto* v : {&full, &shard})
            for (plk_msm_ctx* c : *v)
                if (c) msm_ctx_delete(c);
        (void)ensure_device();
        return rc;
    }
    plk_msm_ctx* ctx = full[0];
    for (int d = 1; d < world; ++d) msm_ctx_peers(ctx).push_back(full[(size_t)d]);
    msm_ctx_shards(ctx) = shard;
    *out_ctx = ctx;
    return PLK_OK;
}

// `batch` scalar vectors (vecs[b]: n scalars of 32 bytes, host memory or memory of the caller's device) against a context built
// by msm_precompute_multi.  Results: device memory of the caller's device, asynchronous on caller_stream (which also orders the
// inputs of the device-source form).  The workers only enqueue; a host-source caller synchronises caller_stream before it lets
// go of the vectors.
int msm_execute_multi(plk_msm_ctx* ctx, unsigned batch, const void* const* vecs, bool host_src, size_t n, void* d_out_xy, void* d_out_zero,
                      hipStream_t caller_stream) {
    if (!ctx || !msm_ctx_is_multi(ctx)) return set_error(PLK_ERR_INVALID_ARG, "not a multi-device context");
    if (n != msm_ctx_len(ctx))
        return set_error(PLK_ERR_SIZE_MISMATCH, "scalars.len() = %zu but the precomputation holds %zu generators (curve_msm.rs:67)", n, msm_ctx_len(ctx));
    if (batch == 0) return PLK_OK;
    if (!vecs || !d_out_xy || !d_out_zero) return set_error(PLK_ERR_INVALID_ARG, "null pointer");
    std::vector<plk_msm_ctx*>& peers = msm_ctx_peers(ctx);
    std::vector<plk_msm_ctx*>& shards = msm_ctx_shards(ctx);
    const int world = (int)shards.size();
    if (world != group_size() || (int)peers.size() != world - 1)
        return set_error(PLK_ERR_INVALID_ARG, "the context was built for %d devices, the library now runs %d", world, group_size());
    const int curve = msm_ctx_curve(ctx);
    const size_t L = (size_t)curve_limbs(curve);
    const unsigned whole = batch / (unsigned)world, rem = batch - whole * (unsigned)world, slots = whole + rem;
    const size_t rec = msm_partials_bytes(curve, slots);
    PLK_TRY(ensure_device());
    const int src_logical = thread_logical_device();
    const int src_phys = group_phys(src_logical);
    std::lock_guard<std::mutex> dl(g_dispatch_mu);
    PLK_TRY(workers_ensure());
    uint8_t* gathered = (uint8_t*)scratch_acquire(rec * (size_t)world, caller_stream);
    if (!gathered) return PLK_ERR_OOM;
    struct Release {
        void* p;
        hipStream_t s;
        ~Release() { scratch_release(p, s); }
    } release_gathered{gathered, caller_stream};
    // the workers' streams start after whatever the caller's stream holds now (device-resident inputs; the previous use of d_out)
    static thread_local hipEvent_t t_ev_in = nullptr;
    static thread_local int t_ev_in_dev = -1;
    if (t_ev_in && t_ev_in_dev != src_phys) {
        (void)hipEventDestroy(t_ev_in);
        t_ev_in = nullptr;
    }
    if (!t_ev_in) {
        PLK_HIP_TRY(hipEventCreateWithFlags(&t_ev_in, hipEventDisableTiming));
        t_ev_in_dev = src_phys;
    }
    const hipEvent_t ev_in = t_ev_in;  // a plain copy for the workers: a thread_local named inside the lambda would be THEIR instance
    PLK_HIP_TRY(hipEventRecord(ev_in, caller_stream));
    auto job = [&](int d) -> int {
        HostLane* l = nullptr;
        PLK_TRY(lane_get(l));
        test_jitter();
        // ev_in was recorded on the caller's stream, possibly on another device: an event without hipEventDisableSystemFence releases
        // at system scope when it is recorded and the waiting stream acquires - the caller's inputs are visible to this device
        PLK_HIP_TRY(hipStreamWaitEvent(l->stream, ev_in, 0));
        PLK_TRY(lane_fork(*l));  // the copy stream starts after the main one, i.e. after the caller's inputs
        while (l->ev_ready.size() < slots) {
            hipEvent_t e = nullptr;
            PLK_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            l->ev_ready.push_back(e);
        }
        // this device's slots: its whole vectors over all the generators, then its base range of every sharded vector
        const bool pure_shard = whole == 0;  // a single MSM (or fewer vectors than devices): the context over this device's own range
        std::vector<uint64_t> first(slots), count(slots);
        std::vector<unsigned> vec(slots);
        std::vector<const void*> ptr(slots);
        size_t need = 0;
        for (unsigned s = 0; s < slots; ++s) {
            size_t f = 0, c = 0;
            multi_plan_slot(world, batch, n, d, s, &vec[s], &f, &c);
            first[s] = f;
            count[s] = c;
            need += c;
        }
        // the vectors already live on this device (PLK_PEER_MODE=host: only the caller's own logical device reads them in place)
        const bool direct = !host_src && group_phys(d) == src_phys && (d == src_logical || !group_force_host_copies());
        LaneBuf sbuf, rbuf;
        // declared AFTER the buffers, so it runs BEFORE they go back to the pool: a job that fails half way has uploads in flight on
        // aux[0] into sbuf, and the pool orders a released buffer after l->stream only
        struct DrainOnFailure {
            HostLane* l;
            bool armed = true;
            ~DrainOnFailure() {
                if (armed) (void)lane_join(*l);
            }
        } drain_guard{l};
        if (!direct) PLK_TRY(sbuf.alloc(need * 32, l->stream));
        PLK_TRY(rbuf.alloc(rec, l->stream));
        size_t off = 0;
        for (unsigned s = 0; s < slots; ++s) {
            const uint8_t* src = (const uint8_t*)vecs[vec[s]] + first[s] * 32;
            if (direct) {
                ptr[s] = src;
            } else {
                uint8_t* dst = (uint8_t*)sbuf.p + off * 32;
                // host memory crosses this device's own PCIe link; memory of the caller's device crosses xGMI
                if (host_src) PLK_HIP_TRY(hipMemcpyAsync(dst, src, count[s] * 32, hipMemcpyHostToDevice, l->aux[0]));
                else PLK_TRY(group_copy(d, src_logical, true, dst, src, count[s] * 32, l->aux[0]));
                ptr[s] = dst;
                off += count[s];
            }
            PLK_HIP_TRY(hipEventRecord(l->ev_ready[s], l->aux[0]));
            if (pure_shard) first[s] = 0;  // the shard context counts from its own first generator
        }
        MsmParts parts{first.data(), count.data(), ptr.data()};
        plk_msm_ctx* c = pure_shard ? shards[(size_t)d] : (d == 0 ? ctx : peers[(size_t)d - 1]);
        uint8_t* r = (uint8_t*)rbuf.p;
        PLK_TRY(msm_execute_dev_impl(c, slots, nullptr, 0, r, r + (size_t)slots * 2 * L * 8, l->stream, l->ev_ready.data(), &parts));
        // this device's record into its slot of the gathered buffer on the caller's device; the event below is recorded after it on
        // the same stream (system-scope release), the caller's stream waits for it, and k_combine_partials reads at system scope
        PLK_TRY(group_copy(d, src_logical, false, gathered + (size_t)d * rec, r, rec, l->stream));
        test_jitter();
        // the aux stream's copies feed kernels of l->stream through ev_ready, so l->stream's event below covers them too; sbuf / rbuf
        // go back to the pool ordered after l->stream
        PLK_HIP_TRY(hipEventRecord((*g_workers)[(size_t)d]->ev_done, l->stream));
        drain_guard.armed = false;
        return PLK_OK;
    };
    // whatever was enqueued must not outlive the caller's vectors, `gathered` or the workers' buffers: EVERY failure after the
    // fan-out has started drains the workers' lanes before this function returns
    auto drain_workers = [&]() {
        auto drain = [&](int) -> int {
            HostLane* l = nullptr;
            if (lane_get(l) == PLK_OK) (void)lane_join(*l);
            return PLK_OK;
        };
        const std::string keep = last_error_ref();
        (void)run_on_devices_locked(world, drain);
        last_error_ref() = keep;
        (void)ensure_device();
    };
    int rc = run_on_devices_locked(world, job);
    if (rc == PLK_OK) {
        auto tail = [&]() -> int {
            PLK_TRY(ensure_device());
            for (int d = 0; d < world; ++d) PLK_HIP_TRY(hipStreamWaitEvent(caller_stream, (*g_workers)[(size_t)d]->ev_done, 0));
            return msm_combine_partials_dev_impl(curve, (unsigned)world, batch, whole, gathered, d_out_xy, d_out_zero, caller_stream);
        };
        rc = tail();
    }
    if (rc != PLK_OK) drain_workers();
    return rc;
}

}  // namespace plk
