// C ABI of libhelm (see include/helm.h) and the host-side Krylov drivers.
//
// The drivers only enqueue kernels: all vectors and all scalar recurrences stay on the device;
// the host looks at the per-RHS status records every `check_every` iterations.
#include "helm_internal.hpp"
#include "direct.hpp"
#include <chrono>
#include <atomic>
#include <unistd.h>
#include <time.h>
#include <mutex>
#include <thread>
#include <map>
#include <cstring>
#include <algorithm>
#include <limits>

// launchers from kernels.hip not in the shared header
int helm_launch_fin_ex(helm_op *op, int which, int nrhs, int nblk_part, const int *mask, double *aux);
int helm_launch_restart_copy_mask(helm_op *op, VecPtrs w, int nrhs, const int *mask);
int helm_launch_norm2(helm_op *op, const cplx *a, int nrhs);
int helm_launch_krylov_init(helm_op *op, const cplx *bvec, VecPtrs w, int nrhs, double rtol);

// one record per host thread: handles are driven from several threads (bench --streams), and the handle-less error
// is read back by the thread that got the failing return code
static thread_local std::string g_last_error;

void helm_set_error(helm_op *op, const char *msg) {
    if (op) op->err = msg;
    g_last_error = msg;
}

extern "C" const char *helm_last_error(const helm_op *op) { return op ? op->err.c_str() : g_last_error.c_str(); }
extern "C" const char *helm_version(void) { return "libhelm 0.1 (gfx950)"; }

// ---- kernel registry and runtime-object bookkeeping (helm_internal.hpp) ---------------------------------------------------------------------
// (function-local statics: kernels register during the static initialisation of whichever translation unit comes first)
namespace {
struct KernelRec { const void *fn; const char *pretty; std::atomic<bool> launched{false}; };
struct KernelRegistry {
    std::mutex mu;
    std::vector<KernelRec *> recs;                 // records are never moved or freed: slots stay valid without the lock
    std::atomic<KernelRec *> fast[2048];
    std::atomic<int> n{0};
};
KernelRegistry &kreg() { static KernelRegistry *r = new KernelRegistry(); return *r; }
struct RuntimeCounters {
    std::atomic<long long> dev_frees{0}, dev_free_us{0}, sync_calls{0}, sync_us{0}, slow_syncs{0}, worst_sync_us{0};
    std::atomic<long long> dev_allocs{0}, dev_alloc_bytes{0}, dev_alloc_us{0}, host_allocs{0}, host_alloc_bytes{0}, host_alloc_us{0},
                           events{0}, streams{0}, first_launches{0}, first_launch_us{0}, resolved{0}, warm_us{0};
};
RuntimeCounters &rtc() { static RuntimeCounters *c = new RuntimeCounters(); return *c; }
double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}
double HelmFirstLaunch::now_ms() { return wall_ms(); }
int helm_kernel_register(const void *fn, const char *pretty) {
    KernelRegistry &r = kreg();
    std::lock_guard<std::mutex> lk(r.mu);
    KernelRec *k = new KernelRec(); k->fn = fn; k->pretty = pretty;
    r.recs.push_back(k);
    const int slot = (int)r.recs.size() - 1;
    if (slot < 2048) r.fast[slot].store(k);
    r.n.store(slot + 1);
    return slot;
}
bool helm_kernel_first_launch(int slot) {
    if (slot < 0 || slot >= 2048) return false;
    KernelRec *k = kreg().fast[slot].load(std::memory_order_relaxed);
    if (!k || k->launched.load(std::memory_order_relaxed)) return false;
    return !k->launched.exchange(true);
}
void helm_kernel_first_launch_done(int slot, double host_ms) {
    (void)slot;
    rtc().first_launches += 1; rtc().first_launch_us += (long long)(host_ms * 1e3);
    static const bool tr = getenv("HELM_LAUNCH_TRACE") && atoi(getenv("HELM_LAUNCH_TRACE"));
    if (tr) { KernelRec *k = kreg().fast[slot].load(); fprintf(stderr, "[helm first launch] %8.3f ms  %s\n", host_ms, k ? k->pretty : "?"); }
}
hipError_t helm_counted_malloc(void **p, size_t bytes) {
    const double t0 = wall_ms();
    const hipError_t e = (hipMalloc)(p, bytes);
    static const int tr = getenv("HELM_ALLOC_TRACE") ? atoi(getenv("HELM_ALLOC_TRACE")) : 0;
    if (tr >= 2) fprintf(stderr, "[helm alloc] hipMalloc %12zu B  %8.3f ms\n", bytes, wall_ms() - t0);
    rtc().dev_allocs += 1; rtc().dev_alloc_bytes += (long long)bytes; rtc().dev_alloc_us += (long long)((wall_ms() - t0) * 1e3);
    return e;
}
namespace {
struct SyncTimer {
    const char *what, *file; int line; double t0;
    SyncTimer(const char *w, const char *f, int l) : what(w), file(f), line(l), t0(wall_ms()) {}
    ~SyncTimer() {
        const double ms = wall_ms() - t0;
        rtc().sync_calls += 1; rtc().sync_us += (long long)(ms * 1e3);
        if (ms >= 10.0) { rtc().slow_syncs += 1; long long us = (long long)(ms * 1e3), prev = rtc().worst_sync_us.load(); while (us > prev && !rtc().worst_sync_us.compare_exchange_weak(prev, us)) {} }
        static const double thr = getenv("HELM_SYNC_TRACE") ? atof(getenv("HELM_SYNC_TRACE")) : 0.0;
        if (thr > 0 && ms >= thr) { const char *b = strrchr(file, '/'); fprintf(stderr, "[helm sync] %-22s %9.3f ms  %s:%d\n", what, ms, b ? b + 1 : file, line); }
    }
};
}
// A wait may poll before it blocks (helm_tuning.sync_spin_ms, default 0 = block at once).  Round 6 built this while hunting 60-80 ms stalls of the config-4
// gradient step in the belief that threads asleep on the runtime's interrupt were woken late; the stalls were the container's CPU quota freezing the process
// (zephyr_amd/problem.py, _norm2), which a polling thread makes worse, not better: under a quota every spinning thread is budget the launching threads
// do not have.  Kept as an option for hosts without one (the poll saves the 20-50 us wake-up of each of the ~30 waits of a work item).
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}
static double sync_spin_budget_ms() { return helm_tuning_now().sync_spin_ms; }
// helm_tuning.sync_sleep_us > 0: a wait polls with a sleep of that many microseconds between two looks instead of the runtime's own wait, which keeps a CPU busy
// for as long as it lasts -- 2.5 CPUs per process in the pipelined bench job (two threads that are nearly always waiting for the GPU).  For N processes on a node
// whose container grants fewer CPUs than 2.5 N (round 6: 16 for the one-GPU boxes): exhausting the quota freezes every thread of every process for the rest of
// the scheduler period (profiles/r06_cpu_quota_stall.txt).  Costs the sleep's granularity per wait (~50 us, ~30 waits per work item).
static bool sleep_wait(hipStream_t s, hipEvent_t e, int sleep_us) {
    struct timespec ts; ts.tv_sec = 0; ts.tv_nsec = (long)sleep_us * 1000L;
    for (;;) {
        const hipError_t q = e ? hipEventQuery(e) : hipStreamQuery(s);
        if (q == hipSuccess) return true;
        (void)hipGetLastError();
        if (q != hipErrorNotReady) return false;
        nanosleep(&ts, nullptr);
    }
}
hipError_t helm_timed_stream_sync(hipStream_t s, const char *file, int line) {
    SyncTimer t("hipStreamSynchronize", file, line);
    { const int su = helm_tuning_now().sync_sleep_us; if (su > 0 && sleep_wait(s, nullptr, su)) return hipSuccess; }
    const double budget = sync_spin_budget_ms();
    if (budget > 0) {
        const double t0 = wall_ms();
        for (;;) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
            (void)hipGetLastError();
            if (wall_ms() - t0 > budget) break;
            for (int i = 0; i < 64; ++i) cpu_relax();
        }
    }
    return (hipStreamSynchronize)(s);
}
hipError_t helm_timed_event_sync(hipEvent_t e, const char *file, int line) {
    SyncTimer t("hipEventSynchronize", file, line);
    { const int su = helm_tuning_now().sync_sleep_us; if (su > 0 && sleep_wait(nullptr, e, su)) return hipSuccess; }
    const double budget = sync_spin_budget_ms();
    if (budget > 0) {
        const double t0 = wall_ms();
        for (;;) {
            const hipError_t q = hipEventQuery(e);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
            (void)hipGetLastError();
            if (wall_ms() - t0 > budget) break;
            for (int i = 0; i < 64; ++i) cpu_relax();
        }
    }
    return (hipEventSynchronize)(e);
}
hipError_t helm_timed_device_sync(const char *file, int line) { SyncTimer t("hipDeviceSynchronize", file, line); return (hipDeviceSynchronize)(); }
hipError_t helm_timed_memcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, const char *file, int line) { SyncTimer t("hipMemcpy", file, line); return (hipMemcpy)(dst, src, bytes, kind); }
hipError_t helm_counted_free(void *p) {
    if (!p) return hipSuccess;
    const double t0 = wall_ms();
    const hipError_t e = (hipFree)(p);                 // (waits for every stream of the device)
    const double ms = wall_ms() - t0;
    rtc().dev_frees += 1; rtc().dev_free_us += (long long)(ms * 1e3);
    static const int tr = getenv("HELM_ALLOC_TRACE") ? atoi(getenv("HELM_ALLOC_TRACE")) : 0;
    if (tr >= 2) fprintf(stderr, "[helm alloc] hipFree   %p  %8.3f ms\n", p, ms);
    return e;
}
hipError_t helm_counted_host_malloc(void **p, size_t bytes, unsigned flags) {
    const double t0 = wall_ms();
    const hipError_t e = (hipHostMalloc)(p, bytes, flags);
    rtc().host_allocs += 1; rtc().host_alloc_bytes += (long long)bytes; rtc().host_alloc_us += (long long)((wall_ms() - t0) * 1e3);
    return e;
}
hipError_t helm_counted_event_create(hipEvent_t *e, unsigned flags) { rtc().events += 1; return flags ? (hipEventCreateWithFlags)(e, flags) : (hipEventCreate)(e); }
hipError_t helm_counted_stream_create(hipStream_t *s, unsigned flags, int prio, bool with_prio) {
    rtc().streams += 1;
    return with_prio ? (hipStreamCreateWithPriority)(s, flags, prio) : (hipStreamCreateWithFlags)(s, flags);
}
extern "C" int helm_debug_runtime_stats(int reset, helm_runtime_stats *out) {
    RuntimeCounters &c = rtc();
    if (out) {
        out->dev_allocs = c.dev_allocs.load(); out->dev_alloc_bytes = (double)c.dev_alloc_bytes.load(); out->dev_alloc_ms = c.dev_alloc_us.load() * 1e-3;
        out->host_allocs = c.host_allocs.load(); out->host_alloc_bytes = (double)c.host_alloc_bytes.load(); out->host_alloc_ms = c.host_alloc_us.load() * 1e-3;
        out->events_created = c.events.load(); out->streams_created = c.streams.load();
        out->first_launches = c.first_launches.load(); out->first_launch_ms = c.first_launch_us.load() * 1e-3;
        out->kernels_registered = kreg().n.load(); out->kernels_resolved = c.resolved.load(); out->warm_ms = c.warm_us.load() * 1e-3;
        out->dev_frees = c.dev_frees.load(); out->dev_free_ms = c.dev_free_us.load() * 1e-3;
        out->sync_calls = c.sync_calls.load(); out->sync_ms = c.sync_us.load() * 1e-3; out->slow_syncs = c.slow_syncs.load(); out->worst_sync_ms = c.worst_sync_us.load() * 1e-3;
    }
    if (reset) { c.dev_allocs = 0; c.dev_alloc_bytes = 0; c.dev_alloc_us = 0; c.host_allocs = 0; c.host_alloc_bytes = 0; c.host_alloc_us = 0;
                 c.events = 0; c.streams = 0; c.first_launches = 0; c.first_launch_us = 0; c.dev_frees = 0; c.dev_free_us = 0; c.sync_calls = 0; c.sync_us = 0; c.slow_syncs = 0; c.worst_sync_us = 0; }
    return HELM_OK;
}
// (diagnostic) a thread of the library that does nothing but read the clock: the longest interval between two readings while it ran.  Tells a stall of the
// PROCESS (every thread stops: the watcher sees it too) from a stall of the GPU or of the runtime (the watcher keeps running).
namespace { std::atomic<bool> g_watch_on{false}; std::atomic<long long> g_watch_worst_us{0}, g_watch_gaps{0}; std::thread *g_watch_thread = nullptr; }
extern "C" int helm_debug_stall_watch(int start, double *worst_gap_ms, long long *gaps_over_5ms) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (start) {
        if (g_watch_thread) return HELM_OK;
        g_watch_worst_us = 0; g_watch_gaps = 0; g_watch_on = true;
        g_watch_thread = new std::thread([] {
            double last = wall_ms();
            while (g_watch_on.load(std::memory_order_relaxed)) {
                const double now = wall_ms(), gap = now - last;
                last = now;
                if (gap > 5.0) g_watch_gaps += 1;
                long long us = (long long)(gap * 1e3), prev = g_watch_worst_us.load();
                while (us > prev && !g_watch_worst_us.compare_exchange_weak(prev, us)) {}
            }
        });
        return HELM_OK;
    }
    if (g_watch_thread) { g_watch_on = false; g_watch_thread->join(); delete g_watch_thread; g_watch_thread = nullptr; }
    if (worst_gap_ms) *worst_gap_ms = g_watch_worst_us.load() * 1e-3;
    if (gaps_over_5ms) *gaps_over_5ms = g_watch_gaps.load();
    return HELM_OK;
}
void helm_pool_slab_reserve(int device);
// Resolve every kernel of the library on `device` (code objects loaded, dispatch records built) without launching anything.  Idempotent; runs by itself
// when the first operator of a device is created (HELM_WARM=0 leaves it to the caller).
extern "C" int helm_warm(int device) {
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); helm_set_error(nullptr, "helm_warm: hipSetDevice failed"); return HELM_ERR_DEVICE; }
    static std::mutex mu; static std::map<int, int> done;
    std::lock_guard<std::mutex> lk(mu);
    KernelRegistry &r = kreg();
    const int n = std::min(r.n.load(), 2048);
    int &upto = done[device];
    const double t0 = wall_ms();
    for (int i = upto; i < n; ++i) {
        hipFuncAttributes at;
        if (hipFuncGetAttributes(&at, r.fast[i].load()->fn) == hipSuccess) rtc().resolved += 1; else (void)hipGetLastError();
    }
    upto = n;
    helm_pool_slab_reserve(device);
    rtc().warm_us += (long long)((wall_ms() - t0) * 1e3);
    return n;
}

// ---- tuning (include/helm.h: helm_tuning) ------------------------------------------------------------------------------------------------
namespace {
std::mutex g_tune_mu;
bool g_tune_set = false;
helm_tuning g_tune_user;
int tune_i(const char *name, int d) { const char *v = getenv(name); return v ? atoi(v) : d; }
double tune_d(const char *name, double d) { const char *v = getenv(name); return v ? atof(v) : d; }
}
// the limits every source of the options goes through (environment, helm_set_tuning): values outside them would switch a path off by accident
// (nd_plans = 0, nd_ws_gb = 0: batch forced to 1) rather than by intent
static void tuning_clamp(helm_tuning &t) {
    t.nd_leaf = std::max(2, t.nd_leaf);
    if (!(t.nd_ws_gb > 0)) t.nd_ws_gb = 32.0;
    t.nd_stable_safety = std::max(1.0, t.nd_stable_safety);
    if (!(t.nd_stable_thr >= 0)) t.nd_stable_thr = 0.0;
    t.nd_fused_leaf_min = std::max(1, t.nd_fused_leaf_min);
    t.nd_gjstep_min = std::max(64, t.nd_gjstep_min);
    t.nd_plans = std::max(1, t.nd_plans);
    t.ws_slots = std::min(4, std::max(1, t.ws_slots));
    t.pf_prio = t.pf_prio > 0 ? 1 : (t.pf_prio < 0 ? -1 : 0);
    if (!(t.mg3_omega > 0) || t.mg3_omega > 2.0) t.mg3_omega = 0.9;
    if (!(t.sync_spin_ms >= 0)) t.sync_spin_ms = 0.0;
    t.sync_spin_ms = std::min(t.sync_spin_ms, 60000.0);
    t.sync_sleep_us = std::min(100000, std::max(0, t.sync_sleep_us));
}
static helm_tuning tuning_from_env() {
    helm_tuning t;
    t.nd_leaf = tune_i("HELM_ND_LEAF", 8);
    t.nd_ws_gb = tune_d("HELM_ND_WS_GB", 32.0);
    t.nd_sparse_rhs = tune_i("HELM_ND_SPARSE_RHS", 1);
    t.nd_stable = tune_i("HELM_ND_STABLE", 1);
    t.nd_stable_thr = tune_d("HELM_ND_STABLE_THR", 0.0);
    t.nd_stable_safety = tune_d("HELM_ND_STABLE_SAFETY", 8.0);
    t.nd_fused_leaf = tune_i("HELM_ND_FUSEDLEAF", 1);
    t.nd_fused_leaf_min = tune_i("HELM_ND_FUSEDLEAF_MIN", 2048);
    t.nd_gjstep = tune_i("HELM_ND_GJSTEP", 1);
    t.nd_gjstep_min = tune_i("HELM_ND_GJSTEP_MIN", 128);
    t.nd_overlap = tune_i("HELM_ND_OVERLAP_NM", 1);
    t.nd_xcd_map = tune_i("HELM_ND_XCDMAP", 2);
    t.nd_plans = tune_i("HELM_ND_PLANS", 6);
    t.nd_direct_out = tune_i("HELM_ND_DIRECT_OUT", 1);
    t.nd_leaf_idle = tune_i("HELM_ND_LEAF_IDLE", 1);
    t.nd_many = tune_i("HELM_ND_MANY", 1);
    t.auto_direct = tune_i("HELM_AUTO_DIRECT", 1);
    t.auto_mg3 = tune_i("HELM_AUTO_MG3", 1);
    t.prof_ext = tune_i("HELM_PROF_EXT", 1);
    t.ws_slots = tune_i("HELM_WS_SLOTS", 3);
    t.pf_prio = tune_i("HELM_PF_PRIO", 1);
    t.mg3_keep = tune_i("HELM_MG3_KEEP", 1);
    t.mg3_keep_levels = tune_i("HELM_MG3_KEEP_LEVELS", -1);
    t.mg3_galerkin = tune_i("HELM_MG3_GALERKIN", 1);
    t.mg3_depth_model = tune_i("HELM_MG3_DEPTH_MODEL", 1);
    t.mg3_bt_f32 = tune_i("HELM_MG3_BT_F32", 1);
    t.mg3_otf = tune_i("HELM_MG3_OTF", 1);
    t.mg3_f32 = tune_i("HELM_MG3_F32", 1);
    t.mg3_omega = tune_d("HELM_MG3_OMEGA", 0.9);
    t.sync_spin_ms = tune_d("HELM_SYNC_SPIN_MS", 0.0);
    t.sync_sleep_us = tune_i("HELM_SYNC_SLEEP_US", 0);
    tuning_clamp(t);
    return t;
}
// The options in force.  helm_set_tuning's structure wins; otherwise defaults + environment, re-read when the HELM_* entries of the environment have changed
// (a test may flip a variable between two calls): the passes ask once per tree level from worker threads, and 27 getenv calls each time raced against exactly
// that setenv.  The environment is compared by a fingerprint of its HELM_* entries, at API entry only (helm_tuning_refresh).
extern char **environ;
static unsigned long long env_fingerprint() {
    unsigned long long h = 1469598103934665603ull;
    for (char **e = environ; e && *e; ++e) {
        const char *s = *e;
        if (s[0] != 'H' || s[1] != 'E' || s[2] != 'L' || s[3] != 'M' || s[4] != '_') continue;
        for (; *s; ++s) { h ^= (unsigned char)*s; h *= 1099511628211ull; }
        h ^= 0xff; h *= 1099511628211ull;
    }
    return h;
}
static bool g_tune_have = false; static unsigned long long g_tune_fp = 0; static helm_tuning g_tune_cached;
// called at the entry of the API calls that start work (create, assemble, prefactor, solve, apply, get_tuning): the environment is looked at THERE, by the
// calling thread, and nowhere below -- a caller that changes a HELM_* variable does so between two calls, as the header says
void helm_tuning_refresh() {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    const unsigned long long now = env_fingerprint();
    if (!g_tune_have || now != g_tune_fp) { g_tune_cached = tuning_from_env(); g_tune_fp = now; g_tune_have = true; }
}
helm_tuning helm_tuning_now() {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (g_tune_set) return g_tune_user;
    if (!g_tune_have) { g_tune_cached = tuning_from_env(); g_tune_fp = env_fingerprint(); g_tune_have = true; }
    return g_tune_cached;
}
extern "C" int helm_get_tuning(helm_tuning *out) { if (!out) return HELM_ERR_ARG; helm_tuning_refresh(); *out = helm_tuning_now(); return HELM_OK; }
extern "C" int helm_set_tuning(const helm_tuning *t) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (t) { g_tune_user = *t; tuning_clamp(g_tune_user); g_tune_set = true; } else g_tune_set = false;
    return HELM_OK;
}

extern "C" int helm_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { helm_set_error(nullptr, hipGetErrorString(e)); return HELM_ERR_DEVICE; }
    return n;
}

// creation failures release everything through helm_destroy (stream, model arrays, planes, the live-handle count)
#define HIP_TRY_NULL(call) do { hipError_t _e = (call); if (_e != hipSuccess) { char _b[256]; \
    snprintf(_b, sizeof(_b), "%s failed: %s", #call, hipGetErrorString(_e)); helm_destroy(op); helm_set_error(nullptr, _b); return nullptr; } } while (0)

// Scratch of the direct path (fronts while factoring, front vectors while solving) is tens of GB at the bench size and
// is only needed during a call, so all handles of a process share one buffer; a handle that finds it taken (another
// host thread is inside a solve) falls back to its own.
// r4: the table is PER DEVICE (HELM_WS_SLOTS slots each, default 3, at most 4).  Round 3 kept one table of 3-4 slots for the whole process,
// tagged with a device: under the in-process dispatcher on an 8-GPU node the first three or four GPUs to ask got them and every other GPU
// allocated its ~30 GB beside running kernels on every solve (the 0.7-1.5 s stalls helm_reserve exists to remove).  A lease is
// (device, slot) packed as device * WS_SLOTS_MAX + slot.
#define WS_SLOTS_MAX 4
struct WsSlot { void *ptr = nullptr; size_t bytes = 0; bool busy = false; };
struct WsDevice { WsSlot slot[WS_SLOTS_MAX]; };
struct SharedWs { std::mutex mu; std::map<int, WsDevice> dev; };
static SharedWs g_shared_ws;
static int shared_ws_slots() { const int n = helm_tuning_now().ws_slots; return n < 1 ? 1 : (n > WS_SLOTS_MAX ? WS_SLOTS_MAX : n); }
static int g_live_handles = 0;     // guarded by g_shared_ws.mu
static std::map<int, int> g_live_per_device;      // guarded by g_shared_ws.mu

// idle device buffers by (device, size); `held` and the cap are per device (r4: one sum over all GPUs hit a single device's cap with the second GPU's buffers)
struct DevPool { std::mutex mu; std::multimap<std::pair<int, size_t>, void *> idle; std::map<int, size_t> held; };
static DevPool g_pool;
static std::map<int, std::vector<hipEvent_t>> g_idle_events;      // per device, guarded by g_pool.mu

int helm_events_grow(helm_op *op, int n) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    for (int i = 0; i < n; ++i) {
        hipEvent_t e;
        std::vector<hipEvent_t> &idle = g_idle_events[op->device];
        if (!idle.empty()) { e = idle.back(); idle.pop_back(); }
        else if (hipEventCreate(&e) != hipSuccess) return -1;
        op->ev_pool.push_back(e);
    }
    return 0;
}
// (small buffers too: hipFree waits for every stream of the device, which would stall a host thread that prepares the next operator
// while another one is solving -- the per-operator scratch of a few KB goes through the pool like the GB-sized buffers)
static const size_t kPoolMinBytes = (size_t)64;
// What the pool may hold idle: half of the device's memory (HELM_POOL_GB overrides; buffers below 1 MB are always kept: their hipFree
// would be a device synchronisation for nothing).  A 16-frequency job at 1024^2 hands back ~70 GB of
// factors when its operators go; with a 64-GB cap the overflow went to hipFree and the next job's hipMalloc calls -- issued while other
// threads had kernels and copies in flight -- took 1.2-1.5 s EACH (HELM_ALLOC_TRACE=1 shows them).
static size_t pool_cap_bytes(int device) {          // (call with g_pool.mu held)
    static std::map<int, size_t> caps;
    auto it = caps.find(device);
    if (it != caps.end()) return it->second;
    size_t cap = (size_t)64 << 30;
    if (const char *e = getenv("HELM_POOL_GB")) cap = (size_t)(atof(e) * 1e9);
    else {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) cap = prop.totalGlobalMem / 2;     // (r6: half, not three quarters -- the caller's own allocator (torch) lives on the same device and cannot make this pool give anything back)
        else (void)hipGetLastError();
    }
    caps[device] = cap;
    return cap;
}

// pinned host buffers (per-handle scalar records) and HIP streams are recycled the same way: a job creates one operator per frequency
// r4: the idle pool is capped by BYTES (HELM_HOSTPOOL_GB, default a quarter of the host's memory, at most 96 GB) and helm_trim / helm_host_trim
// give it back: results of 1 MB or more go through it (4.3 GB per frequency of the 2-D job), and with only an entry-count cap a long-lived
// process that changed nsrc or the split sizes could accumulate hundreds of GB of locked memory in size classes it never used again
struct HostPool { std::mutex mu; std::multimap<size_t, void *> idle; size_t held = 0; };
static HostPool g_hostpool;
static size_t hostpool_cap_bytes() {
    static const size_t cap = [] {
        if (const char *e = getenv("HELM_HOSTPOOL_GB")) return (size_t)(atof(e) * 1e9);
        const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
        const size_t ram = pages > 0 && psz > 0 ? (size_t)pages * (size_t)psz : (size_t)64 << 30;
        return std::min(ram / 4, (size_t)96 << 30);
    }();
    return cap;
}
// HELM_ALLOC_TRACE=1: every allocator call that reaches the driver and takes more than a millisecond is reported on stderr
// (always counted -- helm_debug_alloc_stats -- so that a test can assert that a job issued none after its bookings)
static std::atomic<long long> g_alloc_slow{0};
static std::atomic<long long> g_alloc_worst_us{0};
struct AllocTrace {
    const char *what; size_t bytes; std::chrono::steady_clock::time_point t0; bool on;
    AllocTrace(const char *w, size_t b) : what(w), bytes(b), t0(std::chrono::steady_clock::now()) { static const bool e = getenv("HELM_ALLOC_TRACE") && atoi(getenv("HELM_ALLOC_TRACE")); on = e; }
    ~AllocTrace() {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 1.0) {
            g_alloc_slow += 1;
            long long us = (long long)(ms * 1e3), prev = g_alloc_worst_us.load();
            while (us > prev && !g_alloc_worst_us.compare_exchange_weak(prev, us)) {}
            if (on) fprintf(stderr, "[helm alloc] %-14s %8.3f GB %9.1f ms\n", what, bytes * 1e-9, ms);
        }
    }
};
// allocator calls (hipMalloc / hipFree / hipHostMalloc / pool flushes) that reached the driver and took more than 1 ms since the last reset
extern "C" int helm_debug_alloc_stats(int reset, long long *slow_calls, double *worst_ms) {
    if (slow_calls) *slow_calls = g_alloc_slow.load();
    if (worst_ms) *worst_ms = g_alloc_worst_us.load() * 1e-3;
    if (reset) { g_alloc_slow = 0; g_alloc_worst_us = 0; }
    return HELM_OK;
}

void *helm_hostpool_alloc(size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(g_hostpool.mu);
        auto it = g_hostpool.idle.find(bytes);
        if (it != g_hostpool.idle.end()) { void *p = it->second; g_hostpool.idle.erase(it); g_hostpool.held -= bytes; return p; }
    }
    void *p = nullptr;
    AllocTrace tr("hipHostMalloc", bytes);
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void helm_hostpool_free(void *p, size_t bytes) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_hostpool.mu);
        if (g_hostpool.idle.size() < 1024 && (g_hostpool.held + bytes <= hostpool_cap_bytes() || bytes < ((size_t)1 << 20))) {
            g_hostpool.idle.insert(std::make_pair(bytes, p)); g_hostpool.held += bytes;
            return;
        }
    }
    AllocTrace tr("hipHostFree", bytes);
    hipHostFree(p);
}
// pinned host memory the library holds idle goes back to the system
extern "C" int helm_host_trim(void) {
    helm_tuning_refresh();
    std::lock_guard<std::mutex> lk(g_hostpool.mu);
    for (auto &kv : g_hostpool.idle) hipHostFree(kv.second);
    g_hostpool.idle.clear(); g_hostpool.held = 0;
    return HELM_OK;
}
struct StreamPool { std::mutex mu; std::multimap<std::pair<int, int>, hipStream_t> idle; };     // (device, priority class) -> idle streams
static StreamPool g_streams;
// prio: 0 normal, 1 highest, -1 lowest priority the device offers; the stream comes back idle (synchronised by helm_stream_release)
hipStream_t helm_stream_acquire(int device, int prio) {
    {
        std::lock_guard<std::mutex> lk(g_streams.mu);
        auto it = g_streams.idle.find(std::make_pair(device, prio));
        if (it != g_streams.idle.end()) { hipStream_t s = it->second; g_streams.idle.erase(it); return s; }
    }
    hipStream_t s = nullptr;
    if (prio == 0) { if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr; return s; }
    int plo = 0, phi = 0;
    (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio > 0 ? phi : plo) != hipSuccess) return nullptr;
    return s;
}
void helm_stream_release(int device, int prio, hipStream_t s) {
    if (!s) return;
    hipStreamSynchronize(s);
    std::lock_guard<std::mutex> lk(g_streams.mu);
    if (g_streams.idle.size() < 64) { g_streams.idle.insert(std::make_pair(std::make_pair(device, prio), s)); return; }
    hipStreamDestroy(s);
}

// idle bytes of one device (what hipMemGetInfo's "free" figure does not count although an allocation can have them: the budgets of mg3d.hip add it)
size_t helm_pool_idle_bytes(int device) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.held.find(device);
    return it == g_pool.held.end() ? 0 : it->second;
}
static void pool_forget(void *p);       // (g_pool.mu held) the buffer has gone back to the driver
static std::map<void *, bool> g_carved;  // blocks that are pieces of a slab (see slab_carve; guarded by g_pool.mu)
struct PoolClassStat { int in_use = 0, high = 0, total = 0; };
static std::map<std::pair<int, size_t>, PoolClassStat> g_pool_stats;         // (device, capacity) of big buffers; guarded by g_pool.mu        // (see pool_top_up)
// give this device's idle buffers back to the driver (the current device must be `device`)
static void pool_flush_device(int device) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    AllocTrace trf("pool flush", g_pool.held[device]);
    size_t kept = 0;
    for (auto it = g_pool.idle.lower_bound(std::make_pair(device, (size_t)0)); it != g_pool.idle.end() && it->first.first == device; ) {
        if (g_carved.count(it->second)) { kept += it->first.second; ++it; continue; }       // (a piece of a slab: stays idle)
        hipFree(it->second);
        pool_forget(it->second);
        it = g_pool.idle.erase(it);
    }
    g_pool.held[device] = kept;
    for (auto is = g_pool_stats.begin(); is != g_pool_stats.end(); ) { if (is->first.first == device) { is->second.total = is->second.in_use; is->second.high = is->second.in_use; } ++is; }
}
// hipMalloc that, under memory pressure, empties the device's idle pool and tries once more -- for every allocation of the library that does
// not go through the size-keyed pool itself (scratch slots, temporaries of the host-buffer entry points, plans)
hipError_t helm_malloc_retry(int device, void **p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    pool_flush_device(device);
    e = hipMalloc(p, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; }
    return e;
}
// r6: a request is served by the smallest idle buffer of the device whose capacity is at least the request and at most twice it (+ 1 MB; from 64 MB up: at most
// one size class more, so that the GB-sized factor and wavefield buffers do not take each other's places): the pool used to be
// keyed by the exact size, and sizes that follow the operator -- how many ill-conditioned fronts a frequency has, how many right-hand sides take a
// refinement pass -- missed it at every new frequency: 12 hipMalloc calls inside the timed region of the bench job after a five-item warm-up.  New buffers are
// allocated in size classes (steps of 1/8 of the power of two below, at least 4 KB), and the pool remembers every buffer's capacity, so a buffer
// goes back under what it can hold, not under what it was asked for.
static std::map<void *, size_t> g_pool_capacity;        // every live buffer that came out of helm_pool_alloc: what it can hold (guarded by g_pool.mu)
// Spares beyond the high-water mark (big buffers, 64 MB .. 16 GB): the pool of a class holds what the busiest moment so far needed, and a pipelined job's busiest
// moment is a matter of thread timing -- a job that got by with three factor buffers in its first five items asked for a fourth in its next twenty (round 6: 3 to 5 GB
// of hipMalloc inside the bench's timed region in one run of three; 0.6 ms on one box, 122 ms on another = the stall that cost round 5's driver run a fifth of its
// headline; with pairs of operators factored together a five-item warm-up sees one or two sets of pair buffers alive and the job needs three).  When the last
// operator of a device is destroyed -- every buffer idle, nobody waiting -- each such class whose busiest moment used EVERY buffer it had is topped up to
// high-water + HELM_POOL_SPARE (default 2); a class that kept one unused has its headroom and is left alone (so a job's last destroy adds nothing once the
// pool has settled: a top-up is a hipMalloc too, and the end of one timed pass is the eve of the next).
static const size_t kSpareMin = (size_t)64 << 20, kSpareMax = (size_t)16 << 30;
static void pool_forget(void *p) {
    auto it = g_pool_capacity.find(p);
    if (it == g_pool_capacity.end()) return;
    g_pool_capacity.erase(it);
}
// Small buffers (size class up to 16 MB: per-operator flags, estimates, split-K partials, the pivoted-LU storage of ill-conditioned fronts ...) come out of
// slabs of 512 MB, one hipMalloc each, carved by a bump pointer and recycled through the idle table like every other buffer.  Their sizes follow the operator
// -- how many fronts a frequency has flagged, which products split their inner dimension -- so a job met half a dozen new ones per pass over its frequencies
// however long the warm-up (round 6: 6 hipMalloc calls, 16 MB, in the timed region of every bench run).  A carved block is never handed back to the driver by
// itself; slabs live as long as the process (helm_trim keeps them: 512 MB each, a handful at most).
static const size_t kSlabBytes = (size_t)512 << 20, kSlabMaxBlock = (size_t)16 << 20;
struct Slab { char *base = nullptr; size_t used = 0; };
static std::map<int, std::vector<Slab>> g_slabs;                   // guarded by g_pool.mu
static bool slab_add(int device) {                                 // (g_pool.mu NOT held: the driver call may take milliseconds)
    void *b = nullptr;
    AllocTrace tr("pool slab", kSlabBytes);
    if (helm_malloc_retry(device, &b, kSlabBytes) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    Slab sl; sl.base = (char *)b;
    g_slabs[device].push_back(sl);
    return true;
}
static void *slab_carve(int device, size_t cap) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        {
            std::lock_guard<std::mutex> lk(g_pool.mu);
            std::vector<Slab> &v = g_slabs[device];
            if (!v.empty()) {
                Slab &sl = v.back();
                const size_t off = (sl.used + 255) & ~(size_t)255;
                if (off + cap <= kSlabBytes) { sl.used = off + cap; void *p = sl.base + off; g_pool_capacity[p] = cap; g_carved[p] = true; return p; }
            }
        }
        if (!slab_add(device)) return nullptr;
    }
    return nullptr;
}
// the first slab of a device, brought into being by helm_warm (i.e. when the first operator of the device is created), not by whichever solve first misses the pool
static void slab_reserve(int device) {
    { std::lock_guard<std::mutex> lk(g_pool.mu); if (!g_slabs[device].empty()) return; }
    (void)slab_add(device);
}
static size_t pool_size_class(size_t bytes) {
    if (bytes <= 4096) return 4096;
    int top = 63 - __builtin_clzll((unsigned long long)(bytes - 1));      // bytes - 1 in [2^top, 2^(top+1))
    const int sh = top - 3;
    return (((bytes - 1) >> sh) + 1) << sh;
}
void *helm_pool_alloc(int device, size_t bytes) {
    if (bytes == 0) bytes = 1;
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        // from 64 MB up a buffer of the request's own size class is preferred (the GB-sized factor, scratch and wavefield buffers keep to their own kind);
        // failing that -- and for small requests from the start -- the smallest idle buffer that holds the request and is at most twice its size (+ 1 MB).
        // (Measured, round 6: with the own-class rule alone the bench job allocated 5.2 GB inside its timed region in every run -- 0.6 ms on one box, 122 ms
        // on another, which is the kind of stall that cost round 5's driver run a fifth of its headline; with the fall-back: nothing above 8 MB.)
        static const double slack = getenv("HELM_POOL_SLACK") ? std::max(1.0, atof(getenv("HELM_POOL_SLACK"))) : 2.0;      // (diagnostic: 1 = a request's own size class only)
        static const int ptrace = getenv("HELM_ALLOC_TRACE") ? atoi(getenv("HELM_ALLOC_TRACE")) : 0;
        auto it = g_pool.idle.end();
        if (bytes >= ((size_t)64 << 20)) it = g_pool.idle.find(std::make_pair(device, pool_size_class(bytes)));
        if (it == g_pool.idle.end()) {
            it = g_pool.idle.lower_bound(std::make_pair(device, bytes));
            if (it != g_pool.idle.end() && (it->first.first != device || (double)it->first.second > slack * (double)bytes + (double)((size_t)1 << 20))) it = g_pool.idle.end();
        }
        if (ptrace >= 3 && bytes >= ((size_t)64 << 20))
            fprintf(stderr, "[helm pool] request %9.1f MB (class %9.1f MB): %s %9.1f MB\n", bytes / 1e6, pool_size_class(bytes) / 1e6, it != g_pool.idle.end() ? "served by an idle buffer of" : "MISS, allocating",
                    (it != g_pool.idle.end() ? it->first.second : pool_size_class(bytes)) / 1e6);
        if (it != g_pool.idle.end()) {
            void *p = it->second; g_pool.held[device] -= it->first.second;
            if (it->first.second >= kSpareMin) { PoolClassStat &cs = g_pool_stats[std::make_pair(device, it->first.second)]; cs.in_use += 1; cs.high = std::max(cs.high, cs.in_use); }
            g_pool.idle.erase(it); return p;
        }
    }
    void *p = nullptr;
    const size_t cap = pool_size_class(bytes);
    if (cap <= kSlabMaxBlock) {                 // small buffers are carved out of a slab: no driver call however many new sizes a frequency brings
        p = slab_carve(device, cap);
        if (p) return p;
    }
    AllocTrace tr("pool hipMalloc", cap);
    if (helm_malloc_retry(device, &p, cap) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_pool.mu);
    g_pool_capacity[p] = cap;
    if (cap >= kSpareMin) { PoolClassStat &cs = g_pool_stats[std::make_pair(device, cap)]; cs.total += 1; cs.in_use += 1; cs.high = std::max(cs.high, cs.in_use); }
    return p;
}
// (see PoolClassStat) called with no operator of the device alive
static void pool_top_up(int device, int spare) {
    std::vector<size_t> want;
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        const size_t cap = pool_cap_bytes(device);
        size_t held = g_pool.held[device];
        for (auto &kv : g_pool_stats) {
            if (kv.first.first != device || kv.first.second > kSpareMax) continue;
            PoolClassStat &cs = kv.second;
            if (cs.high < cs.total) continue;                      // the busiest moment left a buffer of this class unused: enough headroom
            for (int k = cs.total; k < cs.high + spare && cs.high > 0; ++k) { if (held + kv.first.second > cap) break; want.push_back(kv.first.second); held += kv.first.second; }
        }
    }
    for (size_t bytes : want) {
        void *p = nullptr;
        AllocTrace tr("pool spare", bytes);
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }        // (a spare is a convenience: no flush-and-retry for it)
        std::lock_guard<std::mutex> lk(g_pool.mu);
        g_pool_capacity[p] = bytes;
        g_pool_stats[std::make_pair(device, bytes)].total += 1;
        g_pool.idle.insert(std::make_pair(std::make_pair(device, bytes), p)); g_pool.held[device] += bytes;
    }
}
void helm_pool_slab_reserve(int device) { slab_reserve(device); }
void helm_pool_free(int device, void *p, size_t bytes) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        auto ic = g_pool_capacity.find(p);
        if (ic != g_pool_capacity.end()) bytes = ic->second;            // (a buffer that did not come from the pool is taken in under the size the caller states)
        else g_pool_capacity[p] = bytes;
        if (bytes >= kSpareMin) { auto is = g_pool_stats.find(std::make_pair(device, bytes)); if (is != g_pool_stats.end() && is->second.in_use > 0) is->second.in_use -= 1; }
        const size_t cap = pool_cap_bytes(device);
        size_t &held = g_pool.held[device];
        if (g_carved.count(p) || (bytes >= kPoolMinBytes && (held + bytes <= cap || bytes < ((size_t)1 << 20)))) {
            g_pool.idle.insert(std::make_pair(std::make_pair(device, bytes), p)); held += bytes;
            return;
        }
        g_pool_capacity.erase(p);
        if (bytes >= kSpareMin) { auto is = g_pool_stats.find(std::make_pair(device, bytes)); if (is != g_pool_stats.end() && is->second.total > 0) is->second.total -= 1; }
    }
    AllocTrace tr("pool hipFree", bytes);
    hipFree(p);
}


// priority class of the factor stream of helm_prefactor (HELM_PF_PRIO: 1 highest, 0 normal, -1 lowest)
static int pf_prio() { return helm_tuning_now().pf_prio; }

static helm_op *create_common(helm_op *op);

extern "C" helm_op *helm_create3d(int device, int nz, int ny, int nx, double dx, double dy, double dz, int nPML) {
    helm_tuning_refresh();
    if (nz < 3 || ny < 3 || nx < 3) { helm_set_error(nullptr, "nz, ny and nx must be >= 3"); return nullptr; }
    if (!(dx > 0) || !(dy > 0) || !(dz > 0)) { helm_set_error(nullptr, "grid spacings must be positive"); return nullptr; }
    helm_op *op = new helm_op();
    op->device = device; op->variant = HELM_3D; op->nz = nz; op->ny = ny; op->nx = nx; op->N = (long long)nz * ny * nx;
    op->dx = dx; op->dy = dy; op->dz = dz; op->nPML = nPML;
    op->nblocks = 1; op->nplanes = 27; op->centre = 13;
    return create_common(op);
}

extern "C" helm_op *helm_create(int device, int variant, int nz, int nx, double dx, double dz, int nPML, const int *freeSurf) {
    helm_tuning_refresh();
    if (nz < 3 || nx < 3) { helm_set_error(nullptr, "nz and nx must be >= 3"); return nullptr; }
    if (variant != HELM_MINIZEPHYR && variant != HELM_EURUS) { helm_set_error(nullptr, "unknown variant"); return nullptr; }
    if (!(dx > 0) || !(dz > 0)) { helm_set_error(nullptr, "dx and dz must be positive"); return nullptr; }
    helm_op *op = new helm_op();
    op->device = device; op->variant = variant; op->nz = nz; op->nx = nx; op->N = (long long)nz * nx;
    op->dx = dx; op->dz = dz; op->nPML = nPML;
    if (freeSurf) for (int i = 0; i < 4; ++i) op->fs[i] = freeSurf[i] ? 1 : 0;
    op->nblocks = variant == HELM_EURUS ? 4 : 1;
    if (nPML < 0) { op->block0_only = true; op->nPML = -nPML; op->nblocks = 1; }   // internal: preconditioner level
    return create_common(op);
}

static helm_op *create_common(helm_op *op) {
    const int device = op->device;
    { std::lock_guard<std::mutex> lk(g_shared_ws.mu); g_live_handles += 1; g_live_per_device[device] += 1; }      // helm_destroy takes it back on every exit
    HIP_TRY_NULL(hipSetDevice(device));
    {   // first operator of this device in the process: resolve the library's kernels now, not one by one inside the first solves of each kind
        static std::mutex wmu; static std::map<int, bool> warmed;
        bool need = false;
        { std::lock_guard<std::mutex> lk(wmu); if (!warmed[device]) { warmed[device] = true; need = true; } }
        if (need && tune_i("HELM_WARM", 1)) (void)helm_warm(device);
    }
    op->stream = helm_stream_acquire(device, 0);
    if (!op->stream) { helm_destroy(op); helm_set_error(nullptr, "hipStreamCreate failed"); return nullptr; }
    op->own_stream = true;
    const size_t N = (size_t)op->N;
    op->Nv = op->N;
    // model arrays come from the size-keyed pool as well: hipMalloc / hipFree of a few MB per operator is a device synchronisation each
    op->d_c = (cplx *)helm_pool_alloc(device, N * sizeof(cplx));
    op->d_rho = (double *)helm_pool_alloc(device, N * sizeof(double));
    if (!op->d_c || !op->d_rho) { helm_destroy(op); helm_set_error(nullptr, "hipMalloc of the model arrays failed"); return nullptr; }
    op->d_C = (cplx *)helm_pool_alloc(device, (size_t)op->nblocks * op->nplanes * N * sizeof(cplx));
    if (!op->d_C) { helm_destroy(op); helm_set_error(nullptr, "hipMalloc of the coefficient planes failed"); return nullptr; }
    return op;
}

extern "C" void helm_destroy(helm_op *op) {
    if (!op) return;
    hipSetDevice(op->device);
    helm_pf_retire(op);
    if (op->stream) hipStreamSynchronize(op->stream);
    helm_pool_free(op->device, op->d_c, (size_t)op->N * sizeof(cplx)); helm_pool_free(op->device, op->d_rho, (size_t)op->N * sizeof(double));
    helm_pool_free(op->device, op->d_theta, (size_t)op->N * sizeof(double)); helm_pool_free(op->device, op->d_eps, (size_t)op->N * sizeof(double)); helm_pool_free(op->device, op->d_delta, (size_t)op->N * sizeof(double));
    helm_pool_free(op->device, op->d_K3, (size_t)op->N * sizeof(cplx)); helm_pool_free(op->device, op->d_b3, (size_t)op->N * sizeof(double));
    helm_pool_free(op->device, op->d_L3, op->l3_elems * sizeof(cplx));
    {
        const size_t pb = (size_t)op->nblocks * op->nplanes * (size_t)op->N * sizeof(cplx);
        helm_pool_free(op->device, op->d_C, pb); helm_pool_free(op->device, op->d_Cs, pb);
        helm_pool_free(op->device, op->d_dinv, (size_t)op->nblocks * (size_t)op->N * sizeof(cplx));
    }
    hipFree(op->d_S); hipFree(op->d_rs);
    if (op->mg || op->mg3) mg_destroy(op);
    for (int b = 0; b < 4; ++b) { nd_free(op->direct[b]); op->direct[b] = nullptr; }
    helm_pool_free(op->device, op->d_ws, op->ws_bytes); helm_pool_free(op->device, op->d_part, op->part_bytes);
    helm_pool_free(op->device, op->sk_buf, op->sk_bytes);
    helm_pool_free(op->device, op->gjp_buf, op->gjp_bytes);
    helm_pool_free(op->device, op->d_scal, (size_t)op->scal_cap * sizeof(RhsScal));
    helm_hostpool_free(op->h_scal, op->h_scal_bytes);
    if (op->pf_done) hipEventDestroy(op->pf_done);
    if (op->pf_t0) hipEventDestroy(op->pf_t0);
    if (op->pf_t1) hipEventDestroy(op->pf_t1);
    if (op->fstream) helm_stream_release(op->device, op->fstream_prio, op->fstream);
    {   // timing events go back to the process-wide free list
        std::lock_guard<std::mutex> lk(g_pool.mu);
        std::vector<hipEvent_t> &idle = g_idle_events[op->device];
        for (hipEvent_t e : op->ev_pool) { if (idle.size() < 65536) idle.push_back(e); else hipEventDestroy(e); }
    }
    if (op->side_stream) helm_stream_release(op->device, -1, op->side_stream);
    if (op->own_stream && op->stream) helm_stream_release(op->device, 0, op->stream);
    const int device = op->device;
    delete op;
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(g_shared_ws.mu);
        g_live_handles -= 1;
        int &n = g_live_per_device[device];
        if (n > 0) n -= 1;
        last = n == 0;
    }
    // (auto: the spares of the big size classes are topped up here, when nobody is waiting for this thread -- HELM_POOL_SPARE_AUTO=0 leaves it to helm_pool_spares,
    // for callers that time the region this destroy ends)
    { const int spare = tune_i("HELM_POOL_SPARE", 2); if (last && spare > 0 && tune_i("HELM_POOL_SPARE_AUTO", 1)) pool_top_up(device, spare); }
}

// Release what the library caches between calls (the shared scratch of the direct path).  The scratch is kept across
// handles on purpose -- allocating tens of GB costs far more than a solve -- so a host that wants the memory back says so.
static void scratch_sweep_all_wait();        // (scratch of enqueued factorisations: waits for them and hands it back)
static void scratch_sweep_fwd(int device);   // (the same for what has finished on one device, without waiting)
extern "C" int helm_pool_spares(int device, int spare) {
    helm_tuning_refresh();
    if (spare < 0) return HELM_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); helm_set_error(nullptr, "helm_pool_spares: hipSetDevice failed"); return HELM_ERR_DEVICE; }
    scratch_sweep_fwd(device);
    if (spare > 0) pool_top_up(device, spare);
    return HELM_OK;
}
extern "C" int helm_trim(void) {
    helm_tuning_refresh();
    int cur = 0;
    (void)hipGetDevice(&cur);
    scratch_sweep_all_wait();
    {
        std::lock_guard<std::mutex> lk(g_shared_ws.mu);
        for (auto &kv : g_shared_ws.dev) for (int i = 0; i < WS_SLOTS_MAX; ++i) if (kv.second.slot[i].busy) return HELM_ERR_STATE;
        for (auto &kv : g_shared_ws.dev)
            for (int i = 0; i < WS_SLOTS_MAX; ++i) {
                WsSlot &w = kv.second.slot[i];
                if (w.ptr) { hipSetDevice(kv.first); hipFree(w.ptr); }
                w.ptr = nullptr; w.bytes = 0;
            }
        std::lock_guard<std::mutex> lp(g_pool.mu);
        std::map<int, size_t> kept;
        for (auto it = g_pool.idle.begin(); it != g_pool.idle.end(); ) {
            if (g_carved.count(it->second)) { kept[it->first.first] += it->first.second; ++it; continue; }       // (pieces of a slab stay idle: slabs live as long as the process)
            hipSetDevice(it->first.first); hipFree(it->second); pool_forget(it->second);
            it = g_pool.idle.erase(it);
        }
        g_pool.held = kept;
        for (auto &kv : g_pool_stats) { kv.second.total = kv.second.in_use; kv.second.high = kv.second.in_use; }
    }
    (void)hipSetDevice(cur);
    return helm_host_trim();
}

// (tests) how many scratch slots of `device` hold a buffer of at least `bytes`; -1: the number of slots per device
extern "C" int helm_debug_ws_slots(int device, long long bytes) {
    helm_tuning_refresh();
    if (device < 0) return shared_ws_slots();
    std::lock_guard<std::mutex> lk(g_shared_ws.mu);
    auto it = g_shared_ws.dev.find(device);
    if (it == g_shared_ws.dev.end()) return 0;
    int n = 0;
    for (int i = 0; i < WS_SLOTS_MAX; ++i) if (it->second.slot[i].ptr && (long long)it->second.slot[i].bytes >= bytes) n += 1;
    return n;
}

extern "C" int helm_set_stream(helm_op *op, void *hip_stream) {
    helm_tuning_refresh();
    if (!op) return HELM_ERR_ARG;
    HIP_TRY(op, hipSetDevice(op->device));
    helm_pf_retire(op);
    if (op->stream) HIP_TRY(op, hipStreamSynchronize(op->stream));
    // the multigrid level operators launch on the stream they were given at setup (mg.hip assemble_child): they are rebuilt on
    // the new stream by the next solve that needs them
    if (op->mg || op->mg3) mg_destroy(op);
    if (op->own_stream && op->stream) { helm_stream_release(op->device, 0, op->stream); op->own_stream = false; }
    if (hip_stream) { op->stream = (hipStream_t)hip_stream; op->own_stream = false; }
    else { op->stream = helm_stream_acquire(op->device, 0); if (!op->stream) HELM_FAIL(op, HELM_ERR_DEVICE, "hipStreamCreate failed"); op->own_stream = true; }
    return HELM_OK;
}

extern "C" int helm_set_profiling(helm_op *op, int on) {
    helm_tuning_refresh(); if (!op) return HELM_ERR_ARG; op->profiling = on != 0; return HELM_OK; }
extern "C" int helm_last_timing(const helm_op *op, helm_timing *out) { if (!op || !out) return HELM_ERR_ARG; *out = op->timing; return HELM_OK; }
extern "C" int helm_num_blocks(const helm_op *op) { return op ? op->nblocks : HELM_ERR_ARG; }
extern "C" long long helm_num_points(const helm_op *op) { return op ? op->N : HELM_ERR_ARG; }

// Host array -> device through pinned buffers of the library (two chunks of 4 MB from the host pool: the memcpy of chunk k+1 runs beside the DMA of chunk k); the
// caller's pages are never handed to the runtime.  A copy of a few MB straight from pageable memory makes HIP pin the caller's pages in place (a user-pointer
// registration); when those pages go back to the system afterwards -- numpy frees an array of that size with munmap -- the kernel driver takes EVERY queue of the
// process off the GPU while it deals with the registration: 15-20 ms in which nothing of this process runs, charged to whatever is submitted next (round 6:
// one dpred(m) of config 4 in three took 55-65 ms instead of 39; with glibc told never to unmap, none did -- profiles/r06_config4_dpred_spread.txt).
// The few small kernels of an operator's set-up (default density, assembly) go to a stream of another priority class than the operator's own for the duration of
// the call.  Streams of one priority class share a handful of hardware queues, each of them in order: on the operator's normal-priority stream an 8-us kernel of
// the NEXT operator's set-up sat behind whatever solve had been queued on the same hardware queue -- helm_set_model / helm_assemble took 10-20 ms of the
// pipeline's prepare thread in every other set (tools/pipeline_timeline.py: the call ended when the other thread's helm_solve_device did).  The low class is
// used by nothing else on the 2-D path.  Top-level operators only: a multigrid level's operator shares its parent's stream and must stay in its order.
namespace {
struct SetupStream {
    helm_op *op; hipStream_t keep, tmp = nullptr; int prio;
    explicit SetupStream(helm_op *o) : op(o), keep(o->stream) {
        static const int pr = getenv("HELM_SETUP_PRIO") ? atoi(getenv("HELM_SETUP_PRIO")) : -1;
        prio = pr;
        if (pr == 0 || !o->own_stream || o->block0_only || !keep) return;
        if (hipStreamSynchronize(keep) != hipSuccess) { (void)hipGetLastError(); return; }        // (nothing of this operator is in flight when its model or frequency changes)
        tmp = helm_stream_acquire(o->device, prio);
        if (tmp) o->stream = tmp;
    }
    ~SetupStream() {
        if (!tmp) return;
        (void)hipStreamSynchronize(tmp);
        op->stream = keep;
        helm_stream_release(op->device, prio, tmp);
    }
};
}
namespace {
std::mutex g_upload_mu;
std::map<int, std::vector<hipEvent_t>> g_upload_events;              // per device, recycled (an event per chunk buffer of an upload in flight)
}
// is this host address memory the runtime already knows as pinned (hipHostMalloc / hipHostRegister, the library's own host pool included)?
static bool host_ptr_is_pinned(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}
// One direction of a staged copy: `up` host -> device, else device -> host; returns when the data is where it was asked to be.
static int copy_staged(helm_op *op, void *dst, const void *src, size_t bytes, bool up) {
    const size_t chunk = (size_t)4 << 20;
    if (bytes == 0) return HELM_OK;
    // The copies run on a stream of another priority class (xs; the low one, which the 2-D path uses for nothing else), not on the operator's: the copy of a chunk is a small kernel (or an SDMA packet behind one), and
    // on the operator's normal-priority stream it waited its turn behind the solve kernels of other operators -- a 24-MB model took 10 ms of the pipeline's prepare
    // thread in every other set (tools/pipeline_timeline.py).  The call returns when the data has arrived, so nothing the operator's stream gets afterwards can
    // overtake it; what that stream has queued BEFORE the call is waited for first (a download reads what those launches produce).
    hipStream_t ops = op->stream;
    if (hipStreamSynchronize(ops) != hipSuccess) { (void)hipGetLastError(); helm_set_error(op, "host / device copy failed"); return HELM_ERR_DEVICE; }
    static const int xprio = getenv("HELM_XFER_PRIO") ? atoi(getenv("HELM_XFER_PRIO")) : -1;
    hipStream_t xs = helm_stream_acquire(op->device, xprio);
    if (!xs) { helm_set_error(op, "host / device copy: no stream"); return HELM_ERR_DEVICE; }
    struct XsGuard { int dev, prio; hipStream_t s; ~XsGuard() { helm_stream_release(dev, prio, s); } } xs_guard{op->device, xprio, xs};
    if (host_ptr_is_pinned(up ? src : dst)) {                     // nothing to protect: the runtime moves it straight
        if (hipMemcpyAsync(dst, src, bytes, up ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, xs) != hipSuccess || hipStreamSynchronize(xs) != hipSuccess) {
            (void)hipGetLastError(); helm_set_error(op, "host / device copy failed"); return HELM_ERR_DEVICE;
        }
        return HELM_OK;
    }
    const int nbuf = bytes > chunk ? 2 : 1;
    char *buf[2] = {(char *)helm_hostpool_alloc(chunk), nbuf > 1 ? (char *)helm_hostpool_alloc(chunk) : nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    {
        std::lock_guard<std::mutex> lk(g_upload_mu);
        std::vector<hipEvent_t> &v = g_upload_events[op->device];
        for (int b = 0; b < nbuf; ++b) if (!v.empty()) { ev[b] = v.back(); v.pop_back(); }
    }
    int rc = HELM_OK;
    for (int b = 0; b < nbuf; ++b) {
        if (!buf[b]) rc = HELM_ERR_DEVICE;
        if (!ev[b] && hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) != hipSuccess) { ev[b] = nullptr; rc = HELM_ERR_DEVICE; }
    }
    bool used[2] = {false, false};
    size_t pend_off[2] = {0, 0}, pend_n[2] = {0, 0};             // (down: the chunk that sits in buf[b] and still has to reach the caller's array)
    for (size_t off = 0, k = 0; off < bytes && rc == HELM_OK; off += chunk, ++k) {
        const int b = (int)(k % nbuf);
        const size_t n = std::min(chunk, bytes - off);
        if (used[b]) {
            if (hipEventSynchronize(ev[b]) != hipSuccess) { rc = HELM_ERR_DEVICE; break; }
            if (!up) memcpy((char *)dst + pend_off[b], buf[b], pend_n[b]);
        }
        if (up) memcpy(buf[b], (const char *)src + off, n);
        const hipError_t e = up ? hipMemcpyAsync((char *)dst + off, buf[b], n, hipMemcpyHostToDevice, xs)
                                : hipMemcpyAsync(buf[b], (const char *)src + off, n, hipMemcpyDeviceToHost, xs);
        if (e != hipSuccess || hipEventRecord(ev[b], xs) != hipSuccess) { rc = HELM_ERR_DEVICE; break; }
        used[b] = true; pend_off[b] = off; pend_n[b] = n;
    }
    // (the chunks still in flight, oldest first)
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    for (int q = 0; q < nbuf; ++q) {
        const int b = (int)((nchunks + q) % nbuf);
        if (used[b]) {
            if (hipEventSynchronize(ev[b]) != hipSuccess) rc = HELM_ERR_DEVICE;
            else if (!up && rc == HELM_OK) memcpy((char *)dst + pend_off[b], buf[b], pend_n[b]);
            used[b] = false;
        }
    }
    for (int b = 0; b < nbuf; ++b) if (buf[b]) helm_hostpool_free(buf[b], chunk);
    {
        std::lock_guard<std::mutex> lk(g_upload_mu);
        for (int b = 0; b < nbuf; ++b) if (ev[b]) g_upload_events[op->device].push_back(ev[b]);
    }
    if (rc) { (void)hipGetLastError(); helm_set_error(op, "host / device copy failed"); }
    return rc;
}
int helm_upload_staged(helm_op *op, void *dst, const void *src, size_t bytes) { return copy_staged(op, dst, src, bytes, true); }
int helm_download_staged(helm_op *op, void *dst, const void *src, size_t bytes) { return copy_staged(op, dst, src, bytes, false); }

extern "C" int helm_set_model(helm_op *op, const double *c, const double *rho, const double *theta, const double *eps, const double *delta) {
    helm_tuning_refresh();
    if (!op || !c) return HELM_ERR_ARG;
    HIP_TRY(op, hipSetDevice(op->device));
    const size_t N = (size_t)op->N;
    SetupStream setup_stream(op);
    if (helm_upload_staged(op, op->d_c, c, N * sizeof(cplx))) return HELM_ERR_DEVICE;
    if (!rho) {   // Gardner default 310 * Re(c)^0.25  (discretization.py:70), evaluated on the device
        const int rcg = helm_launch_gardner_rho(op);
        if (rcg) return rcg;
    } else {
        if (helm_upload_staged(op, op->d_rho, rho, N * sizeof(double))) return HELM_ERR_DEVICE;
    }
    op->aniso = false;
    bool m3zero = true;
    if (op->variant == HELM_EURUS) {
        auto up = [&](double *&dst, const double *src) -> int {
            if (!src) { if (dst) { helm_pool_free(op->device, dst, N * sizeof(double)); dst = nullptr; } return 0; }
            if (!dst) { dst = (double *)helm_pool_alloc(op->device, N * sizeof(double)); if (!dst) return -1; }
            return helm_upload_staged(op, dst, src, N * sizeof(double)) ? -1 : 0;
        };
        if (up(op->d_theta, theta) || up(op->d_eps, eps) || up(op->d_delta, delta)) HELM_FAIL(op, HELM_ERR_DEVICE, "anisotropy upload failed");
        op->aniso = theta || eps || delta;
        for (size_t i = 0; i < N && m3zero; ++i) {
            const double e = eps ? eps[i] : 0.0, d = delta ? delta[i] : 0.0;
            if (e != d) m3zero = false;
        }
    }
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    // host copies are only needed to build the multigrid levels: fetched back from the device if that ever happens
    op->h_c.clear(); op->h_rho.clear(); op->h_theta.clear(); op->h_eps.clear(); op->h_delta.clear();
    op->block_zero[0] = op->block_zero[1] = op->block_zero[3] = false;
    op->block_zero[2] = m3zero;
    op->has_model = true;
    op->assembled = false;
    return HELM_OK;
}

// The model of an operator that lives on the device already (a multigrid level: the caller's arrays, or values a kernel has put into
// dst->d_c / dst->d_rho -- then both pointers are null): device-to-device, nothing visits the host.  Isotropic operators only.
int helm_adopt_model_device(helm_op *dst, const cplx *d_c, const double *d_rho) {
    if (!dst || (d_c == nullptr) != (d_rho == nullptr)) return HELM_ERR_ARG;
    HIP_TRY(dst, hipSetDevice(dst->device));
    const size_t N = (size_t)dst->N;
    if (d_c) {
        HIP_TRY(dst, hipMemcpyAsync(dst->d_c, d_c, N * sizeof(cplx), hipMemcpyDeviceToDevice, dst->stream));
        HIP_TRY(dst, hipMemcpyAsync(dst->d_rho, d_rho, N * sizeof(double), hipMemcpyDeviceToDevice, dst->stream));
    }
    dst->aniso = false;
    dst->h_c.clear(); dst->h_rho.clear(); dst->h_theta.clear(); dst->h_eps.clear(); dst->h_delta.clear();
    dst->block_zero[0] = dst->block_zero[1] = dst->block_zero[3] = false;
    dst->block_zero[2] = true;
    dst->has_model = true;
    dst->assembled = false;
    return HELM_OK;
}

int helm_ensure_host_model(helm_op *op) {
    if (!op->has_model) HELM_FAIL(op, HELM_ERR_STATE, "model not set");
    if (!op->h_c.empty()) return HELM_OK;
    const size_t N = (size_t)op->N;
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    op->h_c.resize(N); op->h_rho.resize(N);
    HIP_TRY(op, hipMemcpy(op->h_c.data(), op->d_c, N * sizeof(cplx), hipMemcpyDeviceToHost));
    HIP_TRY(op, hipMemcpy(op->h_rho.data(), op->d_rho, N * sizeof(double), hipMemcpyDeviceToHost));
    auto down = [&](std::vector<double> &dst, const double *src) -> int {
        dst.clear();
        if (!src) return 0;
        dst.resize(N);
        return hipMemcpy(dst.data(), src, N * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
    };
    if (down(op->h_theta, op->d_theta) || down(op->h_eps, op->d_eps) || down(op->h_delta, op->d_delta)) HELM_FAIL(op, HELM_ERR_DEVICE, "model download failed");
    return HELM_OK;
}

extern "C" int helm_assemble(helm_op *op, double freq_re, double freq_im, double tau, double ky, double cPML) {
    helm_tuning_refresh();
    if (!op) return HELM_ERR_ARG;
    if (!op->has_model) HELM_FAIL(op, HELM_ERR_STATE, "helm_set_model must be called before helm_assemble");
    HIP_TRY(op, hipSetDevice(op->device));
    helm_pf_retire(op);                          // a factorisation still in flight belongs to the operator that is being replaced
    // Eurus with eps == delta is block-triangular and an N-row right-hand side (what the surveys bring, eurus.py:512-533) touches M1 alone: M2 .. M4 -- three
    // quarters of the 604 MB the assembly writes at 1024^2 -- are built when something asks for them (helm_need_all_blocks: a stacked 2N right-hand side,
    // helm_get_diagonals, an apply of another block, the scaled planes of the Krylov paths)
    op->asm_nblk = (op->variant == HELM_EURUS && op->block_zero[2] && !op->block0_only && helm_tuning_now().auto_direct != 0) ? 1 : 4;
    SetupStream setup_stream(op);
    int rc = op->ny > 0 ? helm3d_launch_assemble(op, freq_re, freq_im, tau, cPML) : helm_launch_assemble(op, freq_re, freq_im, tau, ky, cPML);
    if (rc) return rc;
    op->scaled_ok = false;
    if (op->block0_only) { rc = helm_ensure_scaled(op); if (rc) return rc; }     // multigrid levels always smooth with 1/diag
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    op->assembled = true;
    op->a_freq_re = freq_re; op->a_freq_im = freq_im; op->a_tau = tau; op->a_ky = ky; op->a_cpml = cPML;
    if (op->mg || op->mg3) mg_destroy(op);      // preconditioner belongs to the previous frequency
    op->mg3_no_keep = false;
    for (int b = 0; b < 4; ++b) { nd_free(op->direct[b]); op->direct[b] = nullptr; }   // and so do the direct factors
    op->direct_failed = false;
    return HELM_OK;
}

int helm_need_all_blocks(helm_op *op) {
    if (op->variant != HELM_EURUS || op->block0_only || op->ny > 0 || !op->assembled || op->blocks_ready >= op->nblocks) return HELM_OK;
    op->asm_nblk = 4;                            // (M1 is written again with the same values: a factorisation reading it meanwhile sees no change)
    return helm_launch_assemble(op, op->a_freq_re, op->a_freq_im, op->a_tau, op->a_ky, op->a_cpml);
}

int helm_ensure_scaled(helm_op *op) {
    if (op->scaled_ok) return HELM_OK;
    { const int rcb = helm_need_all_blocks(op); if (rcb) return rcb; }
    const size_t N = (size_t)op->N;
    if (!op->d_Cs) op->d_Cs = (cplx *)helm_pool_alloc(op->device, (size_t)op->nblocks * op->nplanes * N * sizeof(cplx));
    if (!op->d_dinv) op->d_dinv = (cplx *)helm_pool_alloc(op->device, (size_t)op->nblocks * N * sizeof(cplx));
    if (!op->d_Cs || !op->d_dinv) HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc of the scaled coefficient planes failed");
    const int rc = helm_launch_scale_planes(op);
    if (rc) return rc;
    op->scaled_ok = true;
    return HELM_OK;
}

extern "C" int helm_get_diagonals(helm_op *op, double *out) {
    helm_tuning_refresh();
    if (!op || !out) return HELM_ERR_ARG;
    if (!op->assembled) HELM_FAIL(op, HELM_ERR_STATE, "operator not assembled");
    HIP_TRY(op, hipSetDevice(op->device));
    { const int rcb = helm_need_all_blocks(op); if (rcb) return rcb; }
    if (helm_download_staged(op, out, op->d_C, (size_t)op->nblocks * op->nplanes * op->N * sizeof(cplx))) return HELM_ERR_DEVICE;
    return HELM_OK;
}

// ---- workspace ------------------------------------------------------------------------------
static int ensure_ws(helm_op *op, size_t bytes) {
    if (op->ws_bytes >= bytes) return HELM_OK;
    // from the size-keyed pool: a job makes one operator per frequency and the Krylov workspace of a 3-D batch is tens of GB
    if (op->d_ws) { hipStreamSynchronize(op->stream); helm_pool_free(op->device, op->d_ws, op->ws_bytes); op->d_ws = nullptr; op->ws_bytes = 0; }
    op->d_ws = helm_pool_alloc(op->device, bytes);
    if (!op->d_ws) HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc of the solver workspace (%.1f GB) failed", bytes / 1e9);
    op->ws_bytes = bytes;
    return HELM_OK;
}
static int ensure_part(helm_op *op, int nrhs) {
    const int nblk = std::max(2 * helm_apply_num_blocks(op), helm_vec_num_blocks(op));
    const size_t bytes = (size_t)nrhs * 4 * nblk * sizeof(double) + (size_t)nrhs * (2 * sizeof(double) + sizeof(int)) + 256;
    if (op->part_bytes < bytes) {
        if (op->d_part) { hipStreamSynchronize(op->stream); helm_pool_free(op->device, op->d_part, op->part_bytes); op->d_part = nullptr; op->part_bytes = 0; }
        op->d_part = helm_pool_alloc(op->device, bytes);
        if (!op->d_part) HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc of the partial-sum buffer failed");
        op->part_bytes = bytes;
    }
    if (op->scal_cap < nrhs) {
        if (op->d_scal || op->h_scal) hipStreamSynchronize(op->stream);
        helm_pool_free(op->device, op->d_scal, (size_t)op->scal_cap * sizeof(RhsScal));
        helm_hostpool_free(op->h_scal, op->h_scal_bytes);
        op->d_scal = nullptr; op->h_scal = nullptr; op->scal_cap = 0; op->h_scal_bytes = 0;
        op->d_scal = (RhsScal *)helm_pool_alloc(op->device, (size_t)nrhs * sizeof(RhsScal));
        const size_t hb = (size_t)nrhs * sizeof(RhsScal) + (size_t)nrhs * (2 * sizeof(double) + sizeof(int)) + 64;
        op->h_scal = (RhsScal *)helm_hostpool_alloc(hb);
        if (!op->d_scal || !op->h_scal) HELM_FAIL(op, HELM_ERR_DEVICE, "allocation of the per-right-hand-side records failed");
        op->h_scal_bytes = hb;
        op->scal_cap = nrhs;
    }
    return HELM_OK;
}

static void timing_begin(helm_op *op) {
    if (!op->pf_pending) {       // (the launches of a factorisation started by helm_prefactor are booked with the solve that uses it)
        op->ev_used = 0;
        op->ev_pending.clear();
        op->ev_pending_gemm.clear(); op->ev_pending_gemm_n.clear(); op->ev_pending_gemm_bytes.clear(); op->ev_pending_gemm_sol.clear(); op->ev_pending_gemm_shape.clear();
    }
    op->timing.apply_ms = 0; op->timing.apply_launches = 0; op->timing.apply_bytes = 0;
    op->timing.factor_ms = 0; op->timing.gemm_ms = 0; op->timing.gemm_launches = 0; op->timing.gemm_flops = 0; op->timing.gemm_bytes = 0; op->timing.gemm_sol_ms = 0;
    op->timing.gemm_big_ms = 0; op->timing.gemm_big_launches = 0; op->timing.gemm_big_flops = 0;
}
static void timing_collect(helm_op *op) {
    for (auto &pr : op->ev_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, op->ev_pool[pr.first], op->ev_pool[pr.first + 1]) == hipSuccess) {
            op->timing.apply_ms += ms; op->timing.apply_launches += 1; op->timing.apply_bytes += pr.second;
        }
    }
    op->ev_pending.clear();
    static const int gemm_log = getenv("HELM_GEMM_LOG") ? atoi(getenv("HELM_GEMM_LOG")) : 0;
    for (size_t i = 0; i < op->ev_pending_gemm.size(); ++i) {
        const std::pair<int, double> &pr = op->ev_pending_gemm[i];
        const int nl = i < op->ev_pending_gemm_n.size() ? op->ev_pending_gemm_n[i] : 1;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, op->ev_pool[pr.first], op->ev_pool[pr.first + 1]) == hipSuccess) {
            if (gemm_log && 5 * i + 4 < op->ev_pending_gemm_shape.size()) {
                const long long *sh = &op->ev_pending_gemm_shape[5 * i];
                fprintf(stderr, "[gemm log] M %lld N %lld K %lld batch %lld mode %lld : %.1f us, %.2f TFLOP/s, %.0f GB/s of operands\n", sh[0], sh[1], sh[2], sh[3], sh[4],
                        1e3 * ms, pr.second / (ms * 1e-3) / 1e12, (i < op->ev_pending_gemm_bytes.size() ? op->ev_pending_gemm_bytes[i] : 0.0) / (ms * 1e-3) / 1e9);
            }
            op->timing.gemm_ms += ms; op->timing.gemm_launches += nl; op->timing.gemm_flops += pr.second;
            if (i < op->ev_pending_gemm_bytes.size()) { op->timing.gemm_bytes += op->ev_pending_gemm_bytes[i]; op->timing.gemm_sol_ms += op->ev_pending_gemm_sol[i]; }
            if (pr.second >= 1e9 * nl) { op->timing.gemm_big_ms += ms; op->timing.gemm_big_launches += nl; op->timing.gemm_big_flops += pr.second; }
        }
    }
    op->ev_pending_gemm.clear(); op->ev_pending_gemm_n.clear(); op->ev_pending_gemm_bytes.clear(); op->ev_pending_gemm_sol.clear(); op->ev_pending_gemm_shape.clear();
    op->ev_used = 0;
}

// ---- apply -----------------------------------------------------------------------------------
extern "C" int helm_apply_device(helm_op *op, int block, int adjoint, const void *dX, void *dY, int nrhs) {
    helm_tuning_refresh();
    if (!op || !dX || !dY || nrhs < 1 || block < 0 || block >= op->nblocks) return HELM_ERR_ARG;
    if (!op->assembled) HELM_FAIL(op, HELM_ERR_STATE, "operator not assembled");
    HIP_TRY(op, hipSetDevice(op->device));
    if (block > 0) { const int rcb = helm_need_all_blocks(op); if (rcb) return rcb; }
    timing_begin(op);
    ApplyArgs a = ApplyArgs();
    a.planes = op->d_C + (long long)block * op->nplanes * op->N; a.X = (const cplx *)dX; a.Y = (cplx *)dY; a.W = nullptr;
    a.ld = op->N; a.nrhs = nrhs; a.scaled = 0; a.adjoint = adjoint ? 1 : 0; a.epi = EPI_NONE; a.scal = nullptr; a.part = nullptr;
    int rc = helm_launch_apply(op, a);
    if (rc) return rc;
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    timing_collect(op);
    return HELM_OK;
}

extern "C" int helm_apply(helm_op *op, int block, int adjoint, const double *X, double *Y, int nrhs) {
    helm_tuning_refresh();
    if (!op || !X || !Y || nrhs < 1) return HELM_ERR_ARG;
    HIP_TRY(op, hipSetDevice(op->device));
    const size_t bytes = (size_t)nrhs * op->N * sizeof(cplx);
    void *dX = nullptr, *dY = nullptr;
    HIP_TRY(op, hipMalloc(&dX, bytes));
    if (hipMalloc(&dY, bytes) != hipSuccess) { hipFree(dX); HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc failed"); }
    int rc = HELM_OK;
    if (helm_upload_staged(op, dX, X, bytes)) rc = HELM_ERR_DEVICE;
    if (!rc) rc = helm_apply_device(op, block, adjoint, dX, dY, nrhs);
    if (!rc && helm_download_staged(op, Y, dY, bytes)) rc = HELM_ERR_DEVICE;
    hipFree(dX); hipFree(dY);
    return rc;
}

// ---- Krylov drivers ---------------------------------------------------------------------------
namespace {

struct Batch {
    int nrhs;
    VecPtrs w;
    cplx *bbar;          // right-hand side of the system being iterated (scaled q' or, with the MG preconditioner, q')
    cplx *bscaled = nullptr;                 // D^-1 q' (right-hand side of the Jacobi-scaled system; CGNR fallback)
    cplx *phat = nullptr, *shat = nullptr;   // preconditioned directions (MG mode)
    bool pre = false;    // true: BiCGSTAB on A right-preconditioned by multigrid; false: Jacobi-scaled system
    bool sys2 = false;   // coupled two-field Eurus system (vectors of length 2N, four stencil launches per apply)
    int nba = 0;         // partial sums written by one (system) apply
    const cplx *planes = nullptr;            // planes of the iterated operator (raw for pre, scaled otherwise)
    int *d_mask; double *d_aux;      // device, nrhs ints / 2*nrhs doubles (inside d_part tail)
    int *h_mask; double *h_aux;      // pinned (inside h_scal tail)
};

int download_scal(helm_op *op, int nrhs) {
    HIP_TRY(op, hipMemcpyAsync(op->h_scal, op->d_scal, (size_t)nrhs * sizeof(RhsScal), hipMemcpyDeviceToHost, op->stream));
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    return HELM_OK;
}
int upload_scal(helm_op *op, int nrhs) {
    HIP_TRY(op, hipMemcpyAsync(op->d_scal, op->h_scal, (size_t)nrhs * sizeof(RhsScal), hipMemcpyHostToDevice, op->stream));
    return HELM_OK;
}

ApplyArgs scaled_apply(helm_op *op, int block, const cplx *X, cplx *Y, const cplx *W, int nrhs, int adjoint, int epi, bool masked) {
    ApplyArgs a = ApplyArgs();
    a.planes = op->d_Cs + (long long)block * op->nplanes * op->N; a.X = X; a.Y = Y; a.W = W; a.ld = op->N; a.nrhs = nrhs;
    a.scaled = 1; a.adjoint = adjoint; a.epi = epi; a.scal = masked ? op->d_scal : nullptr; a.part = (double *)op->d_part;
    return a;
}

// apply of the operator the BiCGSTAB batch iterates on (Jacobi-scaled planes, or raw planes in MG mode)
ApplyArgs batch_apply(helm_op *op, const Batch &B, const cplx *X, cplx *Y, const cplx *W, int epi) {
    ApplyArgs a = ApplyArgs();
    a.planes = B.planes; a.X = X; a.Y = Y; a.W = W; a.ld = op->N; a.nrhs = B.nrhs;
    a.scaled = B.pre ? 0 : 1; a.adjoint = 0; a.epi = epi; a.scal = op->d_scal; a.part = (double *)op->d_part;
    return a;
}

int launch_sys2_apply(helm_op *op, bool raw, int adjoint, const cplx *X, cplx *Y, const cplx *W, int nrhs, int epi, const RhsScal *scal, const cplx *planes_override = nullptr);

// Apply of the coupled Eurus system [[M1, M2], [M3, M4]] (or its conjugate transpose) to vectors [u; v] of length 2N:
// four stencil launches, the second of each output half accumulating into the first and carrying the fused epilogue.
// raw = unscaled planes (true residual), otherwise the row-equilibrated system d_S.
int launch_sys2_apply(helm_op *op, bool raw, int adjoint, const cplx *X, cplx *Y, const cplx *W, int nrhs, int epi, const RhsScal *scal, const cplx *planes_override) {
    const long long N = op->N;
    const int nblk = helm_apply_num_blocks(op);
    const cplx *P = planes_override ? planes_override : (raw ? op->d_C : op->d_S);
    if (epi == EPI_DOT_XY) { epi = EPI_DOT_WY; W = X; }
    for (int half = 0; half < 2; ++half) {
        // forward: out_half = M[half][0] in0 + M[half][1] in1 ; adjoint: out_half = M[0][half]^H in0 + M[1][half]^H in1
        const int blkA = adjoint ? (0 * 2 + half) : (half * 2 + 0), blkB = adjoint ? (1 * 2 + half) : (half * 2 + 1);
        ApplyArgs a = ApplyArgs();
        a.ld = 2 * N; a.nrhs = nrhs; a.scal = scal; a.part = (double *)op->d_part; a.part_stride = 2 * nblk; a.adjoint = adjoint; a.scaled = 0;
        a.planes = P + (long long)blkA * 9 * N; a.X = X; a.Y = Y + half * N; a.epi = EPI_NONE; a.profile = 0;
        int rc = helm_launch_apply(op, a);
        if (rc) return rc;
        a.planes = P + (long long)blkB * 9 * N; a.X = X + N; a.acc = 1; a.epi = epi; a.W = W ? W + half * N : nullptr; a.part_off = half * nblk; a.profile = 1;
        rc = helm_launch_apply(op, a);
        if (rc) return rc;
    }
    return HELM_OK;
}

int launch_batch_apply(helm_op *op, const Batch &B, const cplx *X, cplx *Y, const cplx *W, int epi) {
    if (B.sys2) return launch_sys2_apply(op, false, 0, X, Y, W, B.nrhs, epi, op->d_scal);
    return helm_launch_apply(op, batch_apply(op, B, X, Y, W, epi));
}

// Restart the right-hand sides flagged in h_mask from their current iterate x:
// r = bbar - Abar x, r0 = r, p = v = 0, scalars reset.  Host copy of scal must be fresh.
int restart_masked(helm_op *op, int block, Batch &B) {
    const int n = B.nrhs;
    for (int b = 0; b < n; ++b) {
        if (B.h_mask[b]) op->h_scal[b].status = ST_ACTIVE;
        else if (op->h_scal[b].status == ST_ACTIVE) op->h_scal[b].status = ST_PARKED;
    }
    int rc = upload_scal(op, n);
    if (rc) return rc;
    HIP_TRY(op, hipMemcpyAsync(B.d_mask, B.h_mask, n * sizeof(int), hipMemcpyHostToDevice, op->stream));
    rc = launch_batch_apply(op, B, B.w.x, B.w.r, B.bbar, EPI_RESID);
    if (rc) return rc;
    helm_launch_restart_copy_mask(op, B.w, n, B.d_mask);
    helm_launch_fin_ex(op, FIN_RESTART, n, B.nba, B.d_mask, nullptr);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

int run_bicgstab(helm_op *op, int block, Batch &B, int maxit, int check_every, int max_restarts, std::vector<int> &restarts) {
    const int n = B.nrhs;
    const int nba = B.nba, nbv = helm_vec_num_blocks(op);
    int it_done = 0;
    while (true) {
        int rc = download_scal(op, n);
        if (rc) return rc;
        bool any_active = false, any_break = false;
        int min_iters = std::numeric_limits<int>::max();
        int nactive = 0;
        for (int b = 0; b < n; ++b) {
            B.h_mask[b] = 0;
            RhsScal &S = op->h_scal[b];
            if (S.status == ST_ACTIVE) {
                if (S.iters >= maxit) S.status = ST_FROZEN;   // iteration cap: stop working on it
                else { any_active = true; nactive += 1; min_iters = std::min(min_iters, S.iters); }
            } else if (S.status == ST_BREAKDOWN && restarts[b] < max_restarts && S.iters < maxit) {
                B.h_mask[b] = 1; any_break = true; restarts[b] += 1;
            }
        }
        if (any_break) {
            rc = restart_masked(op, block, B);
            if (rc) return rc;
            continue;     // re-read the status (a restarted RHS may already satisfy the tolerance)
        }
        if (!any_active) {
            // un-freeze bookkeeping for the caller: frozen-by-cap stays FROZEN
            upload_scal(op, n);
            break;
        }
        upload_scal(op, n);
        op->active_hint = nactive;
        const int chunk = std::max(1, std::min(check_every, maxit - min_iters));
        for (int k = 0; k < chunk; ++k) {
            helm_launch_bicg_p(op, B.w, n);
            const cplx *pin = B.w.p, *sin = B.w.s;
            if (B.pre) { rc = mg_apply(op, B.w.p, B.phat, n, op->d_scal); if (rc) return rc; pin = B.phat; }
            rc = launch_batch_apply(op, B, pin, B.w.v, B.w.r0, EPI_DOT_W);
            if (rc) return rc;
            helm_launch_fin(op, FIN_ALPHA, n, nba);
            helm_launch_bicg_s(op, B.w, n);
            if (B.pre) {
                rc = mg_apply(op, B.w.s, B.shat, n, op->d_scal); if (rc) return rc; sin = B.shat;
                rc = launch_batch_apply(op, B, sin, B.w.t, B.w.s, EPI_DOT_WY);
            } else {
                rc = launch_batch_apply(op, B, sin, B.w.t, nullptr, EPI_DOT_XY);
            }
            if (rc) return rc;
            helm_launch_fin(op, FIN_OMEGA, n, nba);
            helm_launch_bicg_xr(op, B.w, pin, sin, n);
            helm_launch_fin(op, FIN_RHO, n, nbv);
        }
        HIP_TRY(op, hipGetLastError());
        it_done += chunk;
    }
    op->active_hint = -1;
    return HELM_OK;
}

// CGNR on the Jacobi-scaled system for the right-hand sides flagged in h_mask (warm start from x).
int run_cgnr(helm_op *op, int block, Batch &B, int maxit, int check_every) {
    const int n = B.nrhs;
    const int nba = B.nba, nbv = helm_vec_num_blocks(op);
    // r = bbar - Abar x for flagged RHS; others frozen
    for (int b = 0; b < n; ++b) {
        RhsScal &S = op->h_scal[b];
        if (B.h_mask[b]) { S.status = ST_ACTIVE; S.iters = 0; }
        else if (S.status == ST_ACTIVE) S.status = ST_FROZEN;
    }
    int rc = upload_scal(op, n);
    if (rc) return rc;
    auto cg_apply = [&](const cplx *X, cplx *Y, const cplx *W, int adjoint, int epi) -> int {
        if (B.sys2) return launch_sys2_apply(op, false, adjoint, X, Y, W, n, epi, op->d_scal);
        return helm_launch_apply(op, scaled_apply(op, block, X, Y, W, n, adjoint, epi, true));
    };
    rc = cg_apply(B.w.x, B.w.r, B.bscaled, 0, EPI_RESID);
    if (rc) return rc;
    helm_launch_fin(op, FIN_CG_RR, n, nba);            // rr (and convergence check); iters becomes 1
    rc = cg_apply(B.w.r, B.w.s, nullptr, 1, EPI_DOT_YY);   // z = A^H r
    if (rc) return rc;
    helm_launch_fin(op, FIN_CG_INIT, n, nba);
    helm_launch_cg_p(op, B.w, n, 1);
    while (true) {
        rc = download_scal(op, n);
        if (rc) return rc;
        bool any_active = false;
        int min_iters = std::numeric_limits<int>::max();
        for (int b = 0; b < n; ++b) {
            RhsScal &S = op->h_scal[b];
            if (S.status == ST_ACTIVE) {
                if (S.iters >= maxit) S.status = ST_FROZEN;
                else { any_active = true; min_iters = std::min(min_iters, S.iters); }
            }
        }
        upload_scal(op, n);
        if (!any_active) break;
        const int chunk = std::max(1, std::min(check_every, maxit - min_iters));
        for (int k = 0; k < chunk; ++k) {
            rc = cg_apply(B.w.p, B.w.v, nullptr, 0, EPI_DOT_YY);   // w = A p
            if (rc) return rc;
            helm_launch_fin(op, FIN_CG_ALPHA, n, nba);
            helm_launch_cg_xr(op, B.w, n);
            helm_launch_fin(op, FIN_CG_RR, n, nbv);
            rc = cg_apply(B.w.r, B.w.s, nullptr, 1, EPI_DOT_YY); // z = A^H r
            if (rc) return rc;
            helm_launch_fin(op, FIN_CG_BETA, n, nba);
            helm_launch_cg_p(op, B.w, n, 0);
        }
        HIP_TRY(op, hipGetLastError());
    }
    return HELM_OK;
}

// Solve M_block X = premul * RHS[:, row_off : row_off+N] - sub   for nrhs right-hand sides.
// dXout: [nrhs][N] (NOT conjugated).  info (optional) is filled per RHS.
struct NvGuard {     // Krylov vector length of the handle for the duration of a solve
    helm_op *op; long long old;
    NvGuard(helm_op *o, long long nv) : op(o), old(o->Nv) { o->Nv = nv; }
    ~NvGuard() { op->Nv = old; }
};


// the slot table's two operations, over any table and allocator (the library's own: g_shared_ws with hipMalloc; helm_debug_ws_selftest: a
// scratch table with malloc, so that the booking logic is testable without a GPU)
typedef void *(*ws_alloc_fn)(int device, size_t bytes);
typedef void (*ws_free_fn)(int device, void *p);
static void *ws_dev_alloc(int device, size_t bytes) { void *p = nullptr; AllocTrace tr("ws slot alloc", bytes); (void)helm_malloc_retry(device, &p, bytes); return p; }
static void ws_dev_free(int, void *p) { hipFree(p); }
// an idle slot of `device` that is already big enough, else any idle one (grown to `bytes`); nullptr when every slot of the device is taken or
// the allocation fails.  *lease = device * WS_SLOTS_MAX + slot.
static void *ws_table_checkout(SharedWs &T, int device, size_t bytes, int *lease, ws_alloc_fn al, ws_free_fn fr) {
    std::lock_guard<std::mutex> lk(T.mu);
    const int ns = shared_ws_slots();
    WsDevice &D = T.dev[device];
    int pick = -1;
    for (int i = 0; i < ns; ++i) { WsSlot &w = D.slot[i]; if (!w.busy && w.ptr && w.bytes >= bytes) { pick = i; break; } }
    if (pick < 0) for (int i = 0; i < ns; ++i) { WsSlot &w = D.slot[i]; if (!w.busy) { pick = i; break; } }
    if (pick < 0) return nullptr;
    WsSlot &w = D.slot[pick];
    if (w.bytes < bytes) {
        if (w.ptr) fr(device, w.ptr);
        w.ptr = al(device, bytes);
        w.bytes = w.ptr ? bytes : 0;
    }
    if (!w.ptr) return nullptr;
    w.busy = true; *lease = device * WS_SLOTS_MAX + pick;
    return w.ptr;
}
static void ws_table_checkin(SharedWs &T, int lease) {
    if (lease < 0) return;
    std::lock_guard<std::mutex> lk(T.mu);
    T.dev[lease / WS_SLOTS_MAX].slot[lease % WS_SLOTS_MAX].busy = false;
}
// make sure `concurrent` slots of `device` hold at least `bytes` each (idle slots that are too small are re-allocated; another device's table is
// never touched); returns the number of slots that are ready
static int ws_table_reserve(SharedWs &T, int device, size_t bytes, int concurrent, ws_alloc_fn al, ws_free_fn fr) {
    std::lock_guard<std::mutex> lk(T.mu);
    const int ns = shared_ws_slots();
    WsDevice &D = T.dev[device];
    int ready = 0;
    for (int i = 0; i < ns; ++i) { const WsSlot &w = D.slot[i]; if (w.ptr && w.bytes >= bytes) ready += 1; }
    for (int i = 0; i < ns && ready < concurrent; ++i) {
        WsSlot &w = D.slot[i];
        if (w.busy || (w.ptr && w.bytes >= bytes)) continue;
        if (w.ptr) { fr(device, w.ptr); w.ptr = nullptr; w.bytes = 0; }
        w.ptr = al(device, bytes);
        if (!w.ptr) break;
        w.bytes = bytes; ready += 1;
    }
    return ready;
}

void *ws_checkout(helm_op *op, size_t bytes, int *slot_out) {
    void *p = ws_table_checkout(g_shared_ws, op->device, bytes, slot_out, ws_dev_alloc, ws_dev_free);
    if (p) return p;
    *slot_out = -1;                        // every slot of this device is inside a solve (or the allocation failed): the handle's own buffer
    if (ensure_ws(op, bytes) != HELM_OK) return nullptr;
    return op->d_ws;
}
void ws_checkin(int slot) { ws_table_checkin(g_shared_ws, slot); }

// (tests, no GPU needed) the slot table with `ndev` logical devices and host memory: every device books `concurrent` slots of `bytes`, then
// `concurrent` leases are taken on every device at once.  Returns 0 when every lease is a booked slot of its own device, no lease needed a
// new allocation, one lease more than the table has slots is refused, and a device's bookings survive the other devices' bookings; a negative
// code says which of these failed.
static int g_selftest_allocs = 0;
static void *ws_host_alloc(int, size_t bytes) { g_selftest_allocs += 1; return malloc(bytes); }
static void ws_host_free(int, void *p) { free(p); }
extern "C" int helm_debug_ws_selftest(int ndev, int concurrent, long long bytes) {
    helm_tuning_refresh();
    if (ndev < 1 || concurrent < 1 || bytes < 1) return HELM_ERR_ARG;
    SharedWs T;
    int rc = 0;
    const int ns = shared_ws_slots();
    const int want = std::min(concurrent, ns);
    g_selftest_allocs = 0;
    for (int d = 0; d < ndev; ++d) if (ws_table_reserve(T, d, (size_t)bytes, concurrent, ws_host_alloc, ws_host_free) != want) rc = -1;
    if (g_selftest_allocs != ndev * want) rc = rc ? rc : -2;
    std::vector<int> leases;
    std::vector<void *> ptrs;
    for (int d = 0; d < ndev && !rc; ++d)
        for (int k = 0; k < want; ++k) {
            int lease = -1;
            void *p = ws_table_checkout(T, d, (size_t)bytes, &lease, ws_host_alloc, ws_host_free);
            if (!p || lease / WS_SLOTS_MAX != d) { rc = -3; break; }
            for (void *q : ptrs) if (q == p) rc = -4;                   // two leases on one buffer
            leases.push_back(lease); ptrs.push_back(p);
        }
    if (!rc && g_selftest_allocs != ndev * want) rc = -5;              // a lease after the booking allocated
    if (!rc && want == ns) { int lease = -1; if (ws_table_checkout(T, 0, (size_t)bytes, &lease, ws_host_alloc, ws_host_free)) rc = -6; }   // all of device 0's slots are out
    for (int l : leases) ws_table_checkin(T, l);
    if (!rc) { int lease = -1; if (!ws_table_checkout(T, ndev - 1, (size_t)bytes / 2 + 1, &lease, ws_host_alloc, ws_host_free) || g_selftest_allocs != ndev * want) rc = -7; else ws_table_checkin(T, lease); }
    for (auto &kv : T.dev) for (int i = 0; i < WS_SLOTS_MAX; ++i) if (kv.second.slot[i].ptr) free(kv.second.slot[i].ptr);
    return rc;
}
struct WsLease {
    int slot = -1; void *ptr = nullptr;
    WsLease(helm_op *op, size_t bytes) { ptr = ws_checkout(op, bytes, &slot); }
    ~WsLease() { ws_checkin(slot); }
};

// statuses of one right-hand side across the blocks / passes of a call, by severity: 0 converged < 3 at the fp64 floor (counted as solved)
// < 1 cap / stalled < 2 breakdown
// fault-injection hooks of the test-suite: honoured only when the process runs with HELM_TESTING=1 (read per call: the tests flip them)
inline int testing_hook(const char *name) {
    const char *t = getenv("HELM_TESTING");
    if (!t || atoi(t) == 0) return 0;
    const char *v = getenv(name);
    return v ? atoi(v) : 0;
}
inline int status_rank(int st) { return st == 0 ? 0 : (st == 3 ? 1 : (st == 1 ? 2 : 3)); }
inline int merge_status(int a, int b) { return status_rank(a) >= status_rank(b) ? a : b; }

// Sparse direct path (direct.hip): factor once per assembled operator, then per batch q' -> x by the multifrontal
// triangular solves and iterative refinement on the true residual q' - A x (stencil kernel) until rtol is met.
// sys2 != 0: the coupled two-field Eurus system [[M1, M2], [M3, M4]] on the stacked unknowns [u; v] (block ignored, two
// unknowns per cell in the elimination tree, factors kept in slot 1); rows_in = N or 2N rows of right-hand side per source.
int solve_block_direct(helm_op *op, int block, const cplx *dRHS, long long rhs_ld, long long row_off, cplx premul,
                       const cplx *sub, cplx *dXout, int nrhs, const helm_solve_opts &o, helm_solve_info *info,
                       int sys2 = 0, long long rows_in = 0, cplx *dUconj = nullptr) {
    // dUconj (single-block systems, N rows per right-hand side): the result is left there already conjugated -- the last
    // transpose of a pass writes conj(x), the residual kernel reads it conjugated -- and dXout is not written
    const long long N = op->N;
    const long long NV = sys2 ? 2 * N : N;
    const int cj = (dUconj && !sys2) ? 1 : 0;
    struct NvScope { helm_op *op; long long old; NvScope(helm_op *o_, long long nv) : op(o_), old(o_->Nv) { o_->Nv = nv; } ~NvScope() { op->Nv = old; } } nvscope(op, NV);
    const int slot = sys2 ? 1 : block;
    int rc;
    NdFactor *f = op->direct[slot];
    const bool need_factor = (f == nullptr);
    // factors enqueued by helm_prefactor on the handle's factor stream: everything this call launches comes after them
    if (op->pf_pending && f && op->pf_done) HIP_TRY(op, hipStreamWaitEvent(op->stream, op->pf_done, 0));
    // fault injection for the tests of the AUTO fallback
    if (testing_hook("HELM_ND_INJECT_FAILURE")) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: injected failure (HELM_ND_INJECT_FAILURE)");
    struct FactorOwner {       // a factor under construction is released on every early return
        NdFactor *p = nullptr;
        ~FactorOwner() { if (p) nd_free(p); }
    } fresh;
    if (need_factor) {
        f = new NdFactor();
        fresh.p = f;
        rc = nd_get_plan(op, helm_tuning_now().nd_leaf, sys2 ? 2 : 1, &f->pd);
        if (rc) return rc;
    }
    // per right-hand side: the node-major pipeline keeps q', x, the stored residual and the correction (4 N) beside the two front-vector
    // regions; the rhs-major path (coupled system) q', r and the solve scratch
    const long long per_rhs = sys2 ? nd_solve_ws_elems(f->pd->plan, 1) + 2 * NV : 4 * N + 2 * f->pd->plan.vregion;
    int Bmax = o.batch > 0 ? o.batch : 256;
    if (Bmax > nrhs) Bmax = nrhs;
    const double cap = helm_tuning_now().nd_ws_gb * 1e9;
    while (Bmax > 1 && (double)per_rhs * Bmax * sizeof(cplx) > cap) Bmax = (Bmax + 1) / 2;
    // factorisation scratch sits behind the solve scratch (they are live together when the forward elimination of the first batch runs
    // beside the factorisation on a second stream, below)
    const long long fws = need_factor ? nd_factor_ws_elems(f->pd->plan) : 0LL;
    const long long ws_elems = per_rhs * Bmax + fws;
    WsLease lease(op, (size_t)ws_elems * sizeof(cplx));
    if (!lease.ptr) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: cannot allocate %.1f GB of scratch", ws_elems * 16e-9);
    cplx *ws_factor = (cplx *)lease.ptr + per_rhs * Bmax;
    if (sys2) {     // the coupled system is factored row-equilibrated (its v rows are orders of magnitude smaller than its u rows,
        rc = helm_launch_rowscaled_system(op);      // which would mislead the magnitude-based pivoting): A_s = D A, A_s x = D q'
        if (rc) return rc;
    }
    // node-major pipeline: the forward elimination of the first batch may run beside the factorisation (HELM_ND_OVERLAP_NM)
    // -- measured on the 16-frequency job: 44.9 -> 43.8 ms per work item; the factorisation itself stretches from 17.6 to 22.2 ms under
    // the competing launches but 5 ms of forward pass disappear behind it.  Not while per-launch profiling is on: HIP events around
    // kernels that share the chip with another stream measure the sharing, not the kernel (HELM_ND_OVERLAP_NM=2 forces it anyway).
    const int overlap_nm = helm_tuning_now().nd_overlap;
    const bool nm_overlap = need_factor && (overlap_nm == 2 || (overlap_nm == 1 && !op->profiling)) && !sys2;
    bool factor_pending = need_factor;
    if (need_factor && !nm_overlap) {
        hipEvent_t f0, f1;
        HIP_TRY(op, hipEventCreate(&f0)); HIP_TRY(op, hipEventCreate(&f1));
        hipEventRecord(f0, op->stream);
        rc = nd_factor(op, block, f, ws_factor, sys2 ? op->d_S : nullptr);
        hipEventRecord(f1, op->stream);
        hipEventSynchronize(f1);
        float ms = 0.f; hipEventElapsedTime(&ms, f0, f1);
        hipEventDestroy(f0); hipEventDestroy(f1);
        if (rc) return rc;
        op->direct[slot] = f; fresh.p = nullptr;
        op->timing.factor_ms += ms;
        factor_pending = false;
    }
    if (factor_pending && !op->side_stream) {
        // lowest priority: the forward pass that runs beside the factorisation must not delay the factorisation's chain of small launches
        op->side_stream = helm_stream_acquire(op->device, -1);
        if (!op->side_stream) HELM_FAIL(op, HELM_ERR_DEVICE, "hipStreamCreate failed");
    }
    rc = ensure_part(op, Bmax);
    if (rc) return rc;
    const int nblk = std::max(2 * helm_apply_num_blocks(op), helm_vec_num_blocks(op));
    char *ptail = (char *)op->d_part + (size_t)Bmax * 4 * nblk * sizeof(double);
    double *d_aux = (double *)ptail;
    char *htail = (char *)op->h_scal + (size_t)op->scal_cap * sizeof(RhsScal);
    double *h_aux = (double *)htail;
    const int max_refine = sys2 ? 40 : 10;      // passes stop earlier when the residual stalls
    static const int nd_debug = getenv("HELM_ND_DEBUG") ? atoi(getenv("HELM_ND_DEBUG")) : 0;
    int unconverged = 0;
    // Node-major pipeline (single-block systems): the right-hand sides are transposed once on the way in (fused with premul /
    // the norm), stay [cell][rhs] through solve, true residual and refinement, and are transposed once on the way out.
    const bool nm = !sys2;        // (single-block systems; the coupled two-field system keeps its vectors rhs-major, below)
    // HELM_NODE_MAJOR (both buffers in the reference's (N, nrhs) layout; helm_solve_device only passes it for one batch of a single-block system):
    // the right-hand sides are used where they lie -- premul moves to the output, u = conj(premul A^-1 q), the relative residual does not see
    // it -- ||q||^2 comes out of the first residual launch, and the wavefield is written by the launch that checks it
    const bool native_nm = nm && (o.flags & HELM_NODE_MAJOR) == HELM_NODE_MAJOR && nrhs <= Bmax && !sub && row_off == 0 && dUconj;
    if ((o.flags & HELM_NODE_MAJOR) && !native_nm) HELM_FAIL(op, HELM_ERR_STATE, "direct solver: node-major buffers reached a path that cannot take them");
    for (int first = 0; nm && first < nrhs; first += Bmax) {
        const int n = std::min(Bmax, nrhs - first);
        const NdPlan &P = f->pd->plan;
        cplx *Qt = (cplx *)lease.ptr, *Xt = Qt + (long long)Bmax * N, *Rt = Xt + (long long)Bmax * N, *Dt = Rt + (long long)Bmax * N, *arenaV = Dt + (long long)Bmax * N;
        (void)P;
        if (native_nm) Qt = const_cast<cplx *>(dRHS);             // read only from here on (the residual is stored to Rt, never over q)
        cplx *xout = cj ? dUconj + (long long)first * N : dXout + (long long)first * N;
        const cplx *rhs_b = dRHS + (long long)first * rhs_ld;
        const cplx *sub_b = sub ? sub + (long long)first * N : nullptr;
        const cplx *planes = op->d_C + (long long)block * op->nplanes * N;
        int *d_cols = (int *)(ptail + (size_t)Bmax * 2 * sizeof(double));
        int *h_cols = (int *)(htail + (size_t)op->scal_cap * 2 * sizeof(double));
        int nb_part = 0;
        if (!native_nm) {
            rc = nd_prep_transpose_norm(op, rhs_b, rhs_ld, row_off, premul, sub_b, Qt, N, n, (double *)op->d_part, nblk, &nb_part);
            if (rc) return rc;
            helm_launch_fin_ex(op, FIN_NORM, n, nb_part, nullptr, d_aux + n);           // ||q'||^2
        }
        bool have_qnorm = !native_nm;
        NdResidExtra rex;
        if (native_nm) { rex.Uout = dUconj; rex.ldu = n; rex.oscale = premul; }
        // direct output (helm_tuning.nd_direct_out; full-width batches, whose residual kernel can read the caller's array): the back substitution writes
        // u = conj(premul x) into dUconj itself and the residual launch below stores nothing -- x_in_u until a refinement pass needs x back in Xt
        NdDirectOut dout;
        // (not when the caller solves in place, dU == dRHS: the back substitution would overwrite q before the residual launch has read it -- that call takes the
        // path of the narrow batches, where the residual launch reads q[cell] and writes u[cell] in the same thread)
        bool x_in_u = native_nm && n > 128 && helm_tuning_now().nd_direct_out != 0 && (const void *)Qt != (const void *)dUconj;
        if (x_in_u) { dout.U = dUconj; dout.oscale = premul; }
        if (factor_pending) {
            float fms = 0.f;
            rc = nd_factor_solve_nm(op, block, f, ws_factor, nullptr, Qt, Xt, n, arenaV, op->side_stream, &fms, x_in_u ? &dout : nullptr);
            if (rc) return rc;
            op->direct[slot] = f; fresh.p = nullptr;
            op->timing.factor_ms += fms;
            factor_pending = false;
        } else {
            rc = nd_solve_nm(op, f, Qt, Xt, n, arenaV, x_in_u ? &dout : nullptr);
            if (rc) return rc;
        }
        auto recover_x = [&]() -> int {            // x of every cell back in Xt (the residual of what follows is evaluated from Xt again, ||q||^2 unscaled)
            if (!x_in_u) return HELM_OK;
            x_in_u = false; have_qnorm = false;
            return nd_recover_x(op, dUconj, Xt, (long long)n * N, premul);
        };
        // where the right-hand sides of this batch can be nonzero at all (the flags of the sparse forward pass just run on Qt): the residual
        // launches read q only there -- every later evaluation too, Qt does not change
        rex.qmask = nd_rhs_mask(op, f);
        std::vector<double> relres(n, 0.0), qq(n, 0.0);
        std::vector<int> extra_solves(n, 0);
        double prev_worst = 0.0;
        // every pass ends with the TRUE residual q' - A x of the vector that is returned (norms only); q' is kept for that
        auto true_residual_norms = [&]() -> int {
            rex.qnorm = have_qnorm ? 0 : 1;
            int r1;
            if (x_in_u) {
                NdResidExtra ru = rex; ru.Uout = nullptr; ru.xin_is_u = 1;
                r1 = nd_resid_nm(op, planes, dUconj, n, Qt, n, nullptr, n, 0, nullptr, (double *)op->d_part, nblk, &nb_part, &ru);
            } else
            r1 = nd_resid_nm(op, planes, Xt, n, Qt, n, nullptr, n, 0, nullptr, (double *)op->d_part, nblk, &nb_part, (native_nm || rex.qmask) ? &rex : nullptr);
            if (r1) return r1;
            helm_launch_fin_ex(op, have_qnorm ? FIN_NORM : FIN_NORM2, n, nb_part, nullptr, d_aux);
            have_qnorm = true;
            HIP_TRY(op, hipMemcpyAsync(h_aux, d_aux, 2 * n * sizeof(double), hipMemcpyDeviceToHost, op->stream));
            HIP_TRY(op, hipStreamSynchronize(op->stream));
            for (int b = 0; b < n; ++b) { qq[b] = h_aux[n + b]; relres[b] = qq[b] > 0 ? sqrt(h_aux[b] / qq[b]) : 0.0; }
            return HELM_OK;
        };
        rc = true_residual_norms();
        if (rc) return rc;
        for (int round = 0; ; ++round) {
            bool all_ok = true;
            double worst = 0.0;
            for (int b = 0; b < n; ++b) {
                if (!(relres[b] <= o.rtol)) all_ok = false;
                if (!(relres[b] <= worst)) worst = relres[b];       // NaN-propagating max
            }
            if (nd_debug) fprintf(stderr, "[helm direct] pass %d: worst true relres %.3e\n", round + 1, worst);
            const bool stalled = round > 0 && !(worst < 0.5 * prev_worst);
            prev_worst = worst;
            if (all_ok || round >= max_refine || stalled) break;
            // r = q' - A x stored (Rt), dx = A^-1 r, x += dx
            rc = recover_x();
            if (rc) return rc;
            rc = nd_resid_nm(op, planes, Xt, n, Qt, n, nullptr, n, 1, Rt, (double *)op->d_part, nblk, &nb_part);
            if (rc) return rc;
            std::vector<int> bad;
            for (int b = 0; b < n; ++b) if (!(relres[b] <= o.rtol)) bad.push_back(b);
            const int k = (int)bad.size();
            if (k < n / 2) {
                // a minority missed rtol: their residual columns are packed to a narrower batch, solved, and the corrections scattered back
                for (int j = 0; j < k; ++j) h_cols[j] = bad[j];
                HIP_TRY(op, hipMemcpyAsync(d_cols, h_cols, k * sizeof(int), hipMemcpyHostToDevice, op->stream));
                cplx *Rp = Dt, *Dp = Dt + N * k;              // k < n / 2: both fit the correction buffer
                rc = nd_pack_cols(op, Rt, n, d_cols, k, Rp, N);
                if (rc) return rc;
                rc = nd_solve_nm(op, f, Rp, Dp, k, arenaV);
                if (rc) return rc;
                rc = nd_scatter_add_cols(op, Xt, n, d_cols, k, Dp, N);
                if (rc) return rc;
                for (int j = 0; j < k; ++j) extra_solves[bad[j]] += 1;
            } else {
                rc = nd_solve_nm(op, f, Rt, Dt, n, arenaV);
                if (rc) return rc;
                nd_axpy_one(op, Xt, Dt, (long long)n * N, 0);
                for (int b = 0; b < n; ++b) extra_solves[b] += 1;
            }
            rc = true_residual_norms();
            if (rc) return rc;
        }
        if (!native_nm) {           // (node-major callers: the last residual launch has written conj(premul x) already)
            rc = nd_transpose_out(op, Xt, N, n, xout, cj);
            if (rc) return rc;
        }
        // Right-hand sides refinement left above rtol: is the residual at the floor fp64 allows (relres ~ eps || |A||x| + |q| || / ||q||,
        // see the coupled-system branch below)?  Evaluated node-major with |planes| and |x|; ||.|| of the sum bounded by the sum of norms.
        std::vector<int> at_floor(n, 0);
        {
            bool any = false;
            for (int b = 0; b < n; ++b) if (!(relres[b] <= o.rtol)) any = true;
            const size_t pbytes = (size_t)op->nplanes * N * sizeof(cplx);
            cplx *absP = any ? (cplx *)helm_pool_alloc(op->device, pbytes) : nullptr;
            if (any && absP) {
                rc = recover_x();
                if (!rc) rc = helm_launch_abs(op, planes, absP, (long long)op->nplanes * N, 1.0);
                if (!rc) rc = helm_launch_abs(op, Xt, Dt, (long long)n * N, 1.0);
                if (!rc && hipMemsetAsync(Rt, 0, (size_t)n * N * sizeof(cplx), op->stream) != hipSuccess) rc = HELM_ERR_DEVICE;
                if (!rc) rc = nd_resid_nm(op, absP, Dt, n, Rt, n, nullptr, n, 0, nullptr, (double *)op->d_part, nblk, &nb_part);      // -|A||x|
                if (!rc) {
                    helm_launch_fin_ex(op, FIN_NORM, n, nb_part, nullptr, d_aux);
                    if (hipMemcpyAsync(h_aux, d_aux, n * sizeof(double), hipMemcpyDeviceToHost, op->stream) != hipSuccess || hipStreamSynchronize(op->stream) != hipSuccess) rc = HELM_ERR_DEVICE;
                }
                helm_pool_free(op->device, absP, pbytes);
                if (rc) return rc;
                for (int b = 0; b < n; ++b) {
                    const double fl = qq[b] > 0 ? 1.1102230246251565e-16 * (sqrt(h_aux[b]) + sqrt(qq[b])) / sqrt(qq[b]) : 0.0;
                    if (!(relres[b] <= o.rtol) && relres[b] <= 8.0 * fl) at_floor[b] = 1;
                    if (nd_debug && !(relres[b] <= o.rtol)) fprintf(stderr, "[helm direct] rhs %d: relres %.3e, fp64 floor %.3e\n", first + b, relres[b], fl);
                }
            }
        }
        const int inject_stall = testing_hook("HELM_ND_INJECT_STALL");
        for (int b = 0; b < n; ++b) {
            const bool ok = (relres[b] <= o.rtol * 1.0000001 || at_floor[b]) && !(first + b < inject_stall);
            if (!ok) unconverged += 1;
            if (info) {
                helm_solve_info &I = info[first + b];
                I.iterations += 1 + extra_solves[b]; I.method = HELM_DIRECT;
                I.relres = std::max(I.relres, relres[b]);
                I.status = merge_status(I.status, ok ? (at_floor[b] ? 3 : 0) : 1);
            }
        }
        HIP_TRY(op, hipStreamSynchronize(op->stream));
    }
    for (int first = 0; !nm && first < nrhs; first += Bmax) {
        const int n = std::min(Bmax, nrhs - first);
        cplx *q = (cplx *)lease.ptr, *r = q + (long long)Bmax * NV, *nws = q + 2LL * Bmax * NV;
        cplx *x = cj ? dUconj + (long long)first * NV : dXout + (long long)first * NV;
        const cplx *rhs_b = dRHS + (long long)first * rhs_ld;
        const cplx *sub_b = sub ? sub + (long long)first * N : nullptr;
        if (sys2) {
            HIP_TRY(op, hipMemsetAsync(q, 0, (size_t)n * NV * sizeof(cplx), op->stream));
            for (int half = 0; half < (rows_in == 2 * N ? 2 : 1); ++half) {
                rc = helm_launch_prep_rhs_ex(op, rhs_b, rhs_ld, half * N, premul, nullptr, q, NV, half * N, n);
                if (rc) return rc;
            }
            helm_launch_norm2(op, q, n);
        } else {
            rc = helm_launch_prep_rhs_norm(op, rhs_b, rhs_ld, row_off, premul, sub_b, q, n);      // q' and the partials of ||q'||^2
            if (rc) return rc;
        }
        helm_launch_fin_ex(op, FIN_NORM, n, helm_vec_num_blocks(op), nullptr, d_aux + n);
        const cplx *xin = q;
        if (sys2) {
            for (int half = 0; half < 2; ++half) {
                rc = helm_launch_prep_rhs_rs(op, q, NV, half * N, cmake(1.0, 0.0), op->d_rs + half * N, x, NV, half * N, n);
                if (rc) return rc;
            }
            xin = x;
        }
        rc = nd_solve(op, f, xin, x, n, nws, cj);           // (the factors exist: the coupled system is factored before its first batch, above)
        if (rc) return rc;
        std::vector<double> relres(n, 0.0);
        std::vector<int> extra_solves(n, 0);
        double prev_worst = 0.0;
        for (int round = 0; ; ++round) {
            if (sys2) {
                rc = launch_sys2_apply(op, true, 0, x, r, q, n, EPI_RESID, nullptr);
                if (rc) return rc;
                helm_launch_fin_ex(op, FIN_NORM, n, 2 * helm_apply_num_blocks(op), nullptr, d_aux);
            } else {
                ApplyArgs a = ApplyArgs();
                a.planes = op->d_C + (long long)block * op->nplanes * N; a.X = x; a.Y = r; a.W = q; a.ld = N; a.nrhs = n;
                a.scaled = 0; a.adjoint = 0; a.epi = EPI_RESID; a.scal = nullptr; a.part = (double *)op->d_part;
                a.xmode = cj ? 3 : 0;
                rc = helm_launch_apply(op, a);
                if (rc) return rc;
                helm_launch_fin_ex(op, FIN_NORM, n, helm_apply_num_blocks(op), nullptr, d_aux);
            }
            HIP_TRY(op, hipMemcpyAsync(h_aux, d_aux, 2 * n * sizeof(double), hipMemcpyDeviceToHost, op->stream));
            HIP_TRY(op, hipStreamSynchronize(op->stream));
            bool all_ok = true;
            double worst = 0.0;
            for (int b = 0; b < n; ++b) {
                const double qq = h_aux[n + b];
                relres[b] = qq > 0 ? sqrt(h_aux[b] / qq) : 0.0;
                if (!(relres[b] <= o.rtol)) all_ok = false;
                if (!(relres[b] <= worst)) worst = relres[b];       // NaN-propagating max
            }
            if (nd_debug) fprintf(stderr, "[helm direct] pass %d: worst true relres %.3e\n", round + 1, worst);
            // refinement contracts by the accuracy of the factorisation per pass; give up when it has stopped doing so
            const bool stalled = round > 0 && !(worst < 0.5 * prev_worst);
            prev_worst = worst;
            if (all_ok || round >= max_refine || stalled) break;
            // refine only the right-hand sides that missed rtol when they are a minority: their residual columns are packed to
            // the front of r (whole 16 MB rows), solved as a narrower batch and added back
            std::vector<int> bad;
            for (int b = 0; b < n; ++b) if (!(relres[b] <= o.rtol)) bad.push_back(b);
            const int k = (int)bad.size();
            if (k < n / 2) {
                for (int j = 0; j < k; ++j)
                    if (bad[j] != j) HIP_TRY(op, hipMemcpyAsync(r + (long long)j * NV, r + (long long)bad[j] * NV, (size_t)NV * sizeof(cplx), hipMemcpyDeviceToDevice, op->stream));
                if (sys2) { rc = helm_launch_rowscale_inplace(op, r, op->d_rs, NV, k); if (rc) return rc; }
                rc = nd_solve(op, f, r, r, k, nws);
                if (rc) return rc;
                for (int j = 0; j < k; ++j) nd_axpy_one(op, x + (long long)bad[j] * NV, r + (long long)j * NV, NV, cj);
                for (int j = 0; j < k; ++j) extra_solves[bad[j]] += 1;
            } else {
                if (sys2) { rc = helm_launch_rowscale_inplace(op, r, op->d_rs, NV, n); if (rc) return rc; }
                rc = nd_solve(op, f, r, r, n, nws);       // dx = A^-1 r
                if (rc) return rc;
                nd_axpy_one(op, x, r, (long long)n * NV, cj);
                for (int b = 0; b < n; ++b) extra_solves[b] += 1;
            }
        }
        // Coupled system: where refinement has stalled above rtol, is that the floor of fp64 itself?  The residual of ANY fp64 vector x
        // near the solution carries rounding of size eps (|A||x| + |q|) componentwise, so ||r|| / ||q|| cannot be pushed below
        // ~ eps || |A||x| + |q| || / ||q||, whatever the solver (a backward-stable sparse LU lands there too).  Evaluated with the
        // stencil kernel on |planes| and |x|; right-hand sides within 8x of it are reported as status 3, not as failures.
        std::vector<int> at_floor(n, 0);
        if (sys2) {
            bool any = false;
            for (int b = 0; b < n; ++b) if (!(relres[b] <= o.rtol)) any = true;
            const size_t pbytes = (size_t)36 * N * sizeof(cplx);
            cplx *absP = any ? (cplx *)helm_pool_alloc(op->device, pbytes) : nullptr;
            if (any && absP) {
                cplx *absx = r, *negq = nws, *yy = nws + (long long)n * NV;
                rc = helm_launch_abs(op, op->d_C, absP, 36LL * N, 1.0);
                if (!rc) rc = helm_launch_abs(op, x, absx, (long long)n * NV, 1.0);
                if (!rc) rc = helm_launch_abs(op, q, negq, (long long)n * NV, -1.0);
                if (!rc) rc = launch_sys2_apply(op, true, 0, absx, yy, negq, n, EPI_RESID, nullptr, absP);      // -(|q| + |A||x|)
                if (!rc) {
                    helm_launch_fin_ex(op, FIN_NORM, n, 2 * helm_apply_num_blocks(op), nullptr, d_aux);
                    if (hipMemcpyAsync(h_aux, d_aux, n * sizeof(double), hipMemcpyDeviceToHost, op->stream) != hipSuccess || hipStreamSynchronize(op->stream) != hipSuccess) rc = HELM_ERR_DEVICE;
                }
                helm_pool_free(op->device, absP, pbytes);
                if (rc) return rc;
                for (int b = 0; b < n; ++b) {
                    // ||q||^2 was left in h_aux[n + b] by the residual rounds
                    const double qq = h_aux[n + b];
                    const double fl = qq > 0 ? 1.1102230246251565e-16 * sqrt(h_aux[b] / qq) : 0.0;
                    if (!(relres[b] <= o.rtol) && relres[b] <= 8.0 * fl) at_floor[b] = 1;
                    if (nd_debug) fprintf(stderr, "[helm direct] rhs %d: relres %.3e, fp64 floor %.3e\n", first + b, relres[b], fl);
                }
            }
        }
        // fault injection for the tests of the partial fallback: report the first k right-hand sides as stalled
        const int inject_stall = testing_hook("HELM_ND_INJECT_STALL");
        for (int b = 0; b < n; ++b) {
            const bool ok = (relres[b] <= o.rtol * 1.0000001 || at_floor[b]) && !(first + b < inject_stall);
            if (!ok) unconverged += 1;
            if (info) {
                helm_solve_info &I = info[first + b];
                I.iterations += 1 + extra_solves[b]; I.method = HELM_DIRECT;
                I.relres = std::max(I.relres, relres[b]);
                I.status = merge_status(I.status, ok ? (at_floor[b] ? 3 : 0) : 1);
            }
        }
        HIP_TRY(op, hipStreamSynchronize(op->stream));
    }
    return unconverged;
}

// sys2 != 0: the coupled two-field Eurus system (block ignored, vectors [u; v] of length 2N, rows_in = N or 2N rows of
// right-hand side per source; sub unused); dXout then holds 2N values per right-hand side.
int solve_block(helm_op *op, int block, const cplx *dRHS, long long rhs_ld, long long row_off, cplx premul,
                const cplx *sub, cplx *dXout, int nrhs, const helm_solve_opts &o, helm_solve_info *info,
                int sys2 = 0, long long rows_in = 0, cplx *dUconj = nullptr, bool *wrote_u = nullptr) {
    // dUconj / wrote_u: the direct path can leave conj(x) straight in the caller's output (then *wrote_u = true and dXout is
    // untouched); every other path fills dXout
    const long long N = op->N;
    if (wrote_u) *wrote_u = false;
    if (o.method == HELM_DIRECT) {
        if (op->ny > 0) HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "the direct solver is 2-D only");
        const int rcd = solve_block_direct(op, block, dRHS, rhs_ld, row_off, premul, sub, dXout, nrhs, o, info, sys2, rows_in, dUconj);
        if (rcd >= 0 && wrote_u && dUconj && !sys2) *wrote_u = true;
        return rcd;
    }
    // AUTO: the sparse direct path wherever it applies (2-D single-block systems that fit), else / on failure the
    // multigrid-preconditioned Krylov path below
    if (o.method == HELM_AUTO && op->ny == 0 && !op->direct_failed) {
        if (helm_tuning_now().auto_direct != 0) {
            std::vector<helm_solve_info> saved;
            if (info) saved.assign(info, info + nrhs);
            // what THIS call's direct pass found, apart from what earlier blocks of the same solve left in `info` (stacked Eurus: block 3, then 0)
            std::vector<helm_solve_info> cur(nrhs);
            for (helm_solve_info &c : cur) { c.iterations = 0; c.status = 0; c.restarts = 0; c.method = HELM_DIRECT; c.relres = 0.0; }
            const int rc = solve_block_direct(op, block, dRHS, rhs_ld, row_off, premul, sub, dXout, nrhs, o, cur.data(), sys2, rows_in, dUconj);
            auto merge_cur = [&](int b) {
                helm_solve_info &I = info[b];
                I.iterations += cur[b].iterations; I.method = cur[b].method;
                I.relres = std::max(I.relres, cur[b].relres);
                I.status = merge_status(I.status, cur[b].status);
            };
            if (rc >= 0 && info && !(rc > 0 && !sys2 && rc < nrhs)) for (int b = 0; b < nrhs; ++b) merge_cur(b);
            if (rc == 0) { if (wrote_u && dUconj && !sys2) *wrote_u = true; return 0; }
            // the coupled system has no better fallback: row-equilibrated CGNR needs 10^4-10^5 iterations and meets the same
            // fp64 floor of the true residual, so right-hand sides that stalled above rtol are reported as such
            if (rc > 0 && sys2) return rc;
            op->direct_failed = true;
            // the factors are of no further use to this handle: give the (multi-GB) storage back now, not at the next assemble
            { const int slot = sys2 ? 1 : block; nd_free(op->direct[slot]); op->direct[slot] = nullptr; }
            if (rc > 0 && info && !sys2 && rc < nrhs) {
                // some right-hand sides stalled above rtol: only those go to the Krylov path (packed into a narrower batch);
                // the converged ones keep the direct result
                // (status 1 or 2 of the direct pass; those at the fp64 floor -- status 3 -- are solved: no Krylov method gets below it either)
                std::vector<int> bad;
                for (int b = 0; b < nrhs; ++b) { if (cur[b].status == 1 || cur[b].status == 2) bad.push_back(b); else merge_cur(b); }
                const int k = (int)bad.size();
                const size_t colb = (size_t)N * sizeof(cplx);
                cplx *tR = (cplx *)helm_pool_alloc(op->device, (size_t)k * colb), *tX = (cplx *)helm_pool_alloc(op->device, (size_t)k * colb);
                cplx *tS = sub ? (cplx *)helm_pool_alloc(op->device, (size_t)k * colb) : nullptr;
                auto release = [&]() { hipStreamSynchronize(op->stream); helm_pool_free(op->device, tR, (size_t)k * colb); helm_pool_free(op->device, tX, (size_t)k * colb);
                                       helm_pool_free(op->device, tS, (size_t)k * colb); };
                if (!tR || !tX || (sub && !tS)) { release(); HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc failed"); }
                for (int j = 0; j < k; ++j) {
                    hipMemcpyAsync(tR + (long long)j * N, dRHS + (long long)bad[j] * rhs_ld + row_off, colb, hipMemcpyDeviceToDevice, op->stream);
                    if (sub) hipMemcpyAsync(tS + (long long)j * N, sub + (long long)bad[j] * N, colb, hipMemcpyDeviceToDevice, op->stream);
                }
                std::vector<helm_solve_info> ki(k);
                for (int j = 0; j < k; ++j) ki[j] = saved[bad[j]];
                const int rck = solve_block(op, block, tR, N, 0, premul, tS, tX, k, o, ki.data());
                if (rck < 0) { release(); return rck; }
                const bool cj = dUconj != nullptr;
                for (int j = 0; j < k; ++j) {
                    info[bad[j]] = ki[j];
                    if (cj) helm_launch_finish_ex(op, tX, N, (long long)j * N, dUconj + (long long)bad[j] * N, N, 0, 1);
                    else hipMemcpyAsync(dXout + (long long)bad[j] * N, tX + (long long)j * N, colb, hipMemcpyDeviceToDevice, op->stream);
                }
                release();
                if (wrote_u && cj) *wrote_u = true;
                return rck;
            }
            if (info) std::copy(saved.begin(), saved.end(), info);
        }
    }
    const long long NV = sys2 ? 2 * N : N;
    NvGuard guard(op, NV);
    { const int rcs = helm_ensure_scaled(op); if (rcs) return rcs; }
    int Bmax = o.batch > 0 ? o.batch : 16;
    if (Bmax > nrhs) Bmax = nrhs;
    int rc = ensure_ws(op, (size_t)11 * Bmax * NV * sizeof(cplx));
    if (rc) return rc;
    if (sys2) {
        // the coupled TTI system is only tractable by the normal-equations method on the row-equilibrated system
        if (o.method == HELM_MG || o.method == HELM_BICGSTAB) HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "the coupled TTI system (eps != delta) is solved with row-equilibrated CGNR only (method 'auto' or 'cgnr')");
        rc = helm_launch_rowscaled_system(op);
        if (rc) return rc;
    }
    rc = ensure_part(op, Bmax);
    if (rc) return rc;
    // preconditioner choice: multigrid for the main block when asked for (or AUTO on Eurus, where it is validated)
    bool use_mg = false;
    const int auto_mg3 = helm_tuning_now().auto_mg3;
    const bool mg3_ok = op->ny > 0 && (o.method == HELM_MG || (o.method == HELM_AUTO && auto_mg3 && std::min(op->nz, std::min(op->ny, op->nx)) >= 24));
    if (!sys2 && block == 0 && (mg3_ok || (op->ny == 0 && (o.method == HELM_MG || (o.method == HELM_AUTO && std::min(op->nz, op->nx) >= 32))))) {
        op->mg3_rhs_hint = nrhs;
        rc = mg_setup(op, Bmax);
        if (rc == HELM_OK) use_mg = true;
        else if (o.method == HELM_MG) return rc;
    }
    // (an iteration of the layer-preserving 3-D cycle costs tens of milliseconds and ten of them are a whole solve: poll after every one)
    auto pick_check_every = [&]() { return o.check_every > 0 ? o.check_every : (use_mg && op->ny > 0 && mg3_is_layer_preserving(op) ? 1 : (use_mg ? 10 : 50)); };
    int check_every = pick_check_every();
    int unconverged = 0;
    for (int first = 0; first < nrhs; first += Bmax) {
        const int n = std::min(Bmax, nrhs - first);
        Batch B;
        B.nrhs = n;
        cplx *base = (cplx *)op->d_ws;
        const long long vs = (long long)Bmax * NV;
        B.sys2 = sys2 != 0;
        B.nba = (sys2 ? 2 : 1) * helm_apply_num_blocks(op);
        B.w.x = base; B.w.r = base + vs; B.w.r0 = base + 2 * vs; B.w.p = base + 3 * vs; B.w.v = base + 4 * vs;
        B.w.s = base + 5 * vs; B.w.t = base + 6 * vs; B.bscaled = base + 7 * vs;
        cplx *qprime = base + 8 * vs;
        B.phat = base + 9 * vs; B.shat = base + 10 * vs;
        B.pre = use_mg;
        B.bbar = use_mg ? qprime : B.bscaled;
        B.planes = use_mg ? op->d_C + (long long)block * op->nplanes * N : op->d_Cs + (long long)block * op->nplanes * N;
        const int nblk = std::max(2 * helm_apply_num_blocks(op), helm_vec_num_blocks(op));
        char *ptail = (char *)op->d_part + (size_t)Bmax * 4 * nblk * sizeof(double);
        B.d_aux = (double *)ptail; B.d_mask = (int *)(ptail + (size_t)Bmax * 2 * sizeof(double));
        char *htail = (char *)op->h_scal + (size_t)op->scal_cap * sizeof(RhsScal);
        B.h_aux = (double *)htail; B.h_mask = (int *)(htail + (size_t)op->scal_cap * 2 * sizeof(double));

        const cplx *rhs_b = dRHS + (long long)first * rhs_ld;
        const cplx *sub_b = sub ? sub + (long long)first * N : nullptr;
        // q' = premul*rhs - sub (unscaled), ||q'||^2 -> aux[n..2n)
        if (sys2) {
            HIP_TRY(op, hipMemsetAsync(qprime, 0, (size_t)n * NV * sizeof(cplx), op->stream));
            HIP_TRY(op, hipMemsetAsync(B.bscaled, 0, (size_t)n * NV * sizeof(cplx), op->stream));
            for (int half = 0; half < (rows_in == 2 * N ? 2 : 1); ++half) {
                rc = helm_launch_prep_rhs_ex(op, rhs_b, rhs_ld, half * N, premul, nullptr, qprime, NV, half * N, n);
                if (rc) return rc;
                rc = helm_launch_prep_rhs_rs(op, rhs_b, rhs_ld, half * N, premul, op->d_rs + half * N, B.bscaled, NV, half * N, n);
                if (rc) return rc;
            }
        } else {
            rc = helm_launch_prep_rhs(op, rhs_b, rhs_ld, row_off, premul, sub_b, qprime, n);
            if (rc) return rc;
        }
        helm_launch_norm2(op, qprime, n);
        helm_launch_fin_ex(op, FIN_NORM, n, helm_vec_num_blocks(op), nullptr, B.d_aux + n);
        // scaled system start
        if (sys2) {
            rc = helm_launch_krylov_init(op, B.bscaled, B.w, n, o.rtol * 0.5);
            if (rc) return rc;
        } else {
            VecPtrs w = B.w;
            w.t = B.bscaled;   // init writes the scaled right-hand side through w.t
            // NB: row offset is applied by giving prep a shifted base pointer
            rc = helm_launch_bicg_init(op, block, rhs_b + row_off, rhs_ld, premul, sub_b, w, n, o.rtol * 0.5);
            if (rc) return rc;
            if (use_mg) {      // iterate on the unscaled system A (M^-1 y) = q'
                rc = helm_launch_krylov_init(op, qprime, B.w, n, o.rtol * 0.9);
                if (rc) return rc;
            }
        }
        std::vector<int> restarts(n, 0);
        std::vector<int> method_used(n, (o.method == HELM_CGNR || sys2) ? HELM_CGNR : (use_mg ? HELM_MG : HELM_BICGSTAB));
        std::vector<int> total_iters(n, 0);
        std::vector<double> relres(n, 0.0);
        const int max_refine = 3;
        for (int round = 0; round <= max_refine; ++round) {
            if (o.method == HELM_CGNR || sys2) {
                rc = download_scal(op, n);
                if (rc) return rc;
                bool any = false;
                for (int b = 0; b < n; ++b) { B.h_mask[b] = (op->h_scal[b].status == ST_ACTIVE); any = any || B.h_mask[b]; }
                if (any) { rc = run_cgnr(op, block, B, o.maxit, check_every); if (rc) return rc; }
            } else {
                // in AUTO mode a preconditioned run that has not converged after 5000 iterations is handed to CGNR
                int cap = (use_mg && o.method == HELM_AUTO) ? std::min(o.maxit, 5000) : o.maxit;
                // the layer-preserving 3-D hierarchy needs tens of iterations; if it has not converged after HELM_MG3_KEEP_CAP (300) the
                // frequency retreats to the standard cycle and goes on from the iterates reached
                const bool keep3 = use_mg && op->ny > 0 && mg3_is_layer_preserving(op);
                // (first round only: the retreat below is what the cap is for, and it is taken there)
                if (keep3 && round == 0) cap = std::min(cap, getenv("HELM_MG3_KEEP_CAP") ? std::max(1, atoi(getenv("HELM_MG3_KEEP_CAP"))) : 300);
                rc = run_bicgstab(op, block, B, cap, check_every, 25, restarts);
                if (rc) return rc;
                if (keep3 && round == 0) {
                    rc = download_scal(op, n);
                    if (rc) return rc;
                    bool any = false;
                    for (int b = 0; b < n; ++b) {
                        const int st = op->h_scal[b].status;
                        B.h_mask[b] = (st == ST_BREAKDOWN || st == ST_FROZEN);
                        any = any || B.h_mask[b];
                    }
                    if (any) {
                        rc = mg3_retreat(op, Bmax);
                        if (rc) return rc;
                        check_every = pick_check_every();          // the standard cycle needs hundreds of iterations: poll every 10, not every one
                        rc = restart_masked(op, block, B);
                        if (rc) return rc;
                        rc = run_bicgstab(op, block, B, o.method == HELM_AUTO ? std::min(o.maxit, 5000) : o.maxit, check_every, 25, restarts);
                        if (rc) return rc;
                    }
                }
                if (o.method == HELM_AUTO && use_mg && round == 0 && op->ny == 0) {      // (no adjoint apply, hence no CGNR, in 3-D)
                    rc = download_scal(op, n);
                    if (rc) return rc;
                    bool any = false;
                    for (int b = 0; b < n; ++b) {
                        const int st = op->h_scal[b].status;
                        B.h_mask[b] = (st == ST_BREAKDOWN || st == ST_FROZEN);
                        any = any || B.h_mask[b];
                    }
                    if (any) {     // safety net: Jacobi-scaled CGNR from the current iterate
                        helm_launch_norm2(op, B.bscaled, n);
                        helm_launch_fin_ex(op, FIN_NORM, n, helm_vec_num_blocks(op), nullptr, B.d_aux);
                        HIP_TRY(op, hipMemcpyAsync(B.h_aux, B.d_aux, n * sizeof(double), hipMemcpyDeviceToHost, op->stream));
                        HIP_TRY(op, hipStreamSynchronize(op->stream));
                        for (int b = 0; b < n; ++b) if (B.h_mask[b]) {
                            RhsScal &S = op->h_scal[b];
                            total_iters[b] += S.iters; method_used[b] = HELM_CGNR;
                            S.bb = B.h_aux[b]; S.tol2 = 0.25 * o.rtol * o.rtol;
                        }
                        rc = run_cgnr(op, block, B, o.maxit, 50);
                        if (rc) return rc;
                    }
                }
                if (o.method == HELM_AUTO && !use_mg && !sys2) {
                    rc = download_scal(op, n);
                    if (rc) return rc;
                    bool any = false;
                    for (int b = 0; b < n; ++b) {
                        B.h_mask[b] = (op->h_scal[b].status == ST_BREAKDOWN);
                        if (B.h_mask[b]) { any = true; method_used[b] = HELM_CGNR; total_iters[b] += op->h_scal[b].iters; }
                    }
                    if (any) { rc = run_cgnr(op, block, B, o.maxit, check_every); if (rc) return rc; }
                }
            }
            // true residual of the UNSCALED system: s = q' - A x
            if (sys2) {
                rc = launch_sys2_apply(op, true, 0, B.w.x, B.w.s, qprime, n, EPI_RESID, nullptr);
            } else {
                ApplyArgs a = ApplyArgs();
                a.planes = op->d_C + (long long)block * op->nplanes * N; a.X = B.w.x; a.Y = B.w.s; a.W = qprime; a.ld = N; a.nrhs = n;
                a.scaled = 0; a.adjoint = 0; a.epi = EPI_RESID; a.scal = nullptr; a.part = (double *)op->d_part;
                rc = helm_launch_apply(op, a);
            }
            if (rc) return rc;
            helm_launch_fin_ex(op, FIN_NORM, n, B.nba, nullptr, B.d_aux);
            HIP_TRY(op, hipMemcpyAsync(B.h_aux, B.d_aux, 2 * n * sizeof(double), hipMemcpyDeviceToHost, op->stream));
            rc = download_scal(op, n);
            if (rc) return rc;
            bool refine = false;
            for (int b = 0; b < n; ++b) {
                const double qq = B.h_aux[n + b];
                relres[b] = qq > 0 ? sqrt(B.h_aux[b] / qq) : 0.0;
                B.h_mask[b] = 0;
                RhsScal &S = op->h_scal[b];
                if (S.status == ST_CONVERGED && relres[b] > o.rtol && round < max_refine && S.iters < o.maxit) {
                    // the scaled criterion was met but the unscaled residual is not there yet: tighten and go on
                    const double f = std::max(1e-3, 0.3 * o.rtol / relres[b]);
                    S.tol2 *= f * f;
                    B.h_mask[b] = 1; refine = true;
                }
            }
            if (!refine) break;
            if (o.method == HELM_CGNR || sys2) {
                for (int b = 0; b < n; ++b) if (B.h_mask[b]) op->h_scal[b].status = ST_ACTIVE;
                upload_scal(op, n);
            } else {
                rc = restart_masked(op, block, B);
                if (rc) return rc;
            }
        }
        // results
        for (int b = 0; b < n; ++b) {
            const RhsScal &S = op->h_scal[b];
            const bool ok = relres[b] <= o.rtol * 1.0000001 || (S.status == ST_CONVERGED && relres[b] <= 10 * o.rtol);
            if (!(relres[b] <= o.rtol * 1.0000001)) unconverged += 1;
            if (info) {
                helm_solve_info &I = info[first + b];
                I.iterations += total_iters[b] + S.iters;
                I.restarts += restarts[b];
                I.method = method_used[b];
                I.relres = std::max(I.relres, relres[b]);
                const int st = (relres[b] <= o.rtol * 1.0000001) ? 0 : (S.status == ST_BREAKDOWN ? 2 : 1);
                I.status = merge_status(I.status, st);
            }
            (void)ok;
        }
        HIP_TRY(op, hipMemcpyAsync(dXout + (long long)first * NV, B.w.x, (size_t)n * NV * sizeof(cplx), hipMemcpyDeviceToDevice, op->stream));
        HIP_TRY(op, hipStreamSynchronize(op->stream));
        if (use_mg && op->ny > 0 && mg3_is_layer_preserving(op)) {      // what this class of hierarchy needed: the depth model's book
            double sum = 0.0;
            for (int b = 0; b < n; ++b) sum += total_iters[b] + op->h_scal[b].iters;
            mg3_record_iterations(op, sum / n, o.rtol);
        }
    }
    return unconverged;
}

}  // namespace

// A factorisation started by helm_prefactor is complete (or abandoned): wait for it, book its time, give its scratch back.
// Scratch of a factorisation that helm_prefactor[_many] has enqueued goes back to the pool when the factorisation has FINISHED on the GPU, not when its operator
// is first solved with: a set is factored long before its turn in the pipeline comes, and held until then the scratch of four sets (4 GB each at 1024^2 x 2) was
// alive at once.  An event recorded behind the factorisation; every later prefactor / retire on the device looks which ones have completed.
namespace {
struct PendingScratch { int device; hipEvent_t ev; void *ws; size_t bytes; };
std::mutex g_ps_mu;
std::vector<PendingScratch> g_pending_scratch;
}
static void scratch_sweep(int device, bool wait) {
    std::vector<PendingScratch> done;
    {
        std::lock_guard<std::mutex> lk(g_ps_mu);
        for (size_t i = 0; i < g_pending_scratch.size(); ) {
            PendingScratch &ps = g_pending_scratch[i];
            bool fin = false;
            if (ps.device == device) {
                if (wait) { (void)hipEventSynchronize(ps.ev); fin = true; }
                else { const hipError_t q = hipEventQuery(ps.ev); if (q == hipSuccess) fin = true; else (void)hipGetLastError(); }
            }
            if (fin) { done.push_back(ps); g_pending_scratch.erase(g_pending_scratch.begin() + i); } else ++i;
        }
    }
    for (PendingScratch &ps : done) { hipEventDestroy(ps.ev); helm_pool_free(ps.device, ps.ws, ps.bytes); }
}
static void scratch_sweep_fwd(int device) { scratch_sweep(device, false); }
static void scratch_sweep_all_wait() {
    std::vector<int> devs;
    { std::lock_guard<std::mutex> lk(g_ps_mu); for (const PendingScratch &ps : g_pending_scratch) devs.push_back(ps.device); }
    for (int d : devs) { hipSetDevice(d); scratch_sweep(d, true); }
}
// ws is handed over: released behind everything enqueued on `st` so far (at once if no event can be had)
static void scratch_defer(int device, hipStream_t st, void *ws, size_t bytes) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, st) != hipSuccess) {
        (void)hipGetLastError();
        if (ev) hipEventDestroy(ev);
        hipStreamSynchronize(st);
        helm_pool_free(device, ws, bytes);
        return;
    }
    std::lock_guard<std::mutex> lk(g_ps_mu);
    g_pending_scratch.push_back(PendingScratch{device, ev, ws, bytes});
}

void helm_pf_retire(helm_op *op) {
    if (!op || !op->pf_pending) return;
    hipSetDevice(op->device);
    if (op->pf_done) hipEventSynchronize(op->pf_done);
    float ms = 0.f;
    if (op->pf_t0 && op->pf_t1 && hipEventElapsedTime(&ms, op->pf_t0, op->pf_t1) == hipSuccess) op->timing.factor_ms += ms / std::max(1, op->pf_share);
    if (op->pf_ws) helm_pool_free(op->device, op->pf_ws, op->pf_ws_bytes);
    op->pf_ws = nullptr; op->pf_ws_bytes = 0; op->pf_share = 1;
    op->pf_pending = false;
    scratch_sweep(op->device, false);          // (this operator's factorisation has finished: its set's scratch, and any older one's, goes back now)
}

// 3-D: what the next solve would build first -- the multigrid hierarchy with its directly solved level (hundreds of ms of GPU work and host logic,
// with waits in between) -- built NOW, in the calling thread.  Meant for a dispatcher's prepare thread: the set-up of frequency k+1 then runs
// beside the Krylov iterations of frequency k on another handle.  nrhs: right-hand sides the solve will bring (batch width, depth decision).
static int prefactor3d(helm_op *op, int nrhs) {
    if (op->mg3 || op->mg3_no_keep) return HELM_OK;
    const int auto_mg3 = helm_tuning_now().auto_mg3;
    if (!auto_mg3 || std::min(op->nz, std::min(op->ny, op->nx)) < 24) return HELM_OK;
    HIP_TRY(op, hipSetDevice(op->device));
    const int Bmax = std::max(1, std::min(nrhs > 0 ? nrhs : 16, 16));
    NvGuard guard(op, op->N);
    if (helm_ensure_scaled(op) != HELM_OK) return HELM_OK;
    if (ensure_ws(op, (size_t)11 * Bmax * op->N * sizeof(cplx)) != HELM_OK || ensure_part(op, Bmax) != HELM_OK) return HELM_OK;
    op->mg3_rhs_hint = nrhs > 0 ? nrhs : 16;
    // on a low-priority stream: the set-up is compute-bound products that would otherwise take the CUs from the (bandwidth-bound, critical-path)
    // iterations of the frequency being solved on another handle
    static const int prio = getenv("HELM_PF3_PRIO") ? atoi(getenv("HELM_PF3_PRIO")) : -1;
    hipStream_t main = op->stream, low = prio < 0 ? helm_stream_acquire(op->device, -1) : nullptr;
    if (low) op->stream = low;
    (void)mg_setup(op, Bmax);            // a hint: a failure here is the solve's to report
    if (low) {
        hipStreamSynchronize(low);
        op->stream = main;
        mg3_retarget_stream(op, main);
        helm_stream_release(op->device, -1, low);
    }
    return HELM_OK;
}

// The tolerance the solves on this operator will ask for, told BEFORE its factors are built (helm_prefactor has no options argument): which
// ill-conditioned fronts get the pivoted-LU treatment follows from it (direct.hip, stabilise_group).  Every solve records its own rtol as well,
// so a factorisation that happens inside a solve needs no hint.
extern "C" int helm_set_tolerance_hint(helm_op *op, double rtol) {
    helm_tuning_refresh();
    if (!op || !(rtol > 0)) return HELM_ERR_ARG;
    op->rtol_hint = rtol; op->rtol_hint_set = true;
    return HELM_OK;
}

// The factorisations of n operators in the same launches (include/helm.h).  The elimination tree is geometry only, so the fronts of n frequencies ride in one
// strided batch each: the latency-bound chain at the top of the tree (88 block steps of ~30 us, the gather-bound small separator levels, products of one to
// four fronts that fill a quarter of the chip) is paid once per set instead of once per frequency.  Operators that do not qualify for the direct path's
// prefactorisation, or that differ in grid / device, are prefactored one by one (the call is a hint, like helm_prefactor).
extern "C" int helm_prefactor_many(helm_op **ops, int n) {
    helm_tuning_refresh();
    if (!ops || n < 1) return HELM_ERR_ARG;
    for (int k = 0; k < n; ++k) if (!ops[k]) return HELM_ERR_ARG;
    auto qualifies = [&](helm_op *op) {
        return op->assembled && op->ny == 0 && !op->direct_failed && !op->direct[0] && !op->pf_pending && !(op->variant == HELM_EURUS && !op->block_zero[2]) &&
               helm_tuning_now().auto_direct != 0 && !testing_hook("HELM_ND_INJECT_FAILURE");
    };
    bool together = n >= 2 && n <= ND_NF_MAX && helm_tuning_now().nd_many != 0;
    for (int k = 0; k < n && together; ++k) {
        helm_op *op = ops[k];
        if (!qualifies(op) || op->device != ops[0]->device || op->nz != ops[0]->nz || op->nx != ops[0]->nx || op->variant != ops[0]->variant) together = false;
        for (int j = 0; j < k; ++j) if (ops[j] == op) together = false;
    }
    if (!together) {
        for (int k = 0; k < n; ++k) { const int rc = helm_prefactor(ops[k]); if (rc) return rc; }
        return HELM_OK;
    }
    helm_op *op0 = ops[0];
    HIP_TRY(op0, hipSetDevice(op0->device));
    if (!op0->fstream) {
        op0->fstream_prio = pf_prio();
        op0->fstream = helm_stream_acquire(op0->device, op0->fstream_prio);
        if (!op0->fstream) HELM_FAIL(op0, HELM_ERR_DEVICE, "hipStreamCreate failed");
    }
    for (int k = 0; k < n; ++k) {
        helm_op *op = ops[k];
        if (!op->pf_done) HIP_TRY(op, hipEventCreateWithFlags(&op->pf_done, hipEventDisableTiming));
        if (!op->pf_t0) HIP_TRY(op, hipEventCreate(&op->pf_t0));
        if (!op->pf_t1) HIP_TRY(op, hipEventCreate(&op->pf_t1));
    }
    NdFactor *fs[ND_NF_MAX] = {nullptr, nullptr, nullptr, nullptr};
    auto drop = [&]() { for (int k = 0; k < n; ++k) { nd_free(fs[k]); fs[k] = nullptr; } };
    for (int k = 0; k < n; ++k) {
        fs[k] = new NdFactor();
        const int rc = nd_get_plan(ops[k], helm_tuning_now().nd_leaf, 1, &fs[k]->pd);
        if (rc) { drop(); return rc; }
    }
    const size_t wsb = (size_t)n * (size_t)nd_factor_ws_elems(fs[0]->pd->plan) * sizeof(cplx);
    void *ws = helm_pool_alloc(op0->device, wsb);
    if (!ws) { drop(); HELM_FAIL(op0, HELM_ERR_DEVICE, "direct solver: cannot allocate %.1f GB of factorisation scratch", wsb / 1e9); }
    scratch_sweep(op0->device, false);           // (scratch of earlier sets whose factorisations have finished)
    { op0->ev_used = 0; op0->ev_pending.clear(); op0->ev_pending_gemm.clear(); op0->ev_pending_gemm_n.clear(); op0->ev_pending_gemm_bytes.clear(); op0->ev_pending_gemm_sol.clear(); op0->ev_pending_gemm_shape.clear(); }
    hipStream_t main = op0->stream;
    op0->stream = op0->fstream;                  // (the assembled planes of every operator are complete: helm_assemble synchronises)
    for (int k = 0; k < n; ++k) hipEventRecord(ops[k]->pf_t0, op0->fstream);
    const int rc = nd_factor_enqueue_many(op0, n, ops, fs, (cplx *)ws);
    for (int k = 0; k < n; ++k) { hipEventRecord(ops[k]->pf_t1, op0->fstream); hipEventRecord(ops[k]->pf_done, op0->fstream); }
    op0->stream = main;
    if (rc) { hipStreamSynchronize(op0->fstream); helm_pool_free(op0->device, ws, wsb); drop(); return rc; }
    scratch_defer(op0->device, op0->fstream, ws, wsb);
    for (int k = 0; k < n; ++k) {
        helm_op *op = ops[k];
        op->direct[0] = fs[k];
        op->pf_ws = nullptr; op->pf_ws_bytes = 0; op->pf_share = n; op->pf_pending = true;
    }
    return HELM_OK;
}

extern "C" int helm_prefactor_n(helm_op *op, int nrhs) {
    helm_tuning_refresh();
    if (!op) return HELM_ERR_ARG;
    if (!op->assembled) HELM_FAIL(op, HELM_ERR_STATE, "operator not assembled");
    if (op->ny > 0) return prefactor3d(op, nrhs);
    return helm_prefactor(op);
}

extern "C" int helm_prefactor(helm_op *op) {
    helm_tuning_refresh();
    if (!op) return HELM_ERR_ARG;
    if (!op->assembled) HELM_FAIL(op, HELM_ERR_STATE, "operator not assembled");
    // a hint: only the single-block 2-D systems the direct path of HELM_AUTO / HELM_DIRECT factors once per frequency
    if (op->ny > 0 || op->direct_failed || op->direct[0] || op->pf_pending) return HELM_OK;
    if (op->variant == HELM_EURUS && !op->block_zero[2]) return HELM_OK;          // coupled TTI: row-equilibrated inside the solve
    if (helm_tuning_now().auto_direct == 0) return HELM_OK;
    if (testing_hook("HELM_ND_INJECT_FAILURE")) return HELM_OK;
    HIP_TRY(op, hipSetDevice(op->device));
    if (!op->fstream) {
        op->fstream_prio = pf_prio();
        op->fstream = helm_stream_acquire(op->device, op->fstream_prio);
        if (!op->fstream) HELM_FAIL(op, HELM_ERR_DEVICE, "hipStreamCreate failed");
    }
    if (!op->pf_done) HIP_TRY(op, hipEventCreateWithFlags(&op->pf_done, hipEventDisableTiming));
    if (!op->pf_t0) HIP_TRY(op, hipEventCreate(&op->pf_t0));
    if (!op->pf_t1) HIP_TRY(op, hipEventCreate(&op->pf_t1));
    NdFactor *f = new NdFactor();
    int rc = nd_get_plan(op, helm_tuning_now().nd_leaf, 1, &f->pd);
    if (rc) { nd_free(f); return rc; }
    const size_t wsb = (size_t)nd_factor_ws_elems(f->pd->plan) * sizeof(cplx);
    void *ws = helm_pool_alloc(op->device, wsb);
    if (!ws) { nd_free(f); HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: cannot allocate %.1f GB of factorisation scratch", wsb / 1e9); }
    { op->ev_used = 0; op->ev_pending.clear(); op->ev_pending_gemm.clear(); op->ev_pending_gemm_n.clear(); op->ev_pending_gemm_bytes.clear(); op->ev_pending_gemm_sol.clear(); op->ev_pending_gemm_shape.clear(); }
    hipStream_t main = op->stream;
    op->stream = op->fstream;                    // (the assembled planes are complete: helm_assemble synchronises)
    hipEventRecord(op->pf_t0, op->fstream);
    rc = nd_factor_enqueue(op, 0, f, (cplx *)ws, nullptr);
    hipEventRecord(op->pf_t1, op->fstream);
    hipEventRecord(op->pf_done, op->fstream);
    op->stream = main;
    if (rc) { hipStreamSynchronize(op->fstream); helm_pool_free(op->device, ws, wsb); nd_free(f); return rc; }
    op->direct[0] = f;
    scratch_defer(op->device, op->fstream, ws, wsb);
    op->pf_ws = nullptr; op->pf_ws_bytes = 0; op->pf_pending = true;
    return HELM_OK;
}

// Scratch for `concurrent` host-array solves (helm_solve / helm_solve_coo) of nrhs right-hand sides running on this handle's GPU at the
// same time, brought into being NOW: the shared scratch slots of the direct path and the device images of the right-hand sides and
// wavefields (three buffers per call).  Everything here is taken lazily anyway; but a hipMalloc issued while other host threads have
// kernels and copies in flight was measured at 0.7-1.5 s (HELM_ALLOC_TRACE=1), so a dispatcher that knows how many workers it is
// about to start on a GPU asks once, before they run.  A hint: errors other than bad arguments are swallowed.
extern "C" int helm_reserve(helm_op *op, int nrhs, long long rows, int concurrent) {
    helm_tuning_refresh();
    if (!op || nrhs < 1 || rows < 1 || concurrent < 1) return HELM_ERR_ARG;
    if (hipSetDevice(op->device) != hipSuccess) { (void)hipGetLastError(); return HELM_OK; }
    if (concurrent > 64) concurrent = 64;
    const size_t bytes = (size_t)nrhs * rows * sizeof(cplx);
    {   // device images: idle buffers of that size the pool holds already count
        size_t have = 0;
        { std::lock_guard<std::mutex> lk(g_pool.mu); have = g_pool.idle.count(std::make_pair(op->device, bytes)); }
        std::vector<void *> got;
        for (size_t k = have; k < (size_t)3 * concurrent; ++k) {
            void *p = nullptr;
            if (helm_malloc_retry(op->device, &p, bytes) != hipSuccess) break;
            got.push_back(p);
        }
        for (void *p : got) helm_pool_free(op->device, p, bytes);
    }
    const bool direct2d = op->assembled && op->ny == 0 && !op->direct_failed && !(op->variant == HELM_EURUS && !op->block_zero[2]);
    const helm_tuning tune = helm_tuning_now();
    if (tune.auto_direct == 0) return HELM_OK;
    if (!direct2d) return HELM_OK;
    std::shared_ptr<NdPlanDev> pd;
    if (nd_get_plan(op, tune.nd_leaf, 1, &pd) != HELM_OK || !pd) return HELM_OK;
    const long long per_rhs = 4 * op->N + 2 * pd->plan.vregion;
    int Bmax = std::min(nrhs, 256);
    const double cap = tune.nd_ws_gb * 1e9;
    while (Bmax > 1 && (double)per_rhs * Bmax * sizeof(cplx) > cap) Bmax = (Bmax + 1) / 2;
    const size_t wsb = (size_t)per_rhs * Bmax * sizeof(cplx);
    const int ready = ws_table_reserve(g_shared_ws, op->device, wsb, concurrent, ws_dev_alloc, ws_dev_free);      // this device's own table: booking for one GPU never touches another's
    // more concurrent solves than slots (several workers per GPU): the others fall back to their handle's own workspace, which comes from the
    // size-keyed pool -- put that many buffers there now, as long as they fit beside everything else (half of what is free)
    if (concurrent > ready) {
        size_t have = 0, freeb = 0, totb = 0;
        { std::lock_guard<std::mutex> lk(g_pool.mu); have = g_pool.idle.count(std::make_pair(op->device, wsb)); }
        if (hipMemGetInfo(&freeb, &totb) != hipSuccess) { (void)hipGetLastError(); freeb = 0; }
        std::vector<void *> got;
        for (size_t k = have; k < (size_t)(concurrent - ready) && (got.size() + 1) * wsb <= freeb / 2; ++k) {
            void *p = nullptr;
            AllocTrace tr("ws fallback", wsb);
            if (hipMalloc(&p, wsb) != hipSuccess) { (void)hipGetLastError(); break; }
            got.push_back(p);
        }
        for (void *p : got) helm_pool_free(op->device, p, wsb);
    }
    return HELM_OK;
}

extern "C" int helm_solve_device(helm_op *op, const void *dRHS, void *dU, int nrhs, long long rows,
                                 double premul_re, double premul_im, const helm_solve_opts *opts, helm_solve_info *info) {
    helm_tuning_refresh();
    if (!op) return HELM_ERR_ARG;
    // a declared support (helm_set_rhs_support) belongs to THIS call's right-hand sides and to no later one -- whichever way the call ends, the
    // early returns below included (the bits usually live in a buffer the caller recycles as soon as this returns)
    struct SupportOneShot { helm_op *o; ~SupportOneShot() { o->rhs_bits = nullptr; o->rhs_bits_q = nullptr; o->rhs_bits_violated = 0; } } support_one_shot{op};
    if (!dRHS || !dU || nrhs < 1) return HELM_ERR_ARG;
    if (!op->assembled) HELM_FAIL(op, HELM_ERR_STATE, "operator not assembled");
    const long long N = op->N;
    const bool stacked = (op->variant == HELM_EURUS && rows == 2 * N);
    if (rows != N && !stacked) HELM_FAIL(op, HELM_ERR_ARG, "dimension mismatch: rhs has %lld rows, operator has %lld%s", rows, N,
                                         op->variant == HELM_EURUS ? " (or 2N stacked)" : "");
    HIP_TRY(op, hipSetDevice(op->device));
    if (stacked) { const int rcb = helm_need_all_blocks(op); if (rcb) return rcb; }      // (u = M1^-1 (q1 - M2 M4^-1 q2): M2 and M4 are needed now)
    if (op->rhs_bits && (op->rhs_bits_rows != rows || op->rhs_bits_nrhs != nrhs)) HELM_FAIL(op, HELM_ERR_ARG, "helm_set_rhs_support was given %lld rows x %d right-hand sides, this solve has %lld x %d", op->rhs_bits_rows, op->rhs_bits_nrhs, rows, nrhs);
    op->rhs_bits_q = op->rhs_bits ? dRHS : nullptr;
    helm_solve_opts o;
    if (opts) o = *opts; else { o.method = HELM_AUTO; o.rtol = 1e-10; o.maxit = 200000; o.check_every = 0; o.batch = 0; o.flags = 0; }
    if (!(o.rtol > 0)) o.rtol = 1e-10;
    if (o.maxit < 1) o.maxit = 200000;
    if (!op->direct[0] && !op->pf_pending) { op->rtol_hint = o.rtol; op->rtol_hint_set = true; }      // (factors that exist, or are on their way, were conditioned for the hint they were given)
    if (info) for (int r = 0; r < nrhs; ++r) { info[r].iterations = 0; info[r].status = 0; info[r].restarts = 0; info[r].method = o.method; info[r].relres = 0.0; }
    const cplx premul = cmake(premul_re, premul_im);

    hipEvent_t e0, e1;
    HIP_TRY(op, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); HELM_FAIL(op, HELM_ERR_DEVICE, "hipEventCreate failed"); }
    timing_begin(op);
    HIP_TRY(op, hipEventRecord(e0, op->stream));

    int result = 0;
    const size_t xbytes = (size_t)nrhs * N * sizeof(cplx);
    cplx *dX = (cplx *)helm_pool_alloc(op->device, xbytes);
    if (!dX) { hipEventDestroy(e0); hipEventDestroy(e1); HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc failed"); }
    // Buffer layouts (opts.flags): HELM_RHS_NODE_MAJOR / HELM_OUT_NODE_MAJOR = the reference's (rows, nrhs) C-order arrays.  The direct path
    // takes both natively for one batch of a single-block system (no transposes at all); every other combination goes through rhs-major
    // temporaries here, so all paths below see one right-hand side per row.
    const int lay = o.flags & HELM_NODE_MAJOR;
    o.flags &= ~HELM_NODE_MAJOR;
    const void *dRHS_use = dRHS; void *dU_use = dU;
    cplx *tR = nullptr, *tU = nullptr;
    const size_t lbytes = (size_t)nrhs * rows * sizeof(cplx);
    auto cleanup = [&]() { hipStreamSynchronize(op->stream); helm_pool_free(op->device, dX, xbytes); helm_pool_free(op->device, tR, lbytes); helm_pool_free(op->device, tU, lbytes);
                           hipEventDestroy(e0); hipEventDestroy(e1); };
    bool native_done = false;
    if (lay == HELM_NODE_MAJOR && op->ny == 0 && rows == N && !op->direct_failed && (o.method == HELM_AUTO || o.method == HELM_DIRECT) &&
        (op->variant == HELM_MINIZEPHYR || op->block_zero[2]) && !testing_hook("HELM_ND_INJECT_FAILURE") && !testing_hook("HELM_ND_INJECT_STALL") &&
        !(helm_tuning_now().auto_direct == 0 && o.method == HELM_AUTO)) {
        helm_solve_opts on = o; on.flags |= HELM_NODE_MAJOR;
        const int rcn = solve_block_direct(op, 0, (const cplx *)dRHS, nrhs, 0, premul, nullptr, nullptr, nrhs, on, info, 0, 0, (cplx *)dU);
        if (op->rhs_bits_violated) { cleanup(); return HELM_ERR_ARG; }          // (HELM_ND_SUPPORT_CHECK: the message is set)
        if (rcn == 0) native_done = true;
        else if (info) for (int r = 0; r < nrhs; ++r) { info[r].iterations = 0; info[r].status = 0; info[r].restarts = 0; info[r].method = o.method; info[r].relres = 0.0; }
        // (anything else -- too many right-hand sides for one batch, a right-hand side above rtol, a failed factorisation: the general path)
    }
    if (lay && !native_done) {
        if (lay & HELM_RHS_NODE_MAJOR) {
            tR = (cplx *)helm_pool_alloc(op->device, lbytes);
            if (!tR) { cleanup(); HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc failed"); }
            const int rct = nd_transpose(op, (const cplx *)dRHS, rows, nrhs, tR);
            if (rct) { cleanup(); return rct; }
            dRHS_use = tR;
        }
        if (lay & HELM_OUT_NODE_MAJOR) {
            tU = (cplx *)helm_pool_alloc(op->device, lbytes);
            if (!tU) { cleanup(); HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc failed"); }
            dU_use = tU;
        }
    }

    if (native_done) {
    } else if (op->variant == HELM_EURUS && !op->block_zero[2]) {
        // eps != delta: M3 != 0, the two fields are coupled -> Jacobi-BiCGSTAB on the full 2N x 2N system
        // (eurus.py:430-464,512-533); N-row right-hand sides are zero-padded and the result clipped
        cplx *dW = nullptr;
        if (helm_malloc_retry(op->device, (void **)&dW, (size_t)nrhs * 2 * N * sizeof(cplx)) != hipSuccess) { cleanup(); HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc failed"); }
        int rc = solve_block(op, 0, (const cplx *)dRHS_use, rows, 0, premul, nullptr, dW, nrhs, o, info, 1, rows);
        if (rc >= 0) {
            result = rc;
            rc = helm_launch_finish_ex(op, dW, 2 * N, 0, (cplx *)dU_use, rows, 0, nrhs);
            if (!rc && stacked) rc = helm_launch_finish_ex(op, dW, 2 * N, N, (cplx *)dU_use, rows, N, nrhs);
            if (!rc && hipStreamSynchronize(op->stream) != hipSuccess) rc = HELM_ERR_DEVICE;
        }
        hipFree(dW);
        if (rc < 0) { cleanup(); return rc; }
    } else if (op->variant == HELM_MINIZEPHYR || !stacked) {
        // Eurus with an N-row right-hand side: zero-padded second field => v = 0 and M1 u = q
        // when M3 == 0 (eurus.py:512-533)
        bool wrote_u = false;        // rows == N here: the direct path writes conj(x) into dU itself
        int rc = solve_block(op, 0, (const cplx *)dRHS_use, rows, 0, premul, nullptr, dX, nrhs, o, info, 0, 0, (cplx *)dU_use, &wrote_u);
        if (rc < 0) { cleanup(); return rc; }
        result = rc;
        if (!wrote_u) {
            rc = helm_launch_finish(op, dX, (cplx *)dU_use, rows, nrhs, 0);
            if (rc) { cleanup(); return rc; }
        }
    } else {
        // block-triangular: v = M4^-1 q2 ; u = M1^-1 (q1 - M2 v)
        cplx *dV = nullptr, *dT = nullptr;
        if (helm_malloc_retry(op->device, (void **)&dV, (size_t)nrhs * N * sizeof(cplx)) != hipSuccess || helm_malloc_retry(op->device, (void **)&dT, (size_t)nrhs * N * sizeof(cplx)) != hipSuccess) {
            hipFree(dV); cleanup(); HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc failed");
        }
        int rc = solve_block(op, 3, (const cplx *)dRHS_use, rows, N, premul, nullptr, dV, nrhs, o, info);
        if (rc >= 0) {
            result = rc;
            ApplyArgs a = ApplyArgs();
            a.planes = op->d_C + 1LL * 9 * N; a.X = dV; a.Y = dT; a.W = nullptr; a.ld = N; a.nrhs = nrhs; a.scaled = 0; a.adjoint = 0;
            a.epi = EPI_NONE; a.scal = nullptr; a.part = nullptr;
            rc = helm_launch_apply(op, a);
            if (!rc) rc = solve_block(op, 0, (const cplx *)dRHS_use, rows, 0, premul, dT, dX, nrhs, o, info);
            if (rc >= 0) {
                result = std::max(result, rc);
                rc = helm_launch_finish(op, dX, (cplx *)dU_use, rows, nrhs, 0);
                if (!rc) rc = helm_launch_finish(op, dV, (cplx *)dU_use, rows, nrhs, N);
            }
        }
        hipFree(dV); hipFree(dT);
        if (rc < 0) { cleanup(); return rc; }
    }
    if (tU) {
        const int rct = nd_transpose(op, tU, nrhs, rows, (cplx *)dU);
        if (rct) { cleanup(); return rct; }
    }
    HIP_TRY(op, hipEventRecord(e1, op->stream));
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    op->timing.solve_ms = ms;
    helm_pf_retire(op);
    timing_collect(op);
    cleanup();
    return result;
}

extern "C" int helm_solve(helm_op *op, const double *RHS, double *U, int nrhs, long long rows,
                          double premul_re, double premul_im, const helm_solve_opts *opts, helm_solve_info *info) {
    helm_tuning_refresh();
    if (!op || !RHS || !U || nrhs < 1) return HELM_ERR_ARG;
    HIP_TRY(op, hipSetDevice(op->device));
    const size_t bytes = (size_t)nrhs * rows * sizeof(cplx);
    void *dR = helm_pool_alloc(op->device, bytes), *dU = helm_pool_alloc(op->device, bytes);
    if (!dR || !dU) { helm_pool_free(op->device, dR, bytes); helm_pool_free(op->device, dU, bytes); HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc failed"); }
    int rc = HELM_OK;
    if (helm_upload_staged(op, dR, RHS, bytes)) rc = HELM_ERR_DEVICE;
    if (!rc) rc = helm_solve_device(op, dR, dU, nrhs, rows, premul_re, premul_im, opts, info);
    if (rc >= 0 && helm_download_staged(op, U, dU, bytes)) rc = HELM_ERR_DEVICE;
    hipStreamSynchronize(op->stream);
    helm_pool_free(op->device, dR, bytes); helm_pool_free(op->device, dU, bytes);
    return rc;
}

extern "C" int helm_rhs_from_coo_device(helm_op *op, const void *d_row, const void *d_col, const void *d_val, long long nnz, void *dR, int nrhs, long long rows) {
    helm_tuning_refresh();
    if (!op || !dR || nrhs < 1 || rows < 1 || nnz < 0 || (nnz > 0 && (!d_row || !d_col || !d_val))) return HELM_ERR_ARG;
    HIP_TRY(op, hipSetDevice(op->device));
    int rc = helm_launch_rhs_from_coo(op, (const long long *)d_row, (const int *)d_col, (const cplx *)d_val, nnz, (cplx *)dR, nrhs, rows);
    if (rc) return rc;
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    return HELM_OK;
}

extern "C" int helm_rhs_from_coo_device_layout(helm_op *op, const void *d_row, const void *d_col, const void *d_val, long long nnz, void *dR, int nrhs, long long rows, int flags) {
    helm_tuning_refresh();
    if (!op || !dR || nrhs < 1 || rows < 1 || nnz < 0 || (nnz > 0 && (!d_row || !d_col || !d_val))) return HELM_ERR_ARG;
    HIP_TRY(op, hipSetDevice(op->device));
    int rc = helm_launch_rhs_from_coo(op, (const long long *)d_row, (const int *)d_col, (const cplx *)d_val, nnz, (cplx *)dR, nrhs, rows, (flags & HELM_RHS_NODE_MAJOR) ? 1 : 0);
    if (rc) return rc;
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    return HELM_OK;
}

// Declared support of right-hand sides (round 4).  The reference hands its sources over as scipy-sparse matrices (survey.py:86-89,162-188): where they are
// nonzero is part of the input, not something to be found.  bits: one byte per row (cell) on the device, bit b set = the right-hand sides of block b of
// 64 columns MAY be nonzero in that row (at most 512 right-hand sides); the caller guarantees zeros everywhere else.  One shot: it applies to the next
// helm_solve_device on this handle (same rows and right-hand-side count, node-major layout) and is forgotten when that call returns.  The direct path then
// sets the leaf flags of its forward pass from the bits instead of reading every right-hand-side row to look for nonzeros (3.2 of 4.3 GB at 1024^2 x 256).
// HELM_ND_SUPPORT_CHECK=1 verifies the guarantee (one pass over q) and fails the solve if it does not hold.
extern "C" int helm_set_rhs_support(helm_op *op, const void *d_bits, long long rows, int nrhs) {
    helm_tuning_refresh();
    if (!op) return HELM_ERR_ARG;
    if (!d_bits) { op->rhs_bits = nullptr; op->rhs_bits_q = nullptr; return HELM_OK; }
    if (rows < 1 || nrhs < 1 || nrhs > 512) HELM_FAIL(op, HELM_ERR_ARG, "helm_set_rhs_support: rows >= 1 and 1 <= nrhs <= 512");
    op->rhs_bits = (const unsigned char *)d_bits; op->rhs_bits_rows = rows; op->rhs_bits_nrhs = nrhs; op->rhs_bits_violated = 0;
    return HELM_OK;
}

namespace {
__global__ __launch_bounds__(256) void k_support_from_coo(const long long *row, const int *col, long long nnz, long long rows, int nrhs, unsigned *words) {
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) {
        const long long r = row[k]; const int c = col[k];
        if (r < 0 || r >= rows || c < 0 || c >= nrhs) continue;
        atomicOr(&words[r >> 2], (1u << (c >> 6)) << (8 * (int)(r & 3)));
    }
}
}  // namespace
// bits (rows bytes, rounded up to a multiple of 4, on the device) from the triplets of a sparse right-hand-side matrix (device arrays as for
// helm_rhs_from_coo_device_layout): what helm_set_rhs_support takes
extern "C" int helm_rhs_support_from_coo(helm_op *op, const void *d_row, const void *d_col, long long nnz, void *d_bits, long long rows, int nrhs) {
    helm_tuning_refresh();
    if (!op || !d_bits || rows < 1 || nrhs < 1 || nrhs > 512 || nnz < 0 || (nnz > 0 && (!d_row || !d_col))) return HELM_ERR_ARG;
    HIP_TRY(op, hipSetDevice(op->device));
    HIP_TRY(op, hipMemsetAsync(d_bits, 0, (size_t)((rows + 3) / 4) * 4, op->stream));
    if (nnz > 0) HELM_LAUNCH(k_support_from_coo, dim3((unsigned)std::min<long long>((nnz + 255) / 256, 4096)), dim3(256), 0, op->stream,
                                    (const long long *)d_row, (const int *)d_col, nnz, rows, nrhs, (unsigned *)d_bits);
    HIP_TRY(op, hipGetLastError());
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    return HELM_OK;
}

// Pinned host memory for the caller's result arrays (recycled by size): device-to-host copies into it run at the PCIe rate, into
// pageable memory at a fraction of it.
extern "C" void *helm_host_alloc(size_t bytes) { return bytes ? helm_hostpool_alloc(bytes) : nullptr; }
extern "C" void helm_host_free(void *p, size_t bytes) { helm_hostpool_free(p, bytes); }

// Host-side sparse right-hand sides (the reference's scipy-sparse source matrices, survey.py:162-169): only the triplets cross PCIe,
// the dense right-hand sides exist on the device alone; the wavefields come back into U (host; pinned memory from helm_host_alloc
// makes that copy run at the PCIe rate).  Layout of U and of the implied dense right-hand sides per opts->flags.
extern "C" int helm_solve_coo(helm_op *op, const long long *row, const int *col, const double *val, long long nnz, double *U, int nrhs, long long rows,
                              double premul_re, double premul_im, const helm_solve_opts *opts, helm_solve_info *info) {
    helm_tuning_refresh();
    if (!op || !U || nrhs < 1 || rows < 1 || nnz < 0 || (nnz > 0 && (!row || !col || !val))) return HELM_ERR_ARG;
    HIP_TRY(op, hipSetDevice(op->device));
    const size_t bytes = (size_t)nrhs * rows * sizeof(cplx);
    // (+ one byte per row behind the triplets: the support of the dense image, see below -- no allocation of its own)
    const size_t tb0 = (((size_t)std::max<long long>(nnz, 1) * (sizeof(long long) + sizeof(int) + sizeof(cplx))) + 15) & ~(size_t)15;
    const size_t bbytes = (size_t)((rows + 3) / 4) * 4;
    const size_t tb = tb0 + bbytes;
    void *dR = helm_pool_alloc(op->device, bytes), *dU = helm_pool_alloc(op->device, bytes), *dT = helm_pool_alloc(op->device, tb);
    auto release = [&]() { hipStreamSynchronize(op->stream); helm_pool_free(op->device, dR, bytes); helm_pool_free(op->device, dU, bytes); helm_pool_free(op->device, dT, tb); };
    if (!dR || !dU || !dT) { release(); HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc failed"); }
    // packed as [val (16-byte entries) | row (8) | col (4)]: every array starts on a multiple of its own element size whatever the parity of nnz
    // (r3 packed row | val | col, which left val on an 8-byte boundary for odd nnz although the expansion kernel reads it as 16-byte vectors)
    cplx *d_val = (cplx *)dT; long long *d_row = (long long *)(d_val + std::max<long long>(nnz, 1)); int *d_col = (int *)(d_row + std::max<long long>(nnz, 1));
    int rc = HELM_OK;
    for (long long k = 0; k < nnz; ++k)              // the scatter trusts its indices: check them where they arrive
        if (row[k] < 0 || row[k] >= rows || col[k] < 0 || col[k] >= nrhs) {
            release();
            HELM_FAIL(op, HELM_ERR_ARG, "sparse right-hand side: entry %lld addresses (row %lld, column %d) outside the %lld x %d right-hand-side matrix", k, row[k], col[k], rows, nrhs);
        }
    if (nnz > 0 && (helm_upload_staged(op, d_row, row, nnz * sizeof(long long)) || helm_upload_staged(op, d_val, val, nnz * sizeof(cplx)) ||
                    helm_upload_staged(op, d_col, col, nnz * sizeof(int)))) rc = HELM_ERR_DEVICE;
    const int flags = opts ? opts->flags : 0;
    if (!rc) rc = helm_launch_rhs_from_coo(op, d_row, d_col, d_val, nnz, (cplx *)dR, nrhs, rows, (flags & HELM_RHS_NODE_MAJOR) ? 1 : 0);
    if (!rc && hipStreamSynchronize(op->stream) != hipSuccess) rc = HELM_ERR_DEVICE;
    // the dense image was made here from the triplets: its support is known exactly, the direct path need not look for it
    void *dBits = (char *)dT + tb0;
    if (!rc && nrhs <= 512 && (flags & HELM_NODE_MAJOR) == HELM_NODE_MAJOR && helm_rhs_support_from_coo(op, d_row, d_col, nnz, dBits, rows, nrhs) == HELM_OK)
        (void)helm_set_rhs_support(op, dBits, rows, nrhs);
    if (!rc) rc = helm_solve_device(op, dR, dU, nrhs, rows, premul_re, premul_im, opts, info);
    (void)helm_set_rhs_support(op, nullptr, 0, 0);          // (the bits live in dT, which goes back to the pool below: never leave a pointer to them behind)
    if (rc >= 0 && helm_download_staged(op, U, dU, bytes)) rc = HELM_ERR_DEVICE;
    release();
    return rc;
}

extern "C" int helm_sample_device(helm_op *op, const void *dU, int nsrc, long long ld, const void *d_rowptr, const void *d_col, const void *d_val, int nrec, void *d_out) {
    helm_tuning_refresh();
    if (!op || !dU || !d_rowptr || !d_col || !d_val || !d_out || nsrc < 1 || nrec < 1) return HELM_ERR_ARG;
    HIP_TRY(op, hipSetDevice(op->device));
    int rc = helm_launch_sample(op, (const cplx *)dU, nsrc, ld, (const long long *)d_rowptr, (const long long *)d_col, (const cplx *)d_val, nrec, (cplx *)d_out);
    if (rc) return rc;
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    return HELM_OK;
}

extern "C" int helm_imaging_accumulate_device(helm_op *op, const void *dUF, const void *dUB, int nsrc, const void *dScaler, void *dG) {
    helm_tuning_refresh();
    if (!op || !dUF || !dUB || !dScaler || !dG || nsrc < 1) return HELM_ERR_ARG;
    HIP_TRY(op, hipSetDevice(op->device));
    int rc = helm_launch_imaging(op, (const cplx *)dUF, (const cplx *)dUB, nsrc, (const cplx *)dScaler, (cplx *)dG);
    if (rc) return rc;
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    return HELM_OK;
}
