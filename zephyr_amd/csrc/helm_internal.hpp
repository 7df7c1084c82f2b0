// Internal declarations shared by the libhelm translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <memory>
#include <cstdio>
#include <cmath>
#include "../../include/helm.h"

typedef double2 cplx;   // interleaved (re, im), 16 B: one dwordx4 per lane

#define HELM_WAVE 64
#define HELM_NXCD 8

// ---- complex helpers (host + device) ----------------------------------------------------
__host__ __device__ inline cplx cmake(double r, double i) { cplx z; z.x = r; z.y = i; return z; }
__host__ __device__ inline cplx cadd(cplx a, cplx b) { return cmake(a.x + b.x, a.y + b.y); }
__host__ __device__ inline cplx csub(cplx a, cplx b) { return cmake(a.x - b.x, a.y - b.y); }
__host__ __device__ inline cplx cmul(cplx a, cplx b) { return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__host__ __device__ inline cplx cscale(cplx a, double s) { return cmake(a.x * s, a.y * s); }
// conj(s * x) with the floating-point contraction pinned: the wavefield u = conj(premul x) is formed by different kernels on different paths (the residual launch,
// the epilogue of the back substitution, the transposes) and has to come out bit for bit the same from all of them
__host__ __device__ inline cplx conj_scaled(cplx s, cplx x) {
    // (explicit fused operations: nothing is left for the compiler to contract one way in one kernel and another way in the next)
    const double re = fma(s.x, x.x, -(s.y * x.y)), im = fma(s.x, x.y, s.y * x.x);
    return cmake(re, -im);
}
__host__ __device__ inline cplx cconj(cplx a) { return cmake(a.x, -a.y); }
__host__ __device__ inline cplx cneg(cplx a) { return cmake(-a.x, -a.y); }
__host__ __device__ inline double cabs2(cplx a) { return a.x * a.x + a.y * a.y; }
__host__ __device__ inline cplx cdiv(cplx a, cplx b) {
    // Smith's algorithm (robust against over/underflow of |b|^2)
    if (fabs(b.x) >= fabs(b.y)) {
        double r = b.y / b.x, d = b.x + b.y * r;
        return cmake((a.x + a.y * r) / d, (a.y - a.x * r) / d);
    } else {
        double r = b.x / b.y, d = b.x * r + b.y;
        return cmake((a.x * r + a.y) / d, (a.y * r - a.x) / d);
    }
}
__host__ __device__ inline cplx crecip(cplx b) { return cdiv(cmake(1.0, 0.0), b); }
// acc += a*b
__host__ __device__ inline void cfma(cplx &acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x); acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y); acc.y = fma(a.y, b.x, acc.y);
}
// acc += conj(a)*b
__host__ __device__ inline void cfma_conj(cplx &acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x); acc.x = fma(a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y); acc.y = fma(-a.y, b.x, acc.y);
}

// ---- single-precision complex (multigrid preconditioner storage) -------------------------------
typedef float2 cplxf;
__host__ __device__ inline cplxf cmakef(float r, float i) { cplxf z; z.x = r; z.y = i; return z; }
__host__ __device__ inline cplxf cadd(cplxf a, cplxf b) { return cmakef(a.x + b.x, a.y + b.y); }
__host__ __device__ inline cplxf csub(cplxf a, cplxf b) { return cmakef(a.x - b.x, a.y - b.y); }
__host__ __device__ inline cplxf cmul(cplxf a, cplxf b) { return cmakef(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__host__ __device__ inline cplxf cscale(cplxf a, double s) { return cmakef(a.x * (float)s, a.y * (float)s); }
__host__ __device__ inline cplxf cconj(cplxf a) { return cmakef(a.x, -a.y); }
__host__ __device__ inline cplxf cneg(cplxf a) { return cmakef(-a.x, -a.y); }
__host__ __device__ inline double cabs2(cplxf a) { return (double)a.x * a.x + (double)a.y * a.y; }
__host__ __device__ inline void cfma(cplxf &acc, cplxf a, cplxf b) {
    acc.x = fmaf(a.x, b.x, acc.x); acc.x = fmaf(-a.y, b.y, acc.x);
    acc.y = fmaf(a.x, b.y, acc.y); acc.y = fmaf(a.y, b.x, acc.y);
}
__host__ __device__ inline cplxf to_f32(cplx a) { return cmakef((float)a.x, (float)a.y); }
__host__ __device__ inline cplx to_f64(cplxf a) { return cmake((double)a.x, (double)a.y); }
template <class V> __host__ __device__ inline V vzero();
template <> __host__ __device__ inline cplx vzero<cplx>() { return cmake(0.0, 0.0); }
template <> __host__ __device__ inline cplxf vzero<cplxf>() { return cmakef(0.f, 0.f); }
template <class V> __host__ __device__ inline V vone();
template <> __host__ __device__ inline cplx vone<cplx>() { return cmake(1.0, 0.0); }
template <> __host__ __device__ inline cplxf vone<cplxf>() { return cmakef(1.f, 0.f); }
template <class V> __host__ __device__ inline V vfrom(cplx a);
template <> __host__ __device__ inline cplx vfrom<cplx>(cplx a) { return a; }
template <> __host__ __device__ inline cplxf vfrom<cplxf>(cplx a) { return to_f32(a); }

// ---- per-RHS solver scalars kept on the device ---------------------------------------------
// One record per right-hand side of the batch; written only by the single-block "finalize"
// kernels, read by every vector kernel.  Plain doubles so the host can memcpy it.
struct RhsScal {
    double rho_re, rho_im;       // BiCGSTAB rho = (r0, r)       | CGNR gamma = (z,z)
    double alpha_re, alpha_im;
    double omega_re, omega_im;
    double beta_re, beta_im;
    double rr;                   // ||r||^2 (recursive residual, scaled system)
    double bb;                   // ||b||^2 (scaled system)
    double tol2;                 // (rtol_eff)^2
    int status;                  // 0 active, 1 converged, 2 breakdown, 3 frozen by host
    int iters;
    int pad0, pad1;
};

enum { ST_ACTIVE = 0, ST_CONVERGED = 1, ST_BREAKDOWN = 2, ST_FROZEN = 3, ST_PARKED = 4 };

// ---- the handle ---------------------------------------------------------------------------
struct helm_op {
    int device = 0, variant = 0, nz = 0, nx = 0, nPML = 10;
    int ny = 0;                  // > 0: 3-D operator on an (nz, ny, nx) grid (27 planes), 0: 2-D (9 planes)
    int nplanes = 9, centre = 4; // planes per block and index of the diagonal plane
    long long N = 0;
    long long Nv = 0;            // length of the Krylov vectors (= N; 2N while the coupled Eurus system is being solved)
    double dx = 1, dz = 1, dy = 1;
    int fs[4] = {0, 0, 0, 0};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    cplx *sk_buf = nullptr; size_t sk_bytes = 0;   // partial products of split-K launches (direct.hip gemm), from the pool
    const unsigned char *rhs_bits = nullptr; const void *rhs_bits_q = nullptr; long long rhs_bits_rows = 0; int rhs_bits_nrhs = 0; int rhs_bits_violated = 0;   // declared support of the next solve's right-hand sides (helm_set_rhs_support; one shot)
    cplx *gjp_buf = nullptr; size_t gjp_bytes = 0; // alternating pivot-block inverses of the one-launch Gauss-Jordan step (direct.hip k_gj_step), from the pool
    hipStream_t side_stream = nullptr;    // second stream of the direct path (forward elimination behind the factorisation), on demand
    // helm_prefactor: the factorisation of the assembled operator enqueued on a high-priority stream of its own, so that it runs beside
    // the solves of ANOTHER handle (the previous frequency of a job); the next solve on this handle waits for pf_done on the device
    hipStream_t fstream = nullptr;
    bool pf_pending = false;
    hipEvent_t pf_done = nullptr, pf_t0 = nullptr, pf_t1 = nullptr;
    void *pf_ws = nullptr; size_t pf_ws_bytes = 0;
    int pf_share = 1;                      // operators whose factorisations share the span pf_t0 .. pf_t1

    // model
    cplx *d_c = nullptr;
    // 3-D operator, coefficients rebuilt on the fly (helm3d.hip, k_stencil3<.., OTF>): K = om^2 / (rho c^2) and b = 1 / rho per point (24 B instead of the 432 B
    // of the 27 stored planes) and the three per-axis factor tables Lx | Ly | Lz (3 n each); valid while otf3 is set (cleared when something else writes d_C)
    cplx *d_K3 = nullptr; double *d_b3 = nullptr; cplx *d_L3 = nullptr; size_t l3_elems = 0;
    bool otf3 = false; double otf_idx2 = 0, otf_idy2 = 0, otf_idz2 = 0, otf_blend = 0.5;
    int fstream_prio = 0;                                  // priority class fstream was acquired with (it goes back to that class's free list)
    double *d_rho = nullptr, *d_theta = nullptr, *d_eps = nullptr, *d_delta = nullptr;
    bool has_model = false, aniso = false;

    // host copies of the model (coarse levels of the multigrid preconditioner are built from them; filled on demand by
    // helm_ensure_host_model)
    std::vector<cplx> h_c; std::vector<double> h_rho, h_theta, h_eps, h_delta;

    // operator
    bool block0_only = false;     // Eurus preconditioner levels: assemble/keep only M1
    double pml_scale = 1.0;       // MiniZephyr preconditioner levels: scales the PML factors (1 = reference)
    std::vector<cplx> lap_override;   // 3-D preconditioner levels on stretched grids: Lx(-1,0,+1)[nx] | Ly[ny] | Lz[nz] including 1/h^2 (helm3d.hip)
    double diag_floor = 0.0;      // preconditioner levels: floor on |diag| as a fraction of the row's absolute sum (smoother safeguard)
    double a_freq_re = 0, a_freq_im = 0, a_tau = 0, a_ky = 0, a_cpml = 0;   // parameters of the last assemble
    struct MgPrecond *mg = nullptr;
    struct Mg3Precond *mg3 = nullptr;    // 3-D multigrid preconditioner (mg3d.hip)
    bool mg3_no_keep = false;            // this frequency retreated from the layer-preserving hierarchy to the standard cycle (capi.hip)
    int mg3_rhs_hint = 0;                // right-hand sides of the solve call that builds the preconditioner (0: unknown): few of them favour a cheap set-up
    struct NdFactor *direct[4] = {nullptr, nullptr, nullptr, nullptr};   // sparse direct factors per block, valid until the next assemble
    bool direct_failed = false;
    int nblocks = 1;
    cplx *d_C = nullptr;      // nblocks * 9 * N   raw planes
    cplx *d_Cs = nullptr;     // nblocks * 9 * N   planes divided by the centre plane (Jacobi-scaled)
    cplx *d_dinv = nullptr;   // nblocks * N       1 / centre plane
    cplx *d_S = nullptr;      // coupled Eurus system: the four blocks scaled by the inverse 2-norm of their system row (36N), on demand
    double *d_rs = nullptr;   // 2N inverse row norms of the 2N x 2N system
    bool assembled = false;
    int asm_nblk = 4, blocks_ready = 0;   // Eurus: blocks the next assembly writes (1: M1 only -- the block-triangular N-row solve needs nothing else) / blocks d_C holds now
    bool scaled_ok = false;       // d_Cs / d_dinv hold the current operator (made on demand: only the Krylov paths need them)
    bool block_zero[4] = {false, false, false, false};   // block is identically zero (e.g. Eurus M3 isotropic)

    // solver workspace (grown on demand)
    void *d_ws = nullptr; size_t ws_bytes = 0;
    void *d_part = nullptr; size_t part_bytes = 0;     // partial sums of the fused dot products
    RhsScal *d_scal = nullptr; RhsScal *h_scal = nullptr; int scal_cap = 0; size_t h_scal_bytes = 0;

    // timing / profiling
    bool profiling = false;
    helm_timing timing = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<int, double>> ev_pending;   // (event-pair index, bytes)
    std::vector<std::pair<int, double>> ev_pending_gemm;   // (event-pair index, flops) of the direct solver's GEMM launches / runs of launches
    std::vector<int> ev_pending_gemm_n;                    // launches covered by each pair
    bool rtol_hint_set = false;                            // a caller or a solve has stated its tolerance on this handle (else rtol_hint is only the default)
    double rtol_hint = 1e-10;                              // tolerance the next factorisation is conditioned for (helm_set_tolerance_hint; every solve records its own)
    std::vector<long long> ev_pending_gemm_shape;          // (HELM_GEMM_LOG=1) M, N, K, batch, addressing mode of each record, five entries apiece
    std::vector<double> ev_pending_gemm_bytes, ev_pending_gemm_sol;   // operand bytes and roofline time (ms) of the same launches
    int gemm_run_depth = 0, gemm_run_launches = 0;         // back-to-back GEMM launches timed with ONE event pair (direct.hip)
    double gemm_run_flops = 0, gemm_run_bytes = 0, gemm_run_sol = 0;
    int gemm_run_pair = -1;
    size_t ev_used = 0;
    int active_hint = -1;        // right-hand sides currently iterating (for the byte count of profiled launches)

    std::string err;
};

#define HELM_FAIL(op, code, ...) do { char _b[512]; snprintf(_b, sizeof(_b), __VA_ARGS__); helm_set_error(op, _b); return code; } while (0)
#define HIP_TRY(op, call) do { hipError_t _e = (call); if (_e != hipSuccess) { \
    HELM_FAIL(op, HELM_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); } } while (0)

void helm_set_error(helm_op *op, const char *msg);

// ---- kernel registry (capi.hip) -------------------------------------------------------------------
// The HIP runtime resolves a kernel lazily: the first launch of a symbol on a device looks it up in the code object and builds its dispatch
// record (tens to hundreds of microseconds of host time, 1.8 ms for the nine kernels of a tree level in a cold process).  A job meets some
// instantiations only at some frequencies (pivoted leaves, the pivoted-LU treatment of ill-conditioned fronts, refinement widths), i.e. possibly
// for the first time in the middle of a timed region.  Every kernel instantiation that has a launch site therefore registers its host handle when
// the library is loaded (HelmKernelReg: a static data member per instantiation), helm_warm() resolves all of them on a device without launching
// anything, and helm_debug_runtime_stats() counts the first launches that still happen (with their host time).
int helm_kernel_register(const void *fn, const char *pretty);
bool helm_kernel_first_launch(int slot);                  // true exactly once per slot
void helm_kernel_first_launch_done(int slot, double host_ms);
template <auto K> struct HelmKernelReg {
    static const char *pretty() { return __PRETTY_FUNCTION__; }
    static inline const int slot = helm_kernel_register((const void *)K, pretty());
};
struct HelmFirstLaunch {
    int slot; bool first; double t0;
    static double now_ms();
    explicit HelmFirstLaunch(int s) : slot(s), first(helm_kernel_first_launch(s)), t0(first ? now_ms() : 0.0) {}
    ~HelmFirstLaunch() { if (first) helm_kernel_first_launch_done(slot, now_ms() - t0); }
};
#define HELM_LAUNCH(KERNEL, ...) do { HelmFirstLaunch fl_(HelmKernelReg<(KERNEL)>::slot); hipLaunchKernelGGL(KERNEL, __VA_ARGS__); } while (0)

// ---- runtime-object bookkeeping (capi.hip) ----------------------------------------------------------
// Every call of the library that makes the HIP runtime create something -- device memory, pinned memory, an event, a stream -- goes through a
// counting wrapper (the function-like macros below catch the calls of every translation unit; `(hipMalloc)(...)` is how the wrappers reach the
// real entry points).  helm_debug_runtime_stats() reports the counts: a job whose pools were booked (helm_reserve, warm-up items) must show
// zeros across its timed region.
hipError_t helm_counted_malloc(void **p, size_t bytes);
hipError_t helm_counted_free(void *p);
hipError_t helm_counted_host_malloc(void **p, size_t bytes, unsigned flags);
hipError_t helm_counted_event_create(hipEvent_t *e, unsigned flags);
hipError_t helm_counted_stream_create(hipStream_t *s, unsigned flags, int prio, bool with_prio);
// host-side waits: how long the calling thread sat in each (HELM_SYNC_TRACE=<ms>: waits longer than that are reported on stderr with their call site;
// helm_debug_runtime_stats counts those of at least 10 ms -- a wait that long on a GPU whose kernels take 0.03-3 ms is a stall, not work)
hipError_t helm_timed_stream_sync(hipStream_t s, const char *file, int line);
hipError_t helm_timed_event_sync(hipEvent_t e, const char *file, int line);
hipError_t helm_timed_device_sync(const char *file, int line);
hipError_t helm_timed_memcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, const char *file, int line);
#define hipStreamSynchronize(S) helm_timed_stream_sync((S), __FILE__, __LINE__)
#define hipEventSynchronize(E) helm_timed_event_sync((E), __FILE__, __LINE__)
#define hipDeviceSynchronize() helm_timed_device_sync(__FILE__, __LINE__)
#define hipMemcpy(D, S, B, K) helm_timed_memcpy((D), (S), (B), (K), __FILE__, __LINE__)
#define hipMalloc(P, B) helm_counted_malloc((void **)(P), (B))
#define hipFree(P) helm_counted_free((void *)(P))
#define hipHostMalloc(P, B, F) helm_counted_host_malloc((void **)(P), (B), (F))
#define hipEventCreate(E) helm_counted_event_create((E), 0u)
#define hipEventCreateWithFlags(E, F) helm_counted_event_create((E), (F))
#define hipStreamCreateWithFlags(S, F) helm_counted_stream_create((S), (F), 0, false)
#define hipStreamCreateWithPriority(S, F, PR) helm_counted_stream_create((S), (F), (PR), true)
helm_tuning helm_tuning_now();                            // the options in force (helm_set_tuning, else defaults + environment; include/helm.h)
void helm_tuning_refresh();                               // look at the environment again (API entry points call this; nothing below them does)

// Size-keyed cache of large device buffers (coefficient planes, factors, per-call temporaries): a job walks through many
// operators of identical shape, and hipMalloc/hipFree of GB-sized buffers cost milliseconds each.  helm_trim() empties it.
void *helm_pool_alloc(int device, size_t bytes);          // nullptr on failure
void helm_pool_free(int device, void *p, size_t bytes);   // the buffer must no longer be in use by any stream
size_t helm_pool_idle_bytes(int device);                  // what the pool holds idle on that device (not in hipMemGetInfo's free figure)
hipError_t helm_malloc_retry(int device, void **p, size_t bytes);   // hipMalloc; on failure the device's idle pool is emptied and the call repeated
void *helm_hostpool_alloc(size_t bytes);                  // pinned host memory, recycled by size
int helm_download_staged(helm_op *op, void *dst, const void *src, size_t bytes);    // device -> caller's host array, the same way
int helm_upload_staged(helm_op *op, void *dst, const void *src, size_t bytes);      // caller's host array -> device on op->stream through the library's pinned chunks (capi.hip); returns when the copy is done
void helm_hostpool_free(void *p, size_t bytes);
hipStream_t helm_stream_acquire(int device, int prio);    // prio 0 normal, 1 highest, -1 lowest; recycled across handles
void helm_stream_release(int device, int prio, hipStream_t s);
void helm_pf_retire(helm_op *op);                         // wait for / book / clean up a factorisation started by helm_prefactor
int helm_ensure_scaled(helm_op *op);
int helm_need_all_blocks(helm_op *op);                   // Eurus operators assembled lazily (M1 only): bring M2..M4 into being before anything reads them
int helm_events_grow(helm_op *op, int n);                 // n more timing events for the handle (recycled across handles)                      // d_Cs, d_dinv for the operator currently assembled

// ---- multigrid preconditioner (mg.hip) ---------------------------------------------------------
struct MgPrecond;
int mg_setup(helm_op *op, int batch);                       // (re)build for the operator's current frequency
void mg_destroy(helm_op *op);
// out[b] = M^-1 in[b] for the active right-hand sides (scal may be null = all)
int mg_apply(helm_op *op, const cplx *in, cplx *out, int nrhs, const RhsScal *scal);

// 3-D counterpart (mg3d.hip): layer-preserving hierarchy with a direct coarse solve on oversampled grids, otherwise a
// shifted-Laplacian V-cycle with damped-Jacobi smoothing and a weak absorbing layer
int mg3_setup(helm_op *op, int batch);
void mg3_retarget_stream(helm_op *op, hipStream_t st);
void mg3_destroy(helm_op *op);
int mg3_apply(helm_op *op, const cplx *in, cplx *out, int nrhs);
void mg3_record_iterations(helm_op *op, double mean_iterations, double rtol);   // books what a solve through the layer-preserving hierarchy needed (depth model)
bool mg3_is_layer_preserving(const helm_op *op);           // the hierarchy in use is the layer-preserving one
int mg3_retreat(helm_op *op, int batch);                   // rebuild as the standard cycle for the rest of this frequency

// ---- launchers implemented in assemble.hip ----------------------------------------------------
int helm_launch_assemble(helm_op *op, double freq_re, double freq_im, double tau, double ky, double cPML);

// ---- launchers implemented in kernels.hip -----------------------------------------------------
// epilogues of the fused stencil kernel
enum { EPI_NONE = 0,      // y = A x
       EPI_DOT_W = 1,     // + partial (w, y)                      [BiCGSTAB (r0, v)]
       EPI_DOT_XY = 2,    // + partials (y, x), (y, y)             [BiCGSTAB (t,s),(t,t)]
       EPI_DOT_YY = 3,    // + partial (y, y)                      [CGNR]
       EPI_RESID = 4,     // y = w - A x, + partial (y, y)         [restart / true residual / MG residual]
       EPI_JACOBI = 5,    // y = x + omega_j * dinv * (w - A x)    [MG smoother sweep]
       EPI_DOT_WY = 6 };  // y = A x, + partials (y, w), (y, y)    [right-preconditioned BiCGSTAB (t,s),(t,t)]

struct ApplyArgs {
    const cplx *planes = nullptr;   // 9 planes (raw or scaled), plane stride = N
    const cplx *X = nullptr; cplx *Y = nullptr; const cplx *W = nullptr;   // rhs stride = ld
    long long ld = 0;
    int nrhs = 0;
    int scaled = 0;           // 1: centre plane is 1 and skipped
    int adjoint = 0;
    int epi = 0;
    const RhsScal *scal = nullptr;  // may be null: all RHS active
    double *part = nullptr;         // partial sums, layout [rhs][q][nblk] doubles (q = 0..3)
    const cplx *dinv = nullptr;     // EPI_JACOBI: 1/diag
    double omega_j = 0.0;           // EPI_JACOBI: damping
    const int *tiles = nullptr;     // optional list of tile ids to process (frame tiles of the strip relaxation)
    int ntiles = 0;
    int planes_tiled = 0;           // 1: `planes` points to the tile-blocked copy of the planes
    int acc = 0;                    // 1: y = Y_old + A x (second half of a two-block row of the coupled Eurus system)
    int part_stride = 0, part_off = 0;   // partial-sum layout override: [rhs][q][part_stride], this launch writes at part_off + block
    int xmode = 0;                  // 1: input = omega_j dinv (.) W (also written to U), with EPI_RESID; 2: input = X + P E, with EPI_JACOBI
    cplx *U = nullptr; const cplx *E = nullptr; int nzc = 0, nxc = 0;
    int f32 = 0;                    // 1: planes / X / Y / W / dinv are single-precision complex (multigrid levels);
                                    //    only EPI_NONE / EPI_RESID / EPI_JACOBI, unscaled, forward
    int profile = 1;                // count this launch in the roofline timing of the handle that owns the solve
    int x32 = 0, w32 = 0, y32 = 0;  // (3-D) X / W / Y hold complex64 (the multigrid cycle's finest-level vectors, mg3d.hip); the arithmetic stays fp64
};
int helm_launch_apply(helm_op *op, const ApplyArgs &a);

// ---- 3-D operator (helm3d.hip) -------------------------------------------------------------------
int helm3d_launch_assemble(helm_op *op, double freq_re, double freq_im, double tau, double cPML);
int helm3d_launch_apply(helm_op *op, const ApplyArgs &a, hipEvent_t e0, hipEvent_t e1);
int helm3d_apply_num_blocks(const helm_op *op);

int helm_apply_num_blocks(const helm_op *op);
int helm_stencil_tile_rows();     // rows of the 64-wide stencil tile (4 * STENCIL_P)

int helm_launch_scale_planes(helm_op *op);   // d_Cs, d_dinv from d_C

struct VecPtrs {   // all [nrhs][N] complex, stride N
    cplx *x, *r, *r0, *p, *v, *s, *t;
};
// BiCGSTAB
int helm_launch_bicg_init(helm_op *op, int block, const cplx *dRHS, long long rhs_ld, cplx premul, const cplx *sub, VecPtrs w, int nrhs, double rtol);
int helm_launch_bicg_p(helm_op *op, VecPtrs w, int nrhs);
int helm_launch_bicg_s(helm_op *op, VecPtrs w, int nrhs);
int helm_launch_bicg_xr(helm_op *op, VecPtrs w, const cplx *xp, const cplx *xs, int nrhs);   // x += alpha xp + omega xs ; r = s - omega t
int helm_launch_fin(helm_op *op, int which, int nrhs, int nblk_part);
enum { FIN_BICG_INIT = 0, FIN_ALPHA = 1, FIN_OMEGA = 2, FIN_RHO = 3, FIN_RESTART = 4,
       FIN_CG_INIT = 5, FIN_CG_ALPHA = 6, FIN_CG_RR = 7, FIN_CG_BETA = 8, FIN_NORM = 9,
       FIN_NORM2 = 10 /* aux[b] = slot 0, aux[nrhs + b] = slot 1 */ };
int helm_vec_num_blocks(const helm_op *op);
// CGNR
int helm_launch_cg_xr(helm_op *op, VecPtrs w, int nrhs);      // x += alpha p ; r -= alpha w(v) ; (r,r)
int helm_launch_cg_p(helm_op *op, VecPtrs w, int nrhs, int first);   // p = z(s) + beta p
// misc
int helm_launch_finish(helm_op *op, const cplx *x, cplx *dU, long long u_ld, int nrhs, long long row_off);   // U = conj(x)
int helm_launch_finish_ex(helm_op *op, const cplx *x, long long x_ld, long long x_off, cplx *dU, long long u_ld, long long row_off, int nrhs);
int helm_launch_prep_rhs_ex(helm_op *op, const cplx *dRHS, long long rhs_ld, long long row_off, cplx premul, const cplx *scale,
                            cplx *out, long long out_ld, long long out_off, int nrhs);
int helm_launch_rowscaled_system(helm_op *op);    // d_S, d_rs
int helm_launch_prep_rhs_rs(helm_op *op, const cplx *dRHS, long long rhs_ld, long long row_off, cplx premul, const double *rs,
                            cplx *out, long long out_ld, long long out_off, int nrhs);
int helm_launch_prep_rhs(helm_op *op, const cplx *dRHS, long long rhs_ld, long long row_off, cplx premul, const cplx *sub, cplx *out, int nrhs); // out = premul*rhs - sub
int helm_launch_prep_rhs_norm(helm_op *op, const cplx *dRHS, long long rhs_ld, long long row_off, cplx premul, const cplx *sub, cplx *out, int nrhs);
int helm_launch_imaging(helm_op *op, const cplx *uf, const cplx *ub, int nsrc, const cplx *scaler, cplx *g);
int helm_launch_zero(helm_op *op, cplx *p, long long n);
int helm_launch_rhs_from_coo(helm_op *op, const long long *row, const int *col, const cplx *val, long long nnz, cplx *R, int nrhs, long long rows, int node_major = 0);
int helm_launch_sample(helm_op *op, const cplx *U, int nsrc, long long ld, const long long *rowptr, const long long *col, const cplx *val, int nrec, cplx *out);
int helm_launch_rowscale_inplace(helm_op *op, cplx *v, const double *rs, long long NV, int nrhs);
int helm_launch_abs(helm_op *op, const cplx *in, cplx *out, long long n, double sign);      // out = sign |in|
int helm_launch_gardner_rho(helm_op *op);     // d_rho = 310 Re(d_c)^0.25
int helm_ensure_host_model(helm_op *op);      // h_c, h_rho, ... (downloaded from the device on first use)
int helm_adopt_model_device(helm_op *dst, const cplx *d_c, const double *d_rho);     // model of a multigrid level from device arrays (capi.hip)
