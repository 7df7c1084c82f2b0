/*
 * helm.h -- C ABI of libhelm: MI355X-native frequency-domain Helmholtz operator
 *           (assemble the 9-point 2-D / 27-point 3-D PML-damped complex stencil on the GPU, apply it,
 *           and solve A(w) u = q for many right-hand sides: in 2-D by a sparse direct factorisation
 *           kept per frequency -- nested dissection, multifrontal, batched dense kernels -- with the
 *           stencil kernel checking the true residual; in 3-D and as fallback by multigrid-
 *           preconditioned / Jacobi-preconditioned BiCGSTAB and CGNR built on the same stencil kernel).
 *
 * This is the drop-in boundary for the hot path of uwoseis/zephyr's backend.  Each entry
 * point names the reference interface it replaces (paths relative to the reference tree):
 *
 *   helm_create / helm_set_model   <- BaseModelDependent / BaseDiscretization config ingestion
 *                                     zephyr/backend/base.py:11-109, discretization.py:23-76
 *   helm_assemble                  <- MiniZephyr._initHelmholtzNinePoint  minizephyr.py:40-298
 *                                     Eurus._initHelmholtzNinePoint       eurus.py:28-485
 *   helm_get_diagonals             <- the `.A` property                   minizephyr.py:300-306, eurus.py:487-492
 *   helm_apply[_device]            <- `A * x` (scipy CSR product)         used by tests / parity
 *   helm_solve[_device]            <- BaseDiscretization.__mul__          discretization.py:78-106
 *                                     (problemo.BestSolver LU + premul + conjugate),
 *                                     Eurus.__mul__ pad/clip              eurus.py:512-533
 *   helm_prefactor                 <- the worker pool of BaseMPDist.__mul__ (distributors.py:80-96,161-168): the LU of the
 *                                     NEXT frequency is built while the current one is being solved
 *   helm_reserve                   <- (same pool: its workers' scratch, allocated before they start)
 *   helm_imaging_accumulate_device <- zero-lag imaging condition in HelmBaseProblem.Jtvec
 *                                     zephyr/middleware/problem.py:152,162
 *   helm_destroy                   <- `del obj.factors` / __del__         discretization.py:86-99
 *
 * Conventions
 *   - complex128 values are interleaved (re, im) doubles; a field is the row-major (nz, nx)
 *     ravel of the reference (linear index iz*nx + ix, base.py:93).
 *   - multi-RHS buffers hold each right-hand side contiguously: X[r*rows + i]   (the Python
 *     host passes np.ascontiguousarray(rhs.T)).
 *   - coefficient planes: C[block][k][iz*nx+ix], k = 3*(dz+1) + (dx+1), meaning
 *     (A x)[iz,ix] = sum_k C[k][iz,ix] * x[iz+dz, ix+dx].  MiniZephyr has 1 block, Eurus 4
 *     (M1, M2, M3, M4 of the 2N x 2N system, eurus.py:430-464).
 *   - every function returns an int status: 0 = ok, > 0 = number of right-hand sides that
 *     did not reach the tolerance (helm_solve only), < 0 = hard error (HELM_ERR_*); the text
 *     is available from helm_last_error().  No exceptions cross this boundary.
 *   - the caller owns host buffers; the library owns all device memory behind the handle.
 *     `_device` variants take device pointers valid on the handle's GPU.
 *   - a handle is bound to one GPU and must be used by one host thread at a time.
 */
#ifndef HELM_H
#define HELM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct helm_op helm_op;

/* discretisation variants */
enum { HELM_MINIZEPHYR = 0, HELM_EURUS = 1, HELM_3D = 2 /* 27-point 3-D operator, created with helm_create3d */ };

/* Krylov methods */
enum {
    HELM_BICGSTAB = 0,   /* Jacobi-preconditioned BiCGSTAB */
    HELM_CGNR = 1,       /* Jacobi-scaled CGNR */
    HELM_AUTO = 2,       /* HELM_DIRECT for every 2-D system; if that fails, and in 3-D: HELM_MG where available, else
                            BiCGSTAB, with CGNR for right-hand sides that break down */
    HELM_MG = 3,         /* BiCGSTAB right-preconditioned by shifted-Laplacian multigrid (damped-Jacobi smoothing)
                            + PML strip line relaxation */
    HELM_DIRECT = 4      /* sparse direct: nested-dissection multifrontal factorisation kept on the handle until the next
                            helm_assemble and re-used for every right-hand side (what the reference's sparse LU does,
                            discretization.py:78-103), plus iterative refinement with the stencil kernel to rtol.
                            2-D: MiniZephyr, Eurus with eps == delta (block-triangular) and the coupled TTI system
                            (eps != delta, two unknowns per cell, row-equilibrated); not the 3-D operator */
};

/* hard errors */
enum {
    HELM_OK = 0,
    HELM_ERR_ARG = -1,         /* bad argument / dimension mismatch (reference: ValueError) */
    HELM_ERR_DEVICE = -2,      /* HIP runtime failure (no GPU, allocation, launch) */
    HELM_ERR_STATE = -3,       /* call order: model/assemble missing */
    HELM_ERR_UNSUPPORTED = -4, /* configuration outside the supported path */
    HELM_ERR_PML = -5          /* Eurus PML profile length hazard (reference raises ValueError, eurus.py:84-91) */
};

/* buffer layouts of the solve entry points (helm_solve_opts.flags).  Default (0): one right-hand side / wavefield per row,
 * X[r*rows + i] (what np.ascontiguousarray(rhs.T) gives).  NODE_MAJOR: the reference's own (rows, nrhs) C-order arrays,
 * X[i*nrhs + r] -- what `Disc * rhs` receives and what lu.solve returns (discretization.py:101-103).  With both set the direct
 * path uses the right-hand sides where they lie and writes the wavefield from the kernel that checks its residual: no transposes. */
enum { HELM_RHS_NODE_MAJOR = 1, HELM_OUT_NODE_MAJOR = 2, HELM_NODE_MAJOR = 3 };

typedef struct helm_solve_opts {
    int method;        /* HELM_BICGSTAB | HELM_CGNR | HELM_AUTO | HELM_MG | HELM_DIRECT */
    double rtol;       /* stop when ||q' - A u||_2 / ||q'||_2 <= rtol (q' = premul*rhs), per RHS */
    int maxit;         /* iteration cap per right-hand side */
    int check_every;   /* iterations between host-side convergence checks (0 = default) */
    int batch;         /* right-hand sides iterated together on the device (0 = default) */
    int flags;         /* buffer layouts: 0, or HELM_RHS_NODE_MAJOR | HELM_OUT_NODE_MAJOR */
} helm_solve_opts;

typedef struct helm_solve_info {
    int iterations;    /* iterations used by this right-hand side (HELM_DIRECT: triangular solves incl. refinement) */
    int status;        /* 0 converged, 1 iteration cap / stalled above rtol, 2 breakdown (not recovered), 3 direct path (in practice the coupled TTI system):
                        * the true residual has reached the floor fp64 allows for this right-hand side, relres <= 8 eps || |A||x| + |q| || / ||q||,
                        * which lies above rtol -- no fp64 vector does better (a backward-stable LU lands on the same floor); counted as solved */
    int restarts;      /* BiCGSTAB restarts taken */
    int method;        /* method that produced the returned wavefield */
    double relres;     /* final TRUE relative residual ||q' - A u|| / ||q'|| (unscaled system) */
} helm_solve_info;

/* --- lifecycle ------------------------------------------------------------------------ */
int helm_device_count(void);                   /* < 0 on HIP failure */
const char *helm_version(void);

/* freeSurf: 4 ints, reference order [iz=0 side, ix=nx-1 side, iz=nz-1 side, ix=0 side]
 * (minizephyr.py:103-115,269-298); ignored by Eurus as in the reference.
 * Returns NULL on failure; helm_last_error(NULL) holds the reason. */
helm_op *helm_create(int device, int variant, int nz, int nx, double dx, double dz,
                     int nPML, const int *freeSurf);
/* 3-D 27-point operator on an (nz, ny, nx) grid, linear index (iz*ny + iy)*nx + ix; 27 planes,
 * k = 9*(oz+1) + 3*(oy+1) + (ox+1).  No reference counterpart (the reference has no 3-D discretisation:
 * zephyr/backend/base.py:20,36-40, source.py:43-44); definition in oracle/helm3d_oracle.py.  All other entry
 * points (set_model with theta/eps/delta NULL, assemble with ky ignored and cPML the layer amplitude,
 * get_diagonals, apply, solve with rows = N) work on the returned handle; solves use the Jacobi-preconditioned
 * BiCGSTAB / CGNR, right-preconditioned by a 3-D shifted-Laplacian multigrid under HELM_MG / HELM_AUTO. */
helm_op *helm_create3d(int device, int nz, int ny, int nx, double dx, double dy, double dz, int nPML);
void helm_destroy(helm_op *op);
const char *helm_last_error(const helm_op *op);

/* Use an existing hipStream_t (e.g. torch's current stream) for all work; NULL = own stream. */
int helm_set_stream(helm_op *op, void *hip_stream);

/* --- model and assembly --------------------------------------------------------------- */
/* c: N complex; rho: N doubles or NULL (Gardner 310*Re(c)^0.25, discretization.py:70);
 * theta/eps/delta: N doubles each or NULL (zeros), Eurus only (base.py:112-149). */
int helm_set_model(helm_op *op, const double *c, const double *rho,
                   const double *theta, const double *eps, const double *delta);

/* freq complex (Hz); tau Laplace damping time constant (inf = none); ky cross-line wavenumber
 * (MiniZephyr only); cPML Eurus C-PML amplitude (eurus.py:500-504). */
int helm_assemble(helm_op *op, double freq_re, double freq_im, double tau, double ky, double cPML);

int helm_num_blocks(const helm_op *op);        /* 1 (MiniZephyr) or 4 (Eurus) */
long long helm_num_points(const helm_op *op);  /* N = nz*nx */

/* out: num_blocks * 9 * N complex, layout C[block][k][i] */
int helm_get_diagonals(helm_op *op, double *out);

/* --- operator application (parity / microbenchmark) ----------------------------------- */
/* Y[r] = M_block X[r], r < nrhs; adjoint != 0 applies the conjugate transpose. */
int helm_apply(helm_op *op, int block, int adjoint, const double *X, double *Y, int nrhs);
int helm_apply_device(helm_op *op, int block, int adjoint, const void *dX, void *dY, int nrhs);

/* --- solve ---------------------------------------------------------------------------- */
/* U[r] = conj( A^-1 (premul * RHS[r]) ).  rows = N, or 2N for Eurus stacked right-hand sides
 * (eurus.py:512-533: N-row input is zero-padded and the result clipped to N rows; 2N-row input
 * returns 2N rows).  info: nrhs entries or NULL.  opts NULL = defaults.  RHS and U may be the same buffer (in place) or disjoint; they
 * must not overlap partially. */
int helm_solve(helm_op *op, const double *RHS, double *U, int nrhs, long long rows,
               double premul_re, double premul_im,
               const helm_solve_opts *opts, helm_solve_info *info);
int helm_solve_device(helm_op *op, const void *dRHS, void *dU, int nrhs, long long rows,
                      double premul_re, double premul_im,
                      const helm_solve_opts *opts, helm_solve_info *info);

/* Sparse right-hand sides given as host COO triplets (row: int64, col: int32 = right-hand side, val: complex128; no duplicates) --
 * the reference's scipy-sparse source matrices (survey.py:162-169), which problemo densifies on the host before the LU solve.
 * Here only the triplets cross PCIe; U (host, nrhs*rows complex128, layout per opts->flags) receives the wavefields. */
int helm_solve_coo(helm_op *op, const long long *row, const int *col, const double *val, long long nnz, double *U, int nrhs,
                   long long rows, double premul_re, double premul_im, const helm_solve_opts *opts, helm_solve_info *info);
/* Pinned host memory for result arrays (device-to-host copies into it run at the PCIe rate); recycled by size.
 * Host arrays in general (r6): every entry point above that takes one (helm_set_model, helm_apply, helm_solve, helm_solve_coo, helm_get_diagonals) moves it through
 * pinned 4-MB chunks of the library unless the array is pinned memory itself (this call, hipHostMalloc, hipHostRegister) -- then it is copied straight, which is
 * what a caller with GBs of wavefields wants.  The caller's pageable pages are never registered with the runtime: memory registered that way and later unmapped
 * (free() of a large array) makes the kernel driver evict every queue of the process for 15-20 ms (profiles/r06_config4_dpred_spread.txt). */
void *helm_host_alloc(size_t bytes);
void helm_host_free(void *p, size_t bytes);

/* Start the factorisation the next helm_solve[_device] on this handle will need, without waiting for it: the launches go to
 * a high-priority stream of the handle and run beside whatever other handles are doing on the GPU (a dispatcher solving
 * frequency k calls this for frequency k+1: the latency-bound top of the elimination tree then hides under the
 * bandwidth-bound triangular solves of the previous frequency).  A hint: returns HELM_OK without doing anything where the
 * direct path does not apply (3-D, coupled TTI) or factors already exist.  Requires helm_assemble.  The reference builds its
 * LU lazily inside the first `Disc * rhs` (discretization.py:78-85) and overlaps frequencies with a process pool
 * (distributors.py:161-168). */
int helm_prefactor(helm_op *op);
/* The same with the number of right-hand sides the next solve will bring (the reference's pool hands a sub-problem its right-hand sides together
 * with the job: distributors.py:161-166).  2-D: helm_prefactor.  3-D: the multigrid hierarchy and the
 * factorisation of its directly solved level are built NOW, in the calling thread (hundreds of ms): a dispatcher's prepare thread calls this
 * for frequency k+1 while another handle iterates on frequency k.  A hint like helm_prefactor. */
int helm_prefactor_n(helm_op *op, int nrhs);
/* helm_prefactor for n operators of the same grid on one GPU at once (2-D; n <= 4): their factorisations are enqueued TOGETHER -- the elimination tree depends
 * on the grid alone, so the fronts of n frequencies ride in the same strided batches, batch index = front x frequency.  The top of the tree is a chain of small
 * dependent launches that leaves most of the chip idle (88 block steps of ~30 us at 1024^2, gather-bound levels of 8-16 unknowns, products over one to four
 * fronts); that chain is now walked once per set.  Each operator's factors come out bit for bit what helm_prefactor would have made of them and are solved with
 * as usual, one operator at a time; the launches go to the high-priority stream of ops[0].  A hint: operators that do not qualify (3-D, coupled TTI, factors
 * already there, different grids) are prefactored one by one.  The reference overlaps frequencies with a process pool (distributors.py:161-168). */
int helm_prefactor_many(helm_op **ops, int n);
/* The relative tolerance the solves on this operator will ask for, told before its factors are built (helm_prefactor takes no options):
 * fronts whose condition estimate exceeds rtol / (8 eps) are re-eliminated with a pivoted LU so that one pass meets rtol.  Default 1e-10;
 * a solve that builds the factors itself uses its own opts->rtol.  Counterpart of the `Solver` the reference receives through its
 * systemConfig (discretization.py:23-31,83-84: SuperLU's pivoting is not negotiable there). */
int helm_set_tolerance_hint(helm_op *op, double rtol);

/* Scratch for `concurrent` host-array solves (helm_solve / helm_solve_coo) of `nrhs` right-hand sides with `rows` rows running at the
 * same time on this handle's GPU, allocated now instead of inside the first solves: the shared scratch slots of the direct path and the
 * device images of right-hand sides and wavefields.  For dispatchers that start several workers per GPU -- the worker pool of
 * BaseMPDist.__mul__ (distributors.py:80-96): a hipMalloc issued beside running kernels and copies can take a second.  A hint:
 * HELM_OK unless the arguments are invalid. */
int helm_reserve(helm_op *op, int nrhs, long long rows, int concurrent);

/* Timing of the last solve/apply on this handle, measured with HIP events on the handle's
 * stream: total ms, and ms / launches / algorithmic bytes of the stencil-apply kernel. */
typedef struct helm_timing {
    double solve_ms;
    double apply_ms;
    long long apply_launches;
    double apply_bytes;      /* sum over launches of N*(32*B + 144) */
    double factor_ms;        /* HELM_DIRECT: factorisation time when this solve had to factor, else 0 */
    double gemm_ms;          /* HELM_DIRECT: ms / launches / flops (8 M N K per batch item) of the dense complex GEMM kernel */
    long long gemm_launches;
    double gemm_flops;
    double gemm_big_ms;      /* the same over the launches of at least 1 GFLOP (throughput-bound ones) */
    long long gemm_big_launches;
    double gemm_big_flops;
    double gemm_bytes;       /* operand bytes of the same launches, every operand once: 16 (M K + K N + M N (1 or 2)) per batch item */
    double gemm_sol_ms;      /* sum over the launches of max(flops / 78.6 TFLOP/s, bytes / 8 TB/s): what the two roofs allow */
} helm_timing;
int helm_last_timing(const helm_op *op, helm_timing *out);
/* enable per-launch HIP-event timing of the stencil kernel inside solves (costs a little) */
int helm_set_profiling(helm_op *op, int on);

/* --- FWI imaging condition ------------------------------------------------------------ */
/* G[i] += scaler[i] * sum_{s<nsrc} UF[s][i] * UB[s][i]   (complex, no conjugation;
 * problem.py:152).  scaler: N complex (= -w^2/c^3).  All device pointers. */
int helm_imaging_accumulate_device(helm_op *op, const void *dUF, const void *dUB, int nsrc,
                                   const void *dScaler, void *dG);

/* --- device-resident callers: sparse sources in, receiver samples out ---------------------------------- */
/* Dense right-hand sides from the COO triplets of the reference's sparse source matrix (survey.py:162-169,
 * source.py:213-317): R (nrhs x rows, zeroed by the call), R[col[k]][row[k]] = val[k]; no duplicate entries.
 * row: int64, col: int32, val: complex128, all device pointers. */
int helm_rhs_from_coo_device(helm_op *op, const void *d_row, const void *d_col, const void *d_val, long long nnz,
                             void *dR, int nrhs, long long rows);
/* the same with the layout of R chosen by `flags` (HELM_RHS_NODE_MAJOR: R[row[k]*nrhs + col[k]]) */
int helm_rhs_from_coo_device_layout(helm_op *op, const void *d_row, const void *d_col, const void *d_val, long long nnz,
                                    void *dR, int nrhs, long long rows, int flags);

/* Declared support of the next solve's right-hand sides (the reference's sources are scipy-sparse matrices, survey.py:86-89,162-188: where they
 * are nonzero is part of the input).  d_bits: one byte per row on the device, bit b = block b of 64 right-hand sides may be nonzero in that row; the
 * caller guarantees zeros elsewhere.  One shot: applies to the next helm_solve_device on this handle (same rows / nrhs, node-major layout), then
 * forgotten; NULL clears it.  HELM_ND_SUPPORT_CHECK=1 verifies the guarantee and fails the solve if it does not hold. */
int helm_set_rhs_support(helm_op *op, const void *d_bits, long long rows, int nrhs);
/* d_bits ((rows + 3) / 4 * 4 bytes on the device) from the triplets of a sparse right-hand-side matrix (device arrays, as for
 * helm_rhs_from_coo_device_layout) */
int helm_rhs_support_from_coo(helm_op *op, const void *d_row, const void *d_col, long long nnz, void *d_bits, long long rows, int nrhs);
/* Receiver sampling data = R u (survey.py:152-160): out[r][s] = sum_k val[k] * U[s][col[k]] over the entries k of CSR
 * row r (rowptr, col: int64; val: complex128; U: nsrc x ld; out: nrec x nsrc complex128; device pointers). */
int helm_sample_device(helm_op *op, const void *dU, int nsrc, long long ld, const void *d_rowptr, const void *d_col,
                       const void *d_val, int nrec, void *d_out);

/* Device buffers are recycled by size class; a class holds as many as the busiest moment so far needed.  How many operators a pipelined job has alive at its
 * busiest is a matter of thread timing, so a job that got by with three buffers of a class in its first items may ask for a fourth later -- a hipMalloc of GBs
 * beside running kernels, 0.5 ms on one box and 120 ms on another.  This call tops every class of 64 MB .. 16 GB whose busiest moment used ALL its buffers up to
 * `spare` more than that (idle ones; classes that kept one unused are left alone).  The library does it by itself when the last operator of a device is destroyed
 * (HELM_POOL_SPARE buffers, default 2; HELM_POOL_SPARE_AUTO=0: only on this call) -- a caller that times the region ending with that destroy calls it itself, before.
 * Counterpart of the reference's pool keeping its workers alive between products (distributors.py:80-96). */
int helm_pool_spares(int device, int spare);
/* Free the scratch memory the library keeps between calls (the direct path's shared workspace, tens of GB at
 * 1024^2 x 256 right-hand sides).  HELM_ERR_STATE while a solve is using it. */
int helm_trim(void);
/* Only the pinned host buffers the library holds idle (results of helm_solve / helm_solve_coo and helm_host_alloc memory that was
 * handed back): the idle pool is capped by bytes (HELM_HOSTPOOL_GB, default a quarter of the host's memory, at most 96 GB); helm_trim
 * calls this too.  Counterpart of dropping the reference's result arrays (discretization.py:101-103 returns fresh numpy arrays). */
int helm_host_trim(void);
/* (diagnostic) number of scratch slots of `device` that hold a buffer of at least `bytes` bytes -- the direct path's shared workspace is
 * kept PER DEVICE, HELM_WS_SLOTS (default 3) slots each, so that every GPU of an in-process multi-GPU job (distributors.py:80-96: one
 * address space per worker in the reference) finds its own; device < 0: the number of slots per device. */
int helm_debug_ws_slots(int device, long long bytes);
/* (diagnostic) number of elimination-tree plans the per-device cache holds for `device` (HELM_ND_PLANS, default 6 per device) */
int helm_debug_plan_cache(int device);
/* (diagnostic, no GPU needed) the scratch-slot table exercised with `ndev` logical devices and host memory: every device books `concurrent`
 * slots of `bytes`, then that many leases are taken on every device at once.  0: every lease was a booked slot of its own device and none
 * allocated; < 0: which check failed. */
int helm_debug_ws_selftest(int ndev, int concurrent, long long bytes);
/* (diagnostic) allocator calls of the library that reached the driver and took more than a millisecond since the last reset (what
 * HELM_ALLOC_TRACE=1 prints): a job that booked its memory with helm_reserve must issue none. */
int helm_debug_alloc_stats(int reset, long long *slow_calls, double *worst_ms);

/* Resolve every kernel of the library on `device` now: the HIP runtime looks a kernel up in its code object and builds its dispatch record the first
 * time it is launched on a device (1.8 ms of host time for the nine kernels of one tree level in a cold process), and a job meets some kernels only at
 * some frequencies -- pivoted leaves, the pivoted-LU treatment of ill-conditioned fronts, refinement widths.  Nothing is launched.  Called by the
 * library itself when the first operator of a device is created (HELM_WARM=0: not); returns the number of kernels the library holds (> 0) or
 * HELM_ERR_DEVICE.  Counterpart of starting the reference's worker pool before the first product (distributors.py:80-96). */
int helm_warm(int device);
/* (diagnostic) what the library made the HIP runtime create since the last reset: device / pinned allocations that reached the driver (count, bytes,
 * host milliseconds), events, streams, kernels launched for the first time in the process (count, host milliseconds of those launch calls); and, not
 * reset, the kernels registered / resolved by helm_warm and its milliseconds.  A job whose memory was booked shows zeros across its timed region. */
typedef struct helm_runtime_stats {
    long long dev_allocs;  double dev_alloc_bytes;  double dev_alloc_ms;
    long long host_allocs; double host_alloc_bytes; double host_alloc_ms;
    long long events_created, streams_created;
    long long first_launches; double first_launch_ms;
    long long kernels_registered, kernels_resolved; double warm_ms;
    long long dev_frees; double dev_free_ms;          /* hipFree calls (each waits for every stream of the device) */
    long long sync_calls; double sync_ms;             /* host-side waits of the library (stream / event / device synchronisations, blocking copies) */
    long long slow_syncs; double worst_sync_ms;       /* those of at least 10 ms (HELM_SYNC_TRACE=<ms> names their call sites on stderr) */
} helm_runtime_stats;
int helm_debug_runtime_stats(int reset, helm_runtime_stats *out);
/* (diagnostic) start != 0: a thread of the library starts reading the clock in a loop; start == 0: it stops, and the longest interval between two of its readings
 * (and how many exceeded 5 ms) come back.  A stall every thread of the process sits through shows here; a stall of the GPU or of the HIP runtime does not. */
int helm_debug_stall_watch(int start, double *worst_gap_ms, long long *gaps_over_5ms);

/* --- tuning ------------------------------------------------------------------------------
 * The options of the library that are real options (round 5: the seventy-odd HELM_* environment switches of rounds 1-4 were the tuning
 * interface; the measured-and-rejected ones are gone, HISTORY.md has what they measured).  Every field can still be given through the
 * environment variable named beside it -- looked at when an API call that starts work is entered (create, assemble, prefactor, solve, apply,
 * get_tuning), by the calling thread and nowhere below it, so a test may flip one BETWEEN two calls, never during one -- and helm_set_tuning
 * replaces the lot for the process (NULL: back to defaults + environment).  Values are clamped to the ranges the code can run with whichever
 * way they arrive (nd_leaf >= 2, nd_plans >= 1, nd_ws_gb > 0, nd_stable_safety >= 1, ws_slots 1..4, pf_prio -1/0/1, mg3_omega in (0, 2]).  Process-wide, like the reference's module-level defaults
 * (distributors.py:28-34); not per handle.  Diagnostics that change no result and no speed stay environment-only: HELM_ND_TRACE,
 * HELM_ND_DEBUG, HELM_GEMM_LOG, HELM_MG3_TRACE, HELM_ALLOC_TRACE, and the test hooks behind HELM_TESTING=1 (HELM_ND_POISON,
 * HELM_ND_SUPPORT_CHECK, HELM_LEAF_DBG, HELM_ND_INJECT_*). */
typedef struct helm_tuning {
    /* 2-D sparse direct path */
    int    nd_leaf;            /* HELM_ND_LEAF           8     regions with both sides <= this are eliminated whole (leaf fronts) */
    double nd_ws_gb;           /* HELM_ND_WS_GB          32    cap on the per-batch scratch of a direct solve; the batch is halved until it fits */
    int    nd_sparse_rhs;      /* HELM_ND_SPARSE_RHS     1     forward pass skips fronts that see nothing but zeros (point sources) */
    int    nd_stable;          /* HELM_ND_STABLE         1     ill-conditioned fronts are re-eliminated with a pivoted LU */
    double nd_stable_thr;      /* HELM_ND_STABLE_THR     0     fixed condition-estimate threshold; 0: rtol / (nd_stable_safety * eps) */
    double nd_stable_safety;   /* HELM_ND_STABLE_SAFETY  8 */
    int    nd_fused_leaf;      /* HELM_ND_FUSEDLEAF      1     leaf level of the factorisation in one kernel */
    int    nd_fused_leaf_min;  /* HELM_ND_FUSEDLEAF_MIN  2048  fewest leaves of a level for which it pays */
    int    nd_gjstep;          /* HELM_ND_GJSTEP         1     one launch per block step of the blocked Gauss-Jordan inversion */
    int    nd_gjstep_min;      /* HELM_ND_GJSTEP_MIN     128   smallest front inverted that way (r5: 512 -> 128, headline +1.3 %) */
    int    nd_overlap;         /* HELM_ND_OVERLAP_NM     1     first forward pass beside the factorisation (2: also while profiling) */
    int    nd_xcd_map;         /* HELM_ND_XCDMAP         2     workgroup ids regrouped so that a front's tiles share an XCD (0 off, 1 column tiles only) */
    int    nd_plans;           /* HELM_ND_PLANS          6     elimination-tree plans cached per device */
    int    nd_direct_out;      /* HELM_ND_DIRECT_OUT     1     back substitution writes the caller's wavefield array itself (node-major calls) */
    int    nd_many;            /* HELM_ND_MANY           1     helm_prefactor_many factors its operators in the same launches (0: one after the other) */
    int    nd_leaf_idle;       /* HELM_ND_LEAF_IDLE      1     sparse right-hand sides: idle leaf blocks and small separator fronts of the back substitution go to kernels that issue all their loads at once; the forward pass deals its separator levels from lists of active (front, block) pairs */
    /* dispatch / memory */
    int    auto_direct;        /* HELM_AUTO_DIRECT       1     HELM_AUTO takes the direct path in 2-D */
    int    auto_mg3;           /* HELM_AUTO_MG3          1     HELM_AUTO takes the multigrid-preconditioned path in 3-D */
    int    prof_ext;           /* HELM_PROF_EXT          1     profiled launches carry their own events (0: event records around them) */
    int    ws_slots;           /* HELM_WS_SLOTS          3     shared scratch slots per device */
    int    pf_prio;            /* HELM_PF_PRIO           1     stream priority of helm_prefactor (1 high, 0 normal, -1 low) */
    /* 3-D multigrid */
    int    mg3_keep;           /* HELM_MG3_KEEP          1     layer-preserving hierarchy (2: fail instead of retreating to the standard cycle) */
    int    mg3_keep_levels;    /* HELM_MG3_KEEP_LEVELS   -1    coarsenings above the directly solved level; -1: chosen by the depth model */
    int    mg3_galerkin;       /* HELM_MG3_GALERKIN      1     Galerkin operator on the directly solved level */
    int    mg3_depth_model;    /* HELM_MG3_DEPTH_MODEL   1     trade set-up seconds against booked iteration counts */
    int    mg3_bt_f32;         /* HELM_MG3_BT_F32        1     single-precision plane inverses of the block-tridiagonal coarse solve */
    int    mg3_otf;            /* HELM_MG3_OTF           1     27-point apply rebuilds its coefficients from c, rho and the PML profiles (1: from 4 right-hand sides per workgroup up, 2: always) */
    int    mg3_f32;            /* HELM_MG3_F32           1     the layer-preserving cycle keeps the work vectors of its finest level in complex64 (input, result and every coarser level stay complex128) */
    double mg3_omega;          /* HELM_MG3_OMEGA         0.9   Jacobi damping of the smoother */
    /* host-side waits */
    double sync_spin_ms;       /* HELM_SYNC_SPIN_MS      0     a wait of the library polls for this long before it blocks on the runtime's interrupt (saves the 20-50 us wake-up of each
                                                               wait; costs a CPU per waiting thread -- leave it off under a container CPU quota, where a spinning thread spends the budget the
                                                               launching threads need) */
    int    sync_sleep_us;      /* HELM_SYNC_SLEEP_US     0     > 0: a wait of the library polls with a sleep of this many microseconds between two looks instead of the runtime's wait, which keeps a
                                                               CPU busy while it lasts (2.5 CPUs per process in the pipelined bench job): for several processes under a container CPU quota */
} helm_tuning;
int helm_get_tuning(helm_tuning *out);          /* the values in force now (environment applied) */
int helm_set_tuning(const helm_tuning *t);      /* NULL: defaults + environment again */

/* --- diagnostics of the direct solver ---------------------------------------------------- */
/* Elimination-tree plan of an (nz, nx) grid (host only, no GPU needed).  out == NULL: returns the number of fronts;
 * else writes 12 ints per front in processing order {z0, z1, x0, x1, cut, pos, s, m, kid0, kid1, smax, mmax}. */
int helm_direct_plan(int nz, int nx, int leaf, int *out, int cap);
/* cells (z*nx + x) of front `node` in local order [separator | ring]; returns s + m, or < 0 if the closed-form
 * index maps disagree with each other */
int helm_direct_plan_front(int nz, int nx, int leaf, int node, long long *cells, int cap);
/* dense kernels on host data (row-major complex128, batch contiguous): C = beta C + alpha A B; in-place inverse */
int helm_debug_zgemm(int device, int M, int N, int K, const double *alpha, const double *A, const double *B,
                     const double *beta, double *C, int batch);
int helm_debug_inverse(int device, int n, double *A, int batch);
/* average milliseconds per launch of one strided-batched GEMM shape (random operands, `reps` timed launches); variant < 16: the tile the
 * library would choose, 16 (t + 1): tile configuration t (0 64x64 ... 7 16x64) forced */
int helm_debug_zgemm_bench(int device, int M, int N, int K, int batch, int variant, int reps, double *ms_out);
int helm_debug_inverse_bench(int device, int n, const double *A, int reps, int recurse_n, double *ms_out);

/* --- diagnostics of the 3-D multigrid hierarchy (host only, no GPU needed; zephyr_amd/csrc/mg3d.hip) ------------------ */
/* One axis of n nodes (spacing h, npml absorbing-layer nodes at each end, damping amplitude cpml) coarsened `level` times by the
 * layer-preserving rule (all layer nodes kept, the interior halved).  Returns the node count nc of that level (< 0: bad arguments)
 * and fills whichever outputs are not NULL: x[nc] coordinates, lay[nc] layer flags, lap[3 nc] complex factors L(-1), L(0), L(+1)
 * of (1/xi) d/dx (1/xi) d/dx on the non-uniform axis for omega = om_re + i om_im; the transfer to the next level: pc / pw [2 nc]
 * coarse nodes and weights every node interpolates from, *n_next its size, rf[n_next] / rw[3 n_next] centre node and weights of
 * the normalised transposed interpolation. */
int helm_mg3_axis(int n, int npml, double h, double cpml, double om_re, double om_im, int level, double *x, int *lay, double *lap,
                  int *pc, double *pw, int *n_next, int *rf, double *rw);

#ifdef __cplusplus
}
#endif
#endif /* HELM_H */
