// Nested-dissection direct solver: what its translation units share (nd_plan / nd_gemm / nd_gj / nd_leaf / nd_factor / nd_passes / nd_resid .hip).
// direct.hpp is the solver's interface to the rest of the library (capi.hip, mg3d.hip); this header is internal to the solver.
#pragma once
#include "helm_internal.hpp"
#include "direct.hpp"
#include <hip/hip_ext.h>
#include <algorithm>

// ---- strided-batched complex GEMM: C = beta C + alpha A B, row-major ----------------------------------------------
// IDX variant (solve phase): rows of B, of the C that is read (beta != 0) and of the C that is written may be taken
// through the plan's row table instead of a dense front buffer, i.e. straight from / to the node-major right-hand sides
// Xt[cell][rhs]:   row r of batch item z  ->  X + tab[z * tab_stride + off + r].x * ldx   (negative: a zero row / not stored).
// This removes the gather / scatter passes (and their HBM round trips) from the lower tree levels.
// arguments of the Gauss-Jordan sweep that rides along with a blocked-inversion update (k_zgemm3_la)
struct GjPivotArgs { const cplx *T0; int ld; long long stride; int n, k0, nb; const cplx *Wc0, *Wr0; long long wstride; cplx *Pb0; long long pstride; int batch; };
struct GemmRows {
    const int4 *tabB = nullptr, *tabCi = nullptr, *tabCo = nullptr;
    int offB = 0, offCi = 0, offCo = 0, tab_stride = 0;
    const cplx *Bx = nullptr, *Cix = nullptr; cplx *Cox = nullptr;
    int ldx = 0;
    int z0 = 0;           // batch index of blockIdx.z == 0 (launches are chunked along z)
    int fwd3 = 0;         // forward-gather mode (k_zgemm3<.., 2, ..>): Bx = right-hand sides, Cix = front-vector arena, Cox = where y_S goes
    int zr0 = 0, zr1 = 0, zc0 = 0, zc1 = 0;   // rows [zr0, zr1) and columns [zc0, zc1) of C are taken as zero on input (beta masked): blocked Gauss-Jordan
    int sk0 = 0, sk1 = 0;                     // the diagonal block [sk0, sk1)^2 of C is neither read nor written (the next pivot block, owned by k_gj_pivot)
    int ksplit = 0, kc = 0; long long pstride = 0;   // split over the inner dimension (dense operands only): z = matrix * ksplit + chunk; chunk c multiplies columns
                                              // [c kc, (c + 1) kc) of A into its own partial product at C0 + c pstride + matrix sc (k_splitk_reduce adds them up)
    int dense = 0;                            // only the masks above are in use: launch the plain (un-indexed) kernel
    const cplx *Bx2 = nullptr; int k2 = 0;    // rows k < k2 of an indexed B come from Bx2 instead of Bx (a leaf's y_S is still in the right-hand sides)
    int tm64 = 0;                             // one 64-row tile per matrix (M <= 64): C may then overwrite B (every workgroup has read all of its B columns
                                              // before it stores, and no other workgroup reads them)
    const GjPivotArgs *la = nullptr;          // (host pointer) fuse this pivot sweep into the launch: 64 x 32 tiles, one extra z-slice
    // Schur-complement mode (k_zgemm3<.., 4, ..>): C = (children's Schur complement entries that land on (r, c), gathered through the row
    // table) + alpha A B -- the ring x ring block of a front is never written by the build pass and never read back here
    int schur4 = 0;
    const NdDev *nodes = nullptr; int first = 0;
    const cplx *arenaS = nullptr;
    int nf = 1;                 // (schur4) batch index = front * nf + frequency: tables and node records go by the front, the children's blocks lie interleaved (direct.hpp)
    // forward pass on sparse right-hand sides (sources of a survey touch a handful of cells): act[front * nct + column / 64] != 0 when that front's
    // outgoing rows were computed for that block of 64 columns -- a front whose own right-hand-side rows and whose children's rows are all zero
    // there has nothing to add, writes zeros for its y_S rows and leaves its ring rows unwritten (its parent reads the flag, not the rows)
    int *act = nullptr; int nct = 0;
    // (fwd3, r5) the (front, block of 64 columns) pairs of this level that have anything to do, found by k_fwd_flags before the launch: workgroup x takes
    // pair list[x] (front = pair / nct, block = pair % nct), workgroups from *lcount on leave at once; 64-column tiles only
    const int *list = nullptr, *lcount = nullptr;
    // back substitution of the leaves, same flags read-only: where a leaf's flag is 0 its first k2 rows of B (its right-hand-side rows y_S) are all
    // zero in that block of columns, and the product starts at row k2 -- x_S = G x_B, 32 of the 81 columns of [F11^-1 | G]
    const int *act_ro = nullptr;
    int idle_done = 0;          // (with act_ro) the blocks whose flag is 0 have been taken by k_leaf_bwd_idle: the tile kernel leaves them alone
    int hint = 0;               // (IDX 1 with act, leaf forward elimination) the flags were set from a declared support of the right-hand sides: a front without one is left
                                // before it reads a byte
    int ntc = 0;                // store C with nontemporal stores (large HBM-bound launches whose output is not read again soon)
    int xcd_map = 0;            // regroup the workgroup ids so that the column tiles of a front share an XCD (zgemm3_body)
    int child_rows = 0;         // (fwd3, host-side bookkeeping) ring rows of a front's two children: what the gather has to read besides q_S
    // Direct output (IDX 1 with tabCo, back substitution of a node-major call): the caller's wavefield array takes u = conj(oscale x) from the launch that
    // computes x -- cj_out: the rows stored through tabCo are written as u (leaves: nobody reads their x on the GPU again but the residual check, which takes
    // it from the caller's array); Cox2: a second, transformed copy of every stored row beside the plain one in Cox (separator rows: the children below still need x)
    int cj_out = 0; cplx oscale = {1.0, 0.0}; cplx *Cox2 = nullptr;
};
#define GB_KIDX 512       // largest K with indexed B rows
// Addressing modes of the tile kernel (template parameter IDX):
//   0  dense operands;
//   1  rows of B / of the C that is read / of the C that is written through the row table (GemmRows), i.e. straight from / to the node-major
//      right-hand sides Xt[cell][rhs];
//   2  "forward gather": a row of B is the SUM the forward pass needs -- the right-hand side of a separator cell plus the children's outgoing rows
//      that land on it (table entries x / y, z) -- and the C that is read is the sum of the children's rows of a ring row; the first row-tile also
//      stores the gathered separator rows (y_S) where the back substitution expects them.  Same additions in the same order as k_nd_fwd_rows + a
//      dense product;
//   4  "Schur gather": C = (children's Schur-complement entries that land on (r, c), gathered through the row table) + alpha A B.
// (Rounds 1-3 ran these products on the vector ALUs -- k_zgemm, k_zgemm2: 4 x 4 complex register blocks, 27-45 TFLOP/s; HISTORY.md -- and round 4
// moved them to the matrix cores; the vector kernels were deleted in round 5.)

#define PNB 32              // pivot block of the blocked Gauss-Jordan inversion
#define LEAF_BW 9            // fused leaf kernel: largest half-bandwidth (w + 1, w <= 8)
#define LEAF_MP 36           // largest ring of a leaf (2 (8 + 2) + 2 x 8)
#define LUS_NMAX 128         // largest front the one-workgroup pivoted LU treats
#define ND_STABLE_CAP 32     // ill-conditioned fronts treated per group at most

// Per-launch timing without extra packets: when gemm() has armed a pair of events, the dispatch itself carries them
// (hipExtLaunchKernelGGL: start / stop timestamps of that kernel), instead of two hipEventRecord markers around it.
extern thread_local hipEvent_t tl_ev0, tl_ev1;
// While a set of nf frequencies is being factored the batches are nf times as long, and everything that CHOOSES by batch size -- tile shape, split of the inner
// dimension, which small-inverse kernel -- must choose as for one frequency: the factors have to come out bit for bit those of the one-at-a-time path.
extern thread_local int tl_nf_div;
struct NfDivScope { int prev; explicit NfDivScope(int nf) : prev(tl_nf_div) { tl_nf_div = nf < 1 ? 1 : nf; } ~NfDivScope() { tl_nf_div = prev; } };
#define ZG_LAUNCH(KERNEL, GRID, ...) do { HelmFirstLaunch fl_(HelmKernelReg<(KERNEL)>::slot); \
                                          if (tl_ev0) hipExtLaunchKernelGGL(KERNEL, GRID, dim3(256), 0, st, tl_ev0, tl_ev1, 0, __VA_ARGS__); \
                                          else hipLaunchKernelGGL(KERNEL, GRID, dim3(256), 0, st, __VA_ARGS__); } while (0)

// What one GEMM launch has to move at the very least -- every operand once: A (M x K), B (K x N), C written (and read when beta != 0) -- and the
// time the part's two roofs allow it: max(flops / 78.6 TFLOP/s, bytes / 8 TB/s).  The thin fronts low in the tree are HBM-bound products
// (a level-13 front multiplies a 48 x 8 block into 256 right-hand sides: 1.6 flop per byte), the big ones fp64-bound; the bench adds both up.
// r4: the two gather modes read more than "every operand once" of a plain product, and that is necessary traffic, not waste --
//   forward gather (fwd3): a B row is q_S plus the children's rows that land on it, the C that is read is the children's rows of a ring row, and the
//     gathered separator rows are written back as y_S: every child ring row (child_rows of them per front) is read once, K more rows are written;
//   Schur gather (schur4): nothing of C is read (beta = 0) but the children's Schur-complement entries that land on the ring x ring block are:
//     about half of its M N entries receive one (both cells on the same child's ring), a few receive two.
inline double gemm_operand_bytes(int M, int Nn, int K, cplx beta, const GemmRows *rows = nullptr) {
    const bool rd = !(beta.x == 0.0 && beta.y == 0.0);
    if (rows && rows->fwd3) return 16.0 * ((double)M * K + (double)K * Nn + (double)rows->child_rows * Nn + (double)M * Nn + (double)K * Nn);
    if (rows && rows->schur4) return 16.0 * ((double)M * K + (double)K * Nn + 1.5 * (double)M * Nn);
    return 16.0 * ((double)M * K + (double)K * Nn + (double)M * Nn * (rd ? 2.0 : 1.0));
}
inline double gemm_sol_ms(double flops, double bytes) { return 1e3 * std::max(flops / 78.6e12, bytes / 8.0e12); }

struct ExtArm {            // arms the per-launch event pair for the launchers below, books it when the launch is out
    helm_op *op; bool on; double fl, by; long long shape[5];
    ExtArm(helm_op *o, bool e, double f, double b, int m_, int n_, int k_, int nb_, int mode_) : op(o), on(false), fl(f), by(b), shape{m_, n_, k_, nb_, mode_} {
        if (!e) return;
        if (op->ev_used + 2 > op->ev_pool.size() && op->ev_pool.size() < 16384) (void)helm_events_grow(op, 64);
        if (op->ev_used + 2 <= op->ev_pool.size()) { tl_ev0 = op->ev_pool[op->ev_used]; tl_ev1 = op->ev_pool[op->ev_used + 1]; on = true; }
    }
    ~ExtArm() {
        if (!on) return;
        op->ev_pending_gemm.push_back(std::make_pair((int)op->ev_used, fl));
        op->ev_pending_gemm_n.push_back(1);
        op->ev_pending_gemm_bytes.push_back(by);
        op->ev_pending_gemm_sol.push_back(gemm_sol_ms(fl, by));
        for (int q = 0; q < 5; ++q) op->ev_pending_gemm_shape.push_back(shape[q]);
        op->ev_used += 2;
        tl_ev0 = tl_ev1 = nullptr;
    }
};

// Consecutive GEMM launches with no other kernel between them (the recursion of the block inversion issues them in runs of
// two and four) share one pair of timing events: the per-launch average stays exact, two thirds of the event traffic go away.
struct GemmRun {
    helm_op *op;
    explicit GemmRun(helm_op *o) : op(o) { if (op && op->gemm_run_depth++ == 0) { op->gemm_run_pair = -1; op->gemm_run_flops = 0; op->gemm_run_bytes = 0; op->gemm_run_sol = 0; op->gemm_run_launches = 0; } }
    ~GemmRun() {
        if (!op || --op->gemm_run_depth != 0) return;
        if (op->gemm_run_pair >= 0 && op->gemm_run_launches > 0) {
            hipEventRecord(op->ev_pool[op->gemm_run_pair + 1], op->stream);
            op->ev_pending_gemm.push_back(std::make_pair(op->gemm_run_pair, op->gemm_run_flops));
            op->ev_pending_gemm_n.push_back(op->gemm_run_launches);
            op->ev_pending_gemm_bytes.push_back(op->gemm_run_bytes);
            op->ev_pending_gemm_sol.push_back(op->gemm_run_sol);
        }
        op->gemm_run_pair = -1;
    }
};

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

#define GJ_MAX 64
// Gauss-Jordan with row pivoting on an n x n matrix held in LDS (all threads of the workgroup call it; a is valid on return
// after the trailing barrier)
template <int NMAX>
__device__ __forceinline__ void gj_lds(cplx (*a)[NMAX + 1], cplx *fcol, int *piv, int n, int tid, int nthreads) {
    for (int k = 0; k < n; ++k) {
        // wave 0 (n <= 64 lanes, lock-step): pivot search in column k, row exchange, scaling of the pivot row, and the
        // column that the elimination needs -- every read of the old values is issued before the writes
        if (tid < 64) {
            double val = (tid >= k && tid < n) ? cabs2(a[tid][k]) : -1.0;
            int idx = tid;
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_down(val, off);
                const int oi = __shfl_down(idx, off);
                if (ov > val) { val = ov; idx = oi; }
            }
            const int p = __shfl(idx, 0);
            if (tid == 0) piv[k] = p;
            if (tid < n) {
                const cplx f = (tid == p) ? a[k][k] : a[tid][k];
                const cplx rk = a[p][tid], rp = a[k][tid];
                const cplx d = crecip(a[p][k]);
                a[p][tid] = rp;
                a[k][tid] = (tid == k) ? d : cmul(rk, d);
                fcol[tid] = f;
            }
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += nthreads) {
            const int i = e / n, j = e % n;
            if (i == k) continue;
            cplx base = (j == k) ? cmake(0.0, 0.0) : a[i][j];
            a[i][j] = csub(base, cmul(fcol[i], a[k][j]));
        }
        __syncthreads();
    }
    if (tid < 64) {      // undo the row exchanges as column exchanges, last first (lock-step within the wave)
        for (int k = n - 1; k >= 0; --k) {
            const int p = piv[k];
            if (p != k && tid < n) { cplx t = a[tid][k]; a[tid][k] = a[tid][p]; a[tid][p] = t; }
        }
    }
    __syncthreads();
}

inline int check_kernels(helm_op *op, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char b[256]; snprintf(b, sizeof(b), "direct solver: %s failed: %s", what, hipGetErrorString(e)); helm_set_error(op, b); return HELM_ERR_DEVICE; }
    return HELM_OK;
}

// HELM_ND_TRACE=1: per-group device time of the factorisation / forward / backward sweeps on stderr (diagnostics only)
struct GroupTrace {
    bool on; hipStream_t st; std::vector<hipEvent_t> ev; const char *what;
    GroupTrace(hipStream_t s, const char *w) : st(s), what(w) { const int t = getenv("HELM_ND_TRACE") ? atoi(getenv("HELM_ND_TRACE")) : 0; on = t != 0; mark(); }
    void mark() { if (!on) return; hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, st); ev.push_back(e); }
    void report(const NdPlan &P, bool reverse) {
        if (!on) return;
        hipStreamSynchronize(st);
        double tot = 0;
        for (size_t i = 0; i + 1 < ev.size(); ++i) {
            float ms = 0.f; hipEventElapsedTime(&ms, ev[i], ev[i + 1]); tot += ms;
            const size_t gi = reverse ? P.groups.size() - 1 - i : i;
            if (gi < P.groups.size()) {
                const NdGroup &g = P.groups[gi];
                fprintf(stderr, "[nd trace] %-8s level %2d %s cnt %6d s %5d m %5d : %8.3f ms\n", what, g.level, g.leaf ? "leaf" : "sep ", g.cnt, g.smax, g.mmax, ms);
            } else fprintf(stderr, "[nd trace] %-8s extra : %8.3f ms\n", what, ms);
        }
        fprintf(stderr, "[nd trace] %-8s total %8.3f ms\n", what, tot);
        for (hipEvent_t e : ev) hipEventDestroy(e);
        ev.clear();
    }
};

// ---- nd_gemm.hip ----
// strided-batched C = beta C + alpha A B on the matrix cores; op may be null (diagnostic entry points): default stream, no profiling
int gemm(helm_op *op, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
         cplx beta, cplx *C, int ldc, long long sc, int batch, const GemmRows *rows = nullptr);
extern int g_gemm_tile;         // >= 0: forces the tile configuration (helm_debug_zgemm_bench)
bool gemm_sep_bwd_small(helm_op *op, int smax, int mmax, int nrhs, const cplx *Finv, const cplx *F12, int lda, long long sa, int batch, const GemmRows &rows);
// ---- nd_gj.hip ----
// in-place inverse of `batch` n x n blocks (row-major, leading dimension ld, batch stride `stride`); W: workspace with batch stride ws, at least n*n elements per matrix
void invert(helm_op *op, cplx *M, int ld, long long stride, int n, int batch, cplx *W, long long ws, int align = 1, int base = 0);
extern int g_recurse_min;       // (helm_debug_inverse_bench overrides the block-recursion threshold)
// the rank-32 update of a blocked Gauss-Jordan step with the pivot sweep of the next block riding in one extra z-slice (gemm() with rows->la)
void launch_zgemm3_la(hipStream_t st, bool latency_tile, int nb, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                      cplx beta, cplx *C, int ldc, long long sc, const GemmRows &R, const GjPivotArgs &pv);
// ---- nd_leaf.hip ----
// the leaf level of the factorisation in one kernel (+ the pivoted re-elimination of the leaves it flags)
void launch_leaf_factor(hipStream_t st, int smax, int nb, const NdDev *d_nodes, int first, cplx *arenaF, cplx *fac, cplx *g21b, const cplx *planes, int nz, int nx, int *flags, int dbg,
                        int nf = 1, int kf = 0);
// ---- nd_factor.hip ----
struct FacSet { int nf = 1; NdFactor *f[ND_NF_MAX] = {nullptr, nullptr, nullptr, nullptr}; const cplx *planes[ND_NF_MAX] = {nullptr, nullptr, nullptr, nullptr}; double rtol[ND_NF_MAX] = {0, 0, 0, 0}; };
int factor_prologue(helm_op *op, int block, NdFactor *f, const cplx *planes_in, const cplx **planes);
int factor_group(helm_op *op, NdFactor *f, size_t gi, cplx *arenaF, cplx *work, const cplx *planes);     // factorisation of one group (tree level x kind) on op->stream
bool merged_group(const NdPlan &P, const NdGroup &g);      // fronts that keep G = -F11^-1 F12 where F12 was (one-product back substitution): the leaves
void launch_lu_solve(hipStream_t st, const cplx *LU, int ld, int n, const int *piv, cplx *B, int ldb, int ncols);    // B <- (L U)^-1 P B, n <= LUS_NMAX
// ---- nd_resid.hip ----
void launch_transpose(hipStream_t st, const cplx *in, long long rows, long long cols, cplx *out, int swap, int conj);
