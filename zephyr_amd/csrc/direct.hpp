// Nested-dissection direct solver: plan structures and the closed-form front index maps (host + device).
#pragma once
#include "helm_internal.hpp"
#include <memory>

struct NdDev {                // one front = one node of the elimination tree
    int z0, z1, x0, x1;       // region of the subtree (half-open)
    int cut, pos;             // -1: leaf, all cells of the region are eliminated; 0: separator row z = pos; 1: separator column x = pos
    int dof;                  // unknowns per cell: 1, or 2 for the coupled two-field Eurus system (u and v of a cell are adjacent)
    int s, m;                 // separator / ring unknowns (cells x dof)
    int ntop, nbot, nleft, nright, xlo;   // ring segments clipped to the grid: row z0-1, row z1 (x from xlo), column x0-1, column x1 (z from z0)
    int kid[2];               // children (indices in processing order), -1: none
    int smax, mmax;           // padded sizes of the node's group
    long long foff;           // [F21 | F22] (mmax x (smax + mmax)) of the front in the factorisation arena (elements)
    long long finv_off, f12_off;   // F11 (smax x smax, inverted in place) and F12 (smax x mmax) of the front in the factor storage
    long long voff;           // front vector offset in the solve arena (rows; x nrhs elements)
    long long roff;           // first row of this front in the row table
};

struct NdGroup {              // nodes of one tree level and kind: one strided batch
    int first = 0, cnt = 0, level = 0;
    bool leaf = false;
    int smax = 0, mmax = 0;
    long long foff = 0, voff = 0;
    long long roff = 0;                        // first row of the group in the row table (cnt * (smax + mmax) rows)
    long long finv = 0, g21 = 0, f12 = 0;      // offsets of F11^-1 [cnt][smax][smax], G21 [cnt][mmax][smax], F12 [cnt][smax][mmax]
};

struct NdPlan {
    int nz = 0, nx = 0, leaf = 8, nlevels = 0, dof = 1;
    std::vector<NdDev> nodes;                  // processing order: deepest level first
    std::vector<NdGroup> groups;
    long long fac_elems = 0, fregion = 0, vregion = 0, work_elems = 0, total_rows = 0;
};

// Row table of the solve phase, one entry per padded front row:
//   x: grid cell of the row (separator or ring cell), -1 for padding
//   y, z: arena rows (front-vector rows, region offset included) of the children's outgoing rows that land here, -1: none
//   w: 1 for separator rows
// Built once per plan by a kernel from the closed-form index maps.
struct NdPlanDev {            // plan + its device-resident tables, shared by every handle with the same grid on a device
    NdPlan plan;
    int device = 0;
    NdDev *d_nodes = nullptr;
    int4 *d_tab = nullptr;
    int *d_cellnode = nullptr;       // (dof 1) per grid cell: the leaf front that eliminates it, -1 for separator cells
    ~NdPlanDev();
};

// A front whose pivot block F11 is too ill-conditioned for its explicit inverse (a near-resonant subdomain: the subtree's region with its
// ring clamped has an eigenvalue close to zero at this frequency; a handful per frequency at most).  The batched path multiplies with
// F11^-1 in three places -- Schur complement, forward and backward pass -- and the rounding of those three is not consistent, which costs
// cond(F11) eps in the residual.  Such a front is eliminated again with ONE pivoted LU used in all three places (what a CPU multifrontal
// code does for every front): the factorisation error then cancels between them.
struct NdStable {
    int node = -1;               // front (plan order)
    size_t group = 0;
    int smax = 0, mmax = 0;
    cplx *lu = nullptr;          // [smax][smax + mmax]: L\U of F11 in the first smax columns, the original F12 beside it
    cplx *f21 = nullptr;         // [mmax][smax]: the original F21
    int *piv = nullptr;          // [smax] row exchanges
    cplx *vs = nullptr; size_t vs_elems = 0;     // back-substitution scratch [smax + mmax][nrhs], grown on demand
};

// Several frequencies of one grid factored by the same launches (helm_prefactor_many; nd_factor.hip): their factors live interleaved in ONE buffer, a front's
// slot of `slot` elements at offset `off` of the single layout at  nf * off + kf * slot  (batch index of the factorisation's launches = front * nf + frequency).
#define ND_NF_MAX 4
struct NdPlanesSet { const cplx *p[ND_NF_MAX]; };       // coefficient planes of the nf operators (a kernel argument)
struct NdFacShared {                                     // what the NdFactors of a set share: released when the last of them goes
    int device = 0;
    cplx *d_fac = nullptr; size_t fac_bytes = 0;
    double *d_est = nullptr; size_t est_elems = 0;
    ~NdFacShared();
};
struct NdFactor {
    std::shared_ptr<NdPlanDev> pd;
    int nf = 1, kf = 0;                                  // this factor is frequency kf of a set of nf (1, 0: on its own)
    std::shared_ptr<NdFacShared> shared;                 // (nf > 1) owner of d_fac / d_est
    cplx *d_fac = nullptr;
    int block = 0;
    double flops = 0;
    std::vector<NdStable> stable;
    double *d_est = nullptr; size_t est_elems = 0;      // per front of a group: max |F11| before, max |F11^-1| after the inversion; flag list
    int act_nct = 0;                                          // blocks of 64 columns the flags of the LAST forward pass cover (0: that pass computed every front)
    unsigned char *d_qmask = nullptr; size_t qmask_elems = 0;  // per grid cell: bit b set when the right-hand sides may be nonzero there in block b of 64 columns
    int *d_act = nullptr; size_t act_elems = 0;              // forward pass on sparse right-hand sides: per front and block of 64 columns, were its outgoing rows computed?
    int *d_leafflag = nullptr; size_t leafflag_elems = 0;     // per leaf of a group: 1 when the fused leaf kernel met a small pivot (the leaf is then re-done with pivoting)
    struct FlagSlot *flag_slot = nullptr; double flag_thr = 0; // pinned buffer + event the list of a group's ill-conditioned fronts is travelling through (flag_group -> stabilise_group), and the threshold it was made with
};

// where a group's array of the single layout (offset `off`, `slot` elements per front) lies for this factor, and the stride between its fronts
inline cplx *nd_fac_at(const NdFactor *f, long long off, long long slot) { return f->d_fac + (long long)f->nf * off + (long long)f->kf * slot; }
inline long long nd_fac_stride(const NdFactor *f, long long slot) { return (long long)f->nf * slot; }

// local index of unknown (cell (z, x), component comp) in the front: [0, s) separator, [s, s+m) ring; -1 when the cell
// is not in the front.  Unknown = cell index * dof + comp within each part.
__host__ __device__ inline int nd_local(const NdDev &n, int nz, int nx, int z, int x, int comp = 0) {
    (void)nz; (void)nx;
    const int d = n.dof;
    if (n.cut < 0) { if (z >= n.z0 && z < n.z1 && x >= n.x0 && x < n.x1) return ((z - n.z0) * (n.x1 - n.x0) + (x - n.x0)) * d + comp; }
    else if (n.cut == 0) { if (z == n.pos && x >= n.x0 && x < n.x1) return (x - n.x0) * d + comp; }
    else { if (x == n.pos && z >= n.z0 && z < n.z1) return (z - n.z0) * d + comp; }
    const int wrow = n.ntop ? n.ntop : n.nbot;
    if (n.ntop && z == n.z0 - 1 && x >= n.xlo && x < n.xlo + wrow) return n.s + (x - n.xlo) * d + comp;
    if (n.nbot && z == n.z1 && x >= n.xlo && x < n.xlo + wrow) return n.s + (n.ntop + (x - n.xlo)) * d + comp;
    if (n.nleft && x == n.x0 - 1 && z >= n.z0 && z < n.z1) return n.s + (n.ntop + n.nbot + (z - n.z0)) * d + comp;
    if (n.nright && x == n.x1 && z >= n.z0 && z < n.z1) return n.s + (n.ntop + n.nbot + n.nleft + (z - n.z0)) * d + comp;
    return -1;
}

// cell and component of local index a (0 <= a < s + m)
__host__ __device__ inline void nd_cell(const NdDev &n, int a, int &z, int &x, int &comp) {
    const int d = n.dof;
    if (a < n.s) {
        comp = a % d; a /= d;
        if (n.cut < 0) { const int w = n.x1 - n.x0; z = n.z0 + a / w; x = n.x0 + a % w; }
        else if (n.cut == 0) { z = n.pos; x = n.x0 + a; }
        else { z = n.z0 + a; x = n.pos; }
        return;
    }
    a -= n.s;
    comp = a % d; a /= d;
    if (a < n.ntop) { z = n.z0 - 1; x = n.xlo + a; return; }
    a -= n.ntop;
    if (a < n.nbot) { z = n.z1; x = n.xlo + a; return; }
    a -= n.nbot;
    if (a < n.nleft) { z = n.z0 + a; x = n.x0 - 1; return; }
    a -= n.nleft;
    z = n.z0 + a; x = n.x1;
}
__host__ __device__ inline void nd_cell(const NdDev &n, int a, int &z, int &x) { int comp; nd_cell(n, a, z, x, comp); }

// row / column of local index a in the padded front (separator block padded to smax)
__host__ __device__ inline int nd_pos(const NdDev &n, int a) { return a < n.s ? a : n.smax + (a - n.s); }

int nd_build_plan(NdPlan &P, int nz, int nx, int leaf, int dof = 1);
long long nd_factor_ws_elems(const NdPlan &P);
int nd_get_plan(helm_op *op, int leaf, int dof, std::shared_ptr<NdPlanDev> *out);    // cached per (device, grid, leaf, dof)
int nd_get_plan_dims(helm_op *op, int nz, int nx, int leaf, int dof, std::shared_ptr<NdPlanDev> *out);
int nd_factor(helm_op *op, int block, NdFactor *f, cplx *ws, const cplx *planes = nullptr);    // f->pd must be set; dof 2: block ignored, all four Eurus blocks
int nd_factor_enqueue(helm_op *op, int block, NdFactor *f, cplx *ws, const cplx *planes = nullptr);    // the same without the final synchronisation: launches only
// nf operators of one grid (block 0 each, one unknown per cell) factored by the same launches on op's stream: fs[k] (pd set, nothing else) receives frequency k's
// factors, ws: nf * nd_factor_ws_elems(plan) elements of scratch.  Every factor comes out bit for bit what nd_factor_enqueue would have made of it.
int nd_factor_enqueue_many(helm_op *op, int nf, helm_op *const *ops, NdFactor *const *fs, cplx *ws);
void nd_free(NdFactor *f);
long long nd_solve_ws_elems(const NdPlan &P, int nrhs);
int nd_solve(helm_op *op, NdFactor *f, const cplx *Xin, cplx *Xout, int nrhs, cplx *ws, int conj_out = 0);   // conj_out: Xout = conj(x)
int nd_axpy_one(helm_op *op, cplx *y, const cplx *x, long long n, int conj = 0);
// node-major pipeline (direct.hip): right-hand sides and solutions as [cell][rhs] between one transpose in and one out
// Direct output (round 5): the back substitution leaves u = conj(oscale x) in the caller's node-major wavefield array U[cell][nrhs] itself -- leaf cells
// there alone (nothing on the GPU needs their x again), separator cells there and in Xt -- so that the residual check reads U and stores nothing
// (4.3 of its 13 GB at 1024^2 x 256).  Afterwards Xt holds x only for the separator cells; nd_recover_x rebuilds the rest from U if a refinement pass needs it.
struct NdDirectOut { cplx *U = nullptr; cplx oscale = {1.0, 0.0}; };
int nd_solve_nm(helm_op *op, NdFactor *f, const cplx *Qt, cplx *Xt, int nrhs, cplx *arenaV, const NdDirectOut *dout = nullptr);
int nd_factor_solve_nm(helm_op *op, int block, NdFactor *f, cplx *ws_factor, const cplx *planes, const cplx *Qt, cplx *Xt, int nrhs, cplx *arenaV,
                       hipStream_t side, float *factor_ms, const NdDirectOut *dout = nullptr);
int nd_recover_x(helm_op *op, const cplx *U, cplx *Xt, long long elems, cplx oscale);       // Xt = conj(U) / oscale
int nd_prep_transpose_norm(helm_op *op, const cplx *rhs, long long rhs_ld, long long row_off, cplx premul, const cplx *sub, cplx *Qt, long long N, int nrhs,
                           double *part, int nblk_cap, int *nblk_out);     // nblk_cap: partials per right-hand side the buffer has room for
// per-cell mask of where the right-hand sides of the last nd_solve_nm / nd_factor_solve_nm on f can be nonzero (leaf cells: the leaf's flag of the
// sparse forward pass; separator cells: always), for nd_resid_nm; null when that pass computed every front
const unsigned char *nd_rhs_mask(helm_op *op, NdFactor *f);
struct NdResidExtra {      // optional by-products of the residual launch (node-major callers)
    int qnorm = 0;                       // also the partials of ||q||^2 (slot 1 of the partial sums)
    cplx *Uout = nullptr; int ldu = 0;   // Uout[cell][j] = conj(oscale * xin[cell][j])
    cplx oscale = {1.0, 0.0};
    const unsigned char *qmask = nullptr;   // nd_rhs_mask of the q passed in: cells and blocks of 64 columns whose bit is 0 are not read (they hold zeros)
    int xin_is_u = 0;                    // Xin is the caller's wavefield array u = conj(oscale x) (direct output): evaluated as oscale q - A conj(u), same relative residual
};
int nd_resid_nm(helm_op *op, const cplx *planes, const cplx *Xin, int ldin, cplx *Q, int ldq, const int *qmap, int ncol, int store, cplx *Rout,
                double *part, int nblk_cap, int *nblk_out, const NdResidExtra *ex = nullptr);      // r = q - A xin; store: r -> Rout (null: over q)
int nd_transpose(helm_op *op, const cplx *in, long long rows, long long cols, cplx *out);
int nd_scatter_add_cols(helm_op *op, cplx *Xt, int ldq, const int *d_cols, int k, const cplx *Dp, long long N);
int nd_pack_cols(helm_op *op, const cplx *Qt, int ldq, const int *d_cols, int k, cplx *Rp, long long N);
int nd_transpose_out(helm_op *op, const cplx *Xt, long long N, int nrhs, cplx *U, int conj);

// dense kernels for other translation units (row-major, single matrices)
int nd_dense_gemm(helm_op *op, int M, int N, int K, cplx alpha, const cplx *A, int lda, const cplx *B, int ldb, cplx beta, cplx *C, int ldc);
// `batch` products C_b = alpha A_b B_b + beta C_b with element strides sa / sb / sc between them (e.g. the K chunks of a split-K product)
int nd_dense_gemm_batched(helm_op *op, int M, int N, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                          cplx beta, cplx *C, int ldc, long long sc, int batch);
int nd_dense_inverse(helm_op *op, cplx *M, int n, cplx *W);     // in place; W: n*n elements of scratch
