// Sparse direct solver for the 2-D 9-point operators: geometric nested dissection + multifrontal
// factorisation with explicit front inverses, batched level by level.  This file: the factorisation.
//
// What it replaces: the reference hands A to a sparse LU (`problemo.BestSolver` -> SuperLU,
// zephyr/backend/discretization.py:78-103) and re-uses the factors for every source.  This is the same
// idea laid out for the GPU: the elimination tree of a regular grid is known in closed form, every front of
// one tree level has the same (padded) shape, so each level is a handful of strided-batched dense kernels.
//
//   tree      recursive bisection of the (nz, nx) rectangle by one-cell-wide separator lines (a one-cell line
//             separates a 9-point stencil); regions with both sides <= LEAF are eliminated whole.        nd_plan.hip
//   front     [separator cells | ring of cells around the subtree's region] -- the ring cells are exactly the
//             ancestors' separator cells the subtree touches.  Index <-> cell maps are closed-form (nd_local /
//             nd_cell, direct.hpp), so no index lists are stored.
//   factor    a front is written once in gather form (k_nd_build_front: stencil entries + the children's Schur complements);
//             F11^-1 in place (nd_gj.hip), G21 = F21 F11^-1, Schur complement S = F22 - G21 F12 gathered by the product that
//             forms it (nd_gemm.hip, IDX 4); the leaf level in one kernel straight from the coefficient planes (nd_leaf.hip);
//             fronts whose F11 is too ill-conditioned for an explicit inverse are eliminated again with a pivoted LU (NdStable).
//   solve     forward: front-local vectors travel up the tree exactly like the Schur complements;
//             backward: x_S = F11^-1 (y_S - F12 x_B), top-down.  Pure GEMMs on node-major right-hand sides
//             X[cell][rhs], no atomics, bit-reproducible.                                                 nd_passes.hip
//   accuracy  the true residual q' - A x of what is returned is evaluated with the stencil (nd_resid.hip); right-hand sides
//             above rtol take a step of iterative refinement (capi.hip).
//
// All dense arithmetic is fp64 complex on the matrix cores (v_mfma_f64_16x16x4_f64, nd_gemm_body.hpp).
#include "nd_internal.hpp"
#include <map>
#include <mutex>
#include <cstring>

// ---- kernels ---------------------------------------------------------------------------------------------------
namespace {

// The front [[F11, F12], [F21, F22]] is assembled where each block is needed afterwards: [F11 | F12] side by side in the factor
// storage (rows of smax + mmax; F11 is inverted in place, F12 is kept -- for leaves it becomes -F11^-1 F12 after the Schur complement, so
// that the back substitution of a leaf is ONE product [F11^-1 | -F11^-1 F12] [y_S; x_B]), [F21 | F22] in the scratch arena (F21 feeds G21, F22 becomes the Schur
// complement the parent picks up).  (r, c): padded front coordinates.
// Several frequencies in the same launches (NdFactor::nf > 1, helm_prefactor_many): the plan is geometry only, so the fronts of nf operators ride in one
// strided batch, batch index = front * nf + frequency.  Every array of the single layout becomes nf interleaved copies of itself -- a front's slot of
// `slot` elements at offset `off` moves to  nf * off + kf * slot  -- so the batched products and inversions see one uniform stride (their kernels do not
// change at all) and the passes of frequency kf see theirs with stride nf * slot.  nf = 1, kf = 0 is the layout of rounds 1-5, bit for bit.
__device__ __forceinline__ long long mf_off(long long off, long long slot, int nf, int kf) { return (long long)nf * off + (long long)kf * slot; }
__device__ __forceinline__ cplx *front_entry(const NdDev &n, cplx *arenaF, cplx *fac, int r, int c, int nf = 1, int kf = 0, int fac_private = 0) {
    const int nmax = n.smax + n.mmax;
    if (r < n.smax) {
        // [F11 | F12] share their rows (fac_private: a buffer of this front alone -- the pivoted-LU storage of an ill-conditioned front)
        return fac + (fac_private ? n.finv_off : mf_off(n.finv_off, (long long)n.smax * nmax, nf, kf)) + (long long)r * nmax + c;
    }
    return arenaF + mf_off(n.foff, (long long)n.mmax * nmax, nf, kf) + (long long)(r - n.smax) * nmax + c;
}

__global__ __launch_bounds__(256) void k_nd_assemble(const NdDev *nodes, int first, cplx *arenaF, cplx *fac, const cplx *planes, int nz, int nx) {
    const NdDev n = nodes[first + blockIdx.y];
    const long long N = (long long)nz * nx;
    const int tot = n.s + n.m;
    for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < n.smax; a += gridDim.x * blockDim.x)
        if (a >= n.s) *front_entry(n, arenaF, fac, a, a) = cmake(1.0, 0.0);     // padded separator slots: identity
    for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < tot; a += gridDim.x * blockDim.x) {
        int z, x, ca;
        nd_cell(n, a, z, x, ca);
        const int ra = nd_pos(n, a);
        // dof > 2 (column mode, the 3-D coarse solve): a cell is a column of dof unknowns along the slowest axis of a (dof, nz, nx) grid with a
        // 27-point operator -- component ca couples to ca - 1, ca, ca + 1 of the nine neighbour columns; plane 9 (dc + 1) + 3 (dz + 1) + dx + 1
        const int cb0 = n.dof > 2 ? max(ca - 1, 0) : 0, cb1 = n.dof > 2 ? min(ca + 1, n.dof - 1) : n.dof - 1;
        for (int cb = cb0; cb <= cb1; ++cb) {
            // dof 2: row component ca, column component cb -> Eurus block 2 ca + cb (M1 M2 / M3 M4), nine planes each
            const cplx *pl = n.dof > 2 ? planes + (long long)(cb - ca + 1) * 9 * N * n.dof + (long long)ca * N
                                       : planes + (long long)(n.dof == 2 ? 2 * ca + cb : 0) * 9 * N;
            const long long pstride = n.dof > 2 ? N * n.dof : N;
            #pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int z2 = z + k / 3 - 1, x2 = x + k % 3 - 1;
                if (z2 < 0 || z2 >= nz || x2 < 0 || x2 >= nx) continue;
                const int b = nd_local(n, nz, nx, z2, x2, cb);
                if (b < 0 || (a >= n.s && b >= n.s)) continue;            // ring x ring entries belong to an ancestor
                *front_entry(n, arenaF, fac, ra, nd_pos(n, b)) = pl[(long long)k * pstride + (long long)z * nx + x];
            }
        }
    }
}

// parent front += Schur complement of child `slot`; the m x m entries of the child's F22 are spread over gridDim.x blocks
// (every block rebuilds the child-ring -> parent-row map in LDS, so a block takes `chunk` entries: large enough to amortise that)
__global__ __launch_bounds__(256) void k_nd_extend_add(const NdDev *nodes, int first, int slot, cplx *arenaF, cplx *fac, int nz, int nx, int chunk) {
    extern __shared__ int map[];
    const NdDev p = nodes[first + blockIdx.y];
    if (p.kid[slot] < 0) return;
    const NdDev c = nodes[p.kid[slot]];
    const long long total = (long long)c.m * c.m;
    if ((long long)blockIdx.x * chunk >= total) return;
    for (int a = threadIdx.x; a < c.m; a += blockDim.x) {
        int z, x, comp;
        nd_cell(c, c.s + a, z, x, comp);
        map[a] = nd_pos(p, nd_local(p, nz, nx, z, x, comp));
    }
    __syncthreads();
    const cplx *Sc = arenaF + c.foff + c.smax;          // child's F22: row a at Sc + a * ldc
    const int ldc = c.smax + c.mmax;
    for (long long e0 = (long long)blockIdx.x * chunk; e0 < total; e0 += (long long)gridDim.x * chunk) {
        const long long e1 = e0 + chunk < total ? e0 + chunk : total;
        for (long long e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
            const int a = (int)(e / c.m), b2 = (int)(e - (long long)a * c.m);
            cplx *dst = front_entry(p, arenaF, fac, map[a], map[b2]);
            *dst = cadd(*dst, Sc[(long long)a * ldc + b2]);
        }
    }
}

// One pass that WRITES every entry of a front exactly once (gather form of k_nd_assemble + the two k_nd_extend_add + the three
// memsets they needed):  entry (r, c) = [stencil coefficient of the two cells, unless both lie on the ring]
//                                       + child 0's Schur complement entry + child 1's, where both cells lie on that child's ring
// (child 0 first: same summation order as the scatter form, bit for bit), identity on the padded separator diagonal, zero on
// every other padded slot.  The inverse maps are the same closed-form nd_cell / nd_local; a workgroup tabulates them for the
// front's rows once in LDS and then streams `rb` rows.  Traffic per level: children's F22 read once, the fronts written once
// (the scatter form read-modify-wrote the parents twice on top of the memsets).
// (OVR: the front's record is the launch argument `ovr`, not nodes[...] -- one front rebuilt with its [F11 | F12] rows redirected, see NdStable.  A template, not a
// run-time choice: `use_ovr ? ovr : nodes[i]` made the compiler keep the 100-byte record in vector registers -- 154 instead of 40, three waves per SIMD instead of
// eight, and the kernel of every level 30-50 % slower: 1.53 -> 2.04 ms per work item between rounds 4 and 5, found in round 6 with round 4's tree beside this one.)
template <bool OVR>
__global__ __launch_bounds__(256) void k_nd_build_front(const NdDev *nodes, int first, cplx *arenaF, cplx *fac, NdPlanesSet pset, int nz, int nx, int rb,
                                                        int skip22, NdDev ovr, int nf, int kfix) {
    constexpr int use_ovr = OVR ? 1 : 0;
    extern __shared__ int2 finfo[];        // per padded row: x = z | x << 16 (-1: padding), y = (k0 + 1) | (k1 + 1) << 14 | comp << 28
    // (nf > 1: batch index = front * nf + frequency; kfix >= 0: one frequency's front alone, the re-elimination of an ill-conditioned one)
    const int kf = kfix >= 0 ? kfix : (int)(blockIdx.y % nf);
    const cplx *planes = pset.p[kf];
    NdDev n;
    if (OVR) n = ovr; else n = nodes[first + (kfix >= 0 ? blockIdx.y : blockIdx.y / nf)];
    const int nmax = n.smax + n.mmax;
    const int r0 = blockIdx.x * rb;
    if (r0 >= nmax) return;
    const bool h0 = n.kid[0] >= 0, h1 = n.kid[1] >= 0;
    NdDev c0 = NdDev(), c1 = NdDev();
    if (h0) c0 = nodes[n.kid[0]];
    if (h1) c1 = nodes[n.kid[1]];
    const int tid = threadIdx.x;
    for (int r = tid; r < nmax; r += 256) {
        int a = -1;
        if (r < n.s) a = r;
        else if (r >= n.smax && r - n.smax < n.m) a = n.s + r - n.smax;
        int2 e = make_int2(-1, 0);
        if (a >= 0) {
            int z, x, comp;
            nd_cell(n, a, z, x, comp);
            int k0 = 0, k1 = 0;
            if (h0) { const int la = nd_local(c0, nz, nx, z, x, comp); if (la >= c0.s) k0 = la - c0.s + 1; }
            if (h1) { const int la = nd_local(c1, nz, nx, z, x, comp); if (la >= c1.s) k1 = la - c1.s + 1; }
            if (n.dof > 2) { e.x = z | (x << 12) | (comp << 24); e.y = k0 | (k1 << 16); }      // (column mode: up to 127 components, rings up to 65534)
            else { e.x = z | (x << 16); e.y = k0 | (k1 << 14) | (comp << 28); }
        }
        finfo[r] = e;
    }
    __syncthreads();
    const long long N = (long long)nz * nx;
    const int ld0 = c0.smax + c0.mmax, ld1 = c1.smax + c1.mmax;
    const cplx *S0 = h0 ? arenaF + mf_off(c0.foff, (long long)c0.mmax * ld0, nf, kf) + c0.smax : nullptr;
    const cplx *S1 = h1 ? arenaF + mf_off(c1.foff, (long long)c1.mmax * ld1, nf, kf) + c1.smax : nullptr;
    const int r1 = r0 + rb < nmax ? r0 + rb : nmax;
    const int tx = tid & 63, ty = tid >> 6;
    const bool colmode = n.dof > 2;
    // entry (r, c) of the front from the row table: stencil coefficient + the two children's Schur-complement entries
    auto value = [&](int r, int c, const int2 ia, const int2 ib) -> cplx {
        cplx v = cmake(0.0, 0.0);
        if (ia.x < 0 || ib.x < 0) { if (r == c && r < n.smax) v = cmake(1.0, 0.0); return v; }
        const int za = colmode ? (ia.x & 0xfff) : (ia.x & 0xffff), xa = colmode ? ((ia.x >> 12) & 0xfff) : (ia.x >> 16);
        const int ca = colmode ? ((ia.x >> 24) & 0x7f) : ((ia.y >> 28) & 1);
        const int a0 = (colmode ? (ia.y & 0xffff) : (ia.y & 0x3fff)) - 1, a1 = (colmode ? ((ia.y >> 16) & 0xffff) : ((ia.y >> 14) & 0x3fff)) - 1;
        if (r < n.smax || c < n.smax) {              // ring x ring entries belong to an ancestor
            const int zb = colmode ? (ib.x & 0xfff) : (ib.x & 0xffff), xb = colmode ? ((ib.x >> 12) & 0xfff) : (ib.x >> 16);
            const int dz = zb - za, dx = xb - xa;
            if (dz >= -1 && dz <= 1 && dx >= -1 && dx <= 1) {
                if (colmode) {
                    const int dc = ((ib.x >> 24) & 0x7f) - ca;
                    if (dc >= -1 && dc <= 1) v = planes[((long long)(dc + 1) * 9 + (dz + 1) * 3 + dx + 1) * N * n.dof + (long long)ca * N + (long long)za * nx + xa];
                } else {
                    const int blk = n.dof == 2 ? 2 * ca + ((ib.y >> 28) & 1) : 0;
                    v = planes[((long long)blk * 9 + (dz + 1) * 3 + dx + 1) * N + (long long)za * nx + xa];
                }
            }
        }
        const int b0 = (colmode ? (ib.y & 0xffff) : (ib.y & 0x3fff)) - 1, b1 = (colmode ? ((ib.y >> 16) & 0xffff) : ((ib.y >> 14) & 0x3fff)) - 1;
        if (a0 >= 0 && b0 >= 0) v = cadd(v, S0[(long long)a0 * ld0 + b0]);
        if (a1 >= 0 && b1 >= 0) v = cadd(v, S1[(long long)a1 * ld1 + b1]);
        return v;
    };
    // skip22: the ring x ring block (the sum of the children's Schur complements, most of a front below the tree top) is not
    // materialised -- the Schur-complement product gathers it itself (k_zgemm3<.., 4, ..>) and writes S where F22 would have been.
    // One entry per lane at a time, 40 registers: eight waves per SIMD cover the gathers' latency.  (Round 5 also gathered four entries per lane before storing
    // any; beside the register blow-up described at the top of the kernel that form measured 2 % ahead of this loop, A/B in one binary -- not worth a second path.)
    for (int r = r0 + ty; r < r1; r += 4) {
        const int2 ia = finfo[r];
        const int cend = (skip22 && r >= n.smax) ? n.smax : nmax;
        for (int c = tx; c < cend; c += 64) *front_entry(n, arenaF, fac, r, c, nf, kf, use_ovr) = value(r, c, ia, finfo[c]);
    }
}

// ---- ill-conditioned fronts (NdStable) -------------------------------------------------------------------------------------------------
// Condition estimate of the s x s pivot block of every front of a group: ||F11||_inf ||F11^-1||_inf (|z| taken as |re| + |im|), two launches around
// the inversion -- with DECOUPLED rows taken out of the norms.  A row of F11 whose only entry is its diagonal (the MiniZephyr system keeps identity rows
// on the outer boundary, norm 1 beside interior rows of norm 1e-5) is eliminated exactly whatever its scale, but counted as it stands it multiplies the
// product of the norms by that scale ratio: every front touching the boundary came out at 1e6 and was handed to the pivoted LU -- all fronts of a small
// model, more than a group treats, and which made it into the list depended on the order the flagging threads ran in (round 5, found through run-to-run
// differences in the last bits).  Such a row j is weighted  w_j = d_j / a  (d_j its norm, a the largest norm among the coupled rows), the others w_j = 1:
//   AFTER = 0 (before the inversion)   out[front] = a,  rows[front * smax + j] = w_j
//   AFTER = 1 (after it)               out[front] = max_i sum_j |F11^-1|_ij w_j
// With no decoupled row (every Eurus front) this IS the product of the two infinity norms the threshold was calibrated on in rounds 3-4.  (Full row
// equilibration, w_j = d_j for every row, was tried first in round 5: it halves the estimate of exactly the fronts that matter -- the 1024^2 operator at
// 8 Hz kept a front at 1.0e5 under the 1.1e5 threshold and paid a refinement pass, first-pass residual 6.8e-9 -- because the pivot search inside a
// 32-wide block is by absolute size, so the gradual row scaling of the PML does cost accuracy and belongs in the estimate.)
// (The max-entry norm was tried first in round 3: for a near-singular front F11^-1 ~ u v^T / sigma with u, v spread over all unknowns, and max |entry|
// underestimates the norm by the front's size.  Over the front's own s x s unknowns: the identity that pads a smaller front is not part of it.)
template <int AFTER>
__global__ __launch_bounds__(256) void k_front_cond(const cplx *M0, int ld, long long stride, const NdDev *nodes, double *out, double *rows, int smax, int nf) {
    __shared__ double red[4];
    __shared__ double dsh[LUS_NMAX];
    __shared__ unsigned char lone[LUS_NMAX];
    const cplx *M = M0 + (long long)blockIdx.x * stride;
    double *w = rows + (long long)blockIdx.x * smax;
    const int n = nodes[blockIdx.x / nf].s;              // (nf > 1: batch index = front * nf + frequency)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double best = 0.0;
    for (int i = wv; i < n; i += 4) {                      // a wave per row: coalesced along the row
        double v = 0.0, off = 0.0;
        for (int j = lane; j < n; j += 64) {
            const cplx a = M[(long long)i * ld + j];
            const double m = fabs(a.x) + fabs(a.y);
            v += AFTER ? m * w[j] : m;
            if (!AFTER && j != i) off += m;
        }
        for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o); if (!AFTER) off += __shfl_down(off, o); }
        v = __shfl(v, 0);
        if (!AFTER && lane == 0) { dsh[i] = v; lone[i] = off == 0.0; }
        best = fmax(best, v);
    }
    if (!AFTER) {
        __syncthreads();
        double a = 0.0, any = 0.0;
        for (int i = 0; i < n; ++i) { any = fmax(any, dsh[i]); if (!lone[i]) a = fmax(a, dsh[i]); }      // (n <= 128, every thread the same loop: no second reduction)
        if (!(a > 0.0)) a = any > 0.0 ? any : 1.0;           // nothing but decoupled rows
        for (int i = threadIdx.x; i < n; i += 256) w[i] = lone[i] ? dsh[i] / a : 1.0;
        if (threadIdx.x == 0) out[blockIdx.x] = a;
        return;
    }
    if (lane == 0) red[wv] = best;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
// list[0] = number of fronts with  scale * a[j] * b[j] > thr  (capped), list[1..] = their positions in the group, worst first is not needed
// (nf, kf: the estimates of nf interleaved frequencies lie at j * nf + kf; the list holds front positions j)
__global__ void k_front_flag(const double *a, const double *b, int cnt, double scale, double thr, int *list, int cap, int nf, int kf) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += gridDim.x * blockDim.x)
        if (!(scale * a[(long long)j * nf + kf] * b[(long long)j * nf + kf] <= thr)) {                   // (NaN counts as flagged)
            const int slot = atomicAdd(list, 1);
            if (slot < cap) list[1 + slot] = j;
        }
}
// the same for n <= 64 with the matrix held in LDS (every elimination step is then a few hundred nanoseconds instead of several global round trips)
__global__ __launch_bounds__(256) void k_lu_factor64(cplx *A0, int ld, int n, int *piv) {
    __shared__ cplx A[64][65];
    __shared__ double rv[256];
    __shared__ int ri[256];
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += 256) A[e / n][e % n] = A0[(long long)(e / n) * ld + e % n];
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        double best = -1.0; int bi = k;
        for (int i = k + tid; i < n; i += 256) { const double v = cabs2(A[i][k]); if (v > best) { best = v; bi = i; } }
        rv[tid] = best; ri[tid] = bi;
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) {          // (at most 64 candidates: the first wave's entries)
            if (tid < off && (rv[tid + off] > rv[tid] || (rv[tid + off] == rv[tid] && ri[tid + off] < ri[tid]))) { rv[tid] = rv[tid + off]; ri[tid] = ri[tid + off]; }
            __syncthreads();
        }
        const int p = ri[0];
        if (tid == 0) piv[k] = p;
        if (p != k && tid < n) { const cplx t = A[k][tid]; A[k][tid] = A[p][tid]; A[p][tid] = t; }
        __syncthreads();
        const cplx d = crecip(A[k][k]);
        if (tid > k && tid < n) A[tid][k] = cmul(A[tid][k], d);
        __syncthreads();
        const int w = n - k - 1;
        for (int e = tid; e < w * w; e += 256) {
            const int i = k + 1 + e / w, j = k + 1 + e % w;
            const cplx l = A[i][k], u = A[k][j];
            cplx a = A[i][j];
            a.x = fma(-l.x, u.x, a.x); a.x = fma(l.y, u.y, a.x);
            a.y = fma(-l.x, u.y, a.y); a.y = fma(-l.y, u.x, a.y);
            A[i][j] = a;
        }
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += 256) A0[(long long)(e / n) * ld + e % n] = A[e / n][e % n];
}
// LU with partial pivoting of one n x n matrix in global memory (one workgroup; the matrix is small and lives in L2)
__global__ __launch_bounds__(256) void k_lu_factor(cplx *A, int ld, int n, int *piv) {
    __shared__ double rv[256];
    __shared__ int ri[256];
    __shared__ int psh;
    const int tid = threadIdx.x;
    for (int k = 0; k < n; ++k) {
        double best = -1.0; int bi = k;
        for (int i = k + tid; i < n; i += 256) { const double v = cabs2(A[(long long)i * ld + k]); if (v > best) { best = v; bi = i; } }
        rv[tid] = best; ri[tid] = bi;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off && (rv[tid + off] > rv[tid] || (rv[tid + off] == rv[tid] && ri[tid + off] < ri[tid]))) { rv[tid] = rv[tid + off]; ri[tid] = ri[tid + off]; }
            __syncthreads();
        }
        if (tid == 0) { psh = ri[0]; piv[k] = ri[0]; }
        __syncthreads();
        const int p = psh;
        if (p != k) for (int j = tid; j < n; j += 256) { const cplx t = A[(long long)k * ld + j]; A[(long long)k * ld + j] = A[(long long)p * ld + j]; A[(long long)p * ld + j] = t; }
        __syncthreads();
        const cplx d = crecip(A[(long long)k * ld + k]);
        for (int i = k + 1 + tid; i < n; i += 256) A[(long long)i * ld + k] = cmul(A[(long long)i * ld + k], d);
        __syncthreads();
        const int w = n - k - 1;
        for (int e = tid; e < w * w; e += 256) {
            const int i = k + 1 + e / w, j = k + 1 + e % w;
            const cplx l = A[(long long)i * ld + k], u = A[(long long)k * ld + j];
            cplx a = A[(long long)i * ld + j];
            a.x = fma(-l.x, u.x, a.x); a.x = fma(l.y, u.y, a.x);
            a.y = fma(-l.x, u.y, a.y); a.y = fma(-l.y, u.x, a.y);
            A[(long long)i * ld + j] = a;
        }
        __syncthreads();
    }
}
// B <- (L U)^-1 P B in place (B row-major, leading dimension ldb; n <= 128).  A workgroup takes 16 columns into LDS; a wave owns four of
// them, 16 lanes per column: a row's dot product against the rows already solved is split over the 16 lanes (k = p, p + 16, ...) and summed
// with four shuffles.  Everything a wave touches in LDS is its own four columns, so there is no barrier inside the substitutions (LDS
// operations of a wave execute in order); the factor's row i + 1 is in flight from L2 while row i is being reduced.
__global__ __launch_bounds__(256) void k_lu_solve(const cplx *__restrict__ LU, int ld, int n, const int *__restrict__ piv, cplx *B, int ldb, int ncols) {
    __shared__ cplx Bt[LUS_NMAX][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cl = wv * 4 + (lane >> 4), p = lane & 15;          // local column 0..15, part 0..15
    const int j0 = blockIdx.x * 16;
    for (int e = tid; e < n * 16; e += 256) { const int i = e >> 4, c = e & 15; Bt[i][c] = j0 + c < ncols ? B[(long long)i * ldb + j0 + c] : cmake(0.0, 0.0); }
    __syncthreads();
    if (p == 0) for (int k = 0; k < n; ++k) { const int q = piv[k]; if (q != k) { const cplx t = Bt[k][cl]; Bt[k][cl] = Bt[q][cl]; Bt[q][cl] = t; } }
    __builtin_amdgcn_wave_barrier();
    constexpr int NK = LUS_NMAX / 16;
    // The factor's rows come from L2 (a 128-wide front's L\U is 256 KB: no room in LDS beside the columns), one row per elimination step: with one row of
    // prefetch the step waited for its row (160 us per launch at n = 128, 2 n dependent steps of ~0.6 us).  Four rows are kept in flight in a ring of
    // register sets, slot u of an unrolled group of four steps.
    constexpr int PD = 4;
    cplx rb[PD][NK];
    auto load_row = [&](cplx (&dst)[NK], int i, int klo, int khi) {        // entries k in [klo, khi), k = p + 16 q
        #pragma unroll
        for (int q = 0; q < NK; ++q) { const int k = p + 16 * q; dst[q] = (i >= 0 && i < n && k >= klo && k < khi) ? LU[(long long)i * ld + k] : cmake(0.0, 0.0); }
    };
    // L y = P b (unit lower triangle): rows top-down
    #pragma unroll
    for (int u = 0; u < PD; ++u) load_row(rb[u], 1 + u, 0, 1 + u);
    for (int i0 = 1; i0 < n; i0 += PD) {
        #pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int i = i0 + u;
            if (i < n) {                                                     // (uniform across the workgroup)
                cplx acc = cmake(0.0, 0.0);
                #pragma unroll
                for (int q = 0; q < NK; ++q) { const int k = p + 16 * q; if (k < i) cfma(acc, rb[u][q], Bt[k][cl]); }
                load_row(rb[u], i + PD, 0, i + PD);
                #pragma unroll
                for (int off = 8; off > 0; off >>= 1) { acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); }
                if (p == 0) Bt[i][cl] = csub(Bt[i][cl], acc);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    // U x = y: rows bottom-up
    #pragma unroll
    for (int u = 0; u < PD; ++u) load_row(rb[u], n - 1 - u, n - 1 - u, n);
    for (int i0 = n - 1; i0 >= 0; i0 -= PD) {
        #pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int i = i0 - u;
            if (i >= 0) {
                cplx acc = cmake(0.0, 0.0), d = cmake(1.0, 0.0);
                #pragma unroll
                for (int q = 0; q < NK; ++q) { const int k = p + 16 * q; if (k > i && k < n) cfma(acc, rb[u][q], Bt[k][cl]); if (k == i) d = rb[u][q]; }
                load_row(rb[u], i - PD, i - PD, n);
                #pragma unroll
                for (int off = 8; off > 0; off >>= 1) { acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); }
                // the diagonal entry sits with the lane whose k == i: hand it to lane 0 of the group
                const int src = (lane & 48) | (i & 15);
                const double dx = __shfl(d.x, src), dy = __shfl(d.y, src);
                if (p == 0) Bt[i][cl] = cmul(csub(Bt[i][cl], acc), crecip(cmake(dx, dy)));
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < n * 16; e += 256) { const int i = e >> 4, c = e & 15; if (j0 + c < ncols) B[(long long)i * ldb + j0 + c] = Bt[i][c]; }
}

}  // namespace

void launch_lu_solve(hipStream_t st, const cplx *LU, int ld, int n, const int *piv, cplx *B, int ldb, int ncols) {
    HELM_LAUNCH(k_lu_solve, dim3((ncols + 15) / 16), dim3(256), 0, st, LU, ld, n, piv, B, ldb, ncols);
}

// ---- factorisation ---------------------------------------------------------------------------------------------
static void stable_free(NdFactor *f) {
    const int dev = f->pd ? f->pd->device : 0;
    for (NdStable &st : f->stable) {
        const size_t nmax = (size_t)st.smax + st.mmax;
        helm_pool_free(dev, st.lu, (size_t)st.smax * nmax * sizeof(cplx));
        helm_pool_free(dev, st.f21, (size_t)std::max(1, st.mmax) * st.smax * sizeof(cplx));
        helm_pool_free(dev, st.piv, (size_t)st.smax * sizeof(int));
        helm_pool_free(dev, st.vs, st.vs_elems * sizeof(cplx));
    }
    f->stable.clear();
}

// r6: the slots live in a process-wide free list per device (they used to be thread_local: every fresh prepare thread of the dispatcher paid a hipHostMalloc
// on its first factorisation's critical path and never gave the page or the event back); a factorisation holds one between flag_group and stabilise_group.
struct FlagSlot { int *host = nullptr; hipEvent_t ev = nullptr; int device = 0; };
namespace { std::mutex g_flag_mu; std::map<int, std::vector<FlagSlot *>> g_flag_idle; }
FlagSlot *flag_slot_acquire(int device) {
    {
        std::lock_guard<std::mutex> lk(g_flag_mu);
        std::vector<FlagSlot *> &idle = g_flag_idle[device];
        if (!idle.empty()) { FlagSlot *s = idle.back(); idle.pop_back(); return s; }
    }
    FlagSlot *s = new FlagSlot(); s->device = device;
    s->host = (int *)helm_hostpool_alloc((ND_STABLE_CAP + 1) * sizeof(int));
    if (!s->host) { delete s; return nullptr; }
    if (hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); helm_hostpool_free(s->host, (ND_STABLE_CAP + 1) * sizeof(int)); delete s; return nullptr; }
    return s;
}
void flag_slot_release(FlagSlot *s) {
    if (!s) return;
    std::lock_guard<std::mutex> lk(g_flag_mu);
    g_flag_idle[s->device].push_back(s);
}

void nd_free(NdFactor *f) {
    if (!f) return;
    stable_free(f);
    if (f->flag_slot) { (void)hipEventSynchronize(f->flag_slot->ev); flag_slot_release(f->flag_slot); f->flag_slot = nullptr; }
    if (f->d_est && !f->shared) helm_pool_free(f->pd ? f->pd->device : 0, f->d_est, f->est_elems * sizeof(double));
    if (f->d_leafflag) helm_pool_free(f->pd ? f->pd->device : 0, f->d_leafflag, f->leafflag_elems * sizeof(int));
    if (f->d_act) helm_pool_free(f->pd ? f->pd->device : 0, f->d_act, f->act_elems * sizeof(int));
    if (f->d_qmask) helm_pool_free(f->pd ? f->pd->device : 0, f->d_qmask, f->qmask_elems);
    if (f->d_fac && !f->shared) helm_pool_free(f->pd ? f->pd->device : 0, f->d_fac, (size_t)f->pd->plan.fac_elems * sizeof(cplx));
    delete f;                      // (a factor of a set: the shared buffers go with the last of them, ~NdFacShared)
}
NdFacShared::~NdFacShared() {
    if (d_fac) helm_pool_free(device, d_fac, fac_bytes);
    if (d_est) helm_pool_free(device, d_est, est_elems * sizeof(double));
}

// Fronts that keep G = -F11^-1 F12 where F12 was, so that their back substitution is ONE product [F11^-1 | G] [y_S; x_B]: the leaves.
// (Round 4 measured the same for every separator front of at most 64 unknowns: the 64-row tile pads the 8-, 16- and 32-row fronts of levels
// 13-10 eight-, four- and two-fold on the matrix cores, 5.5 -> 6.7 ms per pass; removed.)
bool merged_group(const NdPlan &P, const NdGroup &g) { (void)P; return g.mmax > 0 && g.leaf; }

namespace {

// ---- ill-conditioned fronts: detection and re-elimination with a pivoted LU (see NdStable in direct.hpp) -------------------------------------
// ON by default (HELM_ND_STABLE=0 switches it off).  Measured on the 16-frequency bench job (MI355X, round 3): every wavefield meets rtol
// 1e-10 in ONE pass (passes per wavefield 1.15 -> 1.00, worst first-pass residual 7e-9 -> 3e-11) and the job runs at 9180 against 8170
// wavefields/s: detection costs 0.5-0.9 ms per factorisation (two norm kernels and one 4-byte read-back per watched tree level), a treated
// front ~0.15 ms at factor time and per pass, 0-15 fronts per frequency are taken, and 5 of 16 frequencies save a refinement pass of 8-20 ms.
// (A first version with one thread per column in the triangular solves and a threshold of 2e4 that also watched the leaves was correct but
// slower than doing nothing: 7775 wavefields/s.)
// A front is taken when its condition estimate ||F11||_inf ||F11^-1||_inf exceeds rtol / (8 eps) (see stabilise_group; 1.1e5 at rtol 1e-10: at
// 1024^2 / 9 Hz a handful of 32 767 fronts, which between them are the difference between a first-pass residual of 4e-9 and 2e-12; a typical
// leaf sits at 20-40, the tree top at 50-1000).
// Fronts of more than HELM_ND_STABLE_SMAX (128) separator unknowns are left alone: the one-workgroup LU would cost more than the
// refinement pass it saves, and none that large has been seen ill-conditioned (the tree top sits at cond 50-1000)
bool stable_enabled(const NdPlan &P) {
    // one unknown per cell only: the fix-ups of the passes gather a front's right-hand-side rows again, which the in-place passes of the
    // coupled system's rhs-major path have overwritten by then
    return helm_tuning_now().nd_stable != 0 && P.dof == 1;
}
// d_est of a set of nf frequencies (nf = 1: a factor on its own): [a: nf cnt][b: nf cnt] of the group at hand, entry front * nf + frequency; from
// est_list_off on nf flag lists of ND_STABLE_CAP + 1 ints; from est_rows_off on the row sums of its pivot blocks (nf cnt x smax, same order)
const size_t kEstList = (ND_STABLE_CAP + 2) / 2 + 2;            // doubles per flag list
size_t est_list_off(const NdPlan &P, int nf) {
    int maxcnt = 1;
    for (const NdGroup &g : P.groups) maxcnt = std::max(maxcnt, g.cnt);
    return 2 * (size_t)nf * maxcnt;
}
size_t est_rows_off(const NdPlan &P, int nf) { return est_list_off(P, nf) + (size_t)nf * kEstList + 8; }
// (the estimates belong to the set: frequency 0's factor holds the pointer for all of them, NdFacShared owns the buffer when nf > 1)
int ensure_est(helm_op *op, NdFactor *f, int stable_smax) {
    const NdPlan &P = f->pd->plan;
    const int nf = f->nf;
    size_t rows = 0;
    for (const NdGroup &g : P.groups) if (!g.leaf && g.mmax > 0 && g.smax <= stable_smax) rows = std::max(rows, (size_t)g.cnt * g.smax);
    const size_t need = est_rows_off(P, nf) + (size_t)nf * rows;
    if (f->est_elems >= need) return HELM_OK;
    if (f->shared) {
        if (f->shared->d_est) { hipStreamSynchronize(op->stream); helm_pool_free(op->device, f->shared->d_est, f->shared->est_elems * sizeof(double)); f->shared->d_est = nullptr; f->shared->est_elems = 0; }
        f->shared->d_est = (double *)helm_pool_alloc(op->device, need * sizeof(double));
        if (!f->shared->d_est) return HELM_ERR_DEVICE;
        f->shared->est_elems = need;
        f->d_est = f->shared->d_est; f->est_elems = need;
        return HELM_OK;
    }
    if (f->d_est) { hipStreamSynchronize(op->stream); helm_pool_free(op->device, f->d_est, f->est_elems * sizeof(double)); f->d_est = nullptr; f->est_elems = 0; }
    f->d_est = (double *)helm_pool_alloc(op->device, need * sizeof(double));
    if (!f->d_est) return HELM_ERR_DEVICE;
    f->est_elems = need;
    return HELM_OK;
}

// The list of a group's ill-conditioned fronts comes back through a pinned buffer and an event of the calling thread (one per device): flag_group
// enqueues the flagging and the copy right after the inversion, the group's products are enqueued behind them, and stabilise_group waits for the
// EVENT only -- the list is on the host while the products still run, and with nothing flagged (the usual case) the next group is enqueued without the
// stream ever running dry.  (Rounds 3-4: hipStreamSynchronize after the products, once per watched group -- the factorisation stream idled for a host
// round trip 11 times per operator, and in the pipelined job the factorisation span is what a step waits for.)
// after the inversion of group gi: flag its ill-conditioned fronts and start the list on its way to the host
// (f: the factor of frequency f->kf of a set of f->nf; est: the SET's estimates, ensure_est; rt: the tolerance that frequency's factors are built for)
int flag_group(helm_op *op, NdFactor *f, size_t gi, double *est, double rt_hint) {
    const NdPlan &P = f->pd->plan;
    const NdGroup &g = P.groups[gi];
    hipStream_t st = op->stream;
    // Which fronts are taken follows from the tolerance the factors are built for, not from a constant fitted to one model: an explicit inverse of a
    // front with condition number kappa leaves a first-pass residual of about kappa * eps (measured front by front, DESIGN.md 5.1: 8.8e5 -> 4e-9),
    // and it has to stay below rtol with a margin for the handful of such fronts that add up and for what the estimate misses:
    //   kappa_max = rtol / (HELM_ND_STABLE_SAFETY * eps),  safety 8 by default  ->  1.1e5 at the 1e-10 of the reference parity tests, 1.1e7
    // at 1e-8, 1.1e3 at 1e-12 (where most fronts of the middle levels would be taken: the floor of 2e3 keeps the treatment a handful-of-fronts
    // affair and leaves the rest to the refinement pass, which always exists).  HELM_ND_STABLE_THR overrides with a fixed number.
    const helm_tuning tune = helm_tuning_now();
    const double safety = tune.nd_stable_safety;
    const double rt = rt_hint > 0 ? rt_hint : 1e-10;
    const double thr = tune.nd_stable_thr > 0 ? tune.nd_stable_thr : std::min(1e9, std::max(2e3, rt / (safety * 1.1102230246251565e-16)));
    if (!f->flag_slot) f->flag_slot = flag_slot_acquire(op->device);
    FlagSlot *slot = f->flag_slot;
    if (!slot) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: no pinned buffer for the list of ill-conditioned fronts");
    f->flag_thr = thr;
    int *d_list = (int *)(est + est_list_off(P, f->nf) + (size_t)f->kf * kEstList);
    HIP_TRY(op, hipMemsetAsync(d_list, 0, sizeof(int), st));
    HELM_LAUNCH(k_front_flag, dim3((g.cnt + 255) / 256), dim3(256), 0, st, (const double *)est, (const double *)(est + (size_t)f->nf * g.cnt), g.cnt, 1.0, thr, d_list, ND_STABLE_CAP,
                f->nf, f->kf);
    HIP_TRY(op, hipMemcpyAsync(slot->host, d_list, (ND_STABLE_CAP + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(op, hipEventRecord(slot->ev, st));
    return HELM_OK;
}

// after the batched elimination of group gi (whose list flag_group has requested): eliminate each of its ill-conditioned fronts again
// (f: frequency f->kf of a set of f->nf; arenaF / work: the set's scratch; est: the set's estimates)
int stabilise_group(helm_op *op, NdFactor *f, size_t gi, cplx *arenaF, cplx *work, const cplx *planes, const double *est) {
    const NdPlan &P = f->pd->plan;
    const NdGroup &g = P.groups[gi];
    hipStream_t st = op->stream;
    const int nmax = g.smax + g.mmax;
    const int nf = f->nf, kf = f->kf;
    FlagSlot *slot = f->flag_slot;
    if (!slot) HELM_FAIL(op, HELM_ERR_STATE, "direct solver: stabilise_group without flag_group");
    int h_list[ND_STABLE_CAP + 1];
    {
        const hipError_t es = hipEventSynchronize(slot->ev);
        memcpy(h_list, slot->host, sizeof(h_list));
        f->flag_slot = nullptr; flag_slot_release(slot);        // (the list is on the host: the slot goes back before anything below can fail)
        if (es != hipSuccess) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: waiting for the list of ill-conditioned fronts failed: %s", hipGetErrorString(es));
    }
    const int nflag = std::max(0, std::min(h_list[0], ND_STABLE_CAP));
    if (h_list[0] > ND_STABLE_CAP) {
        // more flagged fronts than are treated per group: which of them made it into the list is up to the order the atomics ran in, and the factors would
        // differ from one factorisation of the same operator to the next.  Take the worst ND_STABLE_CAP of the FLAGGED fronts by the estimate the flag was
        // made from, a[j] * b[j] = ||F11|| ||F11^-1|| (ties: lower position).
        std::vector<double> eh(2 * (size_t)nf * g.cnt);
        HIP_TRY(op, hipStreamSynchronize(st));
        HIP_TRY(op, hipMemcpy(eh.data(), est, eh.size() * sizeof(double), hipMemcpyDeviceToHost));
        const double thr = f->flag_thr;
        std::vector<std::pair<double, int>> cand;
        for (int j = 0; j < g.cnt; ++j) {
            const double e = eh[(size_t)j * nf + kf] * eh[(size_t)nf * g.cnt + (size_t)j * nf + kf];
            if (!(e <= thr)) cand.push_back(std::make_pair(e == e ? e : HUGE_VAL, j));                          // (the kernel's own test; NaN: worst)
        }
        auto worse = [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a.first != b.first ? a.first > b.first : a.second < b.second; };
        const int take = std::min<int>(nflag, (int)cand.size());
        std::partial_sort(cand.begin(), cand.begin() + take, cand.end(), worse);
        for (int q = 0; q < take; ++q) h_list[1 + q] = cand[q].second;
    }
    std::sort(h_list + 1, h_list + 1 + nflag);                                   // (the atomics hand the slots out in no particular order)
    if (getenv("HELM_ND_DEBUG") && atoi(getenv("HELM_ND_DEBUG")) >= 2) {
        std::vector<double> h(2 * (size_t)nf * g.cnt);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), est, h.size() * sizeof(double), hipMemcpyDeviceToHost);
        double worst = 0; int wj = 0;
        for (int j = 0; j < g.cnt; ++j) { const double e = h[(size_t)j * nf + kf] * h[(size_t)nf * g.cnt + (size_t)j * nf + kf]; if (!(e <= worst)) { worst = e; wj = j; } }
        fprintf(stderr, "[helm direct] level %d s %d cnt %d (frequency %d of %d): estimate of front 0 = %.3e * %.3e; worst %.3e at %d; flagged %d\n", g.level, g.smax, g.cnt, kf, nf, h[kf], h[(size_t)nf * g.cnt + kf], worst, wj, h_list[0]);
    }
    if ((long long)g.smax * g.mmax > P.work_elems) return HELM_OK;               // (no room for the s x m solve: leave the group as it is)
    const cplx one = cmake(1, 0), mone = cmake(-1, 0);
    for (int q = 0; q < nflag; ++q) {
        const int j = h_list[1 + q];
        NdStable S;
        S.node = g.first + j; S.group = gi; S.smax = g.smax; S.mmax = g.mmax;
        S.lu = (cplx *)helm_pool_alloc(op->device, (size_t)g.smax * nmax * sizeof(cplx));
        S.f21 = (cplx *)helm_pool_alloc(op->device, (size_t)std::max(1, g.mmax) * g.smax * sizeof(cplx));
        S.piv = (int *)helm_pool_alloc(op->device, (size_t)g.smax * sizeof(int));
        f->stable.push_back(S);                                                   // (owned by the factor from here on: freed by nd_free on every path)
        if (!S.lu || !S.f21 || !S.piv) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: allocation for an ill-conditioned front failed");
        NdDev n = P.nodes[S.node];
        const long long foff = (long long)nf * n.foff + (long long)kf * g.mmax * nmax;       // this frequency's [F21 | F22] of the front in the set's arena
        n.finv_off = 0; n.f12_off = g.smax;                                       // [F11 | F12] rows go to S.lu, [F21 | F22] back to the arena
        const int rb = std::max(std::min(nmax, 4), (nmax + 2047) / 2048);
        NdPlanesSet ps; for (int q2 = 0; q2 < ND_NF_MAX; ++q2) ps.p[q2] = planes;
        HELM_LAUNCH(k_nd_build_front<true>, dim3((nmax + rb - 1) / rb, 1), dim3(256), (size_t)nmax * sizeof(int2), st, f->pd->d_nodes, 0, arenaF, S.lu, ps, P.nz, P.nx, rb,
                           0, n, nf, kf);                                         // (the redirected node travels as a launch argument: no copy, no host wait)
        cplx *F21 = arenaF + foff, *F22 = arenaF + foff + g.smax;
        HIP_TRY(op, hipMemcpy2DAsync(S.f21, (size_t)g.smax * sizeof(cplx), F21, (size_t)nmax * sizeof(cplx), (size_t)g.smax * sizeof(cplx), (size_t)g.mmax, hipMemcpyDeviceToDevice, st));
        if (g.smax <= 64) HELM_LAUNCH(k_lu_factor64, dim3(1), dim3(256), 0, st, S.lu, nmax, g.smax, S.piv);
        else HELM_LAUNCH(k_lu_factor, dim3(1), dim3(256), 0, st, S.lu, nmax, g.smax, S.piv);
        // Schur complement through the same factors: F22 -= F21 (F11^-1 F12)
        HIP_TRY(op, hipMemcpy2DAsync(work, (size_t)g.mmax * sizeof(cplx), S.lu + g.smax, (size_t)nmax * sizeof(cplx), (size_t)g.mmax * sizeof(cplx), (size_t)g.smax, hipMemcpyDeviceToDevice, st));
        HELM_LAUNCH(k_lu_solve, dim3((g.mmax + 15) / 16), dim3(256), 0, st, (const cplx *)S.lu, nmax, g.smax, (const int *)S.piv, work, g.mmax, g.mmax);
        gemm(op, g.mmax, g.mmax, g.smax, mone, S.f21, g.smax, 0, work, g.mmax, 0, one, F22, nmax, 0, 1, nullptr);
        if (getenv("HELM_ND_DEBUG") && atoi(getenv("HELM_ND_DEBUG")) >= 3) {      // reproducibility probe: checksums of every piece of this front's re-elimination
            auto sum2d = [&](const void *src, size_t pitch, size_t wbytes, size_t rows_) {
                std::vector<unsigned char> h(wbytes * rows_);
                hipMemcpy2D(h.data(), wbytes, src, pitch, wbytes, rows_, hipMemcpyDeviceToHost);
                unsigned long long x = 1469598103934665603ull;
                for (unsigned char c : h) { x ^= c; x *= 1099511628211ull; }
                return x;
            };
            hipStreamSynchronize(st);
            fprintf(stderr, "[helm direct] treated front %d (level %d, s %d of %d, m %d of %d): lu %016llx f12w %016llx piv %016llx f21 %016llx work %016llx f22 %016llx\n", S.node, g.level, n.s, g.smax, n.m, g.mmax,
                    sum2d(S.lu, (size_t)nmax * sizeof(cplx), (size_t)g.smax * sizeof(cplx), g.smax), sum2d(S.lu + g.smax, (size_t)nmax * sizeof(cplx), (size_t)g.mmax * sizeof(cplx), g.smax),
                    sum2d(S.piv, (size_t)g.smax * sizeof(int), (size_t)g.smax * sizeof(int), 1),
                    sum2d(S.f21, (size_t)g.smax * sizeof(cplx), (size_t)g.smax * sizeof(cplx), g.mmax), sum2d(work, (size_t)g.mmax * sizeof(cplx), (size_t)g.mmax * sizeof(cplx), g.smax),
                    sum2d(F22, (size_t)nmax * sizeof(cplx), (size_t)g.mmax * sizeof(cplx), g.mmax));
        }
    }
    const int dbg = getenv("HELM_ND_DEBUG") ? atoi(getenv("HELM_ND_DEBUG")) : 0;       // (read per call: a test switches it on)
    if (dbg && nflag) fprintf(stderr, "[helm direct] level %d (%s, s = %d, m = %d): %d ill-conditioned front(s) re-eliminated with a pivoted LU\n", g.level, g.leaf ? "leaves" : "separators", g.smax, g.mmax, nflag);
    return check_kernels(op, "re-elimination of ill-conditioned fronts");
}

}  // namespace

// factorisation of one group (tree level x kind) on op->stream, for the nf frequencies of a set at once (nf = 1: the factorisation of rounds 1-5).
// S.f[k] / S.planes[k] / S.rtol[k]: factor, coefficient planes and tolerance of frequency k; arenaF / work: scratch of nf times the single size.
// Every strided batch below is nf times as long, batch index = front * nf + frequency (direct.hpp); the leaf level, which fills the chip by itself, and the
// re-elimination of ill-conditioned fronts go frequency by frequency.
int factor_group_set(helm_op *op, const FacSet &S, size_t gi, cplx *arenaF, cplx *work) {
    NdFactor *f0 = S.f[0];
    const int nf = S.nf;
    const NdPlan &P = f0->pd->plan;
    const NdDev *d_nodes = f0->pd->d_nodes;
    hipStream_t st = op->stream;
    const cplx one = cmake(1, 0), mone = cmake(-1, 0), zero = cmake(0, 0);
    const NdGroup &g = P.groups[gi];
    const int nmax = g.smax + g.mmax;
    const long long fs = (long long)g.mmax * nmax;              // scratch per front: [F21 | F22]
    const long long s11 = (long long)g.smax * g.smax, s12 = (long long)g.smax * g.mmax, s1 = (long long)g.smax * nmax;   // s1: stride of [F11 | F12]
    const int nbatch = nf * g.cnt;                              // fronts x frequencies
    cplx *F = arenaF + (long long)nf * g.foff;
    cplx *Finv = f0->d_fac + (long long)nf * g.finv, *G21 = f0->d_fac + (long long)nf * g.g21, *F12 = Finv + g.smax;
    NfDivScope nfdiv(nf);                                       // tile shapes / kernel variants are chosen as for ONE frequency: bit for bit the one-at-a-time factors
    // leaves of one unknown per cell and at most 8 x 8 cells: the whole leaf level in one kernel (k_leaf_factor; HELM_ND_FUSEDLEAF=0: the batched path)
    const helm_tuning tune = helm_tuning_now();
    const int fused_leaf = tune.nd_fused_leaf;
    // (a leaf takes ~0.3 ms from end to end in that kernel -- a chain of 49 elimination steps, then 2 x 49 substitution rows -- which thousands of
    // leaves in flight hide and a few hundred do not: the handful of larger leaves of a level stay on the batched path, HELM_ND_FUSEDLEAF_MIN)
    const int fused_leaf_min = tune.nd_fused_leaf_min;
    if (fused_leaf && g.leaf && g.cnt >= fused_leaf_min && P.dof == 1 && P.leaf <= 8 && g.smax <= 64 && g.mmax > 0 && g.mmax <= LEAF_MP) {
        const int leaf_dbg = getenv("HELM_LEAF_DBG") ? atoi(getenv("HELM_LEAF_DBG")) : 0;       // (timing experiments: 1 no LU, 2 no substitution, 4 no G21 / S; test: 8 every leaf re-done by the pivoted kernel)
        for (int k = 0; k < nf; ++k) {
            NdFactor *f = S.f[k];
            if (f->leafflag_elems < (size_t)g.cnt) {
                if (f->d_leafflag) { hipStreamSynchronize(st); helm_pool_free(op->device, f->d_leafflag, f->leafflag_elems * sizeof(int)); f->d_leafflag = nullptr; f->leafflag_elems = 0; }
                int maxcnt = g.cnt;
                for (const NdGroup &q : P.groups) if (q.leaf) maxcnt = std::max(maxcnt, q.cnt);
                f->d_leafflag = (int *)helm_pool_alloc(op->device, (size_t)maxcnt * sizeof(int));
                if (!f->d_leafflag) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: allocation of the leaf flags failed");
                f->leafflag_elems = (size_t)maxcnt;
            }
            for (int j0 = 0; j0 < g.cnt; j0 += 65535) {
                const int nb = std::min(65535, g.cnt - j0);
                cplx *g21b = G21 + (long long)j0 * nf * g.mmax * g.smax;
                launch_leaf_factor(st, g.smax, nb, d_nodes, g.first + j0, arenaF, f0->d_fac, g21b, S.planes[k], P.nz, P.nx, f->d_leafflag + j0, leaf_dbg, nf, k);
            }
            f->flops += (double)g.cnt * 8.0 * (2.0 * LEAF_BW * g.smax * (g.smax + g.mmax) + 3.0 * g.mmax * (g.smax + g.mmax));
        }
        return HELM_OK;
    }
    // the ring x ring block of a non-leaf front stays unbuilt: its Schur-complement product gathers the children's contributions itself
    bool schur_gather = false;
    const bool packs = P.dof <= 2 ? (nmax < (1 << 14) && P.nz < 65536 && P.nx < 32768) : (nmax < 65535 && P.nz < 4096 && P.nx < 4096 && P.dof < 128);
    if ((size_t)nmax * sizeof(int2) <= 64 * 1024 && packs)
        schur_gather = !g.leaf && g.mmax > 0 && P.dof == 1;
    if ((size_t)nmax * sizeof(int2) <= 64 * 1024 && packs) {      // (its row table must fit the 64 KB of LDS a launch gets by default: larger fronts take the unfused path)
        // rows per workgroup: whole fronts while there are thousands of them, a few rows each for the handful of big ones at the top
        const int want = std::max(1, 2048 / g.cnt);
        const int rb = std::max(std::min(nmax, 4), (nmax + want - 1) / want);
        NdPlanesSet ps; for (int k = 0; k < ND_NF_MAX; ++k) ps.p[k] = S.planes[k < nf ? k : 0];
        const int chunk = 65535 / nf;                             // fronts per launch (grid y = fronts x frequencies)
        for (int j0 = 0; j0 < g.cnt; j0 += chunk) {
            const int nb = std::min(chunk, g.cnt - j0);
            HELM_LAUNCH(k_nd_build_front<false>, dim3((nmax + rb - 1) / rb, nb * nf), dim3(256), (size_t)nmax * sizeof(int2), st, d_nodes, g.first + j0, arenaF, f0->d_fac, ps,
                               P.nz, P.nx, rb, schur_gather ? 1 : 0, NdDev(), nf, -1);
        }
    } else {
        if (nf != 1) HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "direct solver: fronts of %d unknowns take the unfused build, which factors one frequency at a time", nmax);
        const cplx *planes = S.planes[0];
        if (fs > 0) HIP_TRY(op, hipMemsetAsync(F, 0, (size_t)g.cnt * fs * sizeof(cplx), st));
        HIP_TRY(op, hipMemsetAsync(Finv, 0, (size_t)g.cnt * s1 * sizeof(cplx), st));
        for (int j0 = 0; j0 < g.cnt; j0 += 65535) {
            const int nb = std::min(65535, g.cnt - j0);
            HELM_LAUNCH(k_nd_assemble, dim3((nmax + 255) / 256, nb), dim3(256), 0, st, d_nodes, g.first + j0, arenaF, f0->d_fac, planes, P.nz, P.nx);
        }
        if (!g.leaf) {
            // children's ring sizes are bounded by this group's front size
            const size_t shm = (size_t)(2 * nmax + 8) * sizeof(int);
            for (int slot = 0; slot < 2; ++slot)
                for (int j0 = 0; j0 < g.cnt; j0 += 65535) {
                    const int nb = std::min(65535, g.cnt - j0);
                    // a chunk is at least 16 entries per map entry the block has to build, and enough chunks to fill the chip
                    const long long total = (long long)nmax * nmax;
                    const int chunk = (int)std::max<long long>(4096, std::min<long long>(total, std::max<long long>(16LL * nmax, total / std::max(1, 2048 / nb))));
                    const int gx = (int)std::max<long long>(1, std::min<long long>((total + chunk - 1) / chunk, 65535));
                    HELM_LAUNCH(k_nd_extend_add, dim3(gx, nb), dim3(256), shm, st, d_nodes, g.first + j0, slot, arenaF, f0->d_fac, P.nz, P.nx, chunk);
                }
        }
    }
    // (wider pivot windows -- 64-wide base blocks for the leaves, the upper levels or the tree top -- were measured in round 2 and do not move the
    // first-pass residual: it is not a pivoting artefact but near-resonant subdomains, which the NdStable treatment below handles)
    const int stable_smax = 128;      // larger fronts are left alone: the one-workgroup LU would cost more than the refinement pass it saves, none that large has been seen ill-conditioned
    // (leaves are not watched: 20-40 typically, below 6e3 in every operator examined, and their level is the one where two more passes over
    // every front cost something)
    const bool watch = stable_enabled(P) && !g.leaf && g.mmax > 0 && g.smax <= std::min(stable_smax, LUS_NMAX) && ensure_est(op, f0, std::min(stable_smax, LUS_NMAX)) == HELM_OK;
    double *est = watch ? f0->d_est : nullptr;
    double *est_rows = watch ? est + est_rows_off(P, nf) : nullptr;
    if (watch) for (int b0 = 0; b0 < nbatch; b0 += 65535 / nf * nf)
        HELM_LAUNCH(k_front_cond<0>, dim3(std::min(65535 / nf * nf, nbatch - b0)), dim3(256), 0, st, Finv + (long long)b0 * s1, nmax, s1, d_nodes + g.first + b0 / nf, est + b0,
                           est_rows + (long long)b0 * g.smax, g.smax, nf);
    invert(op, Finv, nmax, s1, g.smax, nbatch, work, s11, P.dof, 0);      // F11 -> F11^-1 where it stays
    if (watch) for (int b0 = 0; b0 < nbatch; b0 += 65535 / nf * nf)
        HELM_LAUNCH(k_front_cond<1>, dim3(std::min(65535 / nf * nf, nbatch - b0)), dim3(256), 0, st, Finv + (long long)b0 * s1, nmax, s1, d_nodes + g.first + b0 / nf, est + nbatch + b0,
                           est_rows + (long long)b0 * g.smax, g.smax, nf);
    if (watch) for (int k = 0; k < nf; ++k) { const int rcf = flag_group(op, S.f[k], gi, est, S.rtol[k]); if (rcf) return rcf; }      // (the lists travel while the products below run)
    if (g.mmax > 0) {
        // G21 = F21 F11^-1 ; F22 -= G21 F12
        gemm(op, g.mmax, g.smax, g.smax, one, F, nmax, fs, Finv, nmax, s1, zero, G21, g.smax, (long long)g.mmax * g.smax, nbatch);
        if (schur_gather) {
            GemmRows R; R.schur4 = 1; R.nodes = d_nodes; R.first = g.first; R.arenaS = arenaF; R.tabCi = f0->pd->d_tab + g.roff; R.tab_stride = nmax; R.nf = nf;
            gemm(op, g.mmax, g.mmax, g.smax, mone, G21, g.smax, (long long)g.mmax * g.smax, F12, nmax, s1, zero, F + g.smax, nmax, fs, nbatch, &R);
        } else
        gemm(op, g.mmax, g.mmax, g.smax, mone, G21, g.smax, (long long)g.mmax * g.smax, F12, nmax, s1, one, F + g.smax, nmax, fs, nbatch);
        if (merged_group(P, g)) {
            // leaves (and small separator fronts): F12 <- -F11^-1 F12, in place where a front is one 64-row tile (GemmRows::tm64), else through the inversion workspace
            if (g.smax <= 64) {
                GemmRows R; R.dense = 1; R.tm64 = 1;
                gemm(op, g.smax, g.mmax, g.smax, mone, Finv, nmax, s1, F12, nmax, s1, zero, F12, nmax, s1, nbatch, &R);
            } else {
                gemm(op, g.smax, g.mmax, g.smax, mone, Finv, nmax, s1, F12, nmax, s1, zero, work, g.mmax, s12, nbatch);
                HIP_TRY(op, hipMemcpy2DAsync(F12, (size_t)nmax * sizeof(cplx), work, (size_t)g.mmax * sizeof(cplx), (size_t)g.mmax * sizeof(cplx),
                                             (size_t)nbatch * g.smax, hipMemcpyDeviceToDevice, st));
            }
        }
    }
    for (int k = 0; k < nf; ++k)
        S.f[k]->flops += (double)g.cnt * 8.0 * (2.0 * g.smax * g.smax * g.smax + (double)g.smax * g.smax * g.mmax + (double)g.smax * g.mmax * g.mmax);
    if (watch) for (int k = 0; k < nf; ++k) { const int rcs = stabilise_group(op, S.f[k], gi, arenaF, work, S.planes[k], est); if (rcs) return rcs; }
    return HELM_OK;
}

// factorisation of one group for a factor on its own
int factor_group(helm_op *op, NdFactor *f, size_t gi, cplx *arenaF, cplx *work, const cplx *planes) {
    FacSet S; S.nf = 1; S.f[0] = f; S.planes[0] = planes; S.rtol[0] = op->rtol_hint;
    return factor_group_set(op, S, gi, arenaF, work);
}

int factor_prologue(helm_op *op, int block, NdFactor *f, const cplx *planes_in, const cplx **planes) {
    const NdPlan &P = f->pd->plan;
    if (!f->d_fac) {
        f->d_fac = (cplx *)helm_pool_alloc(op->device, (size_t)P.fac_elems * sizeof(cplx));
        if (!f->d_fac) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: cannot allocate %.1f GB for the factors", P.fac_elems * 16e-9);
    }
    *planes = planes_in ? planes_in : (P.dof == 2 ? op->d_C : op->d_C + (long long)block * op->nplanes * op->N);
    f->block = block; f->flops = 0;
    return HELM_OK;
}

long long nd_factor_ws_elems(const NdPlan &P) { return 2 * P.fregion + P.work_elems; }

// ws: nd_factor_ws_elems(plan) elements of scratch (fronts of two adjacent levels + inversion workspace)
int nd_factor(helm_op *op, int block, NdFactor *f, cplx *ws, const cplx *planes_in) {
    const NdPlan &P = f->pd->plan;
    const cplx *planes = nullptr;
    int rc = factor_prologue(op, block, f, planes_in, &planes);
    if (rc) return rc;
    GroupTrace tr(op->stream, "factor");
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        rc = factor_group(op, f, gi, ws, ws + 2 * P.fregion, planes);
        if (rc) return rc;
        tr.mark();
    }
    tr.report(P, false);
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    return check_kernels(op, "factorisation kernels");
}

// Launches of the factorisation on op->stream without waiting for them (helm_prefactor): the caller orders later work behind an event
int nd_factor_enqueue(helm_op *op, int block, NdFactor *f, cplx *ws, const cplx *planes_in) {
    const NdPlan &P = f->pd->plan;
    const cplx *planes = nullptr;
    int rc = factor_prologue(op, block, f, planes_in, &planes);
    if (rc) return rc;
    GroupTrace tr(op->stream, "factor");
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        rc = factor_group(op, f, gi, ws, ws + 2 * P.fregion, planes);
        if (rc) return rc;
        tr.mark();
    }
    tr.report(P, false);
    return check_kernels(op, "factorisation kernels");
}

// The factorisations of nf operators of one grid in the same launches on op's stream (helm_prefactor_many).  fs[k]: pd set, nothing else; afterwards each is a
// factor like any other (its passes address the interleaved storage through nd_fac_at / nd_fac_stride) and the shared buffer goes when the last of them is freed.
int nd_factor_enqueue_many(helm_op *op, int nf, helm_op *const *ops, NdFactor *const *fs, cplx *ws) {
    if (nf < 1 || nf > ND_NF_MAX) HELM_FAIL(op, HELM_ERR_ARG, "direct solver: %d frequencies in one factorisation (1 .. %d)", nf, ND_NF_MAX);
    const NdPlan &P = fs[0]->pd->plan;
    if (P.dof != 1) HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "direct solver: the coupled system is factored one frequency at a time");
    std::shared_ptr<NdFacShared> sh = std::make_shared<NdFacShared>();
    sh->device = op->device;
    sh->fac_bytes = (size_t)nf * (size_t)P.fac_elems * sizeof(cplx);
    sh->d_fac = (cplx *)helm_pool_alloc(op->device, sh->fac_bytes);
    if (!sh->d_fac) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: cannot allocate %.1f GB for the factors of %d frequencies", sh->fac_bytes * 1e-9, nf);
    FacSet S; S.nf = nf;
    for (int k = 0; k < nf; ++k) {
        NdFactor *f = fs[k];
        if (f->pd.get() != fs[0]->pd.get()) HELM_FAIL(op, HELM_ERR_ARG, "direct solver: the operators of one factorisation must share their grid");
        f->nf = nf; f->kf = k; f->shared = sh; f->d_fac = sh->d_fac; f->block = 0; f->flops = 0;
        S.f[k] = f; S.planes[k] = ops[k]->d_C; S.rtol[k] = ops[k]->rtol_hint;
    }
    cplx *arenaF = ws, *work = ws + (long long)nf * 2 * P.fregion;
    GroupTrace tr(op->stream, nf == 2 ? "factor x2" : (nf == 3 ? "factor x3" : "factor x4"));      // (HELM_ND_TRACE=1: per-level device time, and a wait at the end)
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        const int rc = factor_group_set(op, S, gi, arenaF, work);
        if (rc) return rc;
        tr.mark();
    }
    tr.report(P, false);
    for (int k = 1; k < nf; ++k) { fs[k]->d_est = nullptr; fs[k]->est_elems = 0; }      // (the estimates hang off frequency 0's factor and the shared record)
    return check_kernels(op, "factorisation kernels");
}
