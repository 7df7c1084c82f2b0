// Sparse direct solver for the 2-D 9-point operators: geometric nested dissection + multifrontal
// factorisation with explicit front inverses, batched level by level.
//
// What it replaces: the reference hands A to a sparse LU (`problemo.BestSolver` -> SuperLU,
// zephyr/backend/discretization.py:78-103) and re-uses the factors for every source.  This is the same
// idea laid out for the GPU: the elimination tree of a regular grid is known in closed form, every front of
// one tree level has the same (padded) shape, so each level is a handful of strided-batched dense kernels.
//
//   tree      recursive bisection of the (nz, nx) rectangle by one-cell-wide separator lines (a one-cell line
//             separates a 9-point stencil); regions with both sides <= LEAF are eliminated whole.
//   front     [separator cells | ring of cells around the subtree's region] -- the ring cells are exactly the
//             ancestors' separator cells the subtree touches.  Index <-> cell maps are closed-form (nd_local /
//             nd_cell), so no index lists are stored.
//   factor    F11^-1 (recursive 2x2 block inversion on batched GEMMs, 32x32 Gauss-Jordan base with row pivoting),
//             G21 = F21 F11^-1, Schur S = F22 - G21 F12 extend-added into the parent (child 0 then child 1:
//             deterministic).
//   solve     forward: front-local vectors travel up the tree exactly like the Schur complements;
//             backward: x_S = F11^-1 (y_S - F12 x_B), top-down.  Pure GEMMs on node-major right-hand sides
//             X[cell][rhs], no atomics, bit-reproducible.
//   accuracy  pivoting is confined to the 32x32 base blocks, so one or two steps of iterative refinement with
//             the stencil kernel (capi.hip) bring the residual to the requested tolerance.
//
// All dense arithmetic is fp64 complex on the vector ALUs.  Measured (tools/fp64_rate.hip): the fp64 MFMA and the vector FMAs share one
// throughput on MI355X (46-50 against 45-49 TFLOP/s for this instruction mix, and they do not add up when mixed), so an MFMA kernel buys
// nothing -- the one round 1 carried was 10 % slower and is gone.  The lower solve levels are bound by moving ring rows through HBM.
#include "helm_internal.hpp"
#include "direct.hpp"
#include <hip/hip_ext.h>
#include <algorithm>
#include <mutex>
#include <map>

namespace {

// ---- plan ------------------------------------------------------------------------------------------------------
struct Build {
    int nz, nx, leaf, dof;
    std::vector<NdDev> nodes;      // creation order
    std::vector<int> level;
    std::vector<int> parent;
};

void fill_geometry(NdDev &n, int nz, int nx, int dof) {
    const int h = n.z1 - n.z0, w = n.x1 - n.x0;
    n.dof = dof;
    n.s = dof * (n.cut < 0 ? h * w : (n.cut == 0 ? w : h));
    n.xlo = std::max(n.x0 - 1, 0);
    const int xhi = std::min(n.x1, nx - 1);
    const int wrow = xhi - n.xlo + 1;
    n.ntop = n.z0 > 0 ? wrow : 0;
    n.nbot = n.z1 < nz ? wrow : 0;
    n.nleft = n.x0 > 0 ? h : 0;
    n.nright = n.x1 < nx ? h : 0;
    n.m = dof * (n.ntop + n.nbot + n.nleft + n.nright);
}

int build_rec(Build &B, int z0, int z1, int x0, int x1, int lev, int parent) {
    NdDev n = NdDev();
    n.z0 = z0; n.z1 = z1; n.x0 = x0; n.x1 = x1; n.kid[0] = n.kid[1] = -1;
    const int h = z1 - z0, w = x1 - x0;
    const int me = (int)B.nodes.size();
    if (h <= B.leaf && w <= B.leaf) { n.cut = -1; n.pos = -1; }
    else if (h >= w) { n.cut = 0; n.pos = z0 + h / 2; }
    else { n.cut = 1; n.pos = x0 + w / 2; }
    fill_geometry(n, B.nz, B.nx, B.dof);
    B.nodes.push_back(n); B.level.push_back(lev); B.parent.push_back(parent);
    if (n.cut == 0) {
        int k = 0;
        if (n.pos > z0) { int c = build_rec(B, z0, n.pos, x0, x1, lev + 1, me); B.nodes[me].kid[k++] = c; }
        if (z1 > n.pos + 1) { int c = build_rec(B, n.pos + 1, z1, x0, x1, lev + 1, me); B.nodes[me].kid[k++] = c; }
    } else if (n.cut == 1) {
        int k = 0;
        if (n.pos > x0) { int c = build_rec(B, z0, z1, x0, n.pos, lev + 1, me); B.nodes[me].kid[k++] = c; }
        if (x1 > n.pos + 1) { int c = build_rec(B, z0, z1, n.pos + 1, x1, lev + 1, me); B.nodes[me].kid[k++] = c; }
    }
    return me;
}

}  // namespace

int nd_build_plan(NdPlan &P, int nz, int nx, int leaf, int dof) {
    Build B; B.nz = nz; B.nx = nx; B.leaf = std::max(2, leaf); B.dof = dof;
    build_rec(B, 0, nz, 0, nx, 0, -1);
    const int nn = (int)B.nodes.size();
    int maxlev = 0;
    for (int l : B.level) maxlev = std::max(maxlev, l);
    // processing order: deepest level first; within a level the leaves, then the separators.  Leaves come in two size classes per level:
    // on a 2^k grid with one-cell separators all but one leaf interval per axis are 7 cells long (1024 = 127 x 7 + 8 + 127 separators),
    // so 98 % of the leaves have 49 unknowns and a single padded shape of 64 would waste a quarter of the leaf level's inner dimension
    std::vector<int> order; order.reserve(nn);
    P.groups.clear();
    for (int lev = maxlev; lev >= 0; --lev) {
        // most frequent leaf size of this level
        int smode = 0;
        {
            std::vector<int> hist;
            for (int i = 0; i < nn; ++i)
                if (B.level[i] == lev && B.nodes[i].cut < 0) { if ((int)hist.size() <= B.nodes[i].s) hist.resize(B.nodes[i].s + 1, 0); hist[B.nodes[i].s] += 1; }
            for (int v = 0; v < (int)hist.size(); ++v) if (hist[v] > (smode < (int)hist.size() ? hist[smode] : 0)) smode = v;
        }
        for (int kind = 0; kind < 3; ++kind) {      // 0: leaves up to the usual size, 1: larger leaves, 2: separators
            NdGroup g = NdGroup(); g.first = (int)order.size(); g.level = lev; g.leaf = kind < 2;
            for (int i = 0; i < nn; ++i) {
                if (B.level[i] != lev) continue;
                const bool isleaf = B.nodes[i].cut < 0;
                const int k = !isleaf ? 2 : (B.nodes[i].s <= smode ? 0 : 1);
                if (k != kind) continue;
                order.push_back(i);
                g.smax = std::max(g.smax, B.nodes[i].s); g.mmax = std::max(g.mmax, B.nodes[i].m);
            }
            g.cnt = (int)order.size() - g.first;
            if (g.cnt > 0) P.groups.push_back(g);
        }
    }
    std::vector<int> newidx(nn);
    for (int k = 0; k < nn; ++k) newidx[order[k]] = k;
    P.nodes.resize(nn);
    for (int k = 0; k < nn; ++k) {
        NdDev n = B.nodes[order[k]];
        for (int c = 0; c < 2; ++c) if (n.kid[c] >= 0) n.kid[c] = newidx[n.kid[c]];
        P.nodes[k] = n;
    }
    P.nz = nz; P.nx = nx; P.leaf = B.leaf; P.nlevels = maxlev + 1; P.total_rows = 0; P.dof = dof;
    // arenas: fronts (factor) and front vectors (solve) of level L live in region L % 2
    std::vector<long long> lev_f(maxlev + 1, 0), lev_v(maxlev + 1, 0);
    long long fac = 0;
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        NdGroup &g = P.groups[gi];
        const long long nmax = g.smax + g.mmax;
        g.foff = lev_f[g.level]; g.voff = lev_v[g.level];
        g.roff = P.total_rows; P.total_rows += (long long)g.cnt * nmax;
        lev_f[g.level] += (long long)g.cnt * g.mmax * nmax;
        lev_v[g.level] += (long long)g.cnt * nmax;
        g.finv = fac; g.f12 = fac + g.smax; fac += (long long)g.cnt * g.smax * nmax;       // [F11^-1 | F12] rows of nmax per front
        g.g21 = fac; fac += (long long)g.cnt * g.mmax * g.smax;
    }
    P.fac_elems = fac;
    P.fregion = 0; P.vregion = 0; P.work_elems = 0;
    for (int l = 0; l <= maxlev; ++l) { P.fregion = std::max(P.fregion, lev_f[l]); P.vregion = std::max(P.vregion, lev_v[l]); }
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        NdGroup &g = P.groups[gi];
        g.foff += (long long)(g.level & 1) * P.fregion;
        g.voff += (long long)(g.level & 1) * P.vregion;
        // (the inversion's scratch; a leaf level with more than 64 unknowns per front also forms -F11^-1 F12 there: smax x mmax per front)
        P.work_elems = std::max(P.work_elems, (long long)g.cnt * g.smax * (g.leaf ? std::max(g.smax, g.mmax) : g.smax));
        for (int j = 0; j < g.cnt; ++j) {
            NdDev &n = P.nodes[g.first + j];
            const long long nmax = g.smax + g.mmax;
            n.smax = g.smax; n.mmax = g.mmax;
            n.foff = g.foff + (long long)j * g.mmax * nmax;
            n.finv_off = g.finv + (long long)j * g.smax * nmax;
            n.f12_off = n.finv_off + g.smax;
            n.voff = g.voff + (long long)j * nmax;
            n.roff = g.roff + (long long)j * nmax;
        }
    }
    return HELM_OK;
}

// ---- kernels ---------------------------------------------------------------------------------------------------
namespace {

// The front [[F11, F12], [F21, F22]] is assembled where each block is needed afterwards: [F11 | F12] side by side in the factor
// storage (rows of smax + mmax; F11 is inverted in place, F12 is kept -- for leaves it becomes -F11^-1 F12 after the Schur complement, so
// that the back substitution of a leaf is ONE product [F11^-1 | -F11^-1 F12] [y_S; x_B]), [F21 | F22] in the scratch arena (F21 feeds G21, F22 becomes the Schur
// complement the parent picks up).  (r, c): padded front coordinates.
__device__ __forceinline__ cplx *front_entry(const NdDev &n, cplx *arenaF, cplx *fac, int r, int c) {
    if (r < n.smax) {
        return fac + n.finv_off + (long long)r * (n.smax + n.mmax) + c;        // [F11 | F12] share their rows
    }
    return arenaF + n.foff + (long long)(r - n.smax) * (n.smax + n.mmax) + c;
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
__global__ __launch_bounds__(256) void k_nd_build_front(const NdDev *nodes, int first, cplx *arenaF, cplx *fac, const cplx *planes, int nz, int nx, int rb,
                                                        const NdDev *ovr = nullptr, int skip22 = 0) {
    extern __shared__ int2 finfo[];        // per padded row: x = z | x << 16 (-1: padding), y = (k0 + 1) | (k1 + 1) << 14 | comp << 28
    const NdDev n = ovr ? *ovr : nodes[first + blockIdx.y];      // (ovr: one front rebuilt with its [F11 | F12] rows redirected, see NdStable)
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
    const cplx *S0 = h0 ? arenaF + c0.foff + c0.smax : nullptr, *S1 = h1 ? arenaF + c1.foff + c1.smax : nullptr;
    const int ld0 = c0.smax + c0.mmax, ld1 = c1.smax + c1.mmax;
    const int r1 = r0 + rb < nmax ? r0 + rb : nmax;
    const int tx = tid & 63, ty = tid >> 6;
    for (int r = r0 + ty; r < r1; r += 4) {
        const int2 ia = finfo[r];
        const bool colmode = n.dof > 2;
        const int za = colmode ? (ia.x & 0xfff) : (ia.x & 0xffff), xa = colmode ? ((ia.x >> 12) & 0xfff) : (ia.x >> 16);
        const int ca = colmode ? ((ia.x >> 24) & 0x7f) : ((ia.y >> 28) & 1);
        const int a0 = (colmode ? (ia.y & 0xffff) : (ia.y & 0x3fff)) - 1, a1 = (colmode ? ((ia.y >> 16) & 0xffff) : ((ia.y >> 14) & 0x3fff)) - 1;
        // skip22: the ring x ring block (the sum of the children's Schur complements, most of a front below the tree top) is not
        // materialised -- the Schur-complement product gathers it itself (k_zgemm2<.., 4, ..>) and writes S where F22 would have been
        const int cend = (skip22 && r >= n.smax) ? n.smax : nmax;
        for (int c = tx; c < cend; c += 64) {
            const int2 ib = finfo[c];
            cplx v = cmake(0.0, 0.0);
            if (ia.x < 0 || ib.x < 0) { if (r == c && r < n.smax) v = cmake(1.0, 0.0); }
            else {
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
            }
            *front_entry(n, arenaF, fac, r, c) = v;
        }
    }
}

// ---- strided-batched complex GEMM: C = beta C + alpha A B, row-major ----------------------------------------------
// 64x64 tile, K step 8, 256 threads each owning a 4x4 block; the next K slab is fetched into registers while the
// current one is multiplied out of LDS.
//
// IDX variant (solve phase): rows of B, of the C that is read (beta != 0) and of the C that is written may be taken
// through the plan's row table instead of a dense front buffer, i.e. straight from / to the node-major right-hand sides
// Xt[cell][rhs]:   row r of batch item z  ->  X + tab[z * tab_stride + off + r].x * ldx   (negative: a zero row / not stored).
// This removes the gather / scatter passes (and their HBM round trips) from the lower tree levels.
// arguments of the Gauss-Jordan sweep that rides along with a blocked-inversion update (k_zgemm2_la)
struct GjPivotArgs { const cplx *T0; int ld; long long stride; int n, k0, nb; const cplx *Wc0, *Wr0; long long wstride; cplx *Pb0; long long pstride; int batch; };
struct GemmRows {
    const int4 *tabB = nullptr, *tabCi = nullptr, *tabCo = nullptr;
    int offB = 0, offCi = 0, offCo = 0, tab_stride = 0;
    const cplx *Bx = nullptr, *Cix = nullptr; cplx *Cox = nullptr;
    int ldx = 0;
    int z0 = 0;           // batch index of blockIdx.z == 0 (launches are chunked along z)
    int fwd3 = 0;         // forward-gather mode (k_zgemm2<.., 2, ..>): Bx = right-hand sides, Cix = front-vector arena, Cox = where y_S goes
    int zr0 = 0, zr1 = 0, zc0 = 0, zc1 = 0;   // rows [zr0, zr1) and columns [zc0, zc1) of C are taken as zero on input (beta masked): blocked Gauss-Jordan
    int sk0 = 0, sk1 = 0;                     // the diagonal block [sk0, sk1)^2 of C is neither read nor written (the next pivot block, owned by k_gj_pivot)
    int ksplit = 0, kc = 0; long long pstride = 0;   // split over the inner dimension (dense operands only): z = matrix * ksplit + chunk; chunk c multiplies columns
                                              // [c kc, (c + 1) kc) of A into its own partial product at C0 + c pstride + matrix sc (k_splitk_reduce adds them up)
    int dense = 0;                            // only the masks above are in use: launch the plain (un-indexed) kernel
    const cplx *Bx2 = nullptr; int k2 = 0;    // rows k < k2 of an indexed B come from Bx2 instead of Bx (a leaf's y_S is still in the right-hand sides)
    int tm64 = 0;                             // one 64-row tile per matrix (M <= 64): C may then overwrite B (every workgroup has read all of its B columns
                                              // before it stores, and no other workgroup reads them)
    const GjPivotArgs *la = nullptr;          // (host pointer) fuse this pivot sweep into the launch: 64 x 32 tiles, one extra z-slice
    // Schur-complement mode (k_zgemm2<.., 4, ..>): C = (children's Schur complement entries that land on (r, c), gathered through the row
    // table) + alpha A B -- the ring x ring block of a front is never written by the build pass and never read back here
    int schur4 = 0;
    const NdDev *nodes = nullptr; int first = 0;
    const cplx *arenaS = nullptr;
    // forward pass on sparse right-hand sides (sources of a survey touch a handful of cells): act[front * nct + column / 64] != 0 when that front's
    // outgoing rows were computed for that block of 64 columns -- a front whose own right-hand-side rows and whose children's rows are all zero
    // there has nothing to add, writes zeros for its y_S rows and leaves its ring rows unwritten (its parent reads the flag, not the rows)
    int *act = nullptr; int nct = 0;
    // back substitution of the leaves, same flags read-only: where a leaf's flag is 0 its first k2 rows of B (its right-hand-side rows y_S) are all
    // zero in that block of columns, and the product starts at row k2 -- x_S = G x_B, 32 of the 81 columns of [F11^-1 | G]
    const int *act_ro = nullptr;
    int hint = 0;               // (IDX 1 with act, leaf forward elimination) the flags were set from a declared support of the right-hand sides: a front without one is left
                                // before it reads a byte
    int ntc = 0;                // store C with nontemporal stores (large HBM-bound launches whose output is not read again soon)
    int xcd_map = 0;            // regroup the workgroup ids so that the column tiles of a front share an XCD (zgemm3_body)
    int child_rows = 0;         // (fwd3, host-side bookkeeping) ring rows of a front's two children: what the gather has to read besides q_S
};
#define GB_K 8
#define GB_KIDX 512       // largest K with indexed B rows
// next K slab of the A and B tiles into registers (shared by both GEMM kernels)
template <int TM, int TN, int NA, int NB>
__device__ __forceinline__ void zg_fetch(cplx (&ra)[NA], cplx (&rb)[NB], int tid, int k0, int m0, int n0, int M, int Nn, int K,
                                         const cplx *A, int lda, const cplx *B, int ldb, bool idxB, const int *kidx, const cplx *Bx, int ldx) {
    #pragma unroll
    for (int e = 0; e < NA; ++e) {
        const int idx = tid + e * 256;
        const int ar = idx >> 3, ak = idx & 7;
        cplx v = cmake(0.0, 0.0);
        if (idx < TM * GB_K && m0 + ar < M && k0 + ak < K) v = A[(long long)(m0 + ar) * lda + k0 + ak];
        ra[e] = v;
    }
    #pragma unroll
    for (int e = 0; e < NB; ++e) {
        const int idx = tid + e * 256;
        const int bk = idx / TN, bc = idx % TN;
        cplx v = cmake(0.0, 0.0);
        if (idx < TN * GB_K && k0 + bk < K && n0 + bc < Nn) {
            if (idxB) { const int r = kidx[k0 + bk]; if (r >= 0) v = Bx[(long long)r * ldx + n0 + bc]; }
            else v = B[(long long)(k0 + bk) * ldb + n0 + bc];
        }
        rb[e] = v;
    }
}
#define ZG_FETCH(k0_) zg_fetch<TM, TN, NA, NB>(ra, rb, tid, (k0_), m0, n0, M, Nn, K, A, lda, B, ldb, idxB, kidx, R.Bx, R.ldx);

// TM x TN tile with TM * TN = 4096: 64 x 64, 32 x 128 or 16 x 256 -- the fronts of the lower tree levels have 36, 54, 72 ...
// rows against 256 right-hand sides, and a 64-row tile would spend up to 44 % of its flops on padding
template <int TM, bool IDX, int RN = 4>
__global__ __launch_bounds__(256) void k_zgemm(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                               const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, GemmRows R) {
    constexpr int TXN = 1024 / TM, TN = TXN * RN;      // 256 threads = (TM / 4) x TXN, each owning 4 x RN outputs
    constexpr int NA = (TM * GB_K + 255) / 256, NB = TN * GB_K / 256;
    __shared__ cplx As[GB_K][TM + 1];
    __shared__ cplx Bs[GB_K][TN];
    __shared__ int kidx[IDX ? GB_KIDX : 1];
    const cplx *A = A0 + (long long)blockIdx.z * sa;
    const cplx *B = B0 + (long long)blockIdx.z * sb;
    cplx *C = C0 + (long long)blockIdx.z * sc;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int tid = threadIdx.x, ty = tid / TXN, tx = tid % TXN;
    const long long trow = IDX ? (long long)(R.z0 + blockIdx.z) * R.tab_stride : 0;
    const bool idxB = IDX && R.tabB != nullptr;
    if (idxB) {
        for (int k = tid; k < K; k += 256) kidx[k] = R.tabB[trow + R.offB + k].x;
        __syncthreads();
    }
    cplx acc[4][RN];
    #pragma unroll
    for (int i = 0; i < 4; ++i)
        #pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = cmake(0.0, 0.0);
    cplx ra[NA], rb[NB];
    ZG_FETCH(0)
    for (int k0 = 0; k0 < K; k0 += GB_K) {
        #pragma unroll
        for (int e = 0; e < NA; ++e) { const int idx = tid + e * 256; if (idx < TM * GB_K) As[idx & 7][idx >> 3] = ra[e]; }
        #pragma unroll
        for (int e = 0; e < NB; ++e) { const int idx = tid + e * 256; Bs[idx / TN][idx % TN] = rb[e]; }
        __syncthreads();
        if (k0 + GB_K < K) { ZG_FETCH(k0 + GB_K) }
        #pragma unroll
        for (int k = 0; k < GB_K; ++k) {
            cplx a[4], b[RN];
            #pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
            #pragma unroll
            for (int j = 0; j < RN; ++j) b[j] = Bs[k][tx * RN + j];
            #pragma unroll
            for (int i = 0; i < 4; ++i)
                #pragma unroll
                for (int j = 0; j < RN; ++j) cfma(acc[i][j], a[i], b[j]);
        }
        __syncthreads();
    }
    const bool b0 = (beta.x == 0.0 && beta.y == 0.0);
    #pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = m0 + ty * 4 + i;
        if (r >= M) continue;
        cplx *dst = C + (long long)r * ldc;
        const cplx *cin = dst;
        if (IDX && R.tabCo) {
            const int ix = R.tabCo[trow + R.offCo + r].x;
            if (ix < 0) continue;
            dst = R.Cox + (long long)ix * R.ldx;
        }
        if (IDX && R.tabCi && !b0) {
            const int ix = R.tabCi[trow + R.offCi + r].x;
            cin = ix >= 0 ? R.Cix + (long long)ix * R.ldx : nullptr;
        }
        #pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int cc = n0 + tx * RN + j;
            if (cc >= Nn) continue;
            cplx v = cmul(alpha, acc[i][j]);
            if (!b0 && cin) v = cadd(v, cmul(beta, cin[cc]));
            dst[cc] = v;
        }
    }
}

template <int TM, int RN = 4>
void launch_vec(hipStream_t st, int idx, int nb, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                cplx beta, cplx *C, int ldc, long long sc, const GemmRows &R) {
    constexpr int TN = (1024 / TM) * RN;
    dim3 grid((Nn + TN - 1) / TN, (M + TM - 1) / TM, nb);
    if (idx) hipLaunchKernelGGL((k_zgemm<TM, true, RN>), grid, dim3(256), 0, st, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else hipLaunchKernelGGL((k_zgemm<TM, false, RN>), grid, dim3(256), 0, st, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
}

// ---- second-generation tile kernel ------------------------------------------------------------------------------------
// Same tile shapes and the same IDX addressing as k_zgemm, with the two LDS problems of that kernel removed:
//   * a thread's RN output columns are interleaved (column j * TXN + tx instead of tx * RN + j): the 16-byte Bs reads of
//     consecutive lanes are consecutive in LDS (conflict-free: 16 lanes x 16 B = all 64 banks once), where the blocked
//     layout put lanes 64 B apart -- a 4-way bank conflict on every B operand read, which is what held the big launches at
//     45 TFLOP/s; global stores of C become 256-byte contiguous runs per wave row as well;
//   * the LDS tiles are double-buffered: the next K slab goes from registers into the other buffer while the current one is
//     multiplied, ONE barrier per slab instead of two.
// IDX: 0 dense operands; 1 rows of B / C through the row table (GemmRows); 2 "forward gather": a row of B is the SUM the forward pass
// needs -- the right-hand side of a separator cell plus the children's outgoing rows that land on it (table entries x / y, z) -- and
// the C that is read is the sum of the children's rows of a ring row; the first row-tile also stores the gathered separator rows
// (y_S) where the back substitution expects them.  Same additions in the same order as k_nd_fwd_rows + the dense GEMM.
template <int TM, int IDX, int RN, int KS, int UNR>
__device__ __forceinline__ void zgemm2_body(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                            const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, const GemmRows &R,
                                            cplx *lds_a, cplx *lds_b) {
    constexpr int TXN = 1024 / TM, TN = TXN * RN;
    constexpr int NA = (TM * KS + 255) / 256, NB = (TN * KS + 255) / 256;
    cplx (&As)[2][KS][TM + 1] = *reinterpret_cast<cplx (*)[2][KS][TM + 1]>(lds_a);      // the two operand tiles live in the caller's LDS
    cplx (&Bs)[2][KS][TN] = *reinterpret_cast<cplx (*)[2][KS][TN]>(lds_b);
    __shared__ int kidx[IDX == 1 ? GB_KIDX : 1];
    __shared__ int4 kidx4[IDX == 2 ? GB_KIDX : 1];
    int zb = blockIdx.z;
    if (IDX == 0 && R.ksplit > 1) {                      // this workgroup's share of the inner dimension
        const int kch = zb % R.ksplit;
        zb /= R.ksplit;
        const int kbeg = kch * R.kc;
        A0 += kbeg; B0 += (long long)kbeg * ldb; C0 += (long long)kch * R.pstride;
        K = K - kbeg < R.kc ? (K - kbeg > 0 ? K - kbeg : 0) : R.kc;
    }
    const cplx *A = A0 + (long long)zb * sa;
    const cplx *B = B0 + (long long)zb * sb;
    cplx *C = C0 + (long long)zb * sc;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int tid = threadIdx.x, ty = tid / TXN, tx = tid % TXN;
    const long long trow = IDX ? (long long)(R.z0 + blockIdx.z) * R.tab_stride : 0;
    const bool idxB = IDX == 1 && R.tabB != nullptr;
    if (idxB) {
        for (int k = tid; k < K; k += 256) kidx[k] = R.tabB[trow + R.offB + k].x;
        __syncthreads();
    }
    if (IDX == 2) {
        for (int k = tid; k < K; k += 256) kidx4[k] = R.tabB[trow + R.offB + k];
        __syncthreads();
    }
    // IDX == 4: child-local ring indices (child 0, child 1; -1: not on that child's ring) of the tile's rows and columns
    __shared__ int2 sgr[IDX == 4 ? TM : 1], sgc[IDX == 4 ? TN : 1];
    const cplx *S0 = nullptr, *S1 = nullptr;
    int ld0 = 0, ld1 = 0;
    if (IDX == 4) {
        const NdDev nd = R.nodes[R.first + R.z0 + blockIdx.z];
        int base0 = 0, base1 = 0;
        if (nd.kid[0] >= 0) { const NdDev c0 = R.nodes[nd.kid[0]]; S0 = R.arenaS + c0.foff + c0.smax; ld0 = c0.smax + c0.mmax; base0 = (int)(c0.voff + c0.smax); }
        if (nd.kid[1] >= 0) { const NdDev c1 = R.nodes[nd.kid[1]]; S1 = R.arenaS + c1.foff + c1.smax; ld1 = c1.smax + c1.mmax; base1 = (int)(c1.voff + c1.smax); }
        for (int t = tid; t < TM + TN; t += 256) {
            const int q = t < TM ? m0 + t : n0 + (t - TM);           // row / column of the ring block
            int2 e = make_int2(-1, -1);
            if (q < (t < TM ? M : Nn)) {
                const int4 t4 = R.tabCi[trow + nd.smax + q];
                if (t4.y >= 0 && S0) e.x = t4.y - base0;
                if (t4.z >= 0 && S1) e.y = t4.z - base1;
            }
            if (t < TM) sgr[t] = e; else sgc[t - TM] = e;
        }
        __syncthreads();
    }
    cplx acc[4][RN];
    #pragma unroll
    for (int i = 0; i < 4; ++i)
        #pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = cmake(0.0, 0.0);
    cplx ra[NA], rb[NB];
    auto fetch = [&](int k0) {
        #pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int idx = tid + e * 256;
            const int ar = idx / KS, ak = idx % KS;
            cplx v = cmake(0.0, 0.0);
            if (idx < TM * KS && m0 + ar < M && k0 + ak < K) v = A[(long long)(m0 + ar) * lda + k0 + ak];
            ra[e] = v;
        }
        #pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int idx = tid + e * 256;
            const int bk = idx / TN, bc = idx % TN;
            cplx v = cmake(0.0, 0.0);
            if (idx < TN * KS && k0 + bk < K && n0 + bc < Nn) {
                if (IDX == 2) {
                    const int4 t4 = kidx4[k0 + bk];
                    if (t4.w) v = R.Bx[(long long)t4.x * R.ldx + n0 + bc];
                    if (t4.y >= 0) v = cadd(v, R.Cix[(long long)t4.y * R.ldx + n0 + bc]);
                    if (t4.z >= 0) v = cadd(v, R.Cix[(long long)t4.z * R.ldx + n0 + bc]);
                    if (blockIdx.y == 0 && t4.w && R.Cox) R.Cox[(long long)t4.x * R.ldx + n0 + bc] = v;      // y_S
                }
                else if (idxB) { const int r = kidx[k0 + bk]; if (r >= 0) v = (k0 + bk < R.k2 ? R.Bx2 : R.Bx)[(long long)r * R.ldx + n0 + bc]; }
                else v = B[(long long)(k0 + bk) * ldb + n0 + bc];
            }
            rb[e] = v;
        }
    };
    auto stash = [&](int buf) {
        #pragma unroll
        for (int e = 0; e < NA; ++e) { const int idx = tid + e * 256; if (idx < TM * KS) As[buf][idx % KS][idx / KS] = ra[e]; }
        #pragma unroll
        for (int e = 0; e < NB; ++e) { const int idx = tid + e * 256; if (idx < TN * KS) Bs[buf][idx / TN][idx % TN] = rb[e]; }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += KS) {
        const bool more = k0 + KS < K;
        if (more) fetch(k0 + KS);
        #pragma unroll UNR
        for (int k = 0; k < KS; ++k) {
            cplx a[4], b[RN];
            #pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[cur][k][ty * 4 + i];
            #pragma unroll
            for (int j = 0; j < RN; ++j) b[j] = Bs[cur][k][j * TXN + tx];
            #pragma unroll
            for (int i = 0; i < 4; ++i)
                #pragma unroll
                for (int j = 0; j < RN; ++j) cfma(acc[i][j], a[i], b[j]);
        }
        if (more) { stash(cur ^ 1); __syncthreads(); cur ^= 1; }
    }
    const bool b0 = (beta.x == 0.0 && beta.y == 0.0);
    #pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = m0 + ty * 4 + i;
        if (r >= M) continue;
        cplx *dst = C + (long long)r * ldc;
        const cplx *cin = dst;
        const cplx *cin2 = nullptr;
        if (IDX == 2) {
            const int4 t4 = R.tabCi[trow + R.offCi + r];
            cin = t4.y >= 0 ? R.Cix + (long long)t4.y * R.ldx : nullptr;
            cin2 = t4.z >= 0 ? R.Cix + (long long)t4.z * R.ldx : nullptr;
        }
        if (IDX == 1 && R.tabCo) {
            const int ix = R.tabCo[trow + R.offCo + r].x;
            if (ix < 0) continue;
            dst = R.Cox + (long long)ix * R.ldx;
        }
        if (IDX == 1 && R.tabCi && !b0) {
            const int ix = R.tabCi[trow + R.offCi + r].x;
            cin = ix >= 0 ? R.Cix + (long long)ix * R.ldx : nullptr;
        }
        const bool zrow = r >= R.zr0 && r < R.zr1;
        const bool srow = r >= R.sk0 && r < R.sk1;
        #pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int cc = n0 + j * TXN + tx;
            if (cc >= Nn) continue;
            if (srow && cc >= R.sk0 && cc < R.sk1) continue;
            cplx v = cmul(alpha, acc[i][j]);
            if (IDX == 4) {
                const int2 er = sgr[r - m0], ec = sgc[cc - n0];
                cplx c = cmake(0.0, 0.0);
                if (er.x >= 0 && ec.x >= 0) c = S0[(long long)er.x * ld0 + ec.x];
                if (er.y >= 0 && ec.y >= 0) c = cadd(c, S1[(long long)er.y * ld1 + ec.y]);
                v = cadd(c, v);
            }
            else if (IDX == 2) {
                cplx c = cin ? cin[cc] : cmake(0.0, 0.0);
                if (cin2) c = cadd(c, cin2[cc]);
                v = cadd(v, cmul(beta, c));
            }
            else if (!b0 && cin && !zrow && !(cc >= R.zc0 && cc < R.zc1)) v = cadd(v, cmul(beta, cin[cc]));
            dst[cc] = v;
        }
    }
}

template <int TM, int IDX, int RN, int KS, int UNR, int OCC>
__global__ __launch_bounds__(256, OCC) void k_zgemm2(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                                const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, GemmRows R) {
    __shared__ cplx As[2][KS][TM + 1];
    __shared__ cplx Bs[2][KS][1024 / TM * RN];
    zgemm2_body<TM, IDX, RN, KS, UNR>(M, Nn, K, alpha, A0, lda, sa, B0, ldb, sb, beta, C0, ldc, sc, R, &As[0][0][0], &Bs[0][0][0]);
}

// Per-launch timing without extra packets: when gemm() has armed a pair of events, the dispatch itself carries them
// (hipExtLaunchKernelGGL: start / stop timestamps of that kernel), instead of two hipEventRecord markers around it.
thread_local hipEvent_t tl_ev0 = nullptr, tl_ev1 = nullptr;
#define ZG_LAUNCH(KERNEL, GRID, ...) do { if (tl_ev0) hipExtLaunchKernelGGL(KERNEL, GRID, dim3(256), 0, st, tl_ev0, tl_ev1, 0, __VA_ARGS__); \
                                          else hipLaunchKernelGGL(KERNEL, GRID, dim3(256), 0, st, __VA_ARGS__); } while (0)

template <int TM, int RN, int KS, int UNR = 1, int OCC = 1>
void launch_vec2(hipStream_t st, int idx, int nb, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                 cplx beta, cplx *C, int ldc, long long sc, const GemmRows &R) {
    constexpr int TN = (1024 / TM) * RN;
    dim3 grid((Nn + TN - 1) / TN, (M + TM - 1) / TM, nb);
    if (idx == 4) ZG_LAUNCH((k_zgemm2<TM, 4, RN, KS, UNR, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else if (idx == 2) ZG_LAUNCH((k_zgemm2<TM, 2, RN, KS, UNR, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else if (idx) ZG_LAUNCH((k_zgemm2<TM, 1, RN, KS, UNR, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else ZG_LAUNCH((k_zgemm2<TM, 0, RN, KS, UNR, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
}

// ---- third-generation tile kernel: the same products on the matrix cores (v_mfma_f64_16x16x4_f64) -----------------------
// Round 4.  tools/fp64_clock.hip with the occupancy pinned (profiles/r04_fp64_clock_probe.txt) settled what the fp64 units give: the
// vector FMAs of the 4 x 4 complex register block saturate at 55-56 TFLOP/s at 2, 4 and 8 waves per SIMD (the chip holds 2.04-2.09 GHz
// under that load), whereas v_mfma_f64_16x16x4_f64 runs at 77.8 TFLOP/s = 99 % of nominal from two waves per SIMD up at 2.38 GHz, 73.4
// in the complex product's own mix of 16 MFMAs on 8 rotating operand registers.  (Rounds 2 and 3 measured 35-47 for the MFMA and
// concluded it could not win: that was one wave per SIMD under __launch_bounds__(256), where hipcc parks the accumulators in AGPRs
// and copies them around every instruction.)  One MFMA replaces 16 vector FMAs per lane-pair of operands, so the inner loop issues
// (MT + NT) 16-byte LDS reads for 4 MT NT matrix instructions of 64 cycles each: LDS and VALU issue are out of the picture.
//   tile      256 threads = WM x WN waves; a wave owns MT x NT blocks of 16 x 16 outputs, real and imaginary accumulators apart
//             (C = (Ar Br - Ai Bi) + i (Ar Bi + Ai Br): four real MFMAs per block and k step of 4, -Ai formed once per fragment)
//   LDS       fragment-ordered: every (k group of 4, 16-row block of A | 16-column block of B) is one 1-KB run that a wave reads with
//             lane l at offset 16 l -- conflict-free by construction -- A[row l & 15][k l >> 4], B[k l >> 4][col l & 15].  The B
//             fragment IS a row-major 4 x 16 piece of B.  A arrives k-contiguous (8 lanes = 8 k of one row): its slot inside the
//             fragment is XOR-swizzled with (k & 3) ^ 4 (k / 4 & 1) so that those 8 lanes hit 8 different 16-byte bank groups on the
//             store and the 16-lane groups of ds_read_b128 still cover all 64 banks on the load.
//   C / D     lane l holds rows (l >> 4) + 4 q, q = 0..3, of column l & 15: a store is four 256-byte row segments.
// Addressing modes (IDX), masks, split-K and the fused gathers are those of zgemm2_body, operand for operand.
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));
__device__ cplx g_zero_page[4];       // 64 bytes of zeros: what masked lanes load instead of branching around a load
// XR = 1 (WM == 1 only): the tile has ONE more row than its 16 MT rows of matrix-core blocks -- row 16 MT goes through the vector ALUs, which the
// MFMA loop leaves idle (thread = column x share of the k range, partial sums added up through LDS at the end).  98 % of the leaves of a 2^k grid
// have 49 = 3 x 16 + 1 unknowns: a fourth block of 16 rows for the 49th would spend a quarter of the leaf level's matrix instructions on padding.
template <int WM, int WN, int MT, int NT, int IDX, int KS, int XR = 0>
__device__ __forceinline__ void zgemm3_body(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                            const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, const GemmRows &R,
                                            cplx *lds) {
    static_assert(WM * WN == 4 && KS % 4 == 0, "four waves per workgroup, k groups of 4");
    static_assert(!XR || (WM == 1 && (IDX == 0 || IDX == 1)), "the extra row needs all four waves side by side and plain / row-table addressing");
    constexpr int TMM = 16 * MT * WM;                                   // rows on the matrix cores
    constexpr int TM = TMM + (XR ? 1 : 0), TN = 16 * NT * WN;           // rows / columns of C per workgroup
    constexpr int FA = TMM / 16 + (XR ? 1 : 0), FB = TN / 16, KG = KS / 4;
    constexpr int NA = (FA * 16 * KS + 255) / 256, NB = (TN * KS + 255) / 256;
    constexpr int XP = 256 / TN;                                        // (XR) shares of the k range
    constexpr int ABUF = KG * FA * 64, BBUF = KG * FB * 64;           // elements per buffer
    cplx *As = lds, *Bs = lds + 2 * ABUF;
    __shared__ int kidx[IDX == 1 ? GB_KIDX : 1];
    __shared__ int4 kidx4[IDX == 2 ? GB_KIDX : 1];
    // Which (front, column tile) this workgroup takes.  Workgroups are dealt round-robin over the eight XCDs in launch order (x fastest), so the
    // column tiles of one front -- which all read the same A operand, the front's factors -- would land on different XCDs with different L2s and the
    // factors would cross the fabric once per tile.  With one row tile per front the ids are regrouped in blocks of eight fronts: ids L and L + 8 are
    // the same front's neighbouring column tiles, i.e. the same XCD (speed only: nothing depends on where a workgroup runs).
    int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
    if (gridDim.x * gridDim.y > 1 && gridDim.z >= 16 && R.la == nullptr && !(IDX == 0 && R.ksplit > 1) && R.xcd_map && (gridDim.y == 1 || R.xcd_map > 1)) {
        // (xcd_map > 1: fronts with several row tiles too -- every row tile repeats the gather of the B rows, which then comes out of that XCD's L2)
        const int nxt = gridDim.x, nt = gridDim.x * gridDim.y, nbz = gridDim.z;
        const int L = blockIdx.x + nxt * blockIdx.y + nt * blockIdx.z;
        const int full = (nbz / 8) * 8 * nt;                          // ids covered by whole blocks of eight fronts
        int tile;
        if (L < full) { const int grp = L / (8 * nt), w = L % (8 * nt); bzi = grp * 8 + (w & 7); tile = w >> 3; }
        else { bzi = (nbz / 8) * 8 + (L - full) / nt; tile = (L - full) % nt; }
        bxi = tile % nxt; byi = tile / nxt;
    }
    int zb = bzi;
    if (IDX == 0 && R.ksplit > 1) {                      // this workgroup's share of the inner dimension
        const int kch = zb % R.ksplit;
        zb /= R.ksplit;
        const int kbeg = kch * R.kc;
        A0 += kbeg; B0 += (long long)kbeg * ldb; C0 += (long long)kch * R.pstride;
        K = K - kbeg < R.kc ? (K - kbeg > 0 ? K - kbeg : 0) : R.kc;
    }
    const cplx *A = A0 + (long long)zb * sa;
    const cplx *B = B0 + (long long)zb * sb;
    cplx *C = C0 + (long long)zb * sc;
    const int m0 = byi * TM, n0 = bxi * TN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN, lr = lane & 15, lq = lane >> 4;
    const long long trow = IDX ? (long long)(R.z0 + bzi) * R.tab_stride : 0;
    const bool idxB = IDX == 1 && R.tabB != nullptr;
    if (idxB) {
        for (int k = tid; k < K; k += 256) kidx[k] = R.tabB[trow + R.offB + k].x;
        __syncthreads();
    }
    if (IDX == 2) {
        for (int k = tid; k < K; k += 256) kidx4[k] = R.tabB[trow + R.offB + k];
        __syncthreads();
    }
    // (IDX 2, sparse right-hand sides) bit j of am0 / am1: child 0 / 1 has outgoing rows for the j-th block of 64 columns of this tile
    unsigned am0 = ~0u, am1 = ~0u;
    if (IDX == 2 && R.act) {
        const int node = R.first + R.z0 + bzi;
        const NdDev nd = R.nodes[node];
        const int ct0 = n0 >> 6, ncl = ((Nn - n0 < TN ? Nn - n0 : TN) + 63) >> 6;
        am0 = 0; am1 = 0;
        for (int j = 0; j < ncl; ++j) {
            if (nd.kid[0] >= 0 && R.act[nd.kid[0] * R.nct + ct0 + j]) am0 |= 1u << j;
            if (nd.kid[1] >= 0 && R.act[nd.kid[1] * R.nct + ct0 + j]) am1 |= 1u << j;
        }
        int nzq = 0;                                                     // any nonzero among the front's own right-hand-side rows in these columns?
        // (over whole blocks of 64 columns: the workgroups of a tile narrower than that share a flag and must all come to the same verdict)
        constexpr int TS = TN < 64 ? 64 : TN;
        const int ns0 = TN < 64 ? (n0 & ~63) : n0;
        if (!(am0 | am1))
            for (int e = tid; e < K * TS; e += 256) {
                const int k = e / TS, bc = e % TS;
                const int4 t4 = kidx4[k];
                if (t4.w && ns0 + bc < Nn) { const cplx v = R.Bx[(long long)t4.x * R.ldx + ns0 + bc]; nzq |= (v.x != 0.0 || v.y != 0.0); }
            }
        if (!(am0 | am1) && !__syncthreads_or(nzq)) {
            if (byi == 0 && R.Cox)                                // y_S = 0 where the back substitution will look for it
                for (int e = tid; e < K * TN; e += 256) {
                    const int k = e / TN, bc = e % TN;
                    const int4 t4 = kidx4[k];
                    if (t4.w && n0 + bc < Nn) R.Cox[(long long)t4.x * R.ldx + n0 + bc] = cmake(0.0, 0.0);
                }
            return;
        }
        if (byi == 0 && tid < ncl) R.act[node * R.nct + ct0 + tid] = 1;
    }
    if (IDX == 1 && R.act && R.hint) {                                   // declared support: a leaf none of whose blocks of 64 columns carries a right-hand side is not read
        const int *fl = R.act + (long long)(R.first + R.z0 + bzi) * R.nct + (n0 >> 6);
        const int ncl = ((Nn - n0 < TN ? Nn - n0 : TN) + 63) >> 6;
        int any = 0;
        for (int j = 0; j < ncl; ++j) any |= fl[j];
        if (!any) return;
    }
    if (IDX == 1 && R.act && !R.hint && idxB && TN >= 64 && R.k2 == 0) {
        // No declared support: the leaf has to look at its right-hand-side rows.  Through the slab pipeline below that is one exposed load latency per
        // slab of 8 rows for what is nearly always a block of zeros; here every thread issues its share of the K x TN entries back to back and the
        // workgroup leaves if none of them is nonzero (the flags stay 0, nothing is stored: the state the exit after the pipeline leaves behind).
        int nzq = 0;
        const int tot = K * TN;
        #pragma unroll 7
        for (int e = tid; e < tot; e += 256) {
            const int k = e / TN, bc = e - k * TN;
            const int r = kidx[k];
            if (r >= 0 && n0 + bc < Nn) { const cplx v = R.Bx[(long long)r * R.ldx + n0 + bc]; nzq |= (v.x != 0.0 || v.y != 0.0) ? 1 : 0; }
        }
        if (!__syncthreads_or(nzq)) return;
    }
    int nzb = 0;                                                         // (IDX 1 with act: leaf level) bit j: a nonzero right-hand-side entry in the j-th block of 64 columns
    __shared__ int2 sgr[IDX == 4 ? TM : 1], sgc[IDX == 4 ? TN : 1];
    const cplx *S0 = nullptr, *S1 = nullptr;
    int ld0 = 0, ld1 = 0;
    if (IDX == 4) {
        const NdDev nd = R.nodes[R.first + R.z0 + bzi];
        int base0 = 0, base1 = 0;
        if (nd.kid[0] >= 0) { const NdDev c0 = R.nodes[nd.kid[0]]; S0 = R.arenaS + c0.foff + c0.smax; ld0 = c0.smax + c0.mmax; base0 = (int)(c0.voff + c0.smax); }
        if (nd.kid[1] >= 0) { const NdDev c1 = R.nodes[nd.kid[1]]; S1 = R.arenaS + c1.foff + c1.smax; ld1 = c1.smax + c1.mmax; base1 = (int)(c1.voff + c1.smax); }
        for (int t = tid; t < TM + TN; t += 256) {
            const int q = t < TM ? m0 + t : n0 + (t - TM);
            int2 e = make_int2(-1, -1);
            if (q < (t < TM ? M : Nn)) {
                const int4 t4 = R.tabCi[trow + nd.smax + q];
                if (t4.y >= 0 && S0) e.x = t4.y - base0;
                if (t4.z >= 0 && S1) e.y = t4.z - base1;
            }
            if (t < TM) sgr[t] = e; else sgc[t - TM] = e;
        }
        __syncthreads();
    }
    v4f64 cr[MT][NT], ci[MT][NT];
    #pragma unroll
    for (int i = 0; i < MT; ++i)
        #pragma unroll
        for (int j = 0; j < NT; ++j) { cr[i][j] = (v4f64){0, 0, 0, 0}; ci[i][j] = (v4f64){0, 0, 0, 0}; }
    cplx xacc = cmake(0.0, 0.0);                                       // (XR) this thread's share of row TMM, column tid % TN
    const int xc = tid % TN, xh = tid / TN;
    // (one register stage.  A second one -- the loads of slab k + 2 issued before slab k is multiplied -- was measured: 16-18 more registers take the 49 x 64 tile
    // from four workgroups per compute unit to three, leaf back substitution 1.70 -> 1.96 ms, headline 13 900 -> 13 700: reverted)
    cplx ra[NA], rb[NB];
    auto fetch = [&](int k0) {
        #pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int idx = tid + e * 256;
            const int ar = idx / KS, ak = idx % KS;
            cplx v = cmake(0.0, 0.0);
            if (idx < FA * 16 * KS && ar < TM && m0 + ar < M && k0 + ak < K) v = A[(long long)(m0 + ar) * lda + k0 + ak];
            ra[e] = v;
        }
        #pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int idx = tid + e * 256;
            const int bk = idx / TN, bc = idx % TN;
            cplx v = cmake(0.0, 0.0);
            if (idx < TN * KS && k0 + bk < K && n0 + bc < Nn) {
                if (IDX == 2) {
                    const int4 t4 = kidx4[k0 + bk];
                    if (t4.w) v = R.Bx[(long long)t4.x * R.ldx + n0 + bc];
                    if (t4.y >= 0 && ((am0 >> (bc >> 6)) & 1)) v = cadd(v, R.Cix[(long long)t4.y * R.ldx + n0 + bc]);
                    if (t4.z >= 0 && ((am1 >> (bc >> 6)) & 1)) v = cadd(v, R.Cix[(long long)t4.z * R.ldx + n0 + bc]);
                    if (byi == 0 && t4.w && R.Cox) R.Cox[(long long)t4.x * R.ldx + n0 + bc] = v;      // y_S
                }
                else if (idxB) {
                    const int r = kidx[k0 + bk];
                    if (r >= 0) v = (k0 + bk < R.k2 ? R.Bx2 : R.Bx)[(long long)r * R.ldx + n0 + bc];
                }
                else v = B[(long long)(k0 + bk) * ldb + n0 + bc];
            }
            rb[e] = v;
        }
    };
    auto stash = [&](int buf) {
        #pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int idx = tid + e * 256;
            const int ar = idx / KS, ak = idx % KS;
            if (idx < FA * 16 * KS) As[buf * ABUF + ((ak >> 2) * FA + (ar >> 4)) * 64 + (ak & 3) * 16 + ((ar & 15) ^ (ak & 3) ^ (((ak >> 2) & 1) << 2))] = ra[e];
        }
        #pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int idx = tid + e * 256;
            const int bk = idx / TN, bc = idx % TN;
            if (idx < TN * KS) Bs[buf * BBUF + ((bk >> 2) * FB + (bc >> 4)) * 64 + (bk & 3) * 16 + (bc & 15)] = rb[e];
            // (leaf level of the sparse-right-hand-side pass: looked at HERE, where the slab is in registers anyway -- testing the value in fetch()
            // made every load wait for its data and cost the kernel half of its bandwidth)
            if (IDX == 1 && R.act) { const long long bx = __double_as_longlong(rb[e].x) | __double_as_longlong(rb[e].y); nzb |= (bx << 1) ? 1 << (bc >> 6) : 0; }
        }
    };
    int kbeg = 0;
    if (IDX == 1 && R.act_ro && R.k2 > 0) {                             // (leaf back substitution on sparse right-hand sides, see GemmRows::act_ro)
        const int *fl = R.act_ro + (long long)(R.first + R.z0 + bzi) * R.nct + (n0 >> 6);
        const int ncl = ((Nn - n0 < TN ? Nn - n0 : TN) + 63) >> 6;
        int any = 0;
        for (int j = 0; j < ncl; ++j) any |= fl[j];
        if (!any) kbeg = (R.k2 / KS) * KS;
    }
    fetch(kbeg);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < K; k0 += KS) {
        const bool more = k0 + KS < K;
        if (more) fetch(k0 + KS);
        #pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            if (KG > 1 && k0 + 4 * kg >= K) break;                  // (a k group that is all padding)
            cplx a[MT], b[NT];
            #pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = As[cur * ABUF + (kg * FA + wm * MT + i) * 64 + lq * 16 + (lr ^ lq ^ ((kg & 1) << 2))];
            #pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = Bs[cur * BBUF + (kg * FB + wn * NT + j) * 64 + lane];
            #pragma unroll
            for (int i = 0; i < MT; ++i)
                #pragma unroll
                for (int j = 0; j < NT; ++j) {
                    cr[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, cr[i][j], 0, 0, 0);
                    ci[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].y, ci[i][j], 0, 0, 0);
                }
            #pragma unroll
            for (int i = 0; i < MT; ++i) {
                const double nai = -a[i].y;
                #pragma unroll
                for (int j = 0; j < NT; ++j) {
                    cr[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, b[j].y, cr[i][j], 0, 0, 0);
                    ci[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].x, ci[i][j], 0, 0, 0);
                }
            }
        }
        if (XR) {
            #pragma unroll
            for (int kk = 0; kk < KS / XP; ++kk) {
                const int k = kk * XP + xh;                            // k of the slab
                const int kg = k >> 2, kq = k & 3;
                const cplx av = As[cur * ABUF + (kg * FA + FA - 1) * 64 + kq * 16 + (kq ^ ((kg & 1) << 2))];
                const cplx bv = Bs[cur * BBUF + (kg * FB + (xc >> 4)) * 64 + kq * 16 + (xc & 15)];
                cfma(xacc, av, bv);
            }
        }
        if (more) { stash(cur ^ 1); __syncthreads(); cur ^= 1; }
    }
    unsigned colmask = ~0u;                                              // blocks of 64 columns of this tile whose results are stored
    if (IDX == 1 && R.act) {                                             // leaf level: which blocks of 64 columns carry a right-hand side at all
        const int node = R.first + R.z0 + bzi;
        colmask = 0;
        #pragma unroll
        for (int j = 0; j < (TN + 63) / 64; ++j)
            if (__syncthreads_or((nzb >> j) & 1)) {
                colmask |= 1u << j;
                if (byi == 0 && tid == 0 && (n0 >> 6) + j < R.nct) R.act[node * R.nct + (n0 >> 6) + j] = 1;
            }
        // tiles narrower than a block of 64 columns share its flag with their neighbours: another workgroup may raise it, so this one writes its
        // zeros; from 64 columns up nothing but zeros coming in means the rows stay unwritten and the flag stays 0
        if (TN < 64) colmask = ~0u;
        if (!colmask) return;
    }
    const bool b0 = (beta.x == 0.0 && beta.y == 0.0);
    // Epilogue, one block row of 16 at a time: first the addresses of its four rows (row-table look-ups), then EVERY value of C that has to be
    // read (beta != 0, the forward gather's child rows, the Schur gather's child entries) with the loads issued back to back -- masked elements
    // read a zero page instead of being branched around (a per-element `if (...) load` made hipcc wait vmcnt(0) after each one: up to 16
    // dependent round trips per thread) -- then the arithmetic and the stores.
    #pragma unroll
    for (int i = 0; i < MT; ++i) {
        cplx *dstq[4];
        const cplx *cinq[4], *cin2q[4];
        int2 erq[4];
        bool rowok[4];
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = m0 + (wm * MT + i) * 16 + lq + 4 * q;
            rowok[q] = r < M;
            const int rr = rowok[q] ? r : m0;                              // (a valid row for the look-ups of a masked one)
            cplx *dst = C + (long long)rr * ldc;
            const cplx *cin = dst, *cin2 = nullptr;
            if (IDX == 2) {
                const int4 t4 = R.tabCi[trow + R.offCi + rr];
                cin = t4.y >= 0 ? R.Cix + (long long)t4.y * R.ldx : nullptr;
                cin2 = t4.z >= 0 ? R.Cix + (long long)t4.z * R.ldx : nullptr;
            }
            if (IDX == 1 && R.tabCo) {
                const int ix = R.tabCo[trow + R.offCo + rr].x;
                if (ix < 0) rowok[q] = false;
                dst = R.Cox + (long long)(ix >= 0 ? ix : 0) * R.ldx;
            }
            if (IDX == 1 && R.tabCi && !b0) {
                const int ix = R.tabCi[trow + R.offCi + rr].x;
                cin = ix >= 0 ? R.Cix + (long long)ix * R.ldx : nullptr;
            }
            if (IDX == 4) erq[q] = sgr[rr - m0];
            if ((IDX == 0 || IDX == 1) && (b0 || (rr >= R.zr0 && rr < R.zr1))) cin = nullptr;
            dstq[q] = dst; cinq[q] = cin; cin2q[q] = cin2;
        }
        cplx cv[4][NT];
        #pragma unroll
        for (int q = 0; q < 4; ++q)
            #pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cc = n0 + (wn * NT + j) * 16 + lr;
                const bool ok = rowok[q] && cc < Nn && ((colmask >> (((wn * NT + j) * 16 + lr) >> 6)) & 1);
                if (IDX == 4) {
                    const int2 ec = sgc[cc < Nn ? cc - n0 : 0];
                    const cplx *p0 = (ok && erq[q].x >= 0 && ec.x >= 0) ? S0 + (long long)erq[q].x * ld0 + ec.x : g_zero_page;
                    const cplx *p1 = (ok && erq[q].y >= 0 && ec.y >= 0) ? S1 + (long long)erq[q].y * ld1 + ec.y : g_zero_page;
                    cv[q][j] = cadd(*p0, *p1);
                } else if (IDX == 2) {
                    const int cl = ((wn * NT + j) * 16 + lr) >> 6;          // block of 64 columns inside the tile
                    const cplx *p0 = (ok && cinq[q] && ((am0 >> cl) & 1)) ? cinq[q] + cc : g_zero_page;
                    const cplx *p1 = (ok && cin2q[q] && ((am1 >> cl) & 1)) ? cin2q[q] + cc : g_zero_page;
                    cv[q][j] = cadd(*p0, *p1);
                } else {
                    const cplx *p0 = (ok && cinq[q] && !(cc >= R.zc0 && cc < R.zc1)) ? cinq[q] + cc : g_zero_page;
                    cv[q][j] = *p0;
                }
            }
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = m0 + (wm * MT + i) * 16 + lq + 4 * q;
            const bool srow = r >= R.sk0 && r < R.sk1;
            #pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cc = n0 + (wn * NT + j) * 16 + lr;
                if (!rowok[q] || cc >= Nn || !((colmask >> (((wn * NT + j) * 16 + lr) >> 6)) & 1)) continue;
                if (srow && cc >= R.sk0 && cc < R.sk1) continue;
                cplx v = cmul(alpha, cmake(cr[i][j][q], ci[i][j][q]));
                if (IDX == 4) v = cadd(cv[q][j], v);
                else v = cadd(v, cmul(beta, cv[q][j]));
                if (R.ntc) __builtin_nontemporal_store((v2f64){v.x, v.y}, reinterpret_cast<v2f64 *>(dstq[q] + cc)); else dstq[q][cc] = v;
            }
        }
    }
    if (XR) {
        __syncthreads();                                               // every wave is done with the operand tiles: their LDS holds the partial sums now
        cplx *part = lds;
        if (xh > 0) part[(xh - 1) * TN + xc] = xacc;
        __syncthreads();
        const int r = m0 + TMM, cc = n0 + xc;
        if (xh == 0 && r < M && cc < Nn) {
            #pragma unroll
            for (int h = 1; h < XP; ++h) xacc = cadd(xacc, part[(h - 1) * TN + xc]);
            cplx *dst = C + (long long)r * ldc;
            const cplx *cin = dst;
            bool keep = true;
            if (IDX == 1 && R.tabCo) {
                const int ix = R.tabCo[trow + R.offCo + r].x;
                keep = ix >= 0;
                dst = R.Cox + (long long)(ix >= 0 ? ix : 0) * R.ldx;
            }
            if (IDX == 1 && R.tabCi && !b0) {
                const int ix = R.tabCi[trow + R.offCi + r].x;
                cin = ix >= 0 ? R.Cix + (long long)ix * R.ldx : nullptr;
            }
            if (keep) {
                cplx v = cmul(alpha, xacc);
                if (!b0 && cin) v = cadd(v, cmul(beta, cin[cc]));
                dst[cc] = v;
            }
        }
    }
}

template <int WM, int WN, int MT, int NT, int IDX, int KS, int OCC, int XR = 0>
__global__ __launch_bounds__(256, OCC) void k_zgemm3(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                                     const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, GemmRows R) {
    __shared__ cplx tiles[2 * (KS / 4) * (MT * WM + XR + NT * WN) * 64];
    zgemm3_body<WM, WN, MT, NT, IDX, KS, XR>(M, Nn, K, alpha, A0, lda, sa, B0, ldb, sb, beta, C0, ldc, sc, R, tiles);
}

// (Round 4, measured and removed: the same kernel with LDS-DMA staging -- global_load_lds_dwordx4 writing each 1-KB fragment straight into a
// three- or four-stage LDS ring, counted vmcnt, one raw barrier per slab, every LDS read of the loop in inline asm because hipcc makes any
// ds_read it can see wait vmcnt(0) while a DMA is in flight.  Correct on the product shapes, and 3-8 % SLOWER than the register-staged
// kernel above on every shape of tools/zgemm_lab.py (leaf back substitution 2989 against 2839 us, 1024 x 1024 x 256 x 16: 619 against 597 us,
// under-filled 1025 x 256 x 512 x 4: 164 against 163 us): at 3-4 workgroups per CU the other workgroups already cover a slab's load latency, and
// the DMA's per-fragment address arithmetic costs what the staging registers did.  profiles/r04_zgemm_lab_mfma.txt keeps the table.)
// the 16 MT + 1-row tile (XR) for dense and row-table operands
template <int MT, int NT, int KS, int OCC = 2>
void launch_mfma_xr(hipStream_t st, int idx, int nb, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                    cplx beta, cplx *C, int ldc, long long sc, const GemmRows &R) {
    constexpr int TM = 16 * MT + 1, TN = 16 * NT * 4;
    dim3 grid((Nn + TN - 1) / TN, (M + TM - 1) / TM, nb);
    if (idx) ZG_LAUNCH((k_zgemm3<1, 4, MT, NT, 1, KS, OCC, 1>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else ZG_LAUNCH((k_zgemm3<1, 4, MT, NT, 0, KS, OCC, 1>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
}

template <int WM, int WN, int MT, int NT, int KS, int OCC = 2>
void launch_mfma(hipStream_t st, int idx, int nb, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                 cplx beta, cplx *C, int ldc, long long sc, const GemmRows &R) {
    constexpr int TM = 16 * MT * WM, TN = 16 * NT * WN;
    dim3 grid((Nn + TN - 1) / TN, (M + TM - 1) / TM, nb);
    if (idx == 4) ZG_LAUNCH((k_zgemm3<WM, WN, MT, NT, 4, KS, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else if (idx == 2) ZG_LAUNCH((k_zgemm3<WM, WN, MT, NT, 2, KS, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else if (idx) ZG_LAUNCH((k_zgemm3<WM, WN, MT, NT, 1, KS, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else ZG_LAUNCH((k_zgemm3<WM, WN, MT, NT, 0, KS, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
}

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

// ---- fast path for blocks of at most 32 x 32: 256 threads, no serial phase ------------------------------------------------
// The generic routine above spends ~1.7 us per elimination step (4000 cycles: a one-wave pivot search through shuffles, two
// fp64 divisions, integer divisions in the update loop); this one needs two barriers and ~500 cycles:
//   * thread (i = tid / 8, columns 4 (tid % 8) .. +3) owns four entries of row i for the whole elimination;
//   * the pivot of the next column is chosen by EVERY thread from 32 keys in LDS -- float(max(|re|, |im|)) with the row index in
//     the low five bits, so the search is one v_max_u32 reduction over eight 16-byte broadcast reads; the keys are written by
//     the threads that produce that column in the previous step;
//   * no row exchange (implicit pivoting) and no scaling of the pivot row (its 1 / d is applied once at the end); all old values are read
//     before the barrier and all new ones written after it;
//   * the reciprocal of the pivot is v_rcp_f64 + two Newton steps.
// (Measured and reverted: exchanging the pivot row and the multiplier column through small LDS buffers instead of the full matrix --
// fewer LDS bytes, but write -> barrier -> read makes three dependent LDS round trips per step instead of two: 65 -> 88 us per block step.)
struct Gj32 {
    cplx a[32][33];       // the matrix on entry, the result on return
    cplx b[32][33];       // second buffer: step k reads one and writes the other, so a step needs ONE barrier
    unsigned cand[2][32];
    cplx dinv[2][32];     // reciprocal of every row's entry in the column that is eliminated next
    int piv[32];          // sigma: pivot row of step k
    int sinv[32];         // step at which row r was the pivot
};
__device__ __forceinline__ double gj_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ unsigned gj_key(cplx v, int i) {
    const float m = (float)fmax(fabs(v.x), fabs(v.y));
    return (__float_as_uint(m) & ~31u) | (unsigned)i;
}
__device__ __forceinline__ cplx gj_recip(cplx d) {
    const double rr = gj_rcp(d.x * d.x + d.y * d.y);
    return cmake(d.x * rr, -d.y * rr);
}
// S.a must hold the matrix padded with the identity to 32 x 32; all 256 threads call.
// Implicit pivoting (rows stay where they are, S.piv[k] = sigma(k) = pivot row of step k, S.sinv its inverse) and deferred scaling of the
// pivot rows (see k_gj32w_inverse).  A step is a chain of latencies, so it is kept short:
//   * a thread keeps its four entries in registers for the whole elimination and publishes them to the buffer the NEXT step reads:
//     one barrier per step;
//   * the thread that produces row i's entry of column k + 1 also publishes the pivot key and the reciprocal of that entry, so step k + 1
//     starts with   read keys -> p;  read row p, 1 / d, own multiplier   and goes straight to the multiply-adds (the division is off the
//     critical path: it runs beside the other three entries' updates of the previous step).
// On return S.a holds the storage rows R with   inverse[i][sigma(k)] = R[sigma(i)][k].
// Measured with clock64 around the call (2.39 GHz, one workgroup): 48 000 cycles for 32 steps = 1500 per step for ~110 instructions per
// wave -- the step is bound by the number of instructions one wave has to issue one after the other, not by a particular latency.  A form
// in panels of four steps (the four columns of a panel in one half-wave: pivot search and pivot row by v_readlane, the other threads apply
// four steps at once after one barrier; bit-for-bit the same result) was built and measured: 3550 cycles for the four narrow steps + 1700
// for the rank-4 update per panel = the same 46-50 000 cycles; reverted.  The pivot search as an LDS atomic (ds_max_u32 by the 32 threads that hold the
// column, one word read by everybody instead of 32 keys and their maximum): 22.5 -> 29.2 us per block; reverted.
__device__ __forceinline__ void gj32(Gj32 &S, int n, int tid) {
    const int i = tid >> 3, jc = tid & 7, j0 = jc * 4;
    bool used = i >= n;
    cplx srow = cmake(1.0, 0.0);
    cplx out[4];
    #pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = S.a[i][j0 + q];
    if (jc == 0) { S.cand[0][i] = used ? (unsigned)i : gj_key(out[0], i); S.dinv[0][i] = gj_recip(out[0]); S.piv[i] = i; S.sinv[i] = i; }
    __syncthreads();
    cplx (*cur)[33] = S.a, (*nxt)[33] = S.b;
    // four steps per trip, so that the register that holds column k (out[k & 3]) is known at compile time: one select per step instead of four
    // compare-and-select groups, and no select chain for the next column's key
#define GJ32_STEP(Q_) do {                                                                                                         \
        const int k = k4 + (Q_);                                                                                                   \
        if (k >= n) break;                                                                                                         \
        const int pb = k & 1;                                                                                                      \
        const uint4 *c4 = reinterpret_cast<const uint4 *>(S.cand[pb]);                                                             \
        unsigned m = 0;                                                                                                            \
        _Pragma("unroll")                                                                                                          \
        for (int q = 0; q < 8; ++q) { const uint4 v = c4[q]; m = max(m, max(max(v.x, v.y), max(v.z, v.w))); }                       \
        const int p = (int)(m & 31u);                                                                                              \
        const cplx dinv = S.dinv[pb][p], f = cur[i][k];                                                                            \
        cplx pr[4];                                                                                                                \
        _Pragma("unroll")                                                                                                          \
        for (int q = 0; q < 4; ++q) pr[q] = cur[p][j0 + q];                                                                        \
        if (tid == 0) { S.piv[k] = p; S.sinv[p] = k; }                                                                             \
        const cplx fp = (i == p) ? cmake(0.0, 0.0) : cmul(f, dinv);                                                                \
        _Pragma("unroll")                                                                                                          \
        for (int q = 0; q < 4; ++q) {                                                                                              \
            out[q].x = fma(-fp.x, pr[q].x, out[q].x); out[q].x = fma(fp.y, pr[q].y, out[q].x);                                     \
            out[q].y = fma(-fp.x, pr[q].y, out[q].y); out[q].y = fma(-fp.y, pr[q].x, out[q].y);                                    \
        }                                                                                                                          \
        if (jc == (k4 >> 2)) out[Q_] = (i == p) ? cmake(1.0, 0.0) : cneg(fp);        /* column k of the running inverse */            \
        if (i == p) { srow = dinv; used = true; }                                                                                  \
        if (k + 1 < n) {                                                                                                           \
            if (jc == ((k + 1) >> 2)) {        /* key and reciprocal of the next column (unused rows only: their scale is still 1) */ \
                const cplx v = out[((Q_) + 1) & 3];                                                                                \
                S.cand[pb ^ 1][i] = used ? (unsigned)i : gj_key(v, i);                                                             \
                S.dinv[pb ^ 1][i] = gj_recip(v);                                                                                   \
            }                                                                                                                      \
            _Pragma("unroll")                                                                                                      \
            for (int q = 0; q < 4; ++q) nxt[i][j0 + q] = out[q];                                                                   \
        }                                                                                                                          \
        __syncthreads();                                                                                                           \
        cplx (*t_)[33] = cur; cur = nxt; nxt = t_;                                                                                 \
    } while (0)
    for (int k4 = 0; k4 < n; k4 += 4) { GJ32_STEP(0); GJ32_STEP(1); GJ32_STEP(2); GJ32_STEP(3); }
#undef GJ32_STEP
    #pragma unroll
    for (int q = 0; q < 4; ++q) S.a[i][j0 + q] = cmul(out[q], srow);
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_gj32_inverse(cplx *A0, int ld, long long stride, int n) {
    __shared__ Gj32 S;
    cplx *A = A0 + (long long)blockIdx.x * stride;
    const int tid = threadIdx.x, i = tid >> 3, j0 = (tid & 7) * 4;
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        S.a[i][j] = (i < n && j < n) ? A[(long long)i * ld + j] : cmake(i == j ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();
    gj32(S, n, tid);
    // inverse[row][sigma(j)] = R[sigma(row)][j]: this thread holds storage row i = sigma(row), i.e. row = sinv[i]
    const int row = S.sinv[i] & 31;
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        const int col = S.piv[j] & 31;
        if (i < n && j < n && row < n && col < n) A[(long long)row * ld + col] = S.a[i][j];
    }
}

// ---- throughput variant for thousands of small blocks: one WAVE per matrix, the matrix in registers -------------------------------
// Lane l holds half a row: row r = l & 31, columns 16 h .. 16 h + 15 with h = l >> 5 (16 complex = 64 VGPRs).  The 32 elimination steps
// are unrolled so that every register index is static.  Per step the wave exchanges three things through its own 1.5 KB of LDS (LDS
// operations of one wave execute in order, no barrier): the pivot keys of column k, the pivot row (written by its two owner lanes, read
// as broadcasts), and the multipliers a[r][k] for the half that does not hold column k.  Pivoting is implicit -- rows stay where they
// are, sigma(k) records the pivot row of step k -- and is undone when the result is stored: inv[i][sigma(k)] = R[sigma(i)][k].
// ~220 wave-instructions per step against ~840 for the four-wave kernel above: the leaf level's two launches of 16 384 blocks are
// issue-bound, so this is what they cost.
struct Gj32w {
    cplx prow[32];
    cplx fcol[32];
    unsigned cand[32];
    int sigma[32];       // pivot row of step k
    int sinv[32];        // step at which row r was the pivot
};
__global__ __launch_bounds__(256) void k_gj32w_inverse(cplx *A0, int ld, long long stride, int n, int nmat) {
    __shared__ Gj32w SW[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int mat = blockIdx.x * 4 + w;
    if (mat >= nmat) return;                       // whole waves leave together (no block-level barrier below)
    Gj32w &S = SW[w];
    cplx *A = A0 + (long long)mat * stride;
    const int r = lane & 31, h = lane >> 5;
    cplx a[16];
    #pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int j = 16 * h + c;
        a[c] = (r < n && j < n) ? A[(long long)r * ld + j] : cmake(r == j ? 1.0 : 0.0, 0.0);
    }
    bool used = r >= n;                            // padding rows never pivot
    if (h == 0) { S.sigma[r] = r; S.sinv[r] = r; }  // (a singular block may leave entries unset: keep every index in range)
    // The pivot row is NOT scaled when it is chosen: the other rows are eliminated with the multiplier f / d against the unscaled row,
    // and the factor 1 / d of the pivot row rides along in `srow` until the end (every later operation on that row is linear in it).
    // That leaves 16 complex multiply-adds per lane and step -- scaling the row at once would double the fp64 work of a step.
    cplx srow = cmake(1.0, 0.0);
    #pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (k < n) {
            const int kc = k & 15, kh = k >> 4;
            // pivot keys and multipliers of column k, from the half that holds it (candidates are unused rows: their scale is still 1)
            if (h == kh) {
                S.cand[r] = used ? (unsigned)r : gj_key(a[kc], r);
                S.fcol[r] = a[kc];
            }
            __builtin_amdgcn_wave_barrier();
            const uint4 *c4 = reinterpret_cast<const uint4 *>(S.cand);
            unsigned m = 0;
            #pragma unroll
            for (int q = 0; q < 8; ++q) { const uint4 v = c4[q]; m = max(m, max(max(v.x, v.y), max(v.z, v.w))); }
            const int p = (int)(m & 31u);
            const cplx f = S.fcol[r];
            __builtin_amdgcn_wave_barrier();
            // the pivot row, written by its two owner lanes
            if (r == p) {
                #pragma unroll
                for (int c = 0; c < 16; ++c) S.prow[16 * h + c] = a[c];
                if (h == 0) { S.sigma[k] = p; S.sinv[p] = k; }
                used = true;
            }
            __builtin_amdgcn_wave_barrier();
            const cplx d = S.prow[k];
            const double rr = gj_rcp(d.x * d.x + d.y * d.y);
            const cplx dinv = cmake(d.x * rr, -d.y * rr);
            const cplx fp = (r == p) ? cmake(0.0, 0.0) : cmul(f, dinv);
            #pragma unroll
            for (int c = 0; c < 16; ++c) {
                const cplx pr = S.prow[16 * h + c];
                a[c].x = fma(-fp.x, pr.x, a[c].x); a[c].x = fma(fp.y, pr.y, a[c].x);
                a[c].y = fma(-fp.x, pr.y, a[c].y); a[c].y = fma(-fp.y, pr.x, a[c].y);
            }
            if (h == kh) a[kc] = (r == p) ? cmake(1.0, 0.0) : cneg(fp);        // column k of the running inverse
            if (r == p) srow = dinv;
            __builtin_amdgcn_wave_barrier();
        }
    }
    #pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = cmul(a[c], srow);
    // inv[i][sigma(kcol)] = R[sigma(i)][kcol]: this lane holds storage row r = sigma(i), i.e. output row i = sinv[r]
    if (r < n) {
        const int i = S.sinv[r] & 31;
        #pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int kcol = 16 * h + c;
            const int col = S.sigma[kcol & 31] & 31;
            if (kcol < n && i < n && col < n) A[(long long)i * ld + col] = a[c];
        }
    }
}

// ---- in-place inverse of n x n blocks, n <= 64: one workgroup per matrix -------------------------------------------------
// (bottom of the block inversions: 32 by default; 64 (HELM_ND_GJ=64) gains a digit of accuracy, but its 64-step elimination
// is slower overall: 41.8 vs 35.9 ms per factorisation at 1024^2)
template <int NMAX, int NT = 256>
__global__ __launch_bounds__(NT) void k_gj_inverse(cplx *A0, int ld, long long stride, int n) {
    __shared__ cplx a[NMAX][NMAX + 1];
    __shared__ cplx fcol[NMAX];
    __shared__ int piv[NMAX];
    cplx *A = A0 + (long long)blockIdx.x * stride;
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += blockDim.x) a[e / n][e % n] = A[(long long)(e / n) * ld + e % n];
    __syncthreads();
    gj_lds<NMAX>(a, fcol, piv, n, tid, blockDim.x);
    for (int e = tid; e < n * n; e += blockDim.x) A[(long long)(e / n) * ld + e % n] = a[e / n][e % n];
}

// ---- blocked Gauss-Jordan inversion: panel kernel ----------------------------------------------------------------------
// In-place inverse of T (n x n) by block steps of nb <= 32 columns.  Step k with pivot block T_kk (rows / columns [k0, k0+nb)):
//     P = T_kk^-1 ;  R = P T[k, :] with R_k := P ;  C = T[:, k] with C_k := -I ;  T <- Z(T) - C R
// where Z zeroes block row k and block column k (the GEMM's masked beta).  This kernel makes R (nb x n) and C (n x nb) in
// scratch; every workgroup inverts the pivot block for itself (25 us, redundant but parallel) and then produces a 64-wide
// slice of R and a 64-tall slice of C.  Compared with the recursive 2 x 2 block inversion the chain of dependent launches is
// n / nb steps of two fat launches instead of ~6.8 n / 32 thin ones, which is what the upper tree levels were spending
// their time on.  Same pivots (the block-LU Schur complements), same accuracy class.
#define PNB 32
__global__ __launch_bounds__(256) void k_gj_panel(cplx *T0, int ld, long long stride, int n, int k0, int nb, cplx *Wc0, cplx *Wr0, long long wstride) {
    __shared__ Gj32 S;
    __shared__ cplx t[PNB][64 + 1];
    __shared__ int cperm[PNB];
    cplx *T = T0 + (long long)blockIdx.y * stride;
    cplx *Wc = Wc0 + (long long)blockIdx.y * wstride, *Wr = Wr0 + (long long)blockIdx.y * wstride;
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * 64;                     // this workgroup's slice [s0, s0 + 64) of the columns of R / rows of C
    {
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            S.a[i][j] = (i < nb && j < nb) ? T[(long long)(k0 + i) * ld + k0 + j] : cmake(i == j ? 1.0 : 0.0, 0.0);
        }
    }
    // row-panel slice T[k-rows, s0 .. s0+63] -> LDS ; column-panel slice copied out (C_k = -I)
    for (int e = tid; e < PNB * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        t[r][c] = (r < nb && s0 + c < n) ? T[(long long)(k0 + r) * ld + s0 + c] : cmake(0.0, 0.0);
    }
    for (int e = tid; e < 64 * PNB; e += 256) {
        const int r = e >> 5, c = e & 31;
        if (s0 + r >= n || c >= nb) continue;
        const int gr = s0 + r;
        cplx v = T[(long long)gr * ld + k0 + c];
        if (gr >= k0 && gr < k0 + nb) v = (gr - k0 == c) ? cmake(-1.0, 0.0) : cmake(0.0, 0.0);
        Wc[(long long)gr * PNB + c] = v;
    }
    __syncthreads();
    gj32(S, nb, tid);
    if (tid < PNB) cperm[tid] = S.piv[tid] & 31;            // P[r][sigma(j)] = S.a[sigma(r)][j]
    __syncthreads();
    // R slice = P * t  (nb x 64) with P[r][cperm[j]] = S.a[r][j]; columns inside the pivot block get P itself.
    // Thread (r = tid / 8, eight consecutive columns): ten LDS reads per eight complex multiply-adds.
    {
        const int r = tid >> 3, c0 = (tid & 7) * 8;
        cplx acc[8];
        #pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = cmake(0.0, 0.0);
        const int sr = cperm[r];                             // storage row of output row r
        for (int j = 0; j < nb; ++j) {
            const cplx a = S.a[sr][j];
            const int tj = cperm[j];
            #pragma unroll
            for (int i = 0; i < 8; ++i) cfma(acc[i], a, t[tj][c0 + i]);
        }
        if (r < nb) {
            #pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gc = s0 + c0 + i;
                if (gc < n && !(gc >= k0 && gc < k0 + nb)) Wr[(long long)r * n + gc] = acc[i];
            }
        }
    }
    if (s0 < k0 + nb && s0 + 64 > k0)                    // the slice that holds the pivot block's columns
        for (int e = tid; e < nb * nb; e += 256) {
            const int r = e / nb, j = e % nb;
            const int gc = k0 + cperm[j];
            if (gc >= s0 && gc < s0 + 64) Wr[(long long)r * n + gc] = S.a[cperm[r]][j];
        }
}

// Look-ahead form of the panel step.  The Gauss-Jordan sweep of the next pivot block only needs that 32 x 32 block, so
// it runs on a second stream beside the rank-32 update of the whole front: k_gj_pivot applies the pending update to its
// block privately (the GEMM skips it, GemmRows::sk0/sk1), inverts it and leaves P in Pb; k_gj_slices then forms the
// panels R_k = P T[k-rows, :], C_k = T[:, k-cols] from the updated front.
// LDS of the sweep: wc | wr (2 x 32 x 33 complex) while the pending update is applied to the block, then the Gj32 state in the same place
constexpr int GJ_PIVOT_LDS = (int)sizeof(Gj32) + PNB * (int)sizeof(int);
static_assert(sizeof(Gj32) >= 2 * PNB * (PNB + 1) * sizeof(cplx), "the two panels lie over Gj32's buffers");

__device__ __forceinline__ void gj_pivot_body(const cplx *T0, int ld, long long stride, int n, int k0, int nb, const cplx *Wc0, const cplx *Wr0, long long wstride,
                                              cplx *Pb0, long long pstride, int mat, char *lds) {
    cplx (&wc)[PNB][PNB + 1] = *reinterpret_cast<cplx (*)[PNB][PNB + 1]>(lds);
    cplx (&wr)[PNB][PNB + 1] = *reinterpret_cast<cplx (*)[PNB][PNB + 1]>(lds + PNB * (PNB + 1) * sizeof(cplx));
    Gj32 &S = *reinterpret_cast<Gj32 *>(lds);
    int *cperm = reinterpret_cast<int *>(lds + sizeof(Gj32));
    const cplx *T = T0 + (long long)mat * stride;
    const cplx *Wc = Wc0 + (long long)mat * wstride, *Wr = Wr0 + (long long)mat * wstride;
    cplx *Pb = Pb0 + (long long)mat * pstride;
    const int tid = threadIdx.x;
    const int i = tid >> 3, j0 = (tid & 7) * 4;
    const bool pending = k0 > 0;                          // the update of step k-1 (full width PNB) has not touched this block
    if (pending) {
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            wc[i][j] = i < nb ? Wc[(long long)(k0 + i) * PNB + j] : cmake(0.0, 0.0);
            wr[i][j] = j < nb ? Wr[(long long)i * n + k0 + j] : cmake(0.0, 0.0);
        }
        __syncthreads();
    }
    {
        cplx v[4];
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            v[q] = (i < nb && j < nb) ? T[(long long)(k0 + i) * ld + k0 + j] : cmake(i == j ? 1.0 : 0.0, 0.0);
        }
        if (pending) {
            cplx acc[4];
            #pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = cmake(0.0, 0.0);
            for (int p = 0; p < PNB; ++p) {
                const cplx a = wc[i][p];
                #pragma unroll
                for (int q = 0; q < 4; ++q) cfma(acc[q], a, wr[p][j0 + q]);
            }
            #pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = csub(v[q], acc[q]);
            __syncthreads();                              // the panels are read: S may take their place
        }
        #pragma unroll
        for (int q = 0; q < 4; ++q) S.a[i][j0 + q] = v[q];
    }
    __syncthreads();
    gj32(S, nb, tid);
    if (tid < PNB) cperm[tid] = S.piv[tid] & 31;            // P[r][sigma(j)] = S.a[sigma(r)][j]
    __syncthreads();
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        if (i < nb && j < nb) Pb[i * PNB + cperm[j]] = S.a[cperm[i]][j];
    }
}

__global__ __launch_bounds__(256) void k_gj_pivot(const cplx *T0, int ld, long long stride, int n, int k0, int nb, const cplx *Wc0, const cplx *Wr0, long long wstride,
                                                  cplx *Pb0, long long pstride) {
    __shared__ __attribute__((aligned(16))) char lds[GJ_PIVOT_LDS];
    gj_pivot_body(T0, ld, stride, n, k0, nb, Wc0, Wr0, wstride, Pb0, pstride, blockIdx.x, lds);
}

// Rank-32 update of step k and the Gauss-Jordan sweep of pivot block k+1 in ONE launch: the workgroups of one extra z-slice of the grid
// do the sweeps (one per matrix, the rest of that slice leaves at once), all others are tiles of the masked update, which skips the pivot
// block.  The sweep (31 us, one workgroup) hides behind the update without a second stream: cross-stream event hops cost 15-20 us apiece.
template <int TM, int RN, int KS, int UNR>
__global__ __launch_bounds__(256) void k_zgemm2_la(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                                   const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, GemmRows R, GjPivotArgs pv) {
    constexpr int TN = 1024 / TM * RN;
    constexpr int ABYTES = 2 * KS * (TM + 1) * (int)sizeof(cplx), BBYTES = 2 * KS * TN * (int)sizeof(cplx);
    constexpr int LDS = ABYTES + BBYTES > GJ_PIVOT_LDS ? ABYTES + BBYTES : GJ_PIVOT_LDS;
    __shared__ __attribute__((aligned(16))) char lds[LDS];               // one buffer: a workgroup is either a tile or a sweep
    if (blockIdx.z == 0) {                                                 // the sweeps go out first (dispatch order is x, y, z)
        const int mat = blockIdx.y * gridDim.x + blockIdx.x;
        if (mat < pv.batch) gj_pivot_body(pv.T0, pv.ld, pv.stride, pv.n, pv.k0, pv.nb, pv.Wc0, pv.Wr0, pv.wstride, pv.Pb0, pv.pstride, mat, lds);
        return;
    }
    zgemm2_body<TM, 0, RN, KS, UNR>(M, Nn, K, alpha, A0 - sa, lda, sa, B0 - sb, ldb, sb, beta, C0 - sc, ldc, sc, R,
                                  reinterpret_cast<cplx *>(lds), reinterpret_cast<cplx *>(lds + ABYTES));     // (its batch index is blockIdx.z - 1)
}

// the same fused launch with the matrix-core tile body (generation 7)
template <int WM, int WN, int MT, int NT, int KS>
__global__ __launch_bounds__(256, 2) void k_zgemm3_la(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                                      const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, GemmRows R, GjPivotArgs pv) {
    constexpr int TBYTES = 2 * (KS / 4) * (MT * WM + NT * WN) * 64 * (int)sizeof(cplx);
    constexpr int LDS = TBYTES > GJ_PIVOT_LDS ? TBYTES : GJ_PIVOT_LDS;
    __shared__ __attribute__((aligned(16))) char lds[LDS];
    if (blockIdx.z == 0) {
        const int mat = blockIdx.y * gridDim.x + blockIdx.x;
        if (mat < pv.batch) gj_pivot_body(pv.T0, pv.ld, pv.stride, pv.n, pv.k0, pv.nb, pv.Wc0, pv.Wr0, pv.wstride, pv.Pb0, pv.pstride, mat, lds);
        return;
    }
    zgemm3_body<WM, WN, MT, NT, 0, KS>(M, Nn, K, alpha, A0 - sa, lda, sa, B0 - sb, ldb, sb, beta, C0 - sc, ldc, sc, R, reinterpret_cast<cplx *>(lds));
}

// ---- one launch per block step of the blocked Gauss-Jordan inversion (round 4) ---------------------------------------------------------------
// The step   T <- Z(T) - C R   with   R = P T[k rows, :] (R_k := P),  C = T[:, k cols] (C_k := -I)   needed two launches because the update overwrites
// the pivot rows and columns that the other tiles still read: k_gj_slices copied the panels out first (18 us of the 54 us a step of a 1024-wide
// front takes).  With TWO copies of the matrix -- a step reads one and writes the other -- nothing a tile reads is written in the same launch,
// so every 64 x 32 tile forms its own slab of R (P times the raw pivot rows of its 32 columns: 32^3 multiply-adds, redundant across the row
// tiles, hidden behind the sweep) and takes its slab of C straight from the source.  The sweep of the NEXT pivot block rides in the first
// z-slice as before; it applies the step to its 32 x 32 block privately, now from the source matrix and P (two 32^3 products) instead of
// the panels.  The workspace W holds the second copy (n^2 per matrix), the two alternating P buffers come from the handle.
struct GjStepArgs {
    const cplx *Ta; int lda; long long sa;     // source: read-only in this launch
    cplx *Tb; int ldb; long long sb;           // destination
    int n, k0, nb;                             // this step's pivot block: rows / columns [k0, k0 + nb)
    const cplx *P; cplx *Pn; long long sp;     // its inverse (PNB x PNB per matrix) ; where the sweep leaves the next block's
    int k1, nb1;                               // the next pivot block (nb1 == 0: none)
    int batch, nsw;                            // matrices; z-slices of the grid that hold the sweeps (one workgroup per matrix)
};
// LDS: the sweep state (35 KB) -- the two 32 x 32 blocks of the private update and the tiles' P and R slab lie over it.  (A first version kept four
// blocks, 68 KB: alone on the GPU the same speed, but beside the solve kernels of the previous work item a workgroup of that size waits for a
// compute unit with that much LDS free -- the products' launches took 171 instead of 156 us on average inside the pipeline; with 35 KB 156.)
constexpr int GJS_LDS = (int)sizeof(Gj32) + PNB * (int)sizeof(int);
static_assert(GJS_LDS >= 2 * PNB * (PNB + 1) * (int)sizeof(cplx), "two 32 x 32 blocks lie over the sweep state");

__global__ __launch_bounds__(256, 2) void k_gj_step(GjStepArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[GJS_LDS];
    const int tid = threadIdx.x;
    const int k0 = a.k0, nb = a.nb, n = a.n;
    cplx (&X0)[PNB][PNB + 1] = *reinterpret_cast<cplx (*)[PNB][PNB + 1]>(lds);
    cplx (&X1)[PNB][PNB + 1] = *reinterpret_cast<cplx (*)[PNB][PNB + 1]>(lds + PNB * (PNB + 1) * sizeof(cplx));
    if ((int)blockIdx.z < a.nsw) {                           // ---- sweep of the next pivot block (one workgroup per matrix, the first nsw z-slices)
        const int mat = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (mat >= a.batch || a.nb1 == 0) return;
        const cplx *Ta = a.Ta + (long long)mat * a.sa, *P = a.P + (long long)mat * a.sp;
        cplx *Pn = a.Pn + (long long)mat * a.sp;
        const int k1 = a.k1, nb1 = a.nb1, lda = a.lda;
        Gj32 &S = *reinterpret_cast<Gj32 *>(lds);
        int *cperm = reinterpret_cast<int *>(lds + sizeof(Gj32));
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        cplx v[4], lreg[4];
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            X0[i][j] = (i < nb && j < nb1) ? Ta[(long long)(k0 + i) * lda + k1 + j] : cmake(0.0, 0.0);       // pivot rows, columns of the next block
            X1[i][j] = (i < nb && j < nb) ? P[i * PNB + j] : cmake(0.0, 0.0);
            lreg[q] = (i < nb1 && j < nb) ? Ta[(long long)(k1 + i) * lda + k0 + j] : cmake(0.0, 0.0);        // C_k, rows of the next block
            v[q] = (i < nb1 && j < nb1) ? Ta[(long long)(k1 + i) * lda + k1 + j] : cmake(i == j ? 1.0 : 0.0, 0.0);
        }
        __syncthreads();
        cplx acc[4];
        #pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = cmake(0.0, 0.0);
        #pragma unroll 2
        for (int p = 0; p < PNB; ++p) {
            const cplx x = X1[i][p];
            #pragma unroll
            for (int q = 0; q < 4; ++q) cfma(acc[q], x, X0[p][j0 + q]);
        }
        __syncthreads();
        #pragma unroll
        for (int q = 0; q < 4; ++q) { X0[i][j0 + q] = acc[q]; X1[i][j0 + q] = lreg[q]; acc[q] = cmake(0.0, 0.0); }      // R_k (columns of the next block), C_k
        __syncthreads();
        #pragma unroll 2
        for (int p = 0; p < PNB; ++p) {
            const cplx x = X1[i][p];
            #pragma unroll
            for (int q = 0; q < 4; ++q) cfma(acc[q], x, X0[p][j0 + q]);
        }
        #pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = csub(v[q], acc[q]);
        __syncthreads();                                     // the two blocks are read: S takes their place
        #pragma unroll
        for (int q = 0; q < 4; ++q) S.a[i][j0 + q] = v[q];
        __syncthreads();
        gj32(S, nb1, tid);
        if (tid < PNB) cperm[tid] = S.piv[tid] & 31;         // P[r][sigma(j)] = S.a[sigma(r)][j]
        __syncthreads();
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            if (i < nb1 && j < nb1) Pn[i * PNB + cperm[j]] = S.a[cperm[i]][j];
        }
        return;
    }
    // ---- a 64 x 32 tile of the update, both products on the matrix cores (v_mfma_f64_16x16x4_f64: A lane l = A[l % 16][l / 16], B lane l = B[l / 16][l % 16],
    // D register q of lane l = D[l / 16 + 4 q][l % 16]; four real instructions per complex block and k step of 4, as in zgemm3_body)
    const int mat = blockIdx.z - a.nsw;
    const cplx *Ta = a.Ta + (long long)mat * a.sa, *P = a.P + (long long)mat * a.sp;
    cplx *Tb = a.Tb + (long long)mat * a.sb;
    const int lda = a.lda, ldb = a.ldb;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 32;
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    cplx (&Ps)[PNB][PNB + 1] = X0;
    cplx (&Bs)[PNB][PNB + 1] = X1;
    // this lane's fragments of the C slab T[tile rows, k cols] (-I on the pivot rows): row 16 wave + lr, columns 4 ks + lq -- straight from the source
    cplx afr[PNB / 4];
    {
        const int gr = r0 + 16 * wave + lr;
        const bool prow = gr >= k0 && gr < k0 + nb;
        #pragma unroll
        for (int ks = 0; ks < PNB / 4; ++ks) {
            const int j = 4 * ks + lq;
            cplx x = cmake(0.0, 0.0);
            if (gr < n && j < nb) x = prow ? cmake(gr - k0 == j ? -1.0 : 0.0, 0.0) : Ta[(long long)gr * lda + k0 + j];
            afr[ks] = x;
        }
    }
    #pragma unroll
    for (int l = 0; l < 4; ++l) {                            // P, and the raw pivot rows of the tile's columns (the identity where they are pivot columns: R_k = P)
        const int e = tid + 256 * l, r = e >> 5, c = e & 31, gc = c0 + c;
        Ps[r][c] = (r < nb && c < nb) ? P[r * PNB + c] : cmake(0.0, 0.0);
        cplx x = cmake(0.0, 0.0);
        if (r < nb && gc < n) {
            if (gc >= k0 && gc < k0 + nb) x = cmake(gc - k0 == r ? 1.0 : 0.0, 0.0);
            else x = Ta[(long long)(k0 + r) * lda + gc];
        }
        Bs[r][c] = x;
    }
    __syncthreads();
    {
        const int br = wave >> 1, bc = wave & 1;             // wave -> one 16 x 16 block of the 32 x 32 slab R = P * (raw pivot rows)
        v4f64 er = {0.0, 0.0, 0.0, 0.0}, ei = {0.0, 0.0, 0.0, 0.0};
        #pragma unroll
        for (int ks = 0; ks < PNB / 4; ++ks) {
            const cplx x = Ps[16 * br + lr][4 * ks + lq], y = Bs[4 * ks + lq][16 * bc + lr];
            er = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.x, er, 0, 0, 0);
            ei = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.y, ei, 0, 0, 0);
            er = __builtin_amdgcn_mfma_f64_16x16x4f64(-x.y, y.y, er, 0, 0, 0);
            ei = __builtin_amdgcn_mfma_f64_16x16x4f64(x.y, y.x, ei, 0, 0, 0);
        }
        __syncthreads();                                     // the raw rows are read: the slab takes their place
        #pragma unroll
        for (int q = 0; q < 4; ++q) Bs[16 * br + lq + 4 * q][16 * bc + lr] = cmake(er[q], ei[q]);
    }
    __syncthreads();
    v4f64 cr[2], ci[2];                                      // wave -> rows 16 wave .. + 15 of the tile, both 16-column blocks
    #pragma unroll
    for (int j = 0; j < 2; ++j) { cr[j] = v4f64{0.0, 0.0, 0.0, 0.0}; ci[j] = v4f64{0.0, 0.0, 0.0, 0.0}; }
    #pragma unroll
    for (int ks = 0; ks < PNB / 4; ++ks) {
        const cplx x = afr[ks];
        #pragma unroll
        for (int j = 0; j < 2; ++j) {
            const cplx y = Bs[4 * ks + lq][16 * j + lr];
            cr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.x, cr[j], 0, 0, 0);
            ci[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.y, ci[j], 0, 0, 0);
            cr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x.y, y.y, cr[j], 0, 0, 0);
            ci[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x.y, y.x, ci[j], 0, 0, 0);
        }
    }
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int gr = r0 + 16 * wave + lq + 4 * q;
        if (gr >= n) continue;
        const bool prow = gr >= k0 && gr < k0 + nb;
        #pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gc = c0 + 16 * j + lr;
            if (gc >= n) continue;
            const bool zero = prow || (gc >= k0 && gc < k0 + nb);
            const cplx cin = zero ? cmake(0.0, 0.0) : Ta[(long long)gr * lda + gc];
            Tb[(long long)gr * ldb + gc] = csub(cin, cmake(cr[j][q], ci[j][q]));
        }
    }
}

// dst[mat][r][c] = src[mat][r][c] for n x n blocks with different leading dimensions (the odd step count of k_gj_step leaves the result in W)
__global__ __launch_bounds__(256) void k_copy_blocks(const cplx *src, int ld_src, long long ss, cplx *dst, int ldd, long long sd, int n) {
    const cplx *s = src + (long long)blockIdx.y * ss;
    cplx *d = dst + (long long)blockIdx.y * sd;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * 256) {
        const int r = (int)(e / n), c = (int)(e % n);
        d[(long long)r * ldd + c] = s[(long long)r * ld_src + c];
    }
}

__global__ __launch_bounds__(256) void k_gj_slices(const cplx *T0, int ld, long long stride, int n, int k0, int nb, cplx *Wc0, cplx *Wr0, long long wstride,
                                                   const cplx *Pb0, long long pstride) {
    __shared__ cplx P[PNB][PNB + 1];
    __shared__ cplx t[PNB][64 + 1];
    const cplx *T = T0 + (long long)blockIdx.y * stride;
    cplx *Wc = Wc0 + (long long)blockIdx.y * wstride, *Wr = Wr0 + (long long)blockIdx.y * wstride;
    const cplx *Pb = Pb0 + (long long)blockIdx.y * pstride;
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * 64;
    for (int e = tid; e < PNB * PNB; e += 256) {
        const int r = e >> 5, c = e & 31;
        P[r][c] = (r < nb && c < nb) ? Pb[e] : cmake(0.0, 0.0);
    }
    for (int e = tid; e < PNB * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        t[r][c] = (r < nb && s0 + c < n) ? T[(long long)(k0 + r) * ld + s0 + c] : cmake(0.0, 0.0);
    }
    for (int e = tid; e < 64 * PNB; e += 256) {
        const int r = e >> 5, c = e & 31;
        if (s0 + r >= n || c >= nb) continue;
        const int gr = s0 + r;
        cplx v = T[(long long)gr * ld + k0 + c];
        if (gr >= k0 && gr < k0 + nb) v = (gr - k0 == c) ? cmake(-1.0, 0.0) : cmake(0.0, 0.0);
        Wc[(long long)gr * PNB + c] = v;
    }
    __syncthreads();
    const int r = tid >> 3, c0 = (tid & 7) * 8;
    cplx acc[8];
    #pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = cmake(0.0, 0.0);
    for (int j = 0; j < nb; ++j) {
        const cplx a = P[r][j];
        #pragma unroll
        for (int i = 0; i < 8; ++i) cfma(acc[i], a, t[j][c0 + i]);
    }
    if (r < nb) {
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gc = s0 + c0 + i;
            if (gc >= n) continue;
            Wr[(long long)r * n + gc] = (gc >= k0 && gc < k0 + nb) ? P[r][gc - k0] : acc[i];
        }
    }
}

// ---- ill-conditioned fronts (NdStable) -------------------------------------------------------------------------------------------------
// infinity norm (largest absolute row sum, |z| taken as |re| + |im|) of the s x s pivot block of every front of a group -- before the
// inversion: F11, after it: F11^-1; their product is the condition estimate.  (The max-entry norm was tried first: for a near-singular
// front F11^-1 ~ u v^T / sigma with u, v spread over all unknowns, and max |entry| then underestimates the norm by the front's size --
// the worst front of the 8-Hz bench operator, cond 1.1e6, came out as 2.5e4 and stayed below the threshold.)
// (over the front's own s x s unknowns: the identity that pads a smaller front to the group's size is not part of its conditioning)
__global__ __launch_bounds__(256) void k_front_absmax(const cplx *M0, int ld, long long stride, const NdDev *nodes, double *out) {
    __shared__ double red[4];
    const cplx *M = M0 + (long long)blockIdx.x * stride;
    const int n = nodes[blockIdx.x].s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double best = 0.0;
    for (int i = wv; i < n; i += 4) {                      // a wave per row: coalesced along the row
        double v = 0.0;
        for (int j = lane; j < n; j += 64) { const cplx a = M[(long long)i * ld + j]; v += fabs(a.x) + fabs(a.y); }
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        v = __shfl(v, 0);
        best = fmax(best, v);
    }
    if (lane == 0) red[wv] = best;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
// list[0] = number of fronts with  scale * a[j] * b[j] > thr  (capped), list[1..] = their positions in the group, worst first is not needed
__global__ void k_front_flag(const double *a, const double *b, int cnt, double scale, double thr, int *list, int cap) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += gridDim.x * blockDim.x)
        if (!(scale * a[j] * b[j] <= thr)) {                                                               // (NaN counts as flagged)
            const int slot = atomicAdd(list, 1);
            if (slot < cap) list[1 + slot] = j;
        }
}
// ---- the leaf level of the factorisation in one kernel (round 4) ---------------------------------------------------------------------------
// A leaf front is [F11 F12; F21 0] with F11 the 9-point operator of an h x w block of cells (h, w <= 8: a BANDED matrix of half-width w + 1) and
// F12 / F21 the stencil across the block's boundary (at most three entries per ring cell).  What the passes need from it is dense --
// [F11^-1 | G = -F11^-1 F12] (s x (s + m)), G21 = F21 F11^-1 (m x s) and the Schur complement S = F21 G (m x m) -- but getting there does not
// have to be: the batched path wrote every leaf as a dense 81 x 81 front (1.7 GB at 1024^2), inverted F11 by a 25 + 24 block recursion of
// Gauss-Jordan sweeps and small products and formed G21, S and G with three more batched products: 3.0 ms of a 16-ms factorisation, and round 3's
// first fused attempt (a scalar dense Gauss-Jordan in LDS) was slower still (4.2 ms).  Here one workgroup of two waves takes a leaf:
//   1. the band of F11 (SP x 19) and the sparse F12 / F21 tables go straight from the coefficient planes into LDS -- no front in HBM;
//   2. wave 0 factors the band, LU without pivoting (a 9 x 9 window per step); a pivot below 1 % of its row's largest original entry flags
//      the leaf, which k_leaf_factor_pivoted then re-does with a row-pivoted dense Gauss-Jordan (indefinite leaves at few points per
//      wavelength; none at the bench's 17);
//   3. every thread takes ONE column of [I | -F12] and runs the banded forward / backward substitution on it in registers, the factor's
//      entries arriving as LDS broadcasts: 2 x 9 x SP complex multiply-adds per column instead of SP^2 for a dense inverse, 81 columns at once;
//      the result is a column of [F11^-1 | G], stored row by row (consecutive threads = consecutive addresses) and, for the block's boundary
//      cells, kept in LDS;
//   4. G21 and S are the three-term sums  sum_t F21[r][a_t] X[a_t][c]  over those LDS rows.
// SP: padded size of F11 (49 for blocks of up to 7 x 7 cells, 64 up to 8 x 8; identity on the padding).
#define LEAF_BW 9            // largest half-bandwidth (w + 1, w <= 8)
#define LEAF_NB 19           // band row: columns i - 9 .. i + 9
#define LEAF_MP 36           // largest ring of a leaf (2 (8 + 2) + 2 x 8)
#define LEAF_NBR 28          // most boundary cells of a leaf (8 x 8: 64 - 36)
struct LeafTabs {
    int idx[LEAF_MP][3];     // leaf cells (local index) adjacent to ring cell r, -1: none
    cplx f12[LEAF_MP][3];    // F12[idx][r]: row = the leaf cell
    cplx f21[LEAF_MP][3];    // F21[r][idx]: row = the ring cell
};
// band of F11, the boundary-row map and the sparse tables of leaf n into LDS (all threads of the workgroup call; no barrier inside)
template <int SP>
__device__ __forceinline__ void leaf_load(const NdDev &n, const cplx *planes, int nz, int nx, cplx *band, float *rowmax, int *brow, LeafTabs &T, int tid, int nthreads) {
    const long long N = (long long)nz * nx;
    const int w = n.x1 - n.x0, h = n.z1 - n.z0;
    for (int e = tid; e < SP * LEAF_NB; e += nthreads) band[e] = (e % LEAF_NB == LEAF_BW && e / LEAF_NB >= n.s) ? cmake(1.0, 0.0) : cmake(0.0, 0.0);
    for (int a = tid; a < SP; a += nthreads) {
        int br = -1;
        if (a < n.s) {
            const int lz = a / w, lx = a % w;
            if (lz == 0 || lz == h - 1 || lx == 0 || lx == w - 1) {          // rank among the boundary cells, in local order
                int cnt = 0;
                for (int q = 0; q < a; ++q) { const int qz = q / w, qx = q % w; if (qz == 0 || qz == h - 1 || qx == 0 || qx == w - 1) cnt += 1; }
                br = cnt;
            }
        }
        brow[a] = br;
        rowmax[a] = 1.0f;
    }
    for (int e = tid; e < LEAF_MP * 3; e += nthreads) { T.idx[e / 3][e % 3] = -1; T.f12[e / 3][e % 3] = cmake(0.0, 0.0); T.f21[e / 3][e % 3] = cmake(0.0, 0.0); }
}
template <int SP>
__device__ __forceinline__ void leaf_fill(const NdDev &n, const cplx *planes, int nz, int nx, cplx *band, float *rowmax, LeafTabs &T, int tid, int nthreads) {
    const long long N = (long long)nz * nx;
    const int w = n.x1 - n.x0;
    for (int a = tid; a < n.s; a += nthreads) {                              // one leaf cell per thread: its nine stencil entries
        const int z = n.z0 + a / w, x = n.x0 + a % w;
        float mx = 0.f;
        #pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int z2 = z + k / 3 - 1, x2 = x + k % 3 - 1;
            if (z2 < n.z0 || z2 >= n.z1 || x2 < n.x0 || x2 >= n.x1) continue;
            const cplx v = planes[(long long)k * N + (long long)z * nx + x];
            const int b = (z2 - n.z0) * w + (x2 - n.x0);
            band[a * LEAF_NB + (b - a) + LEAF_BW] = v;
            mx = fmaxf(mx, (float)fmax(fabs(v.x), fabs(v.y)));
        }
        rowmax[a] = mx;
    }
    for (int r = tid; r < n.m; r += nthreads) {                              // one ring cell per thread: its (at most three) neighbours inside the block
        int zr, xr;
        nd_cell(n, n.s + r, zr, xr);
        int t = 0;
        for (int k = 0; k < 9; ++k) {
            const int z2 = zr + k / 3 - 1, x2 = xr + k % 3 - 1;
            if (z2 < n.z0 || z2 >= n.z1 || x2 < n.x0 || x2 >= n.x1 || t >= 3) continue;
            const int a = (z2 - n.z0) * w + (x2 - n.x0);
            T.idx[r][t] = a;
            T.f21[r][t] = planes[(long long)k * N + (long long)zr * nx + xr];                        // row = ring cell, neighbour offset k
            T.f12[r][t] = planes[(long long)(8 - k) * N + (long long)z2 * nx + x2];                  // row = leaf cell, the opposite offset
            t += 1;
        }
    }
}

template <int SP>
__global__ __launch_bounds__(128, 4) void k_leaf_factor(const NdDev *nodes, int first, cplx *arenaF, cplx *fac, cplx *g21base, const cplx *planes, int nz, int nx, int *flags, int dbg) {
    __shared__ cplx band[SP * LEAF_NB];
    __shared__ LeafTabs T;
    __shared__ float rowmax[SP];
    __shared__ int brow[SP];
    __shared__ int bad;
    const NdDev n = nodes[first + blockIdx.x];
    const int tid = threadIdx.x;
    const int smax = n.smax, mmax = n.mmax, nmax = smax + mmax;
    if (tid == 0) bad = 0;
    leaf_load<SP>(n, planes, nz, nx, band, rowmax, brow, T, tid, 128);
    __syncthreads();
    leaf_fill<SP>(n, planes, nz, nx, band, rowmax, T, tid, 128);
    __syncthreads();
    if (tid < 64 && !(dbg & 1)) {                                            // wave 0: banded LU, no pivoting, window of 9 x 9 per step
        int mybad = 0;
        for (int k = 0; k < SP; ++k) {
            const cplx p = band[k * LEAF_NB + LEAF_BW];
            const double pm = fmax(fabs(p.x), fabs(p.y));
            if (!(pm >= 0.01 * (double)rowmax[k])) mybad = 1;
            const cplx pi = crecip(p);
            for (int e = tid; e < LEAF_BW * LEAF_BW; e += 64) {
                const int di = e / LEAF_BW + 1, dj = e % LEAF_BW + 1;
                const int i = k + di, j = k + dj;
                if (i < SP && j < SP) {
                    const cplx l = cmul(band[i * LEAF_NB + LEAF_BW - di], pi);
                    const cplx u = band[k * LEAF_NB + LEAF_BW + dj];
                    cplx &aij = band[i * LEAF_NB + (j - i) + LEAF_BW];
                    aij = csub(aij, cmul(l, u));
                }
            }
            // the multipliers of column k replace it (after every lane has read the old column: LDS serves a wave's instructions in order)
            if (tid < LEAF_BW && k + tid + 1 < SP) { cplx &l = band[(k + tid + 1) * LEAF_NB + LEAF_BW - (tid + 1)]; l = cmul(l, pi); }
            if (tid == LEAF_BW) band[k * LEAF_NB + LEAF_BW] = pi;            // (1 / pivot where the pivot was: no later step reads row k's diagonal)
        }
        if (mybad && tid == 0) bad = 1;
    }
    __syncthreads();
    if (tid == 0) flags[blockIdx.x] = (dbg & 8) ? 1 : bad;                  // (8: every leaf through the pivoted kernel -- a test)
    // ---- one column of [I | -F12] per thread through the banded substitutions.  Only a window of nine values lives in registers: the forward
    // pass parks its result in the column's own place in the factor storage, the backward pass picks it up from there (the thread's own stores,
    // L2-resident) and overwrites it with the solution -- ~70 registers, so that several leaves share a SIMD and cover each other's latencies
    // (a first version kept the whole column in registers: 400 of them, one wave per SIMD, and was no faster than the batched path).
    const int c = tid;
    const bool active = c < SP + mmax;
    const int gc = c < SP ? c : smax + (c - SP);                             // column in the [F11^-1 | G] rows of smax + mmax
    const bool stored = active && (c < SP ? c < smax : true);
    cplx *Fcol = fac + n.finv_off + gc;
    // right-hand side of this column: e_c, or -F12[:, c - SP] (three entries at most)
    int ra0 = -1, ra1 = -1, ra2 = -1;
    cplx rv0 = cmake(0.0, 0.0), rv1 = rv0, rv2 = rv0;
    if (c < SP) { ra0 = c; rv0 = cmake(1.0, 0.0); }
    else if (c - SP < n.m) {
        ra0 = T.idx[c - SP][0]; rv0 = cneg(T.f12[c - SP][0]);
        ra1 = T.idx[c - SP][1]; rv1 = cneg(T.f12[c - SP][1]);
        ra2 = T.idx[c - SP][2]; rv2 = cneg(T.f12[c - SP][2]);
    }
#define LEAF_RHS(i_) (ra0 == (i_) ? rv0 : (ra1 == (i_) ? rv1 : (ra2 == (i_) ? rv2 : cmake(0.0, 0.0))))
    // The factor's entries are the same for every column, i.e. wave-uniform: ONE ds_read per row brings the row's 19 band entries into lanes
    // 0..18 (multipliers in 0..8, 1 / pivot in 9, the upper row in 10..18), fetched a row ahead, and v_readlane hands each to the whole wave.
    const int lb = (tid & 63) < LEAF_NB ? (tid & 63) : LEAF_NB - 1;
    auto bcast = [](cplx v, int l) {
        cplx r;
        r.x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v.x), l), __builtin_amdgcn_readlane(__double2loint(v.x), l));
        r.y = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v.y), l), __builtin_amdgcn_readlane(__double2loint(v.y), l));
        return r;
    };
    if (!(dbg & 2)) {
        {
            cplx yw[SP];                                                     // (fully unrolled: each entry is live for nine rows only)
            cplx cur = band[0 * LEAF_NB + lb];
            #pragma unroll
            for (int i = 0; i < SP; ++i) {
                cplx nxt = cur;
                if (i + 1 < SP) nxt = band[(i + 1) * LEAF_NB + lb];
                cplx acc = LEAF_RHS(i);
                #pragma unroll
                for (int d = 1; d <= LEAF_BW; ++d)
                    if (i - d >= 0) { const cplx l = bcast(cur, LEAF_BW - d); acc.x = fma(l.y, yw[i - d].y, fma(-l.x, yw[i - d].x, acc.x)); acc.y = fma(-l.y, yw[i - d].x, fma(-l.x, yw[i - d].y, acc.y)); }
                yw[i] = acc;
                if (stored && i < smax) Fcol[(long long)i * nmax] = acc;
                cur = nxt;
            }
        }
        {
            constexpr int AHEAD = 6;                                         // rows fetched back ahead of the one being worked on
            cplx xw[SP], yq[SP];                                             // (fully unrolled: every entry is a value of its own, live for a few rows)
            #pragma unroll
            for (int i = SP - 1; i >= SP - AHEAD && i >= 0; --i) yq[i] = (stored && i < smax) ? Fcol[(long long)i * nmax] : LEAF_RHS(i);
            cplx cur = band[(SP - 1) * LEAF_NB + lb];
            #pragma unroll
            for (int i = SP - 1; i >= 0; --i) {
                if (i - AHEAD >= 0) yq[i - AHEAD] = (stored && i - AHEAD < smax) ? Fcol[(long long)(i - AHEAD) * nmax] : LEAF_RHS(i - AHEAD);
                cplx nxt = cur;
                if (i > 0) nxt = band[(i - 1) * LEAF_NB + lb];
                cplx acc = yq[i];
                #pragma unroll
                for (int d = 1; d <= LEAF_BW; ++d)
                    if (i + d < SP) { const cplx u = bcast(cur, LEAF_BW + d); acc.x = fma(u.y, xw[i + d].y, fma(-u.x, xw[i + d].x, acc.x)); acc.y = fma(-u.y, xw[i + d].x, fma(-u.x, xw[i + d].y, acc.y)); }
                acc = cmul(acc, bcast(cur, LEAF_BW));
                xw[i] = acc;
                if (stored && i < smax) Fcol[(long long)i * nmax] = acc;
                cur = nxt;
            }
        }
    }
    __syncthreads();                                                         // (every column of this leaf is in place: the other wave's too)
    // G21 = F21 F11^-1 (columns c < smax) and S = F21 G (columns SP .. SP + mmax): three-term sums over rows of [F11^-1 | G] read back
    if (active && !(dbg & 4)) {
        cplx *G21 = g21base + (long long)blockIdx.x * mmax * smax;
        cplx *S = arenaF + n.foff + smax;
        for (int r = 0; r < mmax; ++r) {
            cplx acc = cmake(0.0, 0.0);
            if (r < n.m && stored) {
                #pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int a = T.idx[r][t];
                    if (a >= 0) cfma(acc, T.f21[r][t], Fcol[(long long)a * nmax]);
                }
            }
            if (c < SP) { if (c < smax) G21[(long long)r * smax + c] = acc; }
            else S[(long long)r * nmax + (c - SP)] = acc;
        }
    }
}


// a flagged leaf once more, with row pivoting: dense Gauss-Jordan of F11 in LDS, then the same three sparse products
__global__ __launch_bounds__(256) void k_leaf_factor_pivoted(const NdDev *nodes, int first, cplx *arenaF, cplx *fac, cplx *g21base, const cplx *planes, int nz, int nx, const int *flags) {
    if (!flags[blockIdx.x]) return;
    __shared__ cplx a[GJ_MAX][GJ_MAX + 1];
    __shared__ cplx fcol[GJ_MAX];
    __shared__ int piv[GJ_MAX];
    __shared__ LeafTabs T;
    const NdDev n = nodes[first + blockIdx.x];
    const int tid = threadIdx.x;
    const int smax = n.smax, mmax = n.mmax, nmax = smax + mmax, w = n.x1 - n.x0;
    const long long N = (long long)nz * nx;
    for (int e = tid; e < GJ_MAX * GJ_MAX; e += 256) { const int i = e / GJ_MAX, j = e % GJ_MAX; a[i][j] = (i == j && i >= n.s) ? cmake(1.0, 0.0) : cmake(0.0, 0.0); }
    for (int e = tid; e < LEAF_MP * 3; e += 256) { T.idx[e / 3][e % 3] = -1; T.f12[e / 3][e % 3] = cmake(0.0, 0.0); T.f21[e / 3][e % 3] = cmake(0.0, 0.0); }
    __syncthreads();
    for (int q = tid; q < n.s; q += 256) {
        const int z = n.z0 + q / w, x = n.x0 + q % w;
        for (int k = 0; k < 9; ++k) {
            const int z2 = z + k / 3 - 1, x2 = x + k % 3 - 1;
            if (z2 < n.z0 || z2 >= n.z1 || x2 < n.x0 || x2 >= n.x1) continue;
            a[q][(z2 - n.z0) * w + (x2 - n.x0)] = planes[(long long)k * N + (long long)z * nx + x];
        }
    }
    for (int r = tid; r < n.m; r += 256) {
        int zr, xr;
        nd_cell(n, n.s + r, zr, xr);
        int t = 0;
        for (int k = 0; k < 9; ++k) {
            const int z2 = zr + k / 3 - 1, x2 = xr + k % 3 - 1;
            if (z2 < n.z0 || z2 >= n.z1 || x2 < n.x0 || x2 >= n.x1 || t >= 3) continue;
            T.idx[r][t] = (z2 - n.z0) * w + (x2 - n.x0);
            T.f21[r][t] = planes[(long long)k * N + (long long)zr * nx + xr];
            T.f12[r][t] = planes[(long long)(8 - k) * N + (long long)z2 * nx + x2];
            t += 1;
        }
    }
    __syncthreads();
    gj_lds<GJ_MAX>(a, fcol, piv, smax, tid, 256);                         // a = F11^-1 (identity on the padding)
    cplx *Frow = fac + n.finv_off;
    for (int e = tid; e < smax * nmax; e += 256) {
        const int i = e / nmax, cc = e % nmax;
        cplx v;
        if (cc < smax) v = a[i][cc];
        else {                                                               // G[i][b] = -sum_t F11^-1[i][a_t] F12[a_t][b]
            v = cmake(0.0, 0.0);
            const int b = cc - smax;
            if (b < n.m) for (int t = 0; t < 3; ++t) { const int q = T.idx[b][t]; if (q >= 0) cfma(v, a[i][q], cneg(T.f12[b][t])); }
        }
        Frow[(long long)i * nmax + cc] = v;
    }
    cplx *G21 = g21base + (long long)blockIdx.x * mmax * smax;
    cplx *S = arenaF + n.foff + smax;
    for (int e = tid; e < mmax * nmax; e += 256) {
        const int r = e / nmax, cc = e % nmax;
        cplx v = cmake(0.0, 0.0);
        if (r < n.m) {
            if (cc < smax) { for (int t = 0; t < 3; ++t) { const int q = T.idx[r][t]; if (q >= 0) cfma(v, T.f21[r][t], a[q][cc]); } }
            else {
                const int b = cc - smax;
                if (b < n.m)
                    for (int t = 0; t < 3; ++t) {
                        const int q = T.idx[r][t];
                        if (q < 0) continue;
                        cplx gqb = cmake(0.0, 0.0);
                        for (int u = 0; u < 3; ++u) { const int q2 = T.idx[b][u]; if (q2 >= 0) cfma(gqb, a[q][q2], cneg(T.f12[b][u])); }
                        cfma(v, T.f21[r][t], gqb);
                    }
            }
        }
        if (cc < smax) G21[(long long)r * smax + cc] = v; else S[(long long)r * nmax + (cc - smax)] = v;
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
#define LUS_NMAX 128
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
    cplx cur[NK], nxt[NK];
    auto load_row = [&](cplx (&dst)[NK], int i, int klo, int khi) {        // entries k in [klo, khi), k = p + 16 q
        #pragma unroll
        for (int q = 0; q < NK; ++q) { const int k = p + 16 * q; dst[q] = (i >= 0 && i < n && k >= klo && k < khi) ? LU[(long long)i * ld + k] : cmake(0.0, 0.0); }
    };
    // L y = P b (unit lower triangle): rows top-down
    load_row(cur, 1, 0, 1);
    for (int i = 1; i < n; ++i) {
        load_row(nxt, i + 1, 0, i + 1);
        cplx acc = cmake(0.0, 0.0);
        #pragma unroll
        for (int q = 0; q < NK; ++q) { const int k = p + 16 * q; if (k < i) cfma(acc, cur[q], Bt[k][cl]); }
        #pragma unroll
        for (int off = 8; off > 0; off >>= 1) { acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); }
        if (p == 0) Bt[i][cl] = csub(Bt[i][cl], acc);
        __builtin_amdgcn_wave_barrier();
        #pragma unroll
        for (int q = 0; q < NK; ++q) cur[q] = nxt[q];
    }
    // U x = y: rows bottom-up
    load_row(cur, n - 1, n - 1, n);
    for (int i = n - 1; i >= 0; --i) {
        load_row(nxt, i - 1, i - 1, n);
        cplx acc = cmake(0.0, 0.0), d = cmake(1.0, 0.0);
        #pragma unroll
        for (int q = 0; q < NK; ++q) { const int k = p + 16 * q; if (k > i && k < n) cfma(acc, cur[q], Bt[k][cl]); if (k == i) d = cur[q]; }
        #pragma unroll
        for (int off = 8; off > 0; off >>= 1) { acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); }
        // the diagonal entry sits with the lane whose k == i: hand it to lane 0 of the group
        const int src = (lane & 48) | (i & 15);
        const double dx = __shfl(d.x, src), dy = __shfl(d.y, src);
        if (p == 0) Bt[i][cl] = cmul(csub(Bt[i][cl], acc), crecip(cmake(dx, dy)));
        __builtin_amdgcn_wave_barrier();
        #pragma unroll
        for (int q = 0; q < NK; ++q) cur[q] = nxt[q];
    }
    __syncthreads();
    for (int e = tid; e < n * 16; e += 256) { const int i = e >> 4, c = e & 15; if (j0 + c < ncols) B[(long long)i * ldb + j0 + c] = Bt[i][c]; }
}

// ---- solve-phase data movement -----------------------------------------------------------------------------------
// out[i][r] = in[r][i]   (in: rows x cols)
// (the long dimension always rides on gridDim.x: `swap` exchanges the roles of blockIdx.x / blockIdx.y)
__global__ __launch_bounds__(256) void k_transpose(const cplx *in, long long rows, long long cols, cplx *out, int swap, int conj = 0) {
    __shared__ cplx t[32][33];
    const long long c0 = (long long)(swap ? blockIdx.y : blockIdx.x) * 32, r0 = (long long)(swap ? blockIdx.x : blockIdx.y) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) t[j][tx] = in[(r0 + j) * cols + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) out[(c0 + j) * rows + r0 + tx] = conj ? cconj(t[tx][j]) : t[tx][j];
}

// ---- node-major pipeline around the solve --------------------------------------------------------------------------------
// The triangular solves want the right-hand sides node-major, Xt[cell][rhs].  Everything between the caller's rhs-major
// buffers and the solves stays in that layout: the right-hand-side preparation is fused into the transpose-in, the true
// residual q - A x is evaluated node-major (one lane per right-hand side, the nine coefficients of a cell are uniform
// across the lanes), refinement passes solve on the residual where it lies, and only the final result is transposed out.
//
// Qt[i][r] = premul * rhs[r][row_off + i] - sub[r][i]   and the partials of ||q_r||^2: part[(r * 4) * nblk + block]
__global__ __launch_bounds__(256) void k_prep_transpose_norm(const cplx *__restrict__ rhs, long long rhs_ld, long long row_off, cplx premul,
                                                             const cplx *__restrict__ sub, cplx *__restrict__ Qt, long long N, int nrhs,
                                                             double *__restrict__ part, int nblk) {
    __shared__ cplx t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    const long long ntile = (N + 31) / 32;
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long c0 = tile * 32;
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = ty + 8 * q;
            if (r0 + j < nrhs && c0 + tx < N) {
                cplx v = cmul(premul, rhs[(long long)(r0 + j) * rhs_ld + row_off + c0 + tx]);
                if (sub) v = csub(v, sub[(long long)(r0 + j) * N + c0 + tx]);
                t[j][tx] = v;
                s[q] += cabs2(v);
            }
        }
        __syncthreads();
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = ty + 8 * q;
            if (c0 + j < N && r0 + tx < nrhs) Qt[(c0 + j) * nrhs + r0 + tx] = t[tx][j];
        }
        __syncthreads();
    }
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = s[q];
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);       // over the 32 cells of the tile row (half a wave)
        const int j = ty + 8 * q;
        if (tx == 0 && r0 + j < nrhs) part[((long long)(r0 + j) * 4) * nblk + blockIdx.x] = v;
    }
}

// Node-major stencil with residual epilogue.  Lane = right-hand side j (blockDim.x lanes), blockDim.y row segments per workgroup;
// a thread marches along x over `seg` cells of one grid row with a 3 x 3 register window of the input columns.
//   in  : Xin[cell * ldin + j]                                  (the solution, or a refinement correction)
//   q   : Q[cell * ldq + map(j)]   (map = qmap[j] or j)           r = q - A xin
//   store != 0: r written to Rout (same indexing as q; Rout == null: over q)
//   part[(j * 4) * nblk + block] = partial ||r_j||^2
//   qnorm != 0: part[(j * 4 + 1) * nblk + block] = partial ||q_j||^2 (node-major callers: q is read here anyway, no separate norm pass)
//   Uout != null: Uout[cell * ldu + j] = conj(oscale * xin[cell][j]) -- the wavefield in the reference's (N, nrhs) layout and sign
//                 convention (discretization.py:101-103), written by the launch that checks it (one write instead of a read + write pass)
template <int RPT>
__global__ __launch_bounds__(256) void k_resid_nm(const cplx *__restrict__ planes, int nz, int nx, const cplx *__restrict__ Xin, int ldin,
                                                  cplx *__restrict__ Q, int ldq, const int *__restrict__ qmap, int ncol, int store,
                                                  cplx *__restrict__ Rout, double *__restrict__ part, int nblk, int seg, int ntiles,
                                                  int qnorm, cplx *__restrict__ Uout, int ldu, cplx oscale) {
    // RPT grid rows per thread: the window is (RPT + 2) x 3, so a step along x loads RPT + 2 values for RPT outputs and the rows a tile
    // shares with the tiles above and below (the only HBM re-reads of this kernel: 1.7 x the input at RPT = 1 by the PMC counters) shrink
    // from 2 per output row to 2 / RPT
    __shared__ double red[256];
    // blockDim.x is a multiple of the wave size, so threadIdx.y -- and with it the tile, the row and the cell a thread works on -- is uniform
    // across a wave: saying so (readfirstlane) turns the nine coefficient loads per cell into scalar loads through the constant cache
    // instead of 64 lanes fetching the same 16 bytes through the vector memory pipeline (36 of the 46 loads of a step at RPT = 4)
    const int j = threadIdx.x, ly = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const bool act = j < ncol;
    const int col = act ? (qmap ? qmap[j] : j) : 0;
    const long long N = (long long)nz * nx;
    const int nzt = (nz + RPT - 1) / RPT;
    double acc = 0.0, accq = 0.0;
    // tile order: workgroup b serves band (b % 8) of the tile list, so that the workgroups of one XCD (b, b + 8, ...) walk
    // z-adjacent row segments together and the halo rows are served by that XCD's L2
    const int per = (ntiles + 7) / 8;
    for (int w = blockIdx.x * blockDim.y + ly; w < per * 8; w += gridDim.x * blockDim.y) {
        const int t = (w & 7) * per + (w >> 3);
        if (t >= ntiles) continue;
        if (!act) continue;
        const int sgi = t / nzt, z0 = (t - sgi * nzt) * RPT;      // z fastest: consecutive tiles are vertically adjacent
        const int x0 = sgi * seg, x1 = min(nx, x0 + seg);
        cplx win[RPT + 2][3];                                     // win[d][.] = columns x-1, x, x+1 of row z0-1+d
        // software pipeline: the column that enters the window in the NEXT step (`pre`) and the next column of q (`qn`) are loaded
        // while the current column is being multiplied, so a wave never waits on the loads it has just issued
        cplx pre[RPT + 2], qn[RPT];
        #pragma unroll
        for (int d = 0; d < RPT + 2; ++d) {
            const int zz = z0 - 1 + d;
            const bool zin = zz >= 0 && zz < nz;
            win[d][0] = cmake(0.0, 0.0);
            win[d][1] = (zin && x0 - 1 >= 0) ? Xin[((long long)zz * nx + x0 - 1) * ldin + j] : cmake(0.0, 0.0);
            win[d][2] = zin ? Xin[((long long)zz * nx + x0) * ldin + j] : cmake(0.0, 0.0);
            pre[d] = (zin && x0 + 1 < nx) ? Xin[((long long)zz * nx + x0 + 1) * ldin + j] : cmake(0.0, 0.0);
        }
        #pragma unroll
        for (int o = 0; o < RPT; ++o) qn[o] = (z0 + o < nz) ? Q[((long long)(z0 + o) * nx + x0) * ldq + col] : cmake(0.0, 0.0);
        for (int x = x0; x < x1; ++x) {
            cplx qc[RPT];
            #pragma unroll
            for (int d = 0; d < RPT + 2; ++d) {
                const int zz = z0 - 1 + d;
                win[d][0] = win[d][1]; win[d][1] = win[d][2]; win[d][2] = pre[d];
                cplx v = cmake(0.0, 0.0);
                if (zz >= 0 && zz < nz && x + 1 < x1 && x + 2 < nx) v = Xin[((long long)zz * nx + x + 2) * ldin + j];
                pre[d] = v;
            }
            #pragma unroll
            for (int o = 0; o < RPT; ++o) {
                qc[o] = qn[o];
                cplx v = cmake(0.0, 0.0);
                if (z0 + o < nz && x + 1 < x1) v = Q[((long long)(z0 + o) * nx + x + 1) * ldq + col];
                qn[o] = v;
            }
            #pragma unroll
            for (int o = 0; o < RPT; ++o) {
                const int z = z0 + o;
                if (z >= nz) break;
                const long long cell = (long long)z * nx + x;
                cplx r = qc[o];
                if (qnorm) accq += cabs2(r);
                #pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const cplx c = planes[(long long)k * N + cell];
                    const cplx xv = win[o + k / 3][k % 3];
                    r.x = fma(-c.x, xv.x, r.x); r.x = fma(c.y, xv.y, r.x);
                    r.y = fma(-c.x, xv.y, r.y); r.y = fma(-c.y, xv.x, r.y);
                }
                if (store) (Rout ? Rout : Q)[cell * ldq + col] = r;
                if (Uout) Uout[cell * ldu + j] = cconj(cmul(oscale, win[o + 1][1]));
                acc += cabs2(r);
            }
        }
    }
    if (blockDim.y > 1) {
        red[ly * blockDim.x + j] = acc;
        __syncthreads();
        if (ly == 0) for (int q = 1; q < (int)blockDim.y; ++q) acc += red[q * blockDim.x + j];
        if (qnorm) {
            __syncthreads();
            red[ly * blockDim.x + j] = accq;
            __syncthreads();
            if (ly == 0) for (int q = 1; q < (int)blockDim.y; ++q) accq += red[q * blockDim.x + j];
        }
    }
    if (ly == 0 && act) {
        part[((long long)j * 4) * nblk + blockIdx.x] = acc;
        if (qnorm) part[((long long)j * 4 + 1) * nblk + blockIdx.x] = accq;
    }
}

// The same kernel for full-width batches (blockDim = (256, 1): the four waves of a workgroup share their tile).  The nine coefficients of
// the tile's RPT x 32 cells are staged in LDS once per tile by coalesced loads along x (18 KB at RPT = 4) and read back as broadcasts:
// the per-cell coefficient fetches of k_resid_nm -- 36 of its 46 memory instructions per step at RPT = 4, each a 16-byte request -- leave
// the vector memory pipeline, which then only carries the streams that have to move (x, q, and what is stored).
#define RESID_SEG 32
template <int RPT, int NT_STORE>
__global__ __launch_bounds__(256) void k_resid_nm_lds(const cplx *__restrict__ planes, int nz, int nx, const cplx *__restrict__ Xin, int ldin,
                                                      cplx *__restrict__ Q, int ldq, const int *__restrict__ qmap, int ncol, int store,
                                                      cplx *__restrict__ Rout, double *__restrict__ part, int nblk, int ntiles,
                                                      int qnorm, cplx *__restrict__ Uout, int ldu, cplx oscale, const unsigned char *__restrict__ qm) {
    __shared__ cplx cs[9][RPT][RESID_SEG];
    const int j = threadIdx.x;
    const bool act = j < ncol;
    const int col = act ? (qmap ? qmap[j] : j) : 0;
    const long long N = (long long)nz * nx;
    const int nzt = (nz + RPT - 1) / RPT;
    double acc = 0.0, accq = 0.0;
    const int per = (ntiles + 7) / 8;
    for (int w = blockIdx.x; w < per * 8; w += gridDim.x) {
        const int t = (w & 7) * per + (w >> 3);                   // banded tile order, see k_resid_nm
        if (t >= ntiles) continue;                                // (uniform across the workgroup)
        const int sgi = t / nzt, z0 = (t - sgi * nzt) * RPT;
        const int x0 = sgi * RESID_SEG, x1 = min(nx, x0 + RESID_SEG);
        __syncthreads();                                          // the previous tile's coefficients are no longer being read
        for (int e = j; e < 9 * RPT * RESID_SEG; e += 256) {
            const int xx = e % RESID_SEG, o = (e / RESID_SEG) % RPT, k = e / (RESID_SEG * RPT);
            cplx v = cmake(0.0, 0.0);
            if (z0 + o < nz && x0 + xx < nx) v = planes[(long long)k * N + (long long)(z0 + o) * nx + x0 + xx];
            cs[k][o][xx] = v;
        }
        __syncthreads();
        if (!act) continue;
        cplx win[RPT + 2][3], pre[RPT + 2], qn[RPT];
        #pragma unroll
        for (int d = 0; d < RPT + 2; ++d) {
            const int zz = z0 - 1 + d;
            const bool zin = zz >= 0 && zz < nz;
            win[d][0] = cmake(0.0, 0.0);
            win[d][1] = (zin && x0 - 1 >= 0) ? Xin[((long long)zz * nx + x0 - 1) * ldin + j] : cmake(0.0, 0.0);
            win[d][2] = zin ? Xin[((long long)zz * nx + x0) * ldin + j] : cmake(0.0, 0.0);
            pre[d] = (zin && x0 + 1 < nx) ? Xin[((long long)zz * nx + x0 + 1) * ldin + j] : cmake(0.0, 0.0);
        }
        // qm (sparse right-hand sides): a byte per cell, bit = this wave's block of 64 columns may hold a nonzero there; a 0 bit means q is not read.
        // The bytes run one column ahead of the q loads they gate (mk: column x + 1, fetched while column x is worked on).
        const int qbit = j >> 6;
        unsigned mk[RPT];
        #pragma unroll
        for (int o = 0; o < RPT; ++o) {
            const bool in = z0 + o < nz;
            const unsigned m0 = (qm && in) ? qm[(long long)(z0 + o) * nx + x0] : 0xFFu;
            qn[o] = (in && ((m0 >> qbit) & 1)) ? Q[((long long)(z0 + o) * nx + x0) * ldq + col] : cmake(0.0, 0.0);
            mk[o] = (qm && in && x0 + 1 < x1) ? qm[(long long)(z0 + o) * nx + x0 + 1] : 0xFFu;
        }
        for (int x = x0; x < x1; ++x) {
            cplx qc[RPT];
            #pragma unroll
            for (int d = 0; d < RPT + 2; ++d) {
                const int zz = z0 - 1 + d;
                win[d][0] = win[d][1]; win[d][1] = win[d][2]; win[d][2] = pre[d];
                cplx v = cmake(0.0, 0.0);
                if (zz >= 0 && zz < nz && x + 1 < x1 && x + 2 < nx) v = Xin[((long long)zz * nx + x + 2) * ldin + j];
                pre[d] = v;
            }
            #pragma unroll
            for (int o = 0; o < RPT; ++o) {
                qc[o] = qn[o];
                cplx v = cmake(0.0, 0.0);
                if (z0 + o < nz && x + 1 < x1 && ((mk[o] >> qbit) & 1)) v = Q[((long long)(z0 + o) * nx + x + 1) * ldq + col];
                qn[o] = v;
                mk[o] = (qm && z0 + o < nz && x + 2 < x1) ? qm[(long long)(z0 + o) * nx + x + 2] : 0xFFu;
            }
            #pragma unroll
            for (int o = 0; o < RPT; ++o) {
                const int z = z0 + o;
                if (z >= nz) break;
                const long long cell = (long long)z * nx + x;
                cplx r = qc[o];
                if (qnorm) accq += cabs2(r);
                #pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const cplx c = cs[k][o][x - x0];
                    const cplx xv = win[o + k / 3][k % 3];
                    r.x = fma(-c.x, xv.x, r.x); r.x = fma(c.y, xv.y, r.x);
                    r.y = fma(-c.x, xv.y, r.y); r.y = fma(-c.y, xv.x, r.y);
                }
                if (store) (Rout ? Rout : Q)[cell * ldq + col] = r;
                if (Uout) {                                   // written once, read by nobody on the GPU: past the caches
                    const cplx u = cconj(cmul(oscale, win[o + 1][1]));
                    if (NT_STORE) __builtin_nontemporal_store((v2f64){u.x, u.y}, reinterpret_cast<v2f64 *>(Uout + cell * ldu + j));
                    else Uout[cell * ldu + j] = u;
                }
                acc += cabs2(r);
            }
        }
    }
    if (act) {
        part[((long long)j * 4) * nblk + blockIdx.x] = acc;
        if (qnorm) part[((long long)j * 4 + 1) * nblk + blockIdx.x] = accq;
    }
}

// cellnode[cell] = front of the leaf that eliminates the cell (rows of a leaf group's table with the separator flag)
__global__ __launch_bounds__(256) void k_nd_cellnode(const int4 *tab, long long rows, int nmax, int first, int *cellnode) {
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
        const int4 e = tab[r];
        if (e.w && e.x >= 0) cellnode[e.x] = first + (int)(r / nmax);
    }
}
// mask[cell]: bit b = the right-hand sides may be nonzero at this cell in block b of 64 columns (leaf cells: the leaf's flag; separator cells: always)
__global__ __launch_bounds__(256) void k_nd_qmask(const int *cellnode, const int *act, int nct, long long N, unsigned char *mask) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (long long)gridDim.x * blockDim.x) {
        const int nd = cellnode[c];
        unsigned m = 0xFF;
        if (nd >= 0) { m = 0; for (int b = 0; b < nct && b < 8; ++b) if (act[(long long)nd * nct + b]) m |= 1u << b; }
        mask[c] = (unsigned char)m;
    }
}

// Xt[cell][cols[j]] += Dp[cell][j]  (corrections of the packed minority batch back into the full batch)
__global__ __launch_bounds__(256) void k_scatter_add_cols(cplx *__restrict__ Xt, int ldq, const int *__restrict__ cols, int k, const cplx *__restrict__ Dp, long long N) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < N * k; e += (long long)gridDim.x * blockDim.x) {
        const long long cell = e / k; const int j = (int)(e - cell * k);
        cplx *x = Xt + cell * ldq + cols[j];
        *x = cadd(*x, Dp[e]);
    }
}

// Rp[cell][j] = Qt[cell][cols[j]]  (the right-hand sides that need another pass, packed to a narrower batch)
__global__ __launch_bounds__(256) void k_pack_cols(const cplx *__restrict__ Qt, int ldq, const int *__restrict__ cols, int k, cplx *__restrict__ Rp, long long N) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < N * k; e += (long long)gridDim.x * blockDim.x) {
        const long long cell = e / k; const int j = (int)(e - cell * k);
        Rp[e] = Qt[cell * ldq + cols[j]];
    }
}

// row table (see NdPlanDev): one thread per padded row of the group's fronts
__global__ __launch_bounds__(256) void k_nd_build_tab(const NdDev *nodes, int first, int4 *tab, int nz, int nx) {
    const NdDev n = nodes[first + blockIdx.y];
    const int nmax = n.smax + n.mmax;
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < nmax; row += gridDim.x * blockDim.x) {
        int a = -1;
        if (row < n.s) a = row;
        else if (row >= n.smax && row - n.smax < n.m) a = n.s + row - n.smax;
        int4 e = make_int4(-1, -1, -1, row < n.s ? 1 : 0);
        if (a >= 0) {
            int z, x, comp;
            nd_cell(n, a, z, x, comp);
            e.x = comp * nz * nx + z * nx + x;            // row of Xt: the fields are stacked [u; v] like the right-hand sides
            for (int k = 0; k < 2; ++k) {
                if (n.kid[k] < 0) continue;
                const NdDev c = nodes[n.kid[k]];
                const int la = nd_local(c, nz, nx, z, x, comp);
                if (la >= c.s) { const int src = (int)(c.voff + c.smax + (la - c.s)); if (k == 0) e.y = src; else e.z = src; }
            }
        }
        tab[n.roff + row] = e;
    }
}

// forward pass, one group: V[row] = [separator row: Xt[cell]] + outgoing rows of the children; separator rows of
// non-leaf fronts are final (y_S) and written back to Xt.  blockDim = (LX, 256 / LX), LX lanes over the right-hand sides.
// act (may be null): flags of the sparse-right-hand-side forward pass (GemmRows::act) -- a child's rows count only where its flag is set;
// nodes / first / nmax: the fronts these rows belong to (row / nmax-th front from `first`)
__global__ __launch_bounds__(256) void k_nd_fwd_rows(const int4 *tab, cplx *V, const cplx *arenaV, const cplx *Qt, cplx *Xt, long long rows, int nrhs, int write_back,
                                                     const int *act = nullptr, int nct = 0, const NdDev *nodes = nullptr, int first = 0, int nmax = 1) {
    for (long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y; row < rows; row += (long long)gridDim.x * blockDim.y) {
        const int4 e = tab[row];
        cplx *dst = V + row * nrhs;
        const cplx *s0 = e.w ? Qt + (long long)e.x * nrhs : nullptr;
        const cplx *s1 = e.y >= 0 ? arenaV + (long long)e.y * nrhs : nullptr;
        const cplx *s2 = e.z >= 0 ? arenaV + (long long)e.z * nrhs : nullptr;
        const int *a1 = nullptr, *a2 = nullptr;
        if (act) { const NdDev nd = nodes[first + (int)(row / nmax)]; a1 = nd.kid[0] >= 0 ? act + (long long)nd.kid[0] * nct : nullptr; a2 = nd.kid[1] >= 0 ? act + (long long)nd.kid[1] * nct : nullptr; }
        for (int r = threadIdx.x; r < nrhs; r += blockDim.x) {
            cplx acc = s0 ? s0[r] : cmake(0.0, 0.0);
            if (s1 && (!act || (a1 && a1[r >> 6]))) acc = cadd(acc, s1[r]);
            if (s2 && (!act || (a2 && a2[r >> 6]))) acc = cadd(acc, s2[r]);
            dst[r] = acc;
            if (write_back && e.w) Xt[(long long)e.x * nrhs + r] = acc;
        }
    }
}

// backward pass: V[row] = Xt[cell of the row] (separator and ring rows), 0 for padding
__global__ __launch_bounds__(256) void k_nd_bwd_gather(const int4 *tab, cplx *V, const cplx *XS, const cplx *Xt, long long rows, int nrhs) {
    // XS: where the separator rows' y_S lives (the right-hand sides themselves for leaves, Xt otherwise)
    for (long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y; row < rows; row += (long long)gridDim.x * blockDim.y) {
        const int4 e = tab[row];
        cplx *dst = V + row * nrhs;
        const cplx *src = e.x >= 0 ? (e.w ? XS : Xt) + (long long)e.x * nrhs : nullptr;
        for (int r = threadIdx.x; r < nrhs; r += blockDim.x) dst[r] = src ? src[r] : cmake(0.0, 0.0);
    }
}

// backward pass: Xt[separator cells] = XS (cnt x smax rows)
__global__ __launch_bounds__(256) void k_nd_bwd_store(const int4 *tab, const cplx *XS, cplx *Xt, long long rows, int smax, int nmax, int nrhs) {
    for (long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y; row < rows; row += (long long)gridDim.x * blockDim.y) {
        const long long j = row / smax;
        const int a = (int)(row - j * smax);
        const int4 e = tab[j * nmax + a];
        if (!e.w) continue;
        const cplx *src = XS + row * nrhs;
        cplx *dst = Xt + (long long)e.x * nrhs;
        for (int r = threadIdx.x; r < nrhs; r += blockDim.x) dst[r] = src[r];
    }
}

// y += x, or y += conj(x) when y holds the conjugated wavefield  (refinement update), n elements
__global__ void k_axpy_one(cplx *y, const cplx *x, long long n, int conj) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = cadd(y[i], conj ? cconj(x[i]) : x[i]);
}

int g_gemm_variant = -1;        // >= 0: overrides HELM_ND_GEMMV (helm_debug_zgemm_bench)
int g_gemm_tile = -1;           // >= 0: forces the tile configuration (helm_debug_zgemm_bench)
// C = beta C + alpha (sum of the ksplit partial products of a split launch); parts: [chunk][matrix][M x Nn]
__global__ __launch_bounds__(256) void k_splitk_reduce(const cplx *__restrict__ parts, int ksplit, long long pstride, int M, int Nn, cplx alpha, cplx beta,
                                                       cplx *__restrict__ C, int ldc, long long sc, long long total) {
    const bool rd = !(beta.x == 0.0 && beta.y == 0.0);
    const long long per = (long long)M * Nn;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        cplx sum = parts[e];
        for (int k = 1; k < ksplit; ++k) sum = cadd(sum, parts[(long long)k * pstride + e]);
        const long long b = e / per, rem = e - b * per;
        cplx *dst = C + b * sc + (rem / Nn) * ldc + rem % Nn;
        cplx o = cmul(alpha, sum);
        if (rd) o = cadd(o, cmul(beta, *dst));
        *dst = o;
    }
}

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

int gemm_variant() {
    // 0: first-generation kernel (kept for before / after comparisons); 1: second-generation kernel, K slab 8; 6: K slab 32 on the 1-column
    // tiles.  Measured and dropped (profiles/r02_zgemm_lab_variants.txt): K slab 16 (occupancy 2), k loop unrolled twice (+1-2 %, 166 VGPRs),
    // a 4-waves-per-SIMD register budget (spills), and a 3M complex product (48 FMAs for 64 per k step but 192 VGPRs and twelve more LDS
    // reads: 3 % SLOWER on the large shapes, and its rounding pushed two more of the 16 bench frequencies over rtol into a second pass)
    // 7 (round 4, the default): the matrix-core kernel k_zgemm3 (v_mfma_f64_16x16x4_f64) -- 1.2-1.4 x the vector kernel on every compute-bound shape
    static const int gemm_v = getenv("HELM_ND_GEMMV") ? atoi(getenv("HELM_ND_GEMMV")) : 7;
    return g_gemm_variant >= 0 ? g_gemm_variant : gemm_v;
}

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

// op may be null (diagnostic entry points): default stream, no profiling
int gemm(helm_op *op, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
         cplx beta, cplx *C, int ldc, long long sc, int batch, const GemmRows *rows = nullptr) {
    if (M <= 0 || Nn <= 0 || batch <= 0) return 0;
    hipStream_t st = op ? op->stream : nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // profiling mode 1 (default): every launch carries its own start / stop events (ext launch, see ZG_LAUNCH); 0: hipEventRecord markers
    // around launches / runs of launches (round 1)
    static const int prof_ext = getenv("HELM_PROF_EXT") ? atoi(getenv("HELM_PROF_EXT")) : 1;
    const bool ext = prof_ext && op && op->profiling && gemm_variant() != 0;
    const bool in_run = !ext && op && op->gemm_run_depth > 0;
    if (!ext && op && op->profiling && !(in_run && op->gemm_run_pair >= 0)) {
        if (op->ev_used + 2 > op->ev_pool.size() && op->ev_pool.size() < 16384)
            (void)helm_events_grow(op, 64);
        if (op->ev_used + 2 <= op->ev_pool.size()) { e0 = op->ev_pool[op->ev_used]; e1 = op->ev_pool[op->ev_used + 1]; }
    }
    if (e0) hipEventRecord(e0, st);
    if (in_run && e0) { op->gemm_run_pair = (int)op->ev_used; op->ev_used += 2; }
    // Products with at most 16 columns over a few big matrices (the top of the 3-D coarse level's passes: 16 right-hand sides through fronts of
    // thousands of unknowns) fill a few dozen 128 x 16 tiles: the inner dimension is split over up to 16 workgroups per tile, each writes its
    // partial product to the handle's scratch and a small launch adds them up (with alpha / beta applied there).
    static const int splitk = getenv("HELM_ND_SPLITK") ? atoi(getenv("HELM_ND_SPLITK")) : 1;
    bool split_done = false;
    if (splitk && op && !rows && !ext && Nn <= 16 && K >= 1024 && gemm_variant() != 0 && batch <= 64) {
        const long long tiles = (long long)batch * ((M + 127) / 128);
        static const int sk_tiles = getenv("HELM_ND_SPLITK_TILES") ? atoi(getenv("HELM_ND_SPLITK_TILES")) : 300;
        static const int sk_wgs = getenv("HELM_ND_SPLITK_WGS") ? atoi(getenv("HELM_ND_SPLITK_WGS")) : 768;      // (measured on the 47 x 79 x 79 level: 96 / 192 -> 5.8 ms per coarse solve, 300 / 768 -> 5.0, more changes nothing)
        if (tiles < sk_tiles) {
            const int ks = (int)std::min<long long>(16, std::max<long long>(2, sk_wgs / tiles));
            const int kc = (((K + ks - 1) / ks) + 7) & ~7;
            const long long per = (long long)batch * M * Nn;
            const size_t need = (size_t)ks * per * sizeof(cplx);
            if (op->sk_bytes < need) {
                if (op->sk_buf) { hipStreamSynchronize(st); helm_pool_free(op->device, op->sk_buf, op->sk_bytes); op->sk_buf = nullptr; op->sk_bytes = 0; }
                op->sk_buf = (cplx *)helm_pool_alloc(op->device, need);
                op->sk_bytes = op->sk_buf ? need : 0;
            }
            if (op->sk_buf) {
                GemmRows R; R.dense = 1; R.ksplit = ks; R.kc = kc; R.pstride = per;
                if (gemm_variant() == 7) launch_mfma<4, 1, 2, 1, 8>(st, 0, batch * ks, M, Nn, K, cmake(1, 0), A, lda, sa, B, ldb, sb, cmake(0, 0), op->sk_buf, Nn, (long long)M * Nn, R);
                else launch_vec2<128, 2, 8, 1, 1>(st, 0, batch * ks, M, Nn, K, cmake(1, 0), A, lda, sa, B, ldb, sb, cmake(0, 0), op->sk_buf, Nn, (long long)M * Nn, R);
                hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)std::min<long long>((per + 255) / 256, 4096)), dim3(256), 0, st, (const cplx *)op->sk_buf, ks, per, M, Nn, alpha, beta,
                                   C, ldc, sc, per);
                split_done = true;
            }
        }
    }
    static const int fixed_tm = getenv("HELM_ND_TM") ? atoi(getenv("HELM_ND_TM")) : 0;
    // tile of the vector kernel: 4 x RN outputs per thread, TM x TN = TM x (1024 / TM * RN).  The padded area is weighed by how
    // well a register block re-uses its LDS reads (4 x 4: 1, 4 x 2: 0.7, 4 x 1: 0.45): narrow tiles only win on small outputs
    // (the 32 x 32 blocks at the bottom of the inversion recursion, 17- and 35-wide separators of the middle tree levels)
    // (the 128 x 16 tile: products with at most 16 columns -- the solve passes of the 3-D coarse level push 16 right-hand sides through fronts of
    // thousands of unknowns -- where the 64 x 32 tile leaves half of its columns idle)
    static const int vc_tm[9] = {64, 32, 16, 64, 32, 16, 32, 16, 128}, vc_rn[9] = {4, 4, 4, 2, 2, 2, 1, 1, 2};
    static const double w2 = getenv("HELM_ND_EFF2") ? atof(getenv("HELM_ND_EFF2")) : 0.7, w1 = getenv("HELM_ND_EFF1") ? atof(getenv("HELM_ND_EFF1")) : 0.45;
    static const double vc_eff[9] = {1.0, 1.0, 1.0, w2, w2, w2, w1, w1, w2};
    static const int narrow = getenv("HELM_ND_NARROW") ? atoi(getenv("HELM_ND_NARROW")) : 1;
    int vsel = 0; double vcost = -1;
    static const int tall = getenv("HELM_ND_TALL") ? atoi(getenv("HELM_ND_TALL")) : 1;
    for (int c = 0; c < (narrow ? (tall ? 9 : 8) : 3); ++c) {
        if (c == 8 && (Nn > 16 || gemm_variant() == 0)) continue;
        const int tm = vc_tm[c], tn = 1024 / tm * vc_rn[c];
        double cost = (double)((M + tm - 1) / tm) * tm * ((Nn + tn - 1) / tn) * tn / vc_eff[c];
        // matrix-core kernel: a 16-row tile is one block row per wave column -- right for fronts of 8 and 16 rows, 40-50 % slower than the
        // 64- and 32-row tiles on tall operands, where only the padding of an odd row count (1025, 1281) made it look cheap (tools/tile_lab.py)
        if (gemm_variant() == 7 && tm == 16 && M > 32) cost *= 1.6;
        if (vcost < 0 || cost < vcost * 0.999) { vsel = c; vcost = cost; }
    }
    if (fixed_tm == 64) vsel = 0; else if (fixed_tm == 32) vsel = 1; else if (fixed_tm == 16) vsel = 2;
    // under-filled launches (the few big fronts at the top of the tree: fewer 64 x 64 tiles than compute units): 32 x 32 or 16 x 64
    // tiles quadruple the number of workgroups; their kernels take a K slab of 32 with the k loop unrolled, because with one wave per
    // SIMD nothing else hides the LDS and HBM latencies (measured on 1024 x 256 x 1024: 240 -> 118 us)
    static const int latency_tiles = getenv("HELM_ND_LATTILES") ? atoi(getenv("HELM_ND_LATTILES")) : 1;
    bool latency_mode = false;
    static const int latency_max = getenv("HELM_ND_LATTILES_MAX") ? atoi(getenv("HELM_ND_LATTILES_MAX")) : 256;
    if (latency_tiles && (long long)batch * ((M + 63) / 64) * ((Nn + 63) / 64) < latency_max) {
        const long long a6 = (long long)((M + 31) / 32) * 32 * ((Nn + 31) / 32) * 32, a7 = (long long)((M + 15) / 16) * 16 * ((Nn + 63) / 64) * 64;
        vsel = a7 < a6 ? 7 : 6;
        latency_mode = true;
    }
    // fronts of 8 and 16 rows: the 16 x 64 tile (four workgroups per front) is 4-15 % ahead of 16 x 256 (tools/tile_lab.py low)
    if (!latency_mode && gemm_variant() == 7 && M <= 16 && vsel == 2) vsel = 7;
    // a few hundred 64 x 64 tiles (one or two per compute unit, gone in a single round): 32 x 32 tiles give every unit four to eight
    // workgroups to overlap (1025 x 256 x 512 x 4: 152 -> 106 us, 1025 x 512 x 512 x 4: 219 -> 183, tools/tile_lab.py)
    static const int midfill_max = getenv("HELM_ND_MIDFILL_MAX") ? atoi(getenv("HELM_ND_MIDFILL_MAX")) : 600;
    if (!latency_mode && gemm_variant() == 7 && M > 64 && (long long)batch * ((M + 63) / 64) * ((Nn + 63) / 64) < midfill_max) vsel = 6;
    // forward-gather launches are HBM-bound and every row-tile repeats the three-source gather of the B rows: one row-tile per front
    // wherever the front has at most 64 rows, whatever the padding costs in flops
    if (rows && rows->fwd3 && M <= 64 && !latency_mode) vsel = 0;
    // rank-32 updates of one large matrix (blocked Gauss-Jordan of the 3-D plane inverses): HBM-bound, the 64 x 32 tile is the fastest
    // (3713^2 x 32: 87 us against 93 for 64 x 64, tools/zgemm_tiles.py)
    if (rows && rows->dense && K <= 32 && !latency_mode && batch == 1) vsel = 3;
    // 32-row fronts through the row tables (leaf forward elimination, levels 10 and 9 of the back substitution): HELM_ND_M32_TILE=4 gives them the
    // 32 x 64 tile (five workgroups on a compute unit where 32 x 128 keeps three).  Alone on the GPU that is faster (leaf forward 1.31 -> 1.24 ms, level 10
    // backward 0.295 -> 0.264, every_front_computed +1.5 %); beside the factorisation of the next item it is not (headline 13 580 -> 13 420 in two runs each): off
    static const int m32_tile = getenv("HELM_ND_M32_TILE") ? atoi(getenv("HELM_ND_M32_TILE")) : -1;
    if (m32_tile >= 0 && M == 32 && rows && !rows->dense && !rows->schur4 && !latency_mode && gemm_variant() == 7) vsel = m32_tile;
    if (g_gemm_tile >= 0) { vsel = g_gemm_tile & 7; latency_mode = false; }
    // the fused update + sweep launch exists for two tiles: 64 x 32 (large matrices) and the 32 x 32 latency tile (under-filled launches)
    if (rows && rows->la) { vsel = latency_mode ? 6 : 3; }
    if (rows && rows->tm64 && M <= 64) { vsel = Nn <= 32 ? 3 : 0; latency_mode = false; }
    for (int b0 = 0; b0 < batch && !split_done; b0 += 65535) {
        const int nb = std::min(65535, batch - b0);
        GemmRows R; if (rows) R = *rows;
        R.z0 = b0;
        static const int xcd_env = getenv("HELM_ND_XCDMAP") ? atoi(getenv("HELM_ND_XCDMAP")) : 2;
        R.xcd_map = xcd_env;
        // gathered products over many fronts (the row-table and forward-gather levels of both passes) store C with nontemporal stores: the rows are not read
        // again before a whole level has gone by (headline +1.5 %, every_front_computed +1 %; HELM_ND_NTC = fronts per launch from which, 0 off)
        static const int ntc_env = getenv("HELM_ND_NTC") ? atoi(getenv("HELM_ND_NTC")) : 64;
        if (ntc_env && rows && !rows->dense && !rows->schur4 && nb >= ntc_env) R.ntc = 1;                            // (the Schur complements, read back one level later: no difference either way)
        const cplx *Ab = A + b0 * sa, *Bb = B ? B + b0 * sb : B;
        cplx *Cb = C ? C + b0 * sc : C;
        // (a 4 x 8 register block per thread -- RN = 8, 64 x 128 tile -- was measured too: 230 VGPRs, occupancy 2, 38 % slower)
        const int gv = gemm_variant();
        ExtArm arm(op, ext, 8.0 * M * (double)Nn * K * nb, gemm_operand_bytes(M, Nn, K, beta, rows) * nb, M, Nn, K, nb,
              rows && rows->schur4 ? 4 : (rows && !rows->dense ? (rows->fwd3 ? 2 : 1) : (rows && rows->la ? 5 : 0)));
#define ZG_ARGS st, (rows && rows->schur4 ? 4 : (rows && !rows->dense ? (rows->fwd3 ? 2 : 1) : 0)), nb, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R
#define ZG_VEC(TM_, RN_) do { switch (gv) { \
            case 0: launch_vec<TM_, RN_>(ZG_ARGS); break; \
            case 6: launch_vec2<TM_, RN_, (RN_ == 1 ? 32 : (RN_ == 2 ? 16 : 8)), 1, 1>(ZG_ARGS); break; \
            default: launch_vec2<TM_, RN_, 8, 1, 1>(ZG_ARGS); break; } } while (0)
        // matrix-core kernel (generation 7): the same nine tile shapes as WM x WN waves of MT x NT blocks of 16 x 16
#define ZG_MFMA(WM_, WN_, MT_, NT_, KS_) launch_mfma<WM_, WN_, MT_, NT_, KS_>(ZG_ARGS)
        if (gv == 7 && !(rows && rows->la)) {
            // fronts of 16 a + 1 rows (the 49-unknown leaves): a rows of blocks on the matrix cores, the last row on the vector ALUs
            static const int xr_on = getenv("HELM_ND_XR") ? atoi(getenv("HELM_ND_XR")) : 1;
            const int idxmode = rows && rows->schur4 ? 4 : (rows && !rows->dense ? (rows->fwd3 ? 2 : 1) : 0);
            const bool plain = !rows || (!rows->schur4 && !rows->fwd3 && !rows->ksplit && rows->zr1 == 0 && rows->zc1 == 0 && rows->sk1 == 0);
            if (xr_on && M == 49 && Nn >= 64 && plain && idxmode <= 1 && g_gemm_tile < 0) {
                static const int xr_nt = getenv("HELM_ND_XR_NT") ? atoi(getenv("HELM_ND_XR_NT")) : 1;
                // (49 x 64 tile, four workgroups per compute unit, against 49 x 128 with two: leaf back substitution 3.23 -> 2.90 ms with every front
                // computed, 1.89 -> 1.70 on point sources; a K slab of 16 halves the occupancy again: 3.7 / 4.7 ms)
                if (xr_nt == 2 && (Nn % 128 == 0 || Nn > 192)) launch_mfma_xr<3, 2, 8>(ZG_ARGS); else launch_mfma_xr<3, 1, 8>(ZG_ARGS);
                continue;
            }
            if (latency_mode) { if (vsel == 6) ZG_MFMA(2, 2, 1, 1, 16); else ZG_MFMA(1, 4, 1, 1, 16); continue; }
            switch (vsel) {
                case 0: ZG_MFMA(2, 2, 2, 2, 8); break;
                case 1: ZG_MFMA(1, 4, 2, 2, 8); break;
                case 2: ZG_MFMA(1, 4, 1, 4, 8); break;
                case 3: ZG_MFMA(2, 2, 2, 1, 8); break;
                case 4: ZG_MFMA(2, 2, 1, 2, 8); break;
                case 5: ZG_MFMA(1, 4, 1, 2, 8); break;
                case 6: ZG_MFMA(2, 2, 1, 1, 8); break;
                case 8: ZG_MFMA(4, 1, 2, 1, 8); break;
                default: ZG_MFMA(1, 4, 1, 1, 8); break;
            }
            continue;
        }
#undef ZG_MFMA
        if (rows && rows->la && gv != 0) {           // update + pivot sweep of the next block in one launch
            if (latency_mode) {
                dim3 grid((Nn + 31) / 32, (M + 31) / 32, nb + 1);
                if (gv == 7) ZG_LAUNCH((k_zgemm3_la<2, 2, 1, 1, 16>), grid, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R, *rows->la);
                else ZG_LAUNCH((k_zgemm2_la<32, 1, 32, 4>), grid, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R, *rows->la);
            } else {
                dim3 grid((Nn + 31) / 32, (M + 63) / 64, nb + 1);
                if (gv == 7) ZG_LAUNCH((k_zgemm3_la<2, 2, 2, 1, 8>), grid, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R, *rows->la);
                else ZG_LAUNCH((k_zgemm2_la<64, 2, 8, 1>), grid, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R, *rows->la);
            }
            continue;
        }
        if (latency_mode && gv != 0) {
            if (vsel == 6) launch_vec2<32, 1, 32, 4, 1>(ZG_ARGS); else launch_vec2<16, 1, 32, 4, 1>(ZG_ARGS);
            continue;
        }
        switch (vsel) {
            case 0: ZG_VEC(64, 4); break;
            case 1: ZG_VEC(32, 4); break;
            case 2: ZG_VEC(16, 4); break;
            case 3: ZG_VEC(64, 2); break;
            case 4: ZG_VEC(32, 2); break;
            case 5: ZG_VEC(16, 2); break;
            case 6: ZG_VEC(32, 1); break;
            case 8: launch_vec2<128, 2, 8, 1, 1>(ZG_ARGS); break;      // (second-generation kernel only: see the candidate loop)
            default: ZG_VEC(16, 1); break;
        }
#undef ZG_VEC
    }
    const double flops = 8.0 * M * (double)Nn * K * batch;
    const double obytes = gemm_operand_bytes(M, Nn, K, beta, rows) * batch;
    if (ext) return 0;
    if (in_run) {                       // the run's end event is recorded by GemmRun's destructor
        if (op->gemm_run_pair >= 0) { op->gemm_run_flops += flops; op->gemm_run_bytes += obytes; op->gemm_run_sol += gemm_sol_ms(flops, obytes); op->gemm_run_launches += 1; }
    } else if (e0) {
        hipEventRecord(e1, st);
        op->ev_pending_gemm.push_back(std::make_pair((int)op->ev_used, flops));
        op->ev_pending_gemm_n.push_back(1);
        op->ev_pending_gemm_bytes.push_back(obytes);
        op->ev_pending_gemm_sol.push_back(gemm_sol_ms(flops, obytes));
        op->ev_used += 2;
    }
    return 0;
}

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

// scratch for the alternating pivot-block inverses of k_gj_step (2 PNB^2 per matrix): the handle's own, or one process-wide buffer for the
// diagnostic entry points (single-threaded)
static cplx *gj_pbuf(helm_op *op, int batch) {
    const size_t need = (size_t)batch * 2 * PNB * PNB * sizeof(cplx);
    if (op) {
        if (op->gjp_bytes < need) {
            if (op->gjp_buf) { hipStreamSynchronize(op->stream); helm_pool_free(op->device, op->gjp_buf, op->gjp_bytes); op->gjp_buf = nullptr; op->gjp_bytes = 0; }
            op->gjp_buf = (cplx *)helm_pool_alloc(op->device, need);
            op->gjp_bytes = op->gjp_buf ? need : 0;
        }
        return op->gjp_buf;
    }
    static cplx *buf = nullptr; static size_t bytes = 0;
    if (bytes < need) {
        if (buf) { hipDeviceSynchronize(); hipFree(buf); buf = nullptr; bytes = 0; }
        if (hipMalloc((void **)&buf, need) == hipSuccess) bytes = need; else buf = nullptr;
    }
    return buf;
}

// in-place inverse of `batch` n x n blocks (row-major, leading dimension ld, batch stride `stride`); W: workspace with
// batch stride ws, at least n*n elements per matrix
int g_recurse_min = -1;     // (helm_debug_inverse_bench overrides HELM_ND_RECURSE_N)
void invert(helm_op *op, cplx *M, int ld, long long stride, int n, int batch, cplx *W, long long ws, int align = 1, int base = 0) {
    hipStream_t st = op ? op->stream : nullptr;
    // the coupled two-field system (align == 2) is far worse conditioned: it gets the wider pivoting window by default
    static const int gj_env = getenv("HELM_ND_GJ") ? atoi(getenv("HELM_ND_GJ")) : 0;
    const int gj_base = base ? base : (gj_env == 64 ? 64 : (gj_env == 32 ? 32 : (align == 2 ? 64 : 32)));
    // blocked Gauss-Jordan (k_gj_panel + one masked GEMM per 32 columns) for everything above the base size; the recursive 2 x 2
    // block inversion below is kept for the coupled two-field system (64-wide pivot windows) and as HELM_ND_BLOCKED=0
    static const int blocked = getenv("HELM_ND_BLOCKED") ? atoi(getenv("HELM_ND_BLOCKED")) : 1;
    static const int blocked_max_batch = getenv("HELM_ND_BLOCKED_BATCH") ? atoi(getenv("HELM_ND_BLOCKED_BATCH")) : 1024;
    // Large single matrices (the dense plane inverses of the 3-D coarse solve, n = 3713) take ONE level of the 2 x 2 block recursion first:
    // the same n^3 multiply-adds, but three quarters of them in products with an inner dimension of n / 2 (compute-bound, 44 TFLOP/s)
    // instead of rank-32 updates that read and write the whole matrix every 32 columns (HBM-bound, 30 TFLOP/s); the halves go back to the
    // blocked form.  Measured (helm_debug_inverse_bench, n = 3713): 15.9 -> 14.5 ms; a second level of recursion loses (17.3 ms: the
    // blocked form of a 928^2 quarter is latency-bound, 58 us per block step), and so does recursion at n = 1857 (3.4 -> 4.5 ms).
    static const int recurse_env = getenv("HELM_ND_RECURSE_N") ? atoi(getenv("HELM_ND_RECURSE_N")) : 3000;
    const int recurse_min = g_recurse_min > 0 ? g_recurse_min : recurse_env;
    const bool recurse = align == 1 && n >= recurse_min && (long long)n * n <= ws;
    if (!recurse && blocked && batch <= blocked_max_batch && gemm_variant() != 0 && n > gj_base && gj_base == 32 && (long long)2 * PNB * n <= ws) {
        cplx *Wc = W, *Wr = W + (long long)PNB * n;
        // look-ahead: the pivot block of step k+1 is inverted beside the update of step k, in the same launch (k_zgemm2_la)
        // (the dense plane inverses of the 3-D coarse solve, n = 3713: -7 %; the 2-D fronts of 512 and 1024 unknowns at the top of the tree, with the
        // 32 x 32 latency tile: +0.6 % on the bench; below that the split panel kernels cost more than the sweep hides.
        // A first version ran the sweep on a second stream: same gain at n = 3713, but two cross-stream event hops per step, -5 % in 2-D)
        // one launch per block step (k_gj_step): the second copy of the matrix lives in W, the two P buffers in the handle's scratch
        const int gjstep = getenv("HELM_ND_GJSTEP") ? atoi(getenv("HELM_ND_GJSTEP")) : 1;          // (read per call: the tests compare the two forms)
        const int gjstep_min = getenv("HELM_ND_GJSTEP_MIN") ? atoi(getenv("HELM_ND_GJSTEP_MIN")) : 512;      // (per call, like the switch)
        static const int gjstep_max = getenv("HELM_ND_GJSTEP_MAX") ? atoi(getenv("HELM_ND_GJSTEP_MAX")) : 1536;
        if (gjstep && gemm_variant() == 7 && n >= gjstep_min && n <= gjstep_max && (long long)n * n <= ws && batch <= 32768) {
            cplx *Pb = gj_pbuf(op, batch);
            if (Pb) {
                const long long sp = 2LL * PNB * PNB;
                hipLaunchKernelGGL(k_gj_pivot, dim3(batch), dim3(256), 0, st, M, ld, stride, n, 0, std::min(PNB, n), (const cplx *)nullptr, (const cplx *)nullptr, ws, Pb, sp);
                static const int prof_ext = getenv("HELM_PROF_EXT") ? atoi(getenv("HELM_PROF_EXT")) : 1;
                const bool ext = prof_ext && op && op->profiling;
                int step = 0;
                for (int k0 = 0; k0 < n; k0 += PNB, ++step) {
                    const int nb = std::min(PNB, n - k0);
                    GjStepArgs a;
                    const bool fromM = (step & 1) == 0;
                    a.Ta = fromM ? M : W; a.lda = fromM ? ld : n; a.sa = fromM ? stride : ws;
                    a.Tb = fromM ? W : M; a.ldb = fromM ? n : ld; a.sb = fromM ? ws : stride;
                    a.n = n; a.k0 = k0; a.nb = nb;
                    a.P = Pb + (step & 1) * PNB * PNB; a.Pn = Pb + ((step + 1) & 1) * PNB * PNB; a.sp = sp;
                    a.k1 = k0 + PNB; a.nb1 = k0 + PNB < n ? std::min(PNB, n - k0 - PNB) : 0;
                    a.batch = batch;
                    const int per_slice = ((n + 31) / 32) * ((n + 63) / 64);
                    a.nsw = (batch + per_slice - 1) / per_slice;
                    // booked with the products (mode 5: update + pivot sweep): 8 n^2 nb flop, the matrix read and written once
                    ExtArm arm(op, ext, 8.0 * n * (double)n * nb * batch, 16.0 * (2.0 * n * (double)n + 2.0 * n * nb) * batch, n, n, nb, batch, 5);
                    const dim3 grid((n + 31) / 32, (n + 63) / 64, batch + a.nsw);
                    ZG_LAUNCH(k_gj_step, grid, a);
                }
                if (step & 1) hipLaunchKernelGGL(k_copy_blocks, dim3((unsigned)std::min<long long>(((long long)n * n + 255) / 256, 1024), batch), dim3(256), 0, st, (const cplx *)W, n, ws, M, ld, stride, n);
                return;
            }
        }
        static const int lookahead = getenv("HELM_ND_LOOKAHEAD") ? atoi(getenv("HELM_ND_LOOKAHEAD")) : 1;
        static const int lookahead_min_n = getenv("HELM_ND_LOOKAHEAD_N") ? atoi(getenv("HELM_ND_LOOKAHEAD_N")) : 512;
        // (one sweep per matrix rides in the first z-slice of the update's grid: that slice must have a workgroup for each)
        if (lookahead && n >= lookahead_min_n && (long long)((n + 63) / 64) * ((n + 31) / 32) >= batch && (long long)2 * PNB * n + PNB * PNB <= ws) {
            cplx *Pb = W + (long long)2 * PNB * n;
            hipLaunchKernelGGL(k_gj_pivot, dim3(batch), dim3(256), 0, st, M, ld, stride, n, 0, std::min(PNB, n), Wc, Wr, ws, Pb, ws);
            for (int k0 = 0; k0 < n; k0 += PNB) {
                const int nb = std::min(PNB, n - k0);
                hipLaunchKernelGGL(k_gj_slices, dim3((n + 63) / 64, batch), dim3(256), 0, st, M, ld, stride, n, k0, nb, Wc, Wr, ws, Pb, ws);
                GemmRows R; R.dense = 1; R.zr0 = k0; R.zr1 = k0 + nb; R.zc0 = k0; R.zc1 = k0 + nb;
                GjPivotArgs pv;
                if (k0 + PNB < n) {                                      // the sweep of the next pivot block rides along; the update leaves that block alone
                    const int k1 = k0 + PNB, nb1 = std::min(PNB, n - k1);
                    R.sk0 = k1; R.sk1 = k1 + nb1;
                    pv.T0 = M; pv.ld = ld; pv.stride = stride; pv.n = n; pv.k0 = k1; pv.nb = nb1; pv.Wc0 = Wc; pv.Wr0 = Wr; pv.wstride = ws;
                    pv.Pb0 = Pb; pv.pstride = ws; pv.batch = batch;
                    R.la = &pv;
                }
                gemm(op, n, n, nb, cmake(-1, 0), Wc, PNB, ws, Wr, n, ws, cmake(1, 0), M, ld, stride, batch, &R);
            }
            return;
        }
        for (int k0 = 0; k0 < n; k0 += PNB) {
            const int nb = std::min(PNB, n - k0);
            for (int b0 = 0; b0 < batch; b0 += 65535) {
                const int nbt = std::min(65535, batch - b0);
                hipLaunchKernelGGL(k_gj_panel, dim3((n + 63) / 64, nbt), dim3(256), 0, st, M + b0 * stride, ld, stride, n, k0, nb, Wc + b0 * ws, Wr + b0 * ws, ws);
            }
            GemmRows R; R.dense = 1; R.zr0 = k0; R.zr1 = k0 + nb; R.zc0 = k0; R.zc1 = k0 + nb;
            gemm(op, n, n, nb, cmake(-1, 0), Wc, PNB, ws, Wr, n, ws, cmake(1, 0), M, ld, stride, batch, &R);
        }
        return;
    }
    if (n <= gj_base) {
        for (int b0 = 0; b0 < batch; b0 += 1 << 20) {
            const int nb = std::min(1 << 20, batch - b0);
            // (one wave per matrix -- 64 threads, free barriers, four times the matrices in flight -- was measured: 20 % slower factorisation)
            static const int gj_threads = getenv("HELM_ND_GJ_THREADS") ? atoi(getenv("HELM_ND_GJ_THREADS")) : 256;
            static const int gj_fast = getenv("HELM_ND_GJFAST") ? atoi(getenv("HELM_ND_GJFAST")) : 1;
            // the fast kernel is built for latency (every thread repeats the pivot search): with thousands of matrices in flight the
            // chip is issue-bound and the plain kernel is as fast or faster (8192 blocks of 8 x 8: 54 vs 139 us)
            static const int gj_wave = getenv("HELM_ND_GJWAVE") ? atoi(getenv("HELM_ND_GJWAVE")) : 1;
            static const int gj_wave_min = getenv("HELM_ND_GJWAVE_MIN") ? atoi(getenv("HELM_ND_GJWAVE_MIN")) : 2048;
            if (n <= 32 && gj_wave && batch >= gj_wave_min) hipLaunchKernelGGL(k_gj32w_inverse, dim3((nb + 3) / 4), dim3(256), 0, st, M + b0 * stride, ld, stride, n, nb);
            else if (n <= 32 && gj_fast && batch < gj_wave_min) hipLaunchKernelGGL(k_gj32_inverse, dim3(nb), dim3(256), 0, st, M + b0 * stride, ld, stride, n);
            else if (n <= 32 && gj_threads == 1024) hipLaunchKernelGGL((k_gj_inverse<32, 1024>), dim3(nb), dim3(1024), 0, st, M + b0 * stride, ld, stride, n);
            else if (n <= 32) hipLaunchKernelGGL(k_gj_inverse<32>, dim3(nb), dim3(gj_threads), 0, st, M + b0 * stride, ld, stride, n);
            else hipLaunchKernelGGL(k_gj_inverse<64>, dim3(nb), dim3(256), 0, st, M + b0 * stride, ld, stride, n);
        }
        return;
    }
    // halves split between cells, never between the two unknowns of one cell (their 2 x 2 coupling needs the pivoting of a base block)
    int s1 = ((n / 2 + align - 1) / align) * align;
    if (s1 >= n) s1 = n / 2;
    const int s2 = n - s1;
    cplx *A = M, *B = M + s1, *C = M + (long long)s1 * ld, *D = M + (long long)s1 * ld + s1;
    cplx *T1 = W, *T2 = W + (long long)s1 * s2, *Wn = W + 2LL * s1 * s2;
    const cplx one = cmake(1, 0), mone = cmake(-1, 0), zero = cmake(0, 0);
    invert(op, A, ld, stride, s1, batch, Wn, ws, align, base);
    {
        GemmRun run(op);
        gemm(op, s2, s1, s1, one, C, ld, stride, A, ld, stride, zero, T1, s1, ws, batch);       // T1 = C A^-1
        gemm(op, s2, s2, s1, mone, T1, s1, ws, B, ld, stride, one, D, ld, stride, batch);       // D  = D - T1 B  (Schur)
    }
    invert(op, D, ld, stride, s2, batch, Wn, ws, align, base);
    {
        GemmRun run(op);
        gemm(op, s1, s2, s1, one, A, ld, stride, B, ld, stride, zero, T2, s2, ws, batch);       // T2 = A^-1 B
        gemm(op, s1, s2, s2, mone, T2, s2, ws, D, ld, stride, zero, B, ld, stride, batch);      // B  = -T2 S^-1
        gemm(op, s2, s1, s2, mone, D, ld, stride, T1, s1, ws, zero, C, ld, stride, batch);      // C  = -S^-1 T1
        gemm(op, s1, s1, s2, mone, B, ld, stride, T1, s1, ws, one, A, ld, stride, batch);       // A  = A^-1 - B T1
    }
}

}  // namespace

// ---- plan cache ---------------------------------------------------------------------------------------------------
// Per device, most recently used last, HELM_ND_PLANS (default 6) kept alive per device by the cache (a factor keeps its own plan alive whatever the
// cache does).  r4: round 3 had ONE list of four for the whole process, searched and FILLED under one mutex: with the in-process dispatcher dealing
// operators over eight GPUs (or a 2-D plan beside the 3-D column-dissection plans) every new operator missed, and each miss rebuilt the host
// plan -- tens of milliseconds of recursion -- and ran hipMalloc / hipFree with the lock held, i.e. with every other GPU's prepare thread waiting.
// Now: look-up under the lock, build outside it (two threads that miss on the same key both build; the second finds the first's entry and
// drops its own), tables from the size-keyed device pool.
NdPlanDev::~NdPlanDev() {
    if (d_nodes) helm_pool_free(device, d_nodes, plan.nodes.size() * sizeof(NdDev));
    if (d_tab) helm_pool_free(device, d_tab, (size_t)plan.total_rows * sizeof(int4));
    if (d_cellnode) helm_pool_free(device, d_cellnode, (size_t)plan.nz * plan.nx * sizeof(int));
}

namespace {
std::mutex g_plan_mu;
// (never destroyed: a plan's destructor hands its tables to the device pool of capi.hip, which may be gone first when the process exits)
std::map<int, std::vector<std::shared_ptr<NdPlanDev>>> &g_plans = *new std::map<int, std::vector<std::shared_ptr<NdPlanDev>>>();

std::shared_ptr<NdPlanDev> plan_lookup(int device, int pnz, int pnx, int leaf, int dof) {      // (g_plan_mu held)
    std::vector<std::shared_ptr<NdPlanDev>> &L = g_plans[device];
    for (size_t i = 0; i < L.size(); ++i) {
        const NdPlanDev &c = *L[i];
        if (c.plan.nz == pnz && c.plan.nx == pnx && c.plan.leaf == std::max(2, leaf) && c.plan.dof == dof) {
            std::shared_ptr<NdPlanDev> hit = L[i];
            L.erase(L.begin() + i); L.push_back(hit);
            return hit;
        }
    }
    return nullptr;
}
}

int nd_get_plan(helm_op *op, int leaf, int dof, std::shared_ptr<NdPlanDev> *out) { return nd_get_plan_dims(op, op->nz, op->nx, leaf, dof, out); }

// the same for a grid that is not the handle's own: the 3-D coarse solve runs the 2-D dissection over (ny, nx) columns of nz unknowns (dof = nz)
int nd_get_plan_dims(helm_op *op, int pnz, int pnx, int leaf, int dof, std::shared_ptr<NdPlanDev> *out) {
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        if (std::shared_ptr<NdPlanDev> hit = plan_lookup(op->device, pnz, pnx, leaf, dof)) { *out = hit; return HELM_OK; }
    }
    std::shared_ptr<NdPlanDev> pd(new NdPlanDev());
    pd->device = op->device;
    nd_build_plan(pd->plan, pnz, pnx, leaf, dof);
    const NdPlan &P = pd->plan;
    if (2 * P.vregion >= (1LL << 31) || P.total_rows >= (1LL << 31)) HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "direct solver: grid too large for 32-bit row indices");
    pd->d_nodes = (NdDev *)helm_pool_alloc(op->device, P.nodes.size() * sizeof(NdDev));
    pd->d_tab = (int4 *)helm_pool_alloc(op->device, (size_t)P.total_rows * sizeof(int4));
    if (!pd->d_nodes || !pd->d_tab) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: allocation of the plan tables failed");
    HIP_TRY(op, hipMemcpyAsync(pd->d_nodes, P.nodes.data(), P.nodes.size() * sizeof(NdDev), hipMemcpyHostToDevice, op->stream));
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        const NdGroup &g = P.groups[gi];
        const int nmax = g.smax + g.mmax;
        for (int j0 = 0; j0 < g.cnt; j0 += 65535) {
            const int nb = std::min(65535, g.cnt - j0);
            hipLaunchKernelGGL(k_nd_build_tab, dim3((nmax + 255) / 256, nb), dim3(256), 0, op->stream, pd->d_nodes, g.first + j0, pd->d_tab, P.nz, P.nx);
        }
    }
    if (dof == 1) {                                      // which leaf eliminates a cell (the residual's q mask on sparse right-hand sides)
        pd->d_cellnode = (int *)helm_pool_alloc(op->device, (size_t)pnz * pnx * sizeof(int));
        if (pd->d_cellnode) {
            HIP_TRY(op, hipMemsetAsync(pd->d_cellnode, 0xFF, (size_t)pnz * pnx * sizeof(int), op->stream));
            for (size_t gi = 0; gi < P.groups.size(); ++gi) {
                const NdGroup &g = P.groups[gi];
                if (!g.leaf) continue;
                const long long rows = (long long)g.cnt * (g.smax + g.mmax);
                hipLaunchKernelGGL(k_nd_cellnode, dim3((unsigned)std::min<long long>((rows + 255) / 256, 65535)), dim3(256), 0, op->stream,
                                   (const int4 *)(pd->d_tab + g.roff), rows, g.smax + g.mmax, g.first, pd->d_cellnode);
            }
        }
    }
    HIP_TRY(op, hipStreamSynchronize(op->stream));      // the tables are complete before another handle (another stream) can find them
    static const int keep = getenv("HELM_ND_PLANS") ? std::max(1, atoi(getenv("HELM_ND_PLANS"))) : 6;
    std::shared_ptr<NdPlanDev> evicted;                  // (destroyed after the lock is released)
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        if (std::shared_ptr<NdPlanDev> hit = plan_lookup(op->device, pnz, pnx, leaf, dof)) { *out = hit; return HELM_OK; }      // another thread was faster: ours goes back to the pool
        std::vector<std::shared_ptr<NdPlanDev>> &L = g_plans[op->device];
        L.push_back(pd);
        if ((int)L.size() > keep) { evicted = L.front(); L.erase(L.begin()); }
    }
    *out = pd;
    return HELM_OK;
}

// (tests) number of plans the cache holds for `device`
extern "C" int helm_debug_plan_cache(int device) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plans.find(device);
    return it == g_plans.end() ? 0 : (int)it->second.size();
}

// ---- factorisation ---------------------------------------------------------------------------------------------
static void stable_free(NdFactor *f) {
    const int dev = f->pd ? f->pd->device : 0;
    for (NdStable &st : f->stable) {
        const size_t nmax = (size_t)st.smax + st.mmax;
        helm_pool_free(dev, st.lu, (size_t)st.smax * nmax * sizeof(cplx));
        helm_pool_free(dev, st.f21, (size_t)std::max(1, st.mmax) * st.smax * sizeof(cplx));
        helm_pool_free(dev, st.piv, (size_t)st.smax * sizeof(int));
        helm_pool_free(dev, st.d_node, sizeof(NdDev));
        helm_pool_free(dev, st.vs, st.vs_elems * sizeof(cplx));
    }
    f->stable.clear();
}

void nd_free(NdFactor *f) {
    if (!f) return;
    stable_free(f);
    if (f->d_est) helm_pool_free(f->pd ? f->pd->device : 0, f->d_est, f->est_elems * sizeof(double));
    if (f->d_leafflag) helm_pool_free(f->pd ? f->pd->device : 0, f->d_leafflag, f->leafflag_elems * sizeof(int));
    if (f->d_act) helm_pool_free(f->pd ? f->pd->device : 0, f->d_act, f->act_elems * sizeof(int));
    if (f->d_qmask) helm_pool_free(f->pd ? f->pd->device : 0, f->d_qmask, f->qmask_elems);
    if (f->d_fac) helm_pool_free(f->pd ? f->pd->device : 0, f->d_fac, (size_t)f->pd->plan.fac_elems * sizeof(cplx));
    delete f;
}

namespace {

// HELM_ND_MERGED_LEAF=0: keep F12 and back-substitute the leaves with two products through an intermediate (round-2 first half)
bool merged_leaf_backward() {
    static const int v = getenv("HELM_ND_MERGED_LEAF") ? atoi(getenv("HELM_ND_MERGED_LEAF")) : 1;
    return v != 0 && gemm_variant() != 0;
}

// Fronts that keep G = -F11^-1 F12 where F12 was, so that their back substitution is ONE product [F11^-1 | G] [y_S; x_B]: the leaves (round 2).
// Round 4 tried the same for every separator front of at most 64 unknowns (levels 13-7 at 1024^2; one 64-row tile per front, because the product
// overwrites the y_S rows it has just read): HELM_ND_MERGED_SEP=1.  Measured and left off: the 64-row tile pads the 8-, 16- and 32-row fronts of
// levels 13-10 eight-, four- and two-fold on the matrix cores (back substitution of level 13: 0.59 -> 1.11 ms, 12: 0.44 -> 0.79, 11: 0.30 -> 0.54,
// 10: 0.29 -> 0.40) and the 64-row levels gain 0.02 ms apiece: 5.5 -> 6.7 ms per pass.
bool merged_group(const NdPlan &P, const NdGroup &g) {
    static const int sep = getenv("HELM_ND_MERGED_SEP") ? atoi(getenv("HELM_ND_MERGED_SEP")) : 0;
    if (g.mmax <= 0 || !merged_leaf_backward()) return false;
    return g.leaf || (sep && P.dof == 1 && g.smax <= 64 && g.smax + g.mmax <= GB_KIDX);
}

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
    const char *e = getenv("HELM_ND_STABLE");          // (read per call: the tests switch it)
    // one unknown per cell and the out-of-place node-major passes only: the fix-ups of the passes gather a front's right-hand-side rows again,
    // which an in-place pass (HELM_ND_NM=0, the coupled system's rhs-major path) has overwritten by then
    static const int use_nm = getenv("HELM_ND_NM") ? atoi(getenv("HELM_ND_NM")) : 1;
    return (!e || atoi(e) != 0) && P.dof == 1 && use_nm != 0 && gemm_variant() != 0;
}
#define ND_STABLE_CAP 32
int ensure_est(helm_op *op, NdFactor *f, int cnt) {
    const size_t need = 2 * (size_t)cnt + (ND_STABLE_CAP + 2) / 2 + 8;         // two doubles per front + the flag list (ints) behind them
    if (f->est_elems >= need) return HELM_OK;
    if (f->d_est) { hipStreamSynchronize(op->stream); helm_pool_free(op->device, f->d_est, f->est_elems * sizeof(double)); f->d_est = nullptr; f->est_elems = 0; }
    int maxcnt = cnt;
    for (const NdGroup &g : f->pd->plan.groups) maxcnt = std::max(maxcnt, g.cnt);
    const size_t elems = 2 * (size_t)maxcnt + (ND_STABLE_CAP + 2) / 2 + 8;
    f->d_est = (double *)helm_pool_alloc(op->device, elems * sizeof(double));
    if (!f->d_est) return HELM_ERR_DEVICE;
    f->est_elems = elems;
    return HELM_OK;
}

int check_kernels(helm_op *op, const char *what);

// after the batched elimination of group gi: find its ill-conditioned fronts (one small read-back per group) and eliminate each of them again
int stabilise_group(helm_op *op, NdFactor *f, size_t gi, cplx *arenaF, cplx *work, const cplx *planes) {
    const NdPlan &P = f->pd->plan;
    const NdGroup &g = P.groups[gi];
    hipStream_t st = op->stream;
    // Which fronts are taken follows from the tolerance the factors are built for, not from a constant fitted to one model: an explicit inverse of a
    // front with condition number kappa leaves a first-pass residual of about kappa * eps (measured front by front, DESIGN.md 5.1: 8.8e5 -> 4e-9),
    // and it has to stay below rtol with a margin for the handful of such fronts that add up and for what the estimate ||F11||_inf ||F11^-1||_inf
    // misses:   kappa_max = rtol / (HELM_ND_STABLE_SAFETY * eps),  safety 8 by default  ->  1.1e5 at the 1e-10 of the reference parity tests, 1.1e7
    // at 1e-8, 1.1e3 at 1e-12 (where most fronts of the middle levels would be taken: the floor of 2e3 keeps the treatment a handful-of-fronts
    // affair and leaves the rest to the refinement pass, which always exists).  HELM_ND_STABLE_THR overrides with a fixed number.
    const double safety = getenv("HELM_ND_STABLE_SAFETY") ? std::max(1.0, atof(getenv("HELM_ND_STABLE_SAFETY"))) : 8.0;
    const double rt = op->rtol_hint > 0 ? op->rtol_hint : 1e-10;
    const double thr = getenv("HELM_ND_STABLE_THR") ? atof(getenv("HELM_ND_STABLE_THR")) : std::min(1e9, std::max(2e3, rt / (safety * 1.1102230246251565e-16)));
    const int nmax = g.smax + g.mmax;
    int *d_list = (int *)(f->d_est + 2 * (size_t)g.cnt);
    HIP_TRY(op, hipMemsetAsync(d_list, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_front_flag, dim3((g.cnt + 255) / 256), dim3(256), 0, st, (const double *)f->d_est, (const double *)(f->d_est + g.cnt), g.cnt, 1.0, thr, d_list, ND_STABLE_CAP);
    int h_list[ND_STABLE_CAP + 1];
    HIP_TRY(op, hipMemcpyAsync(h_list, d_list, sizeof(h_list), hipMemcpyDeviceToHost, st));
    HIP_TRY(op, hipStreamSynchronize(st));
    const int nflag = std::max(0, std::min(h_list[0], ND_STABLE_CAP));
    std::sort(h_list + 1, h_list + 1 + nflag);                                   // (the atomics hand the slots out in no particular order)
    if (getenv("HELM_ND_DEBUG") && atoi(getenv("HELM_ND_DEBUG")) >= 2) {
        std::vector<double> h(2 * (size_t)g.cnt);
        hipMemcpy(h.data(), f->d_est, h.size() * sizeof(double), hipMemcpyDeviceToHost);
        double worst = 0; int wj = 0;
        for (int j = 0; j < g.cnt; ++j) { const double e = h[j] * h[g.cnt + j]; if (!(e <= worst)) { worst = e; wj = j; } }
        fprintf(stderr, "[helm direct] level %d s %d cnt %d: estimate of front 0 = %.3e * %.3e; worst %.3e at %d; flagged %d\n", g.level, g.smax, g.cnt, h[0], h[g.cnt], worst, wj, h_list[0]);
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
        S.d_node = (NdDev *)helm_pool_alloc(op->device, sizeof(NdDev));
        f->stable.push_back(S);                                                   // (owned by the factor from here on: freed by nd_free on every path)
        if (!S.lu || !S.f21 || !S.piv || !S.d_node) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: allocation for an ill-conditioned front failed");
        NdDev n = P.nodes[S.node];
        const long long foff = n.foff;
        n.finv_off = 0; n.f12_off = g.smax;                                       // [F11 | F12] rows go to S.lu, [F21 | F22] back to the arena
        HIP_TRY(op, hipMemcpyAsync(S.d_node, &n, sizeof(NdDev), hipMemcpyHostToDevice, st));
        HIP_TRY(op, hipStreamSynchronize(st));                                    // (n is a stack copy)
        const int rb = std::max(std::min(nmax, 4), (nmax + 2047) / 2048);
        hipLaunchKernelGGL(k_nd_build_front, dim3((nmax + rb - 1) / rb, 1), dim3(256), (size_t)nmax * sizeof(int2), st, f->pd->d_nodes, 0, arenaF, S.lu, planes, P.nz, P.nx, rb,
                           (const NdDev *)S.d_node);
        cplx *F21 = arenaF + foff, *F22 = arenaF + foff + g.smax;
        HIP_TRY(op, hipMemcpy2DAsync(S.f21, (size_t)g.smax * sizeof(cplx), F21, (size_t)nmax * sizeof(cplx), (size_t)g.smax * sizeof(cplx), (size_t)g.mmax, hipMemcpyDeviceToDevice, st));
        if (g.smax <= 64) hipLaunchKernelGGL(k_lu_factor64, dim3(1), dim3(256), 0, st, S.lu, nmax, g.smax, S.piv);
        else hipLaunchKernelGGL(k_lu_factor, dim3(1), dim3(256), 0, st, S.lu, nmax, g.smax, S.piv);
        // Schur complement through the same factors: F22 -= F21 (F11^-1 F12)
        HIP_TRY(op, hipMemcpy2DAsync(work, (size_t)g.mmax * sizeof(cplx), S.lu + g.smax, (size_t)nmax * sizeof(cplx), (size_t)g.mmax * sizeof(cplx), (size_t)g.smax, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_lu_solve, dim3((g.mmax + 15) / 16), dim3(256), 0, st, (const cplx *)S.lu, nmax, g.smax, (const int *)S.piv, work, g.mmax, g.mmax);
        gemm(op, g.mmax, g.mmax, g.smax, mone, S.f21, g.smax, 0, work, g.mmax, 0, one, F22, nmax, 0, 1, nullptr);
    }
    const int dbg = getenv("HELM_ND_DEBUG") ? atoi(getenv("HELM_ND_DEBUG")) : 0;       // (read per call: a test switches it on)
    if (dbg && nflag) fprintf(stderr, "[helm direct] level %d (%s, s = %d, m = %d): %d ill-conditioned front(s) re-eliminated with a pivoted LU\n", g.level, g.leaf ? "leaves" : "separators", g.smax, g.mmax, nflag);
    return check_kernels(op, "re-elimination of ill-conditioned fronts");
}

// factorisation of one group (tree level x kind) on op->stream
int factor_group(helm_op *op, NdFactor *f, size_t gi, cplx *arenaF, cplx *work, const cplx *planes) {
    const NdPlan &P = f->pd->plan;
    const NdDev *d_nodes = f->pd->d_nodes;
    hipStream_t st = op->stream;
    const cplx one = cmake(1, 0), mone = cmake(-1, 0), zero = cmake(0, 0);
    const NdGroup &g = P.groups[gi];
    const int nmax = g.smax + g.mmax;
    const long long fs = (long long)g.mmax * nmax;              // scratch per front: [F21 | F22]
    const long long s11 = (long long)g.smax * g.smax, s12 = (long long)g.smax * g.mmax, s1 = (long long)g.smax * nmax;   // s1: stride of [F11 | F12]
    cplx *F = arenaF + g.foff;
    cplx *Finv = f->d_fac + g.finv, *G21 = f->d_fac + g.g21, *F12 = f->d_fac + g.f12;
    // leaves of one unknown per cell and at most 8 x 8 cells: the whole leaf level in one kernel (k_leaf_factor; HELM_ND_FUSEDLEAF=0: the batched path)
    const int fused_leaf = getenv("HELM_ND_FUSEDLEAF") ? atoi(getenv("HELM_ND_FUSEDLEAF")) : 1;          // (read per call: the tests switch both)
    // (a leaf takes ~0.3 ms from end to end in that kernel -- a chain of 49 elimination steps, then 2 x 49 substitution rows -- which thousands of
    // leaves in flight hide and a few hundred do not: the handful of larger leaves of a level stay on the batched path, HELM_ND_FUSEDLEAF_MIN)
    const int fused_leaf_min = getenv("HELM_ND_FUSEDLEAF_MIN") ? atoi(getenv("HELM_ND_FUSEDLEAF_MIN")) : 2048;
    if (fused_leaf && g.leaf && g.cnt >= fused_leaf_min && P.dof == 1 && P.leaf <= 8 && g.smax <= 64 && g.mmax > 0 && g.mmax <= LEAF_MP && merged_leaf_backward()) {
        if (f->leafflag_elems < (size_t)g.cnt) {
            if (f->d_leafflag) { hipStreamSynchronize(st); helm_pool_free(op->device, f->d_leafflag, f->leafflag_elems * sizeof(int)); f->d_leafflag = nullptr; f->leafflag_elems = 0; }
            int maxcnt = g.cnt;
            for (const NdGroup &q : P.groups) if (q.leaf) maxcnt = std::max(maxcnt, q.cnt);
            f->d_leafflag = (int *)helm_pool_alloc(op->device, (size_t)maxcnt * sizeof(int));
            if (!f->d_leafflag) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: allocation of the leaf flags failed");
            f->leafflag_elems = (size_t)maxcnt;
        }
        const int leaf_dbg = getenv("HELM_LEAF_DBG") ? atoi(getenv("HELM_LEAF_DBG")) : 0;       // (timing experiments: 1 no LU, 2 no substitution, 4 no G21 / S; test: 8 every leaf re-done by the pivoted kernel)
        for (int j0 = 0; j0 < g.cnt; j0 += 65535) {
            const int nb = std::min(65535, g.cnt - j0);
            cplx *g21b = G21 + (long long)j0 * g.mmax * g.smax;
            if (g.smax <= 49) hipLaunchKernelGGL(k_leaf_factor<49>, dim3(nb), dim3(128), 0, st, d_nodes, g.first + j0, arenaF, f->d_fac, g21b, planes, P.nz, P.nx, f->d_leafflag + j0, leaf_dbg);
            else hipLaunchKernelGGL(k_leaf_factor<64>, dim3(nb), dim3(128), 0, st, d_nodes, g.first + j0, arenaF, f->d_fac, g21b, planes, P.nz, P.nx, f->d_leafflag + j0, leaf_dbg);
            hipLaunchKernelGGL(k_leaf_factor_pivoted, dim3(nb), dim3(256), 0, st, d_nodes, g.first + j0, arenaF, f->d_fac, g21b, planes, P.nz, P.nx, (const int *)(f->d_leafflag + j0));
        }
        f->flops += (double)g.cnt * 8.0 * (2.0 * LEAF_BW * g.smax * (g.smax + g.mmax) + 3.0 * g.mmax * (g.smax + g.mmax));
        return HELM_OK;
    }
    static const int fused_build = getenv("HELM_ND_FUSEDBUILD") ? atoi(getenv("HELM_ND_FUSEDBUILD")) : 1;
    // the ring x ring block of a non-leaf front stays unbuilt: its Schur-complement product gathers the children's contributions itself
    static const int schur_gather_env = getenv("HELM_ND_SCHURGATHER") ? atoi(getenv("HELM_ND_SCHURGATHER")) : 1;
    bool schur_gather = false;
    const bool packs = P.dof <= 2 ? (nmax < (1 << 14) && P.nz < 65536 && P.nx < 32768) : (nmax < 65535 && P.nz < 4096 && P.nx < 4096 && P.dof < 128);
    if (fused_build && (size_t)nmax * sizeof(int2) <= 64 * 1024 && packs)
        schur_gather = schur_gather_env && !g.leaf && g.mmax > 0 && gemm_variant() != 0 && P.dof == 1;
    if (fused_build && (size_t)nmax * sizeof(int2) <= 64 * 1024 && packs) {      // (its row table must fit the 64 KB of LDS a launch gets by default: larger fronts take the unfused path)
        // rows per workgroup: whole fronts while there are thousands of them, a few rows each for the handful of big ones at the top
        const int want = std::max(1, 2048 / g.cnt);
        const int rb = std::max(std::min(nmax, 4), (nmax + want - 1) / want);
        for (int j0 = 0; j0 < g.cnt; j0 += 65535) {
            const int nb = std::min(65535, g.cnt - j0);
            hipLaunchKernelGGL(k_nd_build_front, dim3((nmax + rb - 1) / rb, nb), dim3(256), (size_t)nmax * sizeof(int2), st, d_nodes, g.first + j0, arenaF, f->d_fac, planes,
                               P.nz, P.nx, rb, (const NdDev *)nullptr, schur_gather ? 1 : 0);
        }
    } else {
        if (fs > 0) HIP_TRY(op, hipMemsetAsync(F, 0, (size_t)g.cnt * fs * sizeof(cplx), st));
        HIP_TRY(op, hipMemsetAsync(Finv, 0, (size_t)g.cnt * s1 * sizeof(cplx), st));
        for (int j0 = 0; j0 < g.cnt; j0 += 65535) {
            const int nb = std::min(65535, g.cnt - j0);
            hipLaunchKernelGGL(k_nd_assemble, dim3((nmax + 255) / 256, nb), dim3(256), 0, st, d_nodes, g.first + j0, arenaF, f->d_fac, planes, P.nz, P.nx);
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
                    hipLaunchKernelGGL(k_nd_extend_add, dim3(gx, nb), dim3(256), shm, st, d_nodes, g.first + j0, slot, arenaF, f->d_fac, P.nz, P.nx, chunk);
                }
        }
    }
    static const int gj_leaf = getenv("HELM_ND_GJ_LEAF") ? atoi(getenv("HELM_ND_GJ_LEAF")) : 0;
    static const int gj_upper = getenv("HELM_ND_GJ_UPPER") ? atoi(getenv("HELM_ND_GJ_UPPER")) : 0;
    // the few huge fronts at the top of the tree are one long chain of single-matrix launches: the wider base block halves it
    static const int gj_top = getenv("HELM_ND_GJ_TOP") ? atoi(getenv("HELM_ND_GJ_TOP")) : 0;
    const int base = g.leaf ? gj_leaf : (g.cnt <= gj_top ? 64 : gj_upper);
    const int stable_smax = getenv("HELM_ND_STABLE_SMAX") ? atoi(getenv("HELM_ND_STABLE_SMAX")) : 128;
    // (leaves are not watched: 20-40 typically, below 6e3 in every operator examined, and their level is the one where two more passes over
    // every front cost something)
    const bool watch = stable_enabled(P) && !g.leaf && g.mmax > 0 && g.smax <= std::min(stable_smax, LUS_NMAX) && ensure_est(op, f, g.cnt) == HELM_OK;
    if (watch) for (int j0 = 0; j0 < g.cnt; j0 += 65535)
        hipLaunchKernelGGL(k_front_absmax, dim3(std::min(65535, g.cnt - j0)), dim3(256), 0, st, Finv + (long long)j0 * s1, nmax, s1, d_nodes + g.first + j0, f->d_est + j0);
    invert(op, Finv, nmax, s1, g.smax, g.cnt, work, s11, P.dof, base);      // F11 -> F11^-1 where it stays
    if (watch) for (int j0 = 0; j0 < g.cnt; j0 += 65535)
        hipLaunchKernelGGL(k_front_absmax, dim3(std::min(65535, g.cnt - j0)), dim3(256), 0, st, Finv + (long long)j0 * s1, nmax, s1, d_nodes + g.first + j0, f->d_est + g.cnt + j0);
    if (g.mmax > 0) {
        // G21 = F21 F11^-1 ; F22 -= G21 F12
        gemm(op, g.mmax, g.smax, g.smax, one, F, nmax, fs, Finv, nmax, s1, zero, G21, g.smax, (long long)g.mmax * g.smax, g.cnt);
        if (schur_gather) {
            GemmRows R; R.schur4 = 1; R.nodes = d_nodes; R.first = g.first; R.arenaS = arenaF; R.tabCi = f->pd->d_tab + g.roff; R.tab_stride = nmax;
            gemm(op, g.mmax, g.mmax, g.smax, mone, G21, g.smax, (long long)g.mmax * g.smax, F12, nmax, s1, zero, F + g.smax, nmax, fs, g.cnt, &R);
        } else
        gemm(op, g.mmax, g.mmax, g.smax, mone, G21, g.smax, (long long)g.mmax * g.smax, F12, nmax, s1, one, F + g.smax, nmax, fs, g.cnt);
        if (merged_group(P, g)) {
            // leaves (and small separator fronts): F12 <- -F11^-1 F12, in place where a front is one 64-row tile (GemmRows::tm64), else through the inversion workspace
            if (g.smax <= 64) {
                GemmRows R; R.dense = 1; R.tm64 = 1;
                gemm(op, g.smax, g.mmax, g.smax, mone, Finv, nmax, s1, F12, nmax, s1, zero, F12, nmax, s1, g.cnt, &R);
            } else {
                gemm(op, g.smax, g.mmax, g.smax, mone, Finv, nmax, s1, F12, nmax, s1, zero, work, g.mmax, s12, g.cnt);
                HIP_TRY(op, hipMemcpy2DAsync(F12, (size_t)nmax * sizeof(cplx), work, (size_t)g.mmax * sizeof(cplx), (size_t)g.mmax * sizeof(cplx),
                                             (size_t)g.cnt * g.smax, hipMemcpyDeviceToDevice, st));
            }
        }
    }
    f->flops += (double)g.cnt * 8.0 * (2.0 * g.smax * g.smax * g.smax + (double)g.smax * g.smax * g.mmax + (double)g.smax * g.mmax * g.mmax);
    if (watch) return stabilise_group(op, f, gi, arenaF, work, planes);
    return HELM_OK;
}

struct SolveCtx {
    const int4 *tab; cplx *Xt, *arenaV; int nrhs; dim3 rb; int use_idx;
    const cplx *Qt;       // node-major right-hand sides (read only); == Xt for an in-place solve
    int *act = nullptr; int nct = 0;     // sparse-right-hand-side flags of the forward pass (null: every front is computed)
    int act_hint = 0;                    // the leaves' flags come from the support the caller declared (helm_set_rhs_support): no scan of q
    dim3 rgrid(long long rows) const { return dim3((unsigned)std::min<long long>((rows + rb.y - 1) / rb.y, 1 << 20)); }
};

SolveCtx solve_ctx(const NdFactor *f, cplx *ws, int nrhs) {
    static const int use_idx = getenv("HELM_ND_IDXGEMM") ? atoi(getenv("HELM_ND_IDXGEMM")) : 1;
    const NdPlan &P = f->pd->plan;
    SolveCtx c;
    c.tab = f->pd->d_tab; c.Xt = ws; c.arenaV = ws + (long long)P.dof * P.nz * P.nx * nrhs; c.nrhs = nrhs; c.use_idx = use_idx;
    c.Qt = c.Xt;
    int lx = 1;
    while (lx < nrhs && lx < 256) lx <<= 1;
    c.rb = dim3(lx, 256 / lx);
    return c;
}

// ill-conditioned fronts of group gi (NdStable): their outgoing rows once more, through the front's own LU
void forward_stable(helm_op *op, NdFactor *f, size_t gi, const SolveCtx &c) {
    const NdPlan &P = f->pd->plan;
    const NdGroup &g = P.groups[gi];
    const cplx one = cmake(1, 0), mone = cmake(-1, 0);
    for (NdStable &S : f->stable) {
        if (S.group != gi) continue;
        const NdDev &n = P.nodes[S.node];
        const int nmax = S.smax + S.mmax, nrhs = c.nrhs;
        cplx *V = c.arenaV + n.voff * nrhs;
        // the front vector gathered again: separator rows q_S + the children's rows (written to Xt as y_S for the back substitution), ring rows
        // the children's rows; then z = F11^-1 y_S through the LU and V_B -= F21 z
        hipLaunchKernelGGL(k_nd_fwd_rows, c.rgrid(nmax), c.rb, 0, op->stream, c.tab + n.roff, V, c.arenaV, c.Qt, c.Xt, (long long)nmax, nrhs, g.leaf ? 0 : 1,
                           (const int *)c.act, c.nct, (const NdDev *)f->pd->d_nodes, S.node, nmax);
        if (c.act) hipMemsetAsync(c.act + (long long)S.node * c.nct, 1, (size_t)c.nct * sizeof(int), op->stream);      // (every row of this front has been written)
        hipLaunchKernelGGL(k_lu_solve, dim3((nrhs + 15) / 16), dim3(256), 0, op->stream, (const cplx *)S.lu, nmax, S.smax, (const int *)S.piv, V, nrhs, nrhs);
        gemm(op, S.mmax, nrhs, S.smax, mone, S.f21, S.smax, 0, V, nrhs, 0, one, V + (long long)S.smax * nrhs, nrhs, 0, 1);
    }
}

void forward_group_batched(helm_op *op, NdFactor *f, size_t gi, const SolveCtx &c);
// forward elimination of one group on op->stream
void forward_group(helm_op *op, NdFactor *f, size_t gi, const SolveCtx &c) {
    forward_group_batched(op, f, gi, c);
    if (!f->stable.empty()) forward_stable(op, f, gi, c);
}

void forward_group_batched(helm_op *op, NdFactor *f, size_t gi, const SolveCtx &c) {
    const NdPlan &P = f->pd->plan;
    const cplx one = cmake(1, 0), mone = cmake(-1, 0), zero = cmake(0, 0);
    const NdGroup &g = P.groups[gi];
    const int nmax = g.smax + g.mmax, nrhs = c.nrhs;
    const long long rows = (long long)g.cnt * nmax;
    cplx *V = c.arenaV + g.voff * nrhs;
    if (c.use_idx && g.leaf && g.mmax > 0 && g.smax <= GB_KIDX) {
        // leaves have no children: the outgoing ring part is -G21 x_S with x_S read straight from Xt
        GemmRows R; R.tabB = c.tab + g.roff; R.offB = 0; R.tab_stride = nmax; R.Bx = c.Qt; R.ldx = nrhs;
        static const int leaf_detect = getenv("HELM_ND_SPARSE_LEAF") ? atoi(getenv("HELM_ND_SPARSE_LEAF")) : 1;
        R.act = (gemm_variant() == 7 && leaf_detect) ? c.act : nullptr; R.nct = c.nct; R.first = g.first; R.hint = R.act ? c.act_hint : 0;
        gemm(op, g.mmax, nrhs, g.smax, mone, f->d_fac + g.g21, g.smax, (long long)g.mmax * g.smax, nullptr, 0, 0, zero,
             V + (long long)g.smax * nrhs, nrhs, (long long)nmax * nrhs, g.cnt, &R);
        if (c.act && !R.act) hipMemsetAsync(c.act + (long long)g.first * c.nct, 1, (size_t)g.cnt * c.nct * sizeof(int), op->stream);
        return;
    }
    static const int fuse_fwd = getenv("HELM_ND_FUSEFWD") ? atoi(getenv("HELM_ND_FUSEFWD")) : 1;
    if (fuse_fwd && c.use_idx && c.Qt != c.Xt && !g.leaf && g.mmax > 0 && g.mmax <= 256 && g.smax <= GB_KIDX && gemm_variant() != 0) {
        // (lower levels only: with many row-tiles per front every one of them repeats the three-source gather -- measured slower from m = 1025 up)
        // (out-of-place solves only: in place, the y_S store of the first row-tile would race with the other row-tiles' reads of q_S)
        // the gather of k_nd_fwd_rows happens inside the GEMM's operand loads: V_B = (children's rows) - G21 (q_S + children's rows),
        // y_S stored to Xt on the way
        GemmRows R; R.fwd3 = 1; R.tabB = c.tab + g.roff; R.offB = 0; R.tabCi = c.tab + g.roff; R.offCi = g.smax; R.tab_stride = nmax;
        R.Bx = c.Qt; R.Cix = c.arenaV; R.Cox = c.Xt; R.ldx = nrhs;
        R.act = gemm_variant() == 7 ? c.act : nullptr; R.nct = c.nct; R.first = g.first; R.nodes = f->pd->d_nodes;
        { const NdDev &n0 = P.nodes[g.first]; R.child_rows = (n0.kid[0] >= 0 ? P.nodes[n0.kid[0]].mmax : 0) + (n0.kid[1] >= 0 ? P.nodes[n0.kid[1]].mmax : 0); }
        gemm(op, g.mmax, nrhs, g.smax, mone, f->d_fac + g.g21, g.smax, (long long)g.mmax * g.smax, nullptr, 0, 0, one,
             V + (long long)g.smax * nrhs, nrhs, (long long)nmax * nrhs, g.cnt, &R);
        if (c.act && !R.act) hipMemsetAsync(c.act + (long long)g.first * c.nct, 1, (size_t)g.cnt * c.nct * sizeof(int), op->stream);
        return;
    }
    hipLaunchKernelGGL(k_nd_fwd_rows, c.rgrid(rows), c.rb, 0, op->stream, c.tab + g.roff, V, c.arenaV, c.Qt, c.Xt, rows, nrhs, g.leaf ? 0 : 1,
                       (const int *)c.act, c.nct, (const NdDev *)f->pd->d_nodes, g.first, nmax);
    if (c.act) hipMemsetAsync(c.act + (long long)g.first * c.nct, 1, (size_t)g.cnt * c.nct * sizeof(int), op->stream);      // (these fronts write every row)
    if (g.mmax > 0)     // outgoing ring part: V_B -= G21 V_S
        gemm(op, g.mmax, nrhs, g.smax, mone, f->d_fac + g.g21, g.smax, (long long)g.mmax * g.smax, V, nrhs, (long long)nmax * nrhs, one,
             V + (long long)g.smax * nrhs, nrhs, (long long)nmax * nrhs, g.cnt);
}

// ill-conditioned fronts of group gk: [y_S; x_B] is set aside before the batched launches overwrite y_S in Xt (pre), and x_S = F11^-1 (y_S - F12 x_B)
// through the front's LU replaces what they wrote (post)
int backward_stable(helm_op *op, NdFactor *f, size_t gk, const SolveCtx &c, bool post) {
    const NdPlan &P = f->pd->plan;
    const NdGroup &g = P.groups[gk];
    const cplx one = cmake(1, 0), mone = cmake(-1, 0);
    for (NdStable &S : f->stable) {
        if (S.group != gk) continue;
        const NdDev &n = P.nodes[S.node];
        const int nmax = S.smax + S.mmax, nrhs = c.nrhs;
        if (!post) {
            const size_t need = (size_t)nmax * nrhs;
            if (S.vs_elems < need) {
                if (S.vs) { hipStreamSynchronize(op->stream); helm_pool_free(op->device, S.vs, S.vs_elems * sizeof(cplx)); S.vs = nullptr; S.vs_elems = 0; }
                S.vs = (cplx *)helm_pool_alloc(op->device, need * sizeof(cplx));
                if (!S.vs) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: scratch for an ill-conditioned front failed");
                S.vs_elems = need;
            }
            hipLaunchKernelGGL(k_nd_bwd_gather, c.rgrid(nmax), c.rb, 0, op->stream, c.tab + n.roff, S.vs, g.leaf ? c.Qt : (const cplx *)c.Xt, (const cplx *)c.Xt, (long long)nmax, nrhs);
        } else {
            if (S.mmax > 0) gemm(op, S.smax, nrhs, S.mmax, mone, S.lu + S.smax, nmax, 0, S.vs + (long long)S.smax * nrhs, nrhs, 0, one, S.vs, nrhs, 0, 1);
            hipLaunchKernelGGL(k_lu_solve, dim3((nrhs + 15) / 16), dim3(256), 0, op->stream, (const cplx *)S.lu, nmax, S.smax, (const int *)S.piv, S.vs, nrhs, nrhs);
            hipLaunchKernelGGL(k_nd_bwd_store, c.rgrid(S.smax), c.rb, 0, op->stream, c.tab + n.roff, (const cplx *)S.vs, c.Xt, (long long)S.smax, S.smax, nmax, nrhs);
        }
    }
    return HELM_OK;
}

void backward_group_batched(helm_op *op, NdFactor *f, size_t gk, const SolveCtx &c);
// back substitution of one group on op->stream
void backward_group(helm_op *op, NdFactor *f, size_t gk, const SolveCtx &c) {
    if (!f->stable.empty()) (void)backward_stable(op, f, gk, c, false);
    backward_group_batched(op, f, gk, c);
    if (!f->stable.empty()) (void)backward_stable(op, f, gk, c, true);
}

void backward_group_batched(helm_op *op, NdFactor *f, size_t gk, const SolveCtx &c) {
    const NdPlan &P = f->pd->plan;
    const cplx one = cmake(1, 0), mone = cmake(-1, 0), zero = cmake(0, 0);
    const NdGroup &g = P.groups[gk];
    const int nmax = g.smax + g.mmax, nrhs = c.nrhs;
    const long long rows = (long long)g.cnt * nmax;
    const long long s1 = (long long)g.smax * nmax;                 // stride of a front's [F11^-1 | F12] rows
    const cplx *Finv = f->d_fac + g.finv, *F12 = f->d_fac + g.f12;
    // leaves under HELM_ND_MERGED_LEAF hold G = -F11^-1 F12 in place of F12: x_S = F11^-1 y_S + G x_B
    const bool gform = merged_group(P, g);
    cplx *V = c.arenaV + g.voff * nrhs;
    // the other region is free in this pass: separator results go there
    const long long xs_off = g.voff + ((g.level & 1) ? -P.vregion : P.vregion);
    cplx *XS = c.arenaV + xs_off * nrhs;
    if (c.use_idx && g.mmax > 0 && g.mmax <= GB_KIDX) {
        // lower tree levels (almost all rows): every Xt row addressed through the row table -- no gather / store pass
        if (gform && c.Qt != c.Xt && nmax <= GB_KIDX) {
            // leaves: x_S = [F11^-1 | G] [y_S; x_B] in ONE product -- y_S rows from the right-hand sides, x_B rows from Xt; the result goes
            // straight to the Xt rows (no intermediate: 2 x 3.2 GB less per pass at 1024^2 x 256)
            GemmRows R; R.tabB = c.tab + g.roff; R.offB = 0; R.tabCo = c.tab + g.roff; R.offCo = 0; R.tab_stride = nmax;
            R.Bx = c.Xt; R.Bx2 = g.leaf ? c.Qt : (const cplx *)c.Xt; R.k2 = g.smax; R.Cox = c.Xt; R.ldx = nrhs;      // (a separator front's y_S was left in Xt by the forward pass)
            if (!g.leaf) R.tm64 = 1;                                                 // one row tile per front: the product overwrites rows it reads
            R.act_ro = (gemm_variant() == 7 && g.leaf) ? c.act : nullptr; R.nct = c.nct; R.first = g.first;
            gemm(op, g.smax, nrhs, nmax, one, Finv, nmax, s1, nullptr, 0, 0, zero, nullptr, 0, 0, g.cnt, &R);
            return;
        }
        if (gform) {        // right-hand sides and wavefields share their rows (Qt == Xt): V = F11^-1 y_S first, then x_S = V + G x_B
            GemmRows R1; R1.tabB = c.tab + g.roff; R1.offB = 0; R1.tab_stride = nmax; R1.Bx = c.Qt; R1.ldx = nrhs;
            gemm(op, g.smax, nrhs, g.smax, one, Finv, nmax, s1, nullptr, 0, 0, zero, V, nrhs, (long long)g.smax * nrhs, g.cnt, &R1);
            GemmRows R2; R2.tabB = c.tab + g.roff; R2.offB = g.smax; R2.tabCo = c.tab + g.roff; R2.offCo = 0; R2.tab_stride = nmax;
            R2.Bx = c.Xt; R2.Cox = c.Xt; R2.ldx = nrhs;
            gemm(op, g.smax, nrhs, g.mmax, one, F12, nmax, s1, nullptr, 0, 0, one, V, nrhs, (long long)g.smax * nrhs, g.cnt, &R2);
            return;
        }
        // T = y_S - F12 x_B and x_S = F11^-1 T
        GemmRows R1; R1.tabB = c.tab + g.roff; R1.offB = g.smax; R1.tabCi = c.tab + g.roff; R1.offCi = 0; R1.tab_stride = nmax;
        R1.Bx = c.Xt; R1.Cix = g.leaf ? c.Qt : c.Xt; R1.ldx = nrhs;      // a leaf's y_S is still the right-hand side itself
        gemm(op, g.smax, nrhs, g.mmax, mone, F12, nmax, s1, nullptr, 0, 0, one, V, nrhs, (long long)g.smax * nrhs, g.cnt, &R1);
        GemmRows R2; R2.tabCo = c.tab + g.roff; R2.offCo = 0; R2.tab_stride = nmax; R2.Cox = c.Xt; R2.ldx = nrhs;
        gemm(op, g.smax, nrhs, g.smax, one, Finv, nmax, s1, V, nrhs, (long long)g.smax * nrhs, zero, nullptr, 0, 0, g.cnt, &R2);
        return;
    }
    hipLaunchKernelGGL(k_nd_bwd_gather, c.rgrid(rows), c.rb, 0, op->stream, c.tab + g.roff, V, g.leaf ? c.Qt : c.Xt, c.Xt, rows, nrhs);
    if (gform) {            // the gathered front vector is [y_S; x_B]: one dense product with [F11^-1 | G]
        gemm(op, g.smax, nrhs, nmax, one, Finv, nmax, s1, V, nrhs, (long long)nmax * nrhs, zero, XS, nrhs, (long long)g.smax * nrhs, g.cnt);
    } else {
        if (g.mmax > 0)
            gemm(op, g.smax, nrhs, g.mmax, mone, F12, nmax, s1, V + (long long)g.smax * nrhs, nrhs, (long long)nmax * nrhs,
                 one, V, nrhs, (long long)nmax * nrhs, g.cnt);
        gemm(op, g.smax, nrhs, g.smax, one, Finv, nmax, s1, V, nrhs, (long long)nmax * nrhs, zero, XS, nrhs, (long long)g.smax * nrhs, g.cnt);
    }
    const long long srows = (long long)g.cnt * g.smax;
    hipLaunchKernelGGL(k_nd_bwd_store, c.rgrid(srows), c.rb, 0, op->stream, c.tab + g.roff, XS, c.Xt, srows, g.smax, nmax, nrhs);
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

int check_kernels(helm_op *op, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char b[256]; snprintf(b, sizeof(b), "direct solver: %s failed: %s", what, hipGetErrorString(e)); helm_set_error(op, b); return HELM_ERR_DEVICE; }
    return HELM_OK;
}

}  // namespace

long long nd_factor_ws_elems(const NdPlan &P) { return 2 * P.fregion + P.work_elems; }

namespace {
// HELM_ND_TRACE=1: per-group device time of the factorisation / forward / backward sweeps on stderr (diagnostics only)
struct GroupTrace {
    bool on; hipStream_t st; std::vector<hipEvent_t> ev; const char *what;
    GroupTrace(hipStream_t s, const char *w) : st(s), what(w) { static const int t = getenv("HELM_ND_TRACE") ? atoi(getenv("HELM_ND_TRACE")) : 0; on = t != 0; mark(); }
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
}  // namespace

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
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        rc = factor_group(op, f, gi, ws, ws + 2 * P.fregion, planes);
        if (rc) return rc;
    }
    return check_kernels(op, "factorisation kernels");
}

// ---- solve: Xin (nrhs x N, each right-hand side contiguous) -> Xout (may alias Xin) --------------------------------------
// ws: workspace of nd_solve_ws_elems(plan, nrhs) elements
long long nd_solve_ws_elems(const NdPlan &P, int nrhs) { return ((long long)P.dof * P.nz * P.nx + 2 * P.vregion) * nrhs; }

int nd_solve(helm_op *op, NdFactor *f, const cplx *Xin, cplx *Xout, int nrhs, cplx *ws, int conj_out) {
    const NdPlan &P = f->pd->plan;
    hipStream_t st = op->stream;
    const long long N = (long long)P.dof * P.nz * P.nx;          // unknowns per right-hand side
    const SolveCtx c = solve_ctx(f, ws, nrhs);
    GroupTrace t0(st, "transpose");
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)((N + 31) / 32), (nrhs + 31) / 32), dim3(256), 0, st, Xin, (long long)nrhs, N, c.Xt, 0);
    t0.mark(); t0.report(P, false);
    GroupTrace tf(st, "forward");
    for (size_t gi = 0; gi < P.groups.size(); ++gi) { forward_group(op, f, gi, c); tf.mark(); }       // leaves to root
    tf.report(P, false);
    GroupTrace tb(st, "backward");
    for (size_t gk = P.groups.size(); gk-- > 0;) { backward_group(op, f, gk, c); tb.mark(); }          // root to leaves
    tb.report(P, true);
    GroupTrace t1(st, "transpose");
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)((N + 31) / 32), (nrhs + 31) / 32), dim3(256), 0, st, c.Xt, N, (long long)nrhs, Xout, 1, conj_out);
    t1.mark(); t1.report(P, false);
    return check_kernels(op, "solve kernels");
}

// act[node][b] = 1 for every leaf that holds a cell whose declared support has bit b (block b of 64 columns) set
__global__ __launch_bounds__(256) void k_nd_support_act(const unsigned char *bits, const int *cellnode, int *act, int nct, long long N) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (long long)gridDim.x * blockDim.x) {
        const unsigned b = bits[c];
        if (!b) continue;
        const int nd = cellnode[c];
        if (nd < 0) continue;                                  // separator cells: their fronts look at q themselves
        for (int j = 0; j < nct; ++j) if ((b >> j) & 1) act[(long long)nd * nct + j] = 1;
    }
}
// (HELM_ND_SUPPORT_CHECK=1) bad[0] = 1 when a right-hand side is nonzero outside the declared support
__global__ __launch_bounds__(256) void k_nd_support_check(const unsigned char *bits, const cplx *Q, int ldq, int nrhs, long long N, int *bad) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < N * nrhs; e += (long long)gridDim.x * blockDim.x) {
        const long long c = e / nrhs; const int j = (int)(e - c * nrhs);
        const cplx v = Q[c * ldq + j];
        if ((v.x != 0.0 || v.y != 0.0) && !((bits[c] >> (j >> 6)) & 1)) bad[0] = 1;
    }
}

// node-major solve: Qt (cells x nrhs, read only) -> Xt (may alias Qt); arenaV: 2 * vregion * nrhs elements
// Flags of the forward pass on sparse right-hand sides, zeroed on `st` (HELM_ND_SPARSE_RHS=0: every front is computed, as before round 4).
// A survey's sources touch a handful of grid cells (81 per Kaiser-windowed source at the surface): below the few fronts that contain them the
// forward elimination multiplies zeros -- the reference hands such right-hand sides over as scipy-sparse matrices for the same reason
// (survey.py:86-89, discretization.py:101-103).  Dense right-hand sides set every flag and cost one flag read per workgroup.
static void arm_sparse_rhs(helm_op *op, NdFactor *f, SolveCtx &c, hipStream_t st) {
    const int on = getenv("HELM_ND_SPARSE_RHS") ? atoi(getenv("HELM_ND_SPARSE_RHS")) : 1;        // (read per call: a test compares both)
    c.act = nullptr; c.nct = 0;
    f->act_nct = 0;
    if (!on || c.Qt == c.Xt || gemm_variant() != 7 || f->pd->plan.dof != 1) return;
    const int nct = (c.nrhs + 63) / 64;
    const size_t need = f->pd->plan.nodes.size() * (size_t)nct;
    if (f->act_elems < need) {
        if (f->d_act) { hipStreamSynchronize(st); helm_pool_free(op->device, f->d_act, f->act_elems * sizeof(int)); f->d_act = nullptr; f->act_elems = 0; }
        const size_t want = f->pd->plan.nodes.size() * (size_t)std::max(4, nct);
        f->d_act = (int *)helm_pool_alloc(op->device, want * sizeof(int));
        if (!f->d_act) return;
        f->act_elems = want;
    }
    if (hipMemsetAsync(f->d_act, 0, need * sizeof(int), st) != hipSuccess) { (void)hipGetLastError(); return; }
    c.act = f->d_act; c.nct = nct;
    f->act_nct = nct;
    // Declared support (helm_set_rhs_support: one byte per cell, bit b = block b of 64 columns may be nonzero there; what helm_rhs_support_from_coo makes of
    // the triplets of a scipy-sparse source matrix): the leaves' flags are set from it and the leaf level of the forward pass no longer reads q to find out --
    // 3.2 of the 4.3 GB of a 1024^2 x 256 batch.  Only for the pass whose right-hand sides are the caller's own array (refinement passes solve for residuals).
    if (op->rhs_bits && c.Qt == op->rhs_bits_q && c.nrhs == op->rhs_bits_nrhs && nct <= 8 && f->pd->d_cellnode && op->rhs_bits_rows == (long long)f->pd->plan.nz * f->pd->plan.nx) {
        const long long N = op->rhs_bits_rows;
        hipLaunchKernelGGL(k_nd_support_act, dim3((unsigned)std::min<long long>((N + 255) / 256, 4096)), dim3(256), 0, st, op->rhs_bits, (const int *)f->pd->d_cellnode, f->d_act, nct, N);
        c.act_hint = 1;
        if (getenv("HELM_ND_SUPPORT_CHECK") && atoi(getenv("HELM_ND_SUPPORT_CHECK"))) {
            int *d_bad = (int *)helm_pool_alloc(op->device, sizeof(int));
            if (d_bad) {
                hipMemsetAsync(d_bad, 0, sizeof(int), st);
                hipLaunchKernelGGL(k_nd_support_check, dim3(4096), dim3(256), 0, st, op->rhs_bits, c.Qt, c.nrhs, c.nrhs, N, d_bad);
                int bad = 0;
                hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st);
                hipStreamSynchronize(st);
                helm_pool_free(op->device, d_bad, sizeof(int));
                if (bad) { helm_set_error(op, "a right-hand side is nonzero outside the support declared with helm_set_rhs_support"); op->rhs_bits_violated = 1; }
            }
        }
    }
    // (tests) HELM_ND_POISON=1: the front-vector arena is filled with NaNs first, so that a read of rows no front has written shows up in the wavefield
    if (getenv("HELM_ND_POISON") && atoi(getenv("HELM_ND_POISON"))) (void)hipMemsetAsync(c.arenaV, 0xFF, (size_t)2 * f->pd->plan.vregion * c.nrhs * sizeof(cplx), st);
}

const unsigned char *nd_rhs_mask(helm_op *op, NdFactor *f) {
    if (!f || !f->act_nct || f->act_nct > 8 || !f->d_act || !f->pd->d_cellnode) return nullptr;
    const long long N = (long long)f->pd->plan.nz * f->pd->plan.nx;
    if (f->qmask_elems < (size_t)N) {
        if (f->d_qmask) { hipStreamSynchronize(op->stream); helm_pool_free(op->device, f->d_qmask, f->qmask_elems); f->d_qmask = nullptr; f->qmask_elems = 0; }
        f->d_qmask = (unsigned char *)helm_pool_alloc(op->device, (size_t)N);
        if (!f->d_qmask) return nullptr;
        f->qmask_elems = (size_t)N;
    }
    hipLaunchKernelGGL(k_nd_qmask, dim3((unsigned)std::min<long long>((N + 255) / 256, 4096)), dim3(256), 0, op->stream, (const int *)f->pd->d_cellnode, (const int *)f->d_act, f->act_nct, N, f->d_qmask);
    return f->d_qmask;
}

int nd_solve_nm(helm_op *op, NdFactor *f, const cplx *Qt, cplx *Xt, int nrhs, cplx *arenaV) {
    const NdPlan &P = f->pd->plan;
    SolveCtx c = solve_ctx(f, Xt, nrhs);
    c.Qt = Qt; c.Xt = Xt; c.arenaV = arenaV;
    arm_sparse_rhs(op, f, c, op->stream);
    GroupTrace tf(op->stream, "forward");
    for (size_t gi = 0; gi < P.groups.size(); ++gi) { forward_group(op, f, gi, c); tf.mark(); }
    tf.report(P, false);
    GroupTrace tb(op->stream, "backward");
    for (size_t gk = P.groups.size(); gk-- > 0;) { backward_group(op, f, gk, c); tb.mark(); }
    tb.report(P, true);
    return check_kernels(op, "solve kernels");
}

// events of a factor + solve sweep, destroyed on every exit (an early error return must not leak the ones already created)
namespace {
struct EventSet {
    std::vector<hipEvent_t> plain, timed;
    bool create(size_t nplain, size_t ntimed) {
        for (size_t i = 0; i < nplain; ++i) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false; plain.push_back(e); }
        for (size_t i = 0; i < ntimed; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return false; timed.push_back(e); }
        return true;
    }
    ~EventSet() { for (hipEvent_t e : plain) hipEventDestroy(e); for (hipEvent_t e : timed) hipEventDestroy(e); }
};
}  // namespace

// Factorisation with the forward elimination of one batch running beside it: the forward pass of a tree level only needs that
// level's factors, so it follows the factorisation level by level on a second, LOW-priority stream.  The top of the tree is a chain of
// small dependent launches (80 block steps of a 16-workgroup panel kernel + one update each) that leaves most of the chip idle; the
// forward pass of the lower levels (big HBM-bound launches) fills it, and the priorities keep it from delaying the chain.
int nd_factor_solve_nm(helm_op *op, int block, NdFactor *f, cplx *ws_factor, const cplx *planes_in, const cplx *Qt, cplx *Xt, int nrhs, cplx *arenaV,
                       hipStream_t side, float *factor_ms) {
    const NdPlan &P = f->pd->plan;
    hipStream_t main = op->stream;
    const cplx *planes = nullptr;
    int rc = factor_prologue(op, block, f, planes_in, &planes);
    if (rc) return rc;
    SolveCtx c = solve_ctx(f, Xt, nrhs);
    c.Qt = Qt; c.Xt = Xt; c.arenaV = arenaV;
    const size_t ng = P.groups.size();
    EventSet evs;
    if (!evs.create(ng + 2, 2)) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: hipEventCreate failed");
    std::vector<hipEvent_t> &ev = evs.plain;
    hipEvent_t t0 = evs.timed[0], t1 = evs.timed[1];
    hipEventRecord(ev[ng], main);                 // the right-hand sides were prepared on the main stream
    hipStreamWaitEvent(side, ev[ng], 0);
    arm_sparse_rhs(op, f, c, side);
    hipEventRecord(t0, main);
    for (size_t gi = 0; gi < ng && !rc; ++gi) {
        rc = factor_group(op, f, gi, ws_factor, ws_factor + 2 * P.fregion, planes);
        hipEventRecord(ev[gi], main);
        hipStreamWaitEvent(side, ev[gi], 0);
        op->stream = side;
        forward_group(op, f, gi, c);
        op->stream = main;
    }
    hipEventRecord(t1, main);
    hipEventRecord(ev[ng + 1], side);
    hipStreamWaitEvent(main, ev[ng + 1], 0);
    if (!rc) for (size_t gk = ng; gk-- > 0;) backward_group(op, f, gk, c);
    hipError_t e = hipStreamSynchronize(main);
    if (factor_ms) { float ms = 0.f; if (hipEventElapsedTime(&ms, t0, t1) == hipSuccess) *factor_ms = ms; }
    if (rc) return rc;
    if (e != hipSuccess) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: factor + solve failed: %s", hipGetErrorString(e));
    return check_kernels(op, "factor + solve kernels");
}

int nd_prep_transpose_norm(helm_op *op, const cplx *rhs, long long rhs_ld, long long row_off, cplx premul, const cplx *sub, cplx *Qt, long long N, int nrhs,
                           double *part, int nblk_cap, int *nblk_out) {
    const int nblk = (int)std::max<long long>(1, std::min<long long>((N + 31) / 32, std::min(nblk_cap, 1024)));
    hipLaunchKernelGGL(k_prep_transpose_norm, dim3(nblk, (nrhs + 31) / 32), dim3(256), 0, op->stream, rhs, rhs_ld, row_off, premul, sub, Qt, N, nrhs, part, nblk);
    *nblk_out = nblk;
    return check_kernels(op, "right-hand-side transpose");
}

// r = q - A xin node-major (see k_resid_nm); ncol columns of Xin (leading dimension ldin); returns the partial count per column
int nd_resid_nm(helm_op *op, const cplx *planes, const cplx *Xin, int ldin, cplx *Q, int ldq, const int *qmap, int ncol, int store, cplx *Rout,
                double *part, int nblk_cap, int *nblk_out, const NdResidExtra *ex) {
    const int qnorm = ex ? ex->qnorm : 0;
    cplx *Uout = ex ? ex->Uout : nullptr;
    const int ldu = ex ? ex->ldu : 0;
    const cplx oscale = ex ? ex->oscale : cmake(1.0, 0.0);
    const unsigned char *qmask = (ex && !qmap) ? ex->qmask : nullptr;
    int lx = 64;
    while (lx < ncol && lx < 256) lx <<= 1;
    const int ly = 256 / lx;
    static const int seg_env = getenv("HELM_ND_RESID_SEG") ? atoi(getenv("HELM_ND_RESID_SEG")) : 32;
    const int seg = std::max(8, seg_env);
    // rows per thread: measured on 1024^2 x 256 (norm-only launch, round 2): 1 -> 2.52 ms, 2 -> 3.71 ms, 4 -> 2.23 ms.  Round 3, launch with the
    // wavefield store (13 GB moved): 3.45 ms -> 2.59 ms (5.0 TB/s, the rate of a plain copy on this part) with the coefficients staged in
    // LDS (k_resid_nm_lds, full-width batches); wave-uniform scalar loads of the coefficients instead: 3.3 ms at RPT = 2, SGPR spills at 4.
    // Round 4, LDS kernel without the wavefield store, q read everywhere / q masked: RPT 4 -> 1.57 / 1.20 ms, 6 -> 1.63 / 1.39, 8 -> 2.73 / 2.62
    // (the window of 10 x 3 values halves the occupancy): 4 stays
    static const int rpt_env = getenv("HELM_ND_RESID_RPT") ? atoi(getenv("HELM_ND_RESID_RPT")) : 4;
    const int rpt = (rpt_env == 1 || rpt_env == 2 || rpt_env == 8) ? rpt_env : 4;
    const int ntiles = ((op->nz + rpt - 1) / rpt) * ((op->nx + seg - 1) / seg);
    const int nblk = std::max(1, std::min((ntiles + ly - 1) / ly, std::min(nblk_cap, 2048)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (op->profiling) {
        if (op->ev_used + 2 > op->ev_pool.size() && op->ev_pool.size() < 16384) (void)helm_events_grow(op, 64);
        if (op->ev_used + 2 <= op->ev_pool.size()) { e0 = op->ev_pool[op->ev_used]; e1 = op->ev_pool[op->ev_used + 1]; }
    }
    if (e0) hipEventRecord(e0, op->stream);
    for (int c0 = 0; c0 < ncol; c0 += 256) {          // more than 256 columns: one launch per 256 (partials of later chunks follow the first)
        const int nc = std::min(256, ncol - c0);
#define RESID_LAUNCH(RPT_) hipLaunchKernelGGL(k_resid_nm<RPT_>, dim3(nblk), dim3(lx, ly), 0, op->stream, planes, op->nz, op->nx, Xin + c0, ldin, Q + (qmap ? 0 : c0), ldq, \
                           qmap ? qmap + c0 : nullptr, nc, store, Rout ? Rout + (qmap ? 0 : c0) : nullptr, part + (long long)c0 * 4 * nblk, nblk, seg, ntiles, \
                           qnorm, Uout ? Uout + c0 : nullptr, ldu, oscale)
        static const int lds_env = getenv("HELM_ND_RESID_LDS") ? atoi(getenv("HELM_ND_RESID_LDS")) : 1;
        if (lds_env && ly == 1 && seg == RESID_SEG && (rpt == 2 || rpt == 4)) {
#define RESID_LDS(RPT_) hipLaunchKernelGGL((k_resid_nm_lds<RPT_, NTS_>), dim3(nblk), dim3(256, 1), 0, op->stream, planes, op->nz, op->nx, Xin + c0, ldin, Q + (qmap ? 0 : c0), ldq, \
                           qmap ? qmap + c0 : nullptr, nc, store, Rout ? Rout + (qmap ? 0 : c0) : nullptr, part + (long long)c0 * 4 * nblk, nblk, ntiles, \
                           qnorm, Uout ? Uout + c0 : nullptr, ldu, oscale, c0 == 0 ? qmask : nullptr)
            // the wavefield is written once and read by nobody on the GPU: nontemporal stores (2578 -> 2549 us with the store, within the run-to-run spread; q loaded
            // nontemporally as well: no difference)
            static const int nt_store = getenv("HELM_ND_RESID_NT") ? atoi(getenv("HELM_ND_RESID_NT")) : 1;
            if (nt_store) {
#define NTS_ 1
                if (rpt == 2) RESID_LDS(2); else RESID_LDS(4);
#undef NTS_
            } else {
#define NTS_ 0
                if (rpt == 2) RESID_LDS(2); else RESID_LDS(4);
#undef NTS_
            }
#undef RESID_LDS
            continue;
        }
        if (rpt == 1) RESID_LAUNCH(1); else if (rpt == 2) RESID_LAUNCH(2); else if (rpt == 8) RESID_LAUNCH(8); else RESID_LAUNCH(4);
#undef RESID_LAUNCH
    }
    if (e0) {
        hipEventRecord(e1, op->stream);
        // algorithmic bytes of what this launch has to move (SURVEY.md 8(d) convention: operands once, halo re-reads not counted):
        // the input columns and q (16 B each per point and column), the nine coefficients (144 B per point); r written only when it
        // is stored (+16)
        op->ev_pending.push_back(std::make_pair((int)op->ev_used, (double)op->N * ((32.0 + (store ? 16.0 : 0.0) + (Uout ? 16.0 : 0.0)) * ncol + 144.0)));
        op->ev_used += 2;
    }
    *nblk_out = nblk;
    return check_kernels(op, "node-major residual");
}

int nd_scatter_add_cols(helm_op *op, cplx *Xt, int ldq, const int *d_cols, int k, const cplx *Dp, long long N) {
    hipLaunchKernelGGL(k_scatter_add_cols, dim3((unsigned)std::min<long long>((N * k + 255) / 256, 1 << 20)), dim3(256), 0, op->stream, Xt, ldq, d_cols, k, Dp, N);
    return check_kernels(op, "column scatter");
}

int nd_pack_cols(helm_op *op, const cplx *Qt, int ldq, const int *d_cols, int k, cplx *Rp, long long N) {
    hipLaunchKernelGGL(k_pack_cols, dim3((unsigned)std::min<long long>((N * k + 255) / 256, 1 << 20)), dim3(256), 0, op->stream, Qt, ldq, d_cols, k, Rp, N);
    return check_kernels(op, "column packing");
}

// out[c][r] = in[r][c] for an (rows x cols) array (layout conversions at the C ABI: the reference's (N, nrhs) arrays <-> one right-hand side per row)
int nd_transpose(helm_op *op, const cplx *in, long long rows, long long cols, cplx *out) {
    const bool swap = rows > cols;            // the long dimension rides on gridDim.x
    dim3 grid = swap ? dim3((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32)) : dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, op->stream, in, rows, cols, out, swap ? 1 : 0, 0);
    return check_kernels(op, "transpose");
}

// Xt (cells x nrhs) -> U (nrhs x N), conjugated on request
int nd_transpose_out(helm_op *op, const cplx *Xt, long long N, int nrhs, cplx *U, int conj) {
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)((N + 31) / 32), (nrhs + 31) / 32), dim3(256), 0, op->stream, Xt, N, (long long)nrhs, U, 1, conj);
    return check_kernels(op, "transpose out");
}

// Factorisation and the first solve in one sweep: the forward elimination of a tree level only needs that level's factors,
// so it follows the factorisation on a second stream, level by level.  The upper tree levels of the factorisation are a chain
// of small latency-bound launches; the forward pass of the lower levels (big batched GEMMs) runs underneath it.
// ws_factor / ws_solve as for nd_factor / nd_solve (disjoint); *factor_ms: time until the last front is factored.
int nd_factor_solve(helm_op *op, int block, NdFactor *f, cplx *ws_factor, const cplx *planes_in, const cplx *Xin, cplx *Xout, int nrhs,
                    cplx *ws_solve, hipStream_t side, float *factor_ms, int conj_out) {
    const NdPlan &P = f->pd->plan;
    hipStream_t main = op->stream;
    const long long N = (long long)P.dof * P.nz * P.nx;
    const cplx *planes = nullptr;
    int rc = factor_prologue(op, block, f, planes_in, &planes);
    if (rc) return rc;
    const SolveCtx c = solve_ctx(f, ws_solve, nrhs);
    const size_t ng = P.groups.size();
    EventSet evs;
    if (!evs.create(ng + 2, 2)) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: hipEventCreate failed");
    std::vector<hipEvent_t> &ev = evs.plain;
    hipEvent_t t0 = evs.timed[0], t1 = evs.timed[1];
    // the right-hand sides were prepared on the main stream
    hipEventRecord(ev[ng], main);
    hipStreamWaitEvent(side, ev[ng], 0);
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)((N + 31) / 32), (nrhs + 31) / 32), dim3(256), 0, side, Xin, (long long)nrhs, N, c.Xt, 0);
    hipEventRecord(t0, main);
    for (size_t gi = 0; gi < ng && !rc; ++gi) {
        rc = factor_group(op, f, gi, ws_factor, ws_factor + 2 * P.fregion, planes);
        hipEventRecord(ev[gi], main);
        hipStreamWaitEvent(side, ev[gi], 0);
        op->stream = side;
        forward_group(op, f, gi, c);
        op->stream = main;
    }
    hipEventRecord(t1, main);
    hipEventRecord(ev[ng + 1], side);
    hipStreamWaitEvent(main, ev[ng + 1], 0);
    if (!rc) {
        for (size_t gk = ng; gk-- > 0;) backward_group(op, f, gk, c);
        hipLaunchKernelGGL(k_transpose, dim3((unsigned)((N + 31) / 32), (nrhs + 31) / 32), dim3(256), 0, main, c.Xt, N, (long long)nrhs, Xout, 1, conj_out);
    }
    hipError_t e = hipStreamSynchronize(main);
    if (factor_ms) { float ms = 0.f; if (hipEventElapsedTime(&ms, t0, t1) == hipSuccess) *factor_ms = ms; }
    if (rc) return rc;
    if (e != hipSuccess) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: factor + solve failed: %s", hipGetErrorString(e));
    return check_kernels(op, "factor + solve kernels");
}

// dense helpers for other translation units (3-D multigrid: coarsest-level inverse and its application)
int nd_dense_gemm(helm_op *op, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, const cplx *B, int ldb, cplx beta, cplx *C, int ldc) {
    gemm(op, M, Nn, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1);
    return check_kernels(op, "dense GEMM");
}
int nd_dense_gemm_batched(helm_op *op, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                          cplx beta, cplx *C, int ldc, long long sc, int batch) {
    gemm(op, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, batch);
    return check_kernels(op, "dense GEMM");
}
int nd_dense_inverse(helm_op *op, cplx *M, int n, cplx *W) {
    invert(op, M, n, (long long)n * n, n, 1, W, (long long)n * n);
    return check_kernels(op, "dense inverse");
}

int nd_axpy_one(helm_op *op, cplx *y, const cplx *x, long long n, int conj) {
    hipLaunchKernelGGL(k_axpy_one, dim3((unsigned)std::min<long long>((n + 255) / 256, 65535)), dim3(256), 0, op->stream, y, x, n, conj);
    return HELM_OK;
}

// ---- diagnostics exported through the C ABI (host side of the plan; dense kernels on small inputs) ---------------------
extern "C" int helm_direct_plan(int nz, int nx, int leaf, int *out, int cap) {
    NdPlan P;
    nd_build_plan(P, nz, nx, leaf);
    const int nn = (int)P.nodes.size();
    if (!out) return nn;
    for (int i = 0; i < nn && i < cap; ++i) {
        const NdDev &n = P.nodes[i];
        int *o = out + 12 * i;
        o[0] = n.z0; o[1] = n.z1; o[2] = n.x0; o[3] = n.x1; o[4] = n.cut; o[5] = n.pos; o[6] = n.s; o[7] = n.m;
        o[8] = n.kid[0]; o[9] = n.kid[1]; o[10] = n.smax; o[11] = n.mmax;
    }
    return nn;
}

// cells (z * nx + x) of the front of node `node` in local order; returns s + m, or a negative value when the
// inverse map nd_local disagrees with nd_cell (self-check of the closed-form index maps)
extern "C" int helm_direct_plan_front(int nz, int nx, int leaf, int node, long long *cells, int cap) {
    NdPlan P;
    nd_build_plan(P, nz, nx, leaf);
    if (node < 0 || node >= (int)P.nodes.size()) return HELM_ERR_ARG;
    const NdDev &n = P.nodes[node];
    for (int a = 0; a < n.s + n.m; ++a) {
        int z, x;
        nd_cell(n, a, z, x);
        if (z < 0 || z >= nz || x < 0 || x >= nx) return -100;
        if (nd_local(n, nz, nx, z, x) != a) return -101;
        if (cells && a < cap) cells[a] = (long long)z * nx + x;
    }
    return n.s + n.m;
}

extern "C" int helm_debug_zgemm(int device, int M, int Nn, int K, const double *alpha, const double *A, const double *B, const double *beta, double *C, int batch) {
    if (hipSetDevice(device) != hipSuccess) return HELM_ERR_DEVICE;
    cplx *dA, *dB, *dC;
    const size_t na = (size_t)batch * M * K, nb = (size_t)batch * K * Nn, nc = (size_t)batch * M * Nn;
    if (hipMalloc((void **)&dA, na * 16) != hipSuccess || hipMalloc((void **)&dB, nb * 16) != hipSuccess || hipMalloc((void **)&dC, nc * 16) != hipSuccess) return HELM_ERR_DEVICE;
    hipMemcpy(dA, A, na * 16, hipMemcpyHostToDevice); hipMemcpy(dB, B, nb * 16, hipMemcpyHostToDevice); hipMemcpy(dC, C, nc * 16, hipMemcpyHostToDevice);
    // (with a handle, like the solver's own calls: the paths that keep scratch on it -- the split over the inner dimension -- are taken too)
    const int fs[4] = {0, 0, 0, 0};
    helm_op *tmp = helm_create(device, 0, 8, 8, 1.0, 1.0, 2, fs);
    gemm(tmp, M, Nn, K, cmake(alpha[0], alpha[1]), dA, K, (long long)M * K, dB, Nn, (long long)K * Nn, cmake(beta[0], beta[1]), dC, Nn, (long long)M * Nn, batch);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(C, dC, nc * 16, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dC);
    if (tmp) helm_destroy(tmp);
    return e == hipSuccess ? HELM_OK : HELM_ERR_DEVICE;
}

extern "C" int helm_debug_inverse(int device, int n, double *A, int batch) {
    if (hipSetDevice(device) != hipSuccess) return HELM_ERR_DEVICE;
    cplx *dA, *dW;
    const size_t na = (size_t)batch * n * n;
    if (hipMalloc((void **)&dA, na * 16) != hipSuccess || hipMalloc((void **)&dW, na * 16) != hipSuccess) return HELM_ERR_DEVICE;
    hipMemcpy(dA, A, na * 16, hipMemcpyHostToDevice);
    invert((helm_op *)nullptr, dA, n, (long long)n * n, n, batch, dW, (long long)n * n);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(A, dA, na * 16, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dW);
    return e == hipSuccess ? HELM_OK : HELM_ERR_DEVICE;
}

// times `reps` in-place inversions of one n x n matrix (the caller's A, uploaded once; an inverse of an inverse is as good a test
// matrix as the original) with the 2 x 2 block recursion applied from `recurse_n` unknowns up (0: the default policy)
extern "C" int helm_debug_inverse_bench(int device, int n, const double *A, int reps, int recurse_n, double *ms_out) {
    if (hipSetDevice(device) != hipSuccess) return HELM_ERR_DEVICE;
    cplx *dA, *dW;
    const size_t na = (size_t)n * n;
    if (hipMalloc((void **)&dA, na * 16) != hipSuccess || hipMalloc((void **)&dW, na * 16) != hipSuccess) return HELM_ERR_DEVICE;
    hipMemcpy(dA, A, na * 16, hipMemcpyHostToDevice);
    g_recurse_min = recurse_n > 0 ? recurse_n : -1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    invert((helm_op *)nullptr, dA, n, (long long)n * n, n, 1, dW, (long long)n * n);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) invert((helm_op *)nullptr, dA, n, (long long)n * n, n, 1, dW, (long long)n * n);
    hipEventRecord(e1, nullptr);
    const hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / std::max(1, reps);
    g_recurse_min = -1;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(dA); hipFree(dW);
    return e == hipSuccess ? HELM_OK : HELM_ERR_DEVICE;
}

// times `reps` launches of one strided-batched GEMM shape on random operands with kernel variant `variant` (see HELM_ND_GEMMV;
// -1: the default); returns the average milliseconds per launch in *ms
extern "C" int helm_debug_zgemm_bench(int device, int M, int Nn, int K, int batch, int variant, int reps, double *ms_out) {
    if (hipSetDevice(device) != hipSuccess) return HELM_ERR_DEVICE;
    cplx *dA, *dB, *dC;
    const size_t na = (size_t)batch * M * K, nb = (size_t)batch * K * Nn, nc = (size_t)batch * M * Nn;
    if (hipMalloc((void **)&dA, na * 16) != hipSuccess || hipMalloc((void **)&dB, nb * 16) != hipSuccess || hipMalloc((void **)&dC, nc * 16) != hipSuccess) return HELM_ERR_DEVICE;
    std::vector<cplx> h(std::max(na, nb));
    unsigned long long st = 88172645463325252ULL;
    for (size_t i = 0; i < h.size(); ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = cmake((double)(st & 0xffff) / 65536.0 - 0.5, (double)((st >> 16) & 0xffff) / 65536.0 - 0.5); }
    hipMemcpy(dA, h.data(), na * 16, hipMemcpyHostToDevice); hipMemcpy(dB, h.data(), nb * 16, hipMemcpyHostToDevice);
    hipMemset(dC, 0, nc * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    g_gemm_variant = variant < 0 ? -1 : (variant & 15);          // variant = kernel generation + 16 * (tile + 1)
    g_gemm_tile = variant >= 16 ? (variant >> 4) - 1 : -1;
    for (int w = 0; w < 2; ++w) gemm((helm_op *)nullptr, M, Nn, K, cmake(1, 0), dA, K, (long long)M * K, dB, Nn, (long long)K * Nn, cmake(0, 0), dC, Nn, (long long)M * Nn, batch);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) gemm((helm_op *)nullptr, M, Nn, K, cmake(1, 0), dA, K, (long long)M * K, dB, Nn, (long long)K * Nn, cmake(0, 0), dC, Nn, (long long)M * Nn, batch);
    hipEventRecord(e1, nullptr);
    hipError_t e = hipEventSynchronize(e1);
    g_gemm_variant = -1; g_gemm_tile = -1;
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = ms / std::max(1, reps);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(dA); hipFree(dB); hipFree(dC);
    return e == hipSuccess ? HELM_OK : HELM_ERR_DEVICE;
}
