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
// All dense arithmetic is fp64 complex on the vector ALUs (MI355X: fp64 MFMA rate == fp64 vector FMA rate, so
// MFMA buys nothing here); the solve phase is bound by reading the factors and the right-hand sides from HBM.
#include "helm_internal.hpp"
#include "direct.hpp"
#include <algorithm>
#include <mutex>

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
    // processing order: deepest level first; within a level the leaves, then the separators
    std::vector<int> order; order.reserve(nn);
    P.groups.clear();
    for (int lev = maxlev; lev >= 0; --lev)
        for (int kind = 0; kind < 2; ++kind) {
            NdGroup g = NdGroup(); g.first = (int)order.size(); g.level = lev; g.leaf = kind == 0;
            for (int i = 0; i < nn; ++i)
                if (B.level[i] == lev && (B.nodes[i].cut < 0) == (kind == 0)) {
                    order.push_back(i);
                    g.smax = std::max(g.smax, B.nodes[i].s); g.mmax = std::max(g.mmax, B.nodes[i].m);
                }
            g.cnt = (int)order.size() - g.first;
            if (g.cnt > 0) P.groups.push_back(g);
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
        g.finv = fac; fac += (long long)g.cnt * g.smax * g.smax;
        g.g21 = fac; fac += (long long)g.cnt * g.mmax * g.smax;
        g.f12 = fac; fac += (long long)g.cnt * g.smax * g.mmax;
    }
    P.fac_elems = fac;
    P.fregion = 0; P.vregion = 0; P.work_elems = 0;
    for (int l = 0; l <= maxlev; ++l) { P.fregion = std::max(P.fregion, lev_f[l]); P.vregion = std::max(P.vregion, lev_v[l]); }
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        NdGroup &g = P.groups[gi];
        g.foff += (long long)(g.level & 1) * P.fregion;
        g.voff += (long long)(g.level & 1) * P.vregion;
        P.work_elems = std::max(P.work_elems, (long long)g.cnt * g.smax * g.smax);
        for (int j = 0; j < g.cnt; ++j) {
            NdDev &n = P.nodes[g.first + j];
            const long long nmax = g.smax + g.mmax;
            n.smax = g.smax; n.mmax = g.mmax;
            n.foff = g.foff + (long long)j * g.mmax * nmax;
            n.finv_off = g.finv + (long long)j * g.smax * g.smax;
            n.f12_off = g.f12 + (long long)j * g.smax * g.mmax;
            n.voff = g.voff + (long long)j * nmax;
            n.roff = g.roff + (long long)j * nmax;
        }
    }
    return HELM_OK;
}

// ---- kernels ---------------------------------------------------------------------------------------------------
namespace {

// The front [[F11, F12], [F21, F22]] is assembled where each block is needed afterwards: F11 and F12 in the factor storage
// (F11 is inverted in place, F12 is kept as it is), [F21 | F22] in the scratch arena (F21 feeds G21, F22 becomes the Schur
// complement the parent picks up).  (r, c): padded front coordinates.
__device__ __forceinline__ cplx *front_entry(const NdDev &n, cplx *arenaF, cplx *fac, int r, int c) {
    if (r < n.smax) {
        if (c < n.smax) return fac + n.finv_off + (long long)r * n.smax + c;
        return fac + n.f12_off + (long long)r * n.mmax + (c - n.smax);
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
        for (int cb = 0; cb < n.dof; ++cb) {
            // dof 2: row component ca, column component cb -> Eurus block 2 ca + cb (M1 M2 / M3 M4), nine planes each
            const cplx *pl = planes + (long long)(n.dof == 2 ? 2 * ca + cb : 0) * 9 * N;
            #pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int z2 = z + k / 3 - 1, x2 = x + k % 3 - 1;
                if (z2 < 0 || z2 >= nz || x2 < 0 || x2 >= nx) continue;
                const int b = nd_local(n, nz, nx, z2, x2, cb);
                if (b < 0 || (a >= n.s && b >= n.s)) continue;            // ring x ring entries belong to an ancestor
                *front_entry(n, arenaF, fac, ra, nd_pos(n, b)) = pl[(long long)k * N + (long long)z * nx + x];
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

// ---- strided-batched complex GEMM: C = beta C + alpha A B, row-major ----------------------------------------------
// 64x64 tile, K step 8, 256 threads each owning a 4x4 block; the next K slab is fetched into registers while the
// current one is multiplied out of LDS.
//
// IDX variant (solve phase): rows of B, of the C that is read (beta != 0) and of the C that is written may be taken
// through the plan's row table instead of a dense front buffer, i.e. straight from / to the node-major right-hand sides
// Xt[cell][rhs]:   row r of batch item z  ->  X + tab[z * tab_stride + off + r].x * ldx   (negative: a zero row / not stored).
// This removes the gather / scatter passes (and their HBM round trips) from the lower tree levels.
struct GemmRows {
    const int4 *tabB = nullptr, *tabCi = nullptr, *tabCo = nullptr;
    int offB = 0, offCi = 0, offCo = 0, tab_stride = 0;
    const cplx *Bx = nullptr, *Cix = nullptr; cplx *Cox = nullptr;
    int ldx = 0;
    int z0 = 0;           // batch index of blockIdx.z == 0 (launches are chunked along z)
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
void launch_vec(hipStream_t st, bool idx, int nb, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
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
template <int TM, bool IDX, int RN, int KS, int UNR, int OCC>
__global__ __launch_bounds__(256, OCC) void k_zgemm2(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                                const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, GemmRows R) {
    constexpr int TXN = 1024 / TM, TN = TXN * RN;
    constexpr int NA = (TM * KS + 255) / 256, NB = (TN * KS + 255) / 256;
    __shared__ cplx As[2][KS][TM + 1];
    __shared__ cplx Bs[2][KS][TN];
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
                if (idxB) { const int r = kidx[k0 + bk]; if (r >= 0) v = R.Bx[(long long)r * R.ldx + n0 + bc]; }
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
            const int cc = n0 + j * TXN + tx;
            if (cc >= Nn) continue;
            cplx v = cmul(alpha, acc[i][j]);
            if (!b0 && cin) v = cadd(v, cmul(beta, cin[cc]));
            dst[cc] = v;
        }
    }
}

template <int TM, int RN, int KS, int UNR = 1, int OCC = 1>
void launch_vec2(hipStream_t st, bool idx, int nb, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                 cplx beta, cplx *C, int ldc, long long sc, const GemmRows &R) {
    constexpr int TN = (1024 / TM) * RN;
    dim3 grid((Nn + TN - 1) / TN, (M + TM - 1) / TM, nb);
    if (idx) hipLaunchKernelGGL((k_zgemm2<TM, true, RN, KS, UNR, OCC>), grid, dim3(256), 0, st, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else hipLaunchKernelGGL((k_zgemm2<TM, false, RN, KS, UNR, OCC>), grid, dim3(256), 0, st, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
}

// ---- the same GEMM on the matrix cores -------------------------------------------------------------------------
// v_mfma_f64_16x16x4_f64: lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15]; result register r of lane l is
// C[(l >> 4) + 4 r][l & 15].  A complex product is four real MFMAs on the (re, im) planes of the same lane data:
//   Cre += Are Bre - Aim Bim ;  Cim += Are Bim + Aim Bre.
// fp64 MFMA issues at the same flop rate as the vector FMAs, but one LDS read of 16 B per lane feeds 1024 real
// multiply-adds instead of 16, so the kernel is no longer bound by LDS bandwidth (the vector tile kernel is).
// Block = 4 waves arranged WM x WN; each wave owns RA x RB blocks of 16 x 16; tile = (16 RA WM) x (16 RB WN).
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int RA, int RB, int WM, int WN, bool IDX>
__global__ __launch_bounds__(256) void k_zgemm_mfma(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                                    const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, GemmRows R) {
    constexpr int TM = 16 * RA * WM, TN = 16 * RB * WN;
    constexpr int NA = (TM * GB_K + 255) / 256, NB = (TN * GB_K + 255) / 256;
    __shared__ cplx As[GB_K][TM + 1];
    __shared__ cplx Bs[GB_K][TN + 1];
    __shared__ int kidx[IDX ? GB_KIDX : 1];
    const cplx *A = A0 + (long long)blockIdx.z * sa;
    const cplx *B = B0 + (long long)blockIdx.z * sb;
    cplx *C = C0 + (long long)blockIdx.z * sc;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long long trow = IDX ? (long long)(R.z0 + blockIdx.z) * R.tab_stride : 0;
    const bool idxB = IDX && R.tabB != nullptr;
    if (idxB) {
        for (int k = tid; k < K; k += 256) kidx[k] = R.tabB[trow + R.offB + k].x;
        __syncthreads();
    }
    v4f64 cre[RA][RB], cim[RA][RB];
    #pragma unroll
    for (int i = 0; i < RA; ++i)
        #pragma unroll
        for (int j = 0; j < RB; ++j) { cre[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0}; cim[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0}; }
    cplx ra[NA], rb[NB];
    ZG_FETCH(0)
    const int lr = lane & 15, lk = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += GB_K) {
        #pragma unroll
        for (int e = 0; e < NA; ++e) { const int idx = tid + e * 256; if (idx < TM * GB_K) As[idx & 7][idx >> 3] = ra[e]; }
        #pragma unroll
        for (int e = 0; e < NB; ++e) { const int idx = tid + e * 256; if (idx < TN * GB_K) Bs[idx / TN][idx % TN] = rb[e]; }
        __syncthreads();
        if (k0 + GB_K < K) { ZG_FETCH(k0 + GB_K) }
        #pragma unroll
        for (int kk = 0; kk < GB_K; kk += 4) {
            cplx a[RA], b[RB];
            #pragma unroll
            for (int i = 0; i < RA; ++i) a[i] = As[kk + lk][(wm * RA + i) * 16 + lr];
            #pragma unroll
            for (int j = 0; j < RB; ++j) b[j] = Bs[kk + lk][(wn * RB + j) * 16 + lr];
            // four sweeps over the RA x RB accumulator blocks, so that two MFMAs on the same accumulator are 2 RA RB - 1
            // instructions apart (back-to-back dependent MFMAs stall for the full pass latency)
            #pragma unroll
            for (int i = 0; i < RA; ++i)
                #pragma unroll
                for (int j = 0; j < RB; ++j) {
                    cre[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, cre[i][j], 0, 0, 0);
                    cim[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].y, cim[i][j], 0, 0, 0);
                }
            #pragma unroll
            for (int i = 0; i < RA; ++i)
                #pragma unroll
                for (int j = 0; j < RB; ++j) {
                    cre[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[i].y, b[j].y, cre[i][j], 0, 0, 0);
                    cim[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].x, cim[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    const bool b0 = (beta.x == 0.0 && beta.y == 0.0);
    #pragma unroll
    for (int i = 0; i < RA; ++i)
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + (wm * RA + i) * 16 + lk + 4 * r;
            if (row >= M) continue;
            cplx *dst = C + (long long)row * ldc;
            const cplx *cin = dst;
            if (IDX && R.tabCo) {
                const int ix = R.tabCo[trow + R.offCo + row].x;
                if (ix < 0) continue;
                dst = R.Cox + (long long)ix * R.ldx;
            }
            if (IDX && R.tabCi && !b0) {
                const int ix = R.tabCi[trow + R.offCi + row].x;
                cin = ix >= 0 ? R.Cix + (long long)ix * R.ldx : nullptr;
            }
            #pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int cc = n0 + (wn * RB + j) * 16 + lr;
                if (cc >= Nn) continue;
                cplx v = cmul(alpha, cmake(cre[i][j][r], cim[i][j][r]));
                if (!b0 && cin) v = cadd(v, cmul(beta, cin[cc]));
                dst[cc] = v;
            }
        }
}

template <int RA, int RB, int WM, int WN>
void launch_mfma(hipStream_t st, dim3 grid, bool idx, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                 cplx beta, cplx *C, int ldc, long long sc, const GemmRows &R) {
    if (idx) hipLaunchKernelGGL((k_zgemm_mfma<RA, RB, WM, WN, true>), grid, dim3(256), 0, st, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else hipLaunchKernelGGL((k_zgemm_mfma<RA, RB, WM, WN, false>), grid, dim3(256), 0, st, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
}

// ---- in-place inverse of n x n blocks, n <= 64: Gauss-Jordan with row pivoting in LDS, one workgroup per matrix ----
// (bottom of the recursive block inversion: 32 by default; 64 (HELM_ND_GJ=64) halves the number of small GEMM launches
// and gains a digit of accuracy, but its 64-step elimination is slower overall: 41.8 vs 35.9 ms per factorisation at 1024^2)
#define GJ_MAX 64
template <int NMAX, int NT = 256>
__global__ __launch_bounds__(NT) void k_gj_inverse(cplx *A0, int ld, long long stride, int n) {
    __shared__ cplx a[NMAX][NMAX + 1];
    __shared__ cplx fcol[NMAX];
    __shared__ int piv[NMAX];
    cplx *A = A0 + (long long)blockIdx.x * stride;
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += blockDim.x) a[e / n][e % n] = A[(long long)(e / n) * ld + e % n];
    __syncthreads();
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
        for (int e = tid; e < n * n; e += blockDim.x) {
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
    for (int e = tid; e < n * n; e += blockDim.x) A[(long long)(e / n) * ld + e % n] = a[e / n][e % n];
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
__global__ __launch_bounds__(256) void k_nd_fwd_rows(const int4 *tab, cplx *V, const cplx *arenaV, cplx *Xt, long long rows, int nrhs, int write_back) {
    for (long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y; row < rows; row += (long long)gridDim.x * blockDim.y) {
        const int4 e = tab[row];
        cplx *dst = V + row * nrhs;
        const cplx *s0 = e.w ? Xt + (long long)e.x * nrhs : nullptr;
        const cplx *s1 = e.y >= 0 ? arenaV + (long long)e.y * nrhs : nullptr;
        const cplx *s2 = e.z >= 0 ? arenaV + (long long)e.z * nrhs : nullptr;
        for (int r = threadIdx.x; r < nrhs; r += blockDim.x) {
            cplx acc = s0 ? s0[r] : cmake(0.0, 0.0);
            if (s1) acc = cadd(acc, s1[r]);
            if (s2) acc = cadd(acc, s2[r]);
            dst[r] = acc;
            if (write_back && e.w) Xt[(long long)e.x * nrhs + r] = acc;
        }
    }
}

// backward pass: V[row] = Xt[cell of the row] (separator and ring rows), 0 for padding
__global__ __launch_bounds__(256) void k_nd_bwd_gather(const int4 *tab, cplx *V, const cplx *Xt, long long rows, int nrhs) {
    for (long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y; row < rows; row += (long long)gridDim.x * blockDim.y) {
        const int4 e = tab[row];
        cplx *dst = V + row * nrhs;
        const cplx *src = e.x >= 0 ? Xt + (long long)e.x * nrhs : nullptr;
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

// op may be null (diagnostic entry points): default stream, no profiling
int gemm(helm_op *op, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
         cplx beta, cplx *C, int ldc, long long sc, int batch, const GemmRows *rows = nullptr) {
    if (M <= 0 || Nn <= 0 || batch <= 0) return 0;
    hipStream_t st = op ? op->stream : nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool in_run = op && op->gemm_run_depth > 0;
    if (op && op->profiling && !(in_run && op->gemm_run_pair >= 0)) {
        if (op->ev_used + 2 > op->ev_pool.size() && op->ev_pool.size() < 16384)
            (void)helm_events_grow(op, 64);
        if (op->ev_used + 2 <= op->ev_pool.size()) { e0 = op->ev_pool[op->ev_used]; e1 = op->ev_pool[op->ev_used + 1]; }
    }
    if (e0) hipEventRecord(e0, st);
    if (in_run && e0) { op->gemm_run_pair = (int)op->ev_used; op->ev_used += 2; }
    static const int use_mfma = getenv("HELM_ND_MFMA") ? atoi(getenv("HELM_ND_MFMA")) : 0;
    static const int fixed_tm = getenv("HELM_ND_TM") ? atoi(getenv("HELM_ND_TM")) : 0;
    // tile of the vector kernel: 4 x RN outputs per thread, TM x TN = TM x (1024 / TM * RN).  The padded area is weighed by how
    // well a register block re-uses its LDS reads (4 x 4: 1, 4 x 2: 0.7, 4 x 1: 0.45): narrow tiles only win on small outputs
    // (the 32 x 32 blocks at the bottom of the inversion recursion, 17- and 35-wide separators of the middle tree levels)
    static const int vc_tm[8] = {64, 32, 16, 64, 32, 16, 32, 16}, vc_rn[8] = {4, 4, 4, 2, 2, 2, 1, 1};
    static const double w2 = getenv("HELM_ND_EFF2") ? atof(getenv("HELM_ND_EFF2")) : 0.7, w1 = getenv("HELM_ND_EFF1") ? atof(getenv("HELM_ND_EFF1")) : 0.45;
    static const double vc_eff[8] = {1.0, 1.0, 1.0, w2, w2, w2, w1, w1};
    static const int narrow = getenv("HELM_ND_NARROW") ? atoi(getenv("HELM_ND_NARROW")) : 1;
    int vsel = 0; double vcost = -1;
    for (int c = 0; c < (narrow ? 8 : 3); ++c) {
        const int tm = vc_tm[c], tn = 1024 / tm * vc_rn[c];
        const double cost = (double)((M + tm - 1) / tm) * tm * ((Nn + tn - 1) / tn) * tn / vc_eff[c];
        if (vcost < 0 || cost < vcost * 0.999) { vsel = c; vcost = cost; }
    }
    if (fixed_tm == 64) vsel = 0; else if (fixed_tm == 32) vsel = 1; else if (fixed_tm == 16) vsel = 2;
    // tile shapes of the MFMA kernel: pick the one that pads M x N the least (ties: the larger tile)
    static const int cfg_tm[4] = {64, 48, 32, 16}, cfg_tn[4] = {64, 128, 128, 256};
    int best = 0; long long best_area = -1;
    for (int c = 0; c < 4; ++c) {
        const long long area = (long long)((M + cfg_tm[c] - 1) / cfg_tm[c]) * cfg_tm[c] * ((Nn + cfg_tn[c] - 1) / cfg_tn[c]) * cfg_tn[c];
        if (best_area < 0 || area < best_area) { best = c; best_area = area; }
    }
    for (int b0 = 0; b0 < batch; b0 += 65535) {
        const int nb = std::min(65535, batch - b0);
        GemmRows R; if (rows) R = *rows;
        R.z0 = b0;
        const cplx *Ab = A + b0 * sa, *Bb = B ? B + b0 * sb : B;
        cplx *Cb = C ? C + b0 * sc : C;
        if (use_mfma) {
            dim3 grid((Nn + cfg_tn[best] - 1) / cfg_tn[best], (M + cfg_tm[best] - 1) / cfg_tm[best], nb);
            switch (best) {
                case 0: launch_mfma<2, 2, 2, 2>(st, grid, rows != nullptr, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R); break;
                case 1: launch_mfma<3, 2, 1, 4>(st, grid, rows != nullptr, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R); break;
                case 2: launch_mfma<2, 2, 1, 4>(st, grid, rows != nullptr, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R); break;
                default: launch_mfma<1, 4, 1, 4>(st, grid, rows != nullptr, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R); break;
            }
            continue;
        }
        // (a 4 x 8 register block per thread -- RN = 8, 64 x 128 tile -- was measured too: 230 VGPRs, occupancy 2, 38 % slower)
        // 0: first-generation kernel; v2 kernel: 1: K slab 8, 2: K slab 16, 3: K slab 8 + k loop unrolled twice, 4: K slab 16 unrolled twice,
        // 5: K slab 8 with the register budget of 4 waves per SIMD
        static const int gemm_v = getenv("HELM_ND_GEMMV") ? atoi(getenv("HELM_ND_GEMMV")) : 1;
        const int gv = g_gemm_variant >= 0 ? g_gemm_variant : gemm_v;
#define ZG_ARGS st, rows != nullptr, nb, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R
#define ZG_VEC(TM_, RN_) do { switch (gv) { \
            case 0: launch_vec<TM_, RN_>(ZG_ARGS); break; \
            case 2: launch_vec2<TM_, RN_, 16, 1, 1>(ZG_ARGS); break; \
            case 3: launch_vec2<TM_, RN_, 8, 2, 1>(ZG_ARGS); break; \
            case 4: launch_vec2<TM_, RN_, 16, 2, 1>(ZG_ARGS); break; \
            case 5: launch_vec2<TM_, RN_, 8, 1, 4>(ZG_ARGS); break; \
            default: launch_vec2<TM_, RN_, 8, 1, 1>(ZG_ARGS); break; } } while (0)
        switch (vsel) {
            case 0: ZG_VEC(64, 4); break;
            case 1: ZG_VEC(32, 4); break;
            case 2: ZG_VEC(16, 4); break;
            case 3: ZG_VEC(64, 2); break;
            case 4: ZG_VEC(32, 2); break;
            case 5: ZG_VEC(16, 2); break;
            case 6: ZG_VEC(32, 1); break;
            default: ZG_VEC(16, 1); break;
        }
#undef ZG_VEC
    }
    const double flops = 8.0 * M * (double)Nn * K * batch;
    if (in_run) {                       // the run's end event is recorded by GemmRun's destructor
        if (op->gemm_run_pair >= 0) { op->gemm_run_flops += flops; op->gemm_run_launches += 1; }
    } else if (e0) {
        hipEventRecord(e1, st);
        op->ev_pending_gemm.push_back(std::make_pair((int)op->ev_used, flops));
        op->ev_pending_gemm_n.push_back(1);
        op->ev_used += 2;
    }
    return 0;
}

// Consecutive GEMM launches with no other kernel between them (the recursion of the block inversion issues them in runs of
// two and four) share one pair of timing events: the per-launch average stays exact, two thirds of the event traffic go away.
struct GemmRun {
    helm_op *op;
    explicit GemmRun(helm_op *o) : op(o) { if (op && op->gemm_run_depth++ == 0) { op->gemm_run_pair = -1; op->gemm_run_flops = 0; op->gemm_run_launches = 0; } }
    ~GemmRun() {
        if (!op || --op->gemm_run_depth != 0) return;
        if (op->gemm_run_pair >= 0 && op->gemm_run_launches > 0) {
            hipEventRecord(op->ev_pool[op->gemm_run_pair + 1], op->stream);
            op->ev_pending_gemm.push_back(std::make_pair(op->gemm_run_pair, op->gemm_run_flops));
            op->ev_pending_gemm_n.push_back(op->gemm_run_launches);
        }
        op->gemm_run_pair = -1;
    }
};

// in-place inverse of `batch` n x n blocks (row-major, leading dimension ld, batch stride `stride`); W: workspace with
// batch stride ws, at least n*n elements per matrix
void invert(helm_op *op, cplx *M, int ld, long long stride, int n, int batch, cplx *W, long long ws, int align = 1, int base = 0) {
    hipStream_t st = op ? op->stream : nullptr;
    // the coupled two-field system (align == 2) is far worse conditioned: it gets the wider pivoting window by default
    static const int gj_env = getenv("HELM_ND_GJ") ? atoi(getenv("HELM_ND_GJ")) : 0;
    const int gj_base = base ? base : (gj_env == 64 ? 64 : (gj_env == 32 ? 32 : (align == 2 ? 64 : 32)));
    if (n <= gj_base) {
        for (int b0 = 0; b0 < batch; b0 += 1 << 20) {
            const int nb = std::min(1 << 20, batch - b0);
            // (one wave per matrix -- 64 threads, free barriers, four times the matrices in flight -- was measured: 20 % slower factorisation)
            static const int gj_threads = getenv("HELM_ND_GJ_THREADS") ? atoi(getenv("HELM_ND_GJ_THREADS")) : 256;
            if (n <= 32 && gj_threads == 1024) hipLaunchKernelGGL((k_gj_inverse<32, 1024>), dim3(nb), dim3(1024), 0, st, M + b0 * stride, ld, stride, n);
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
NdPlanDev::~NdPlanDev() {
    if (d_nodes) hipFree(d_nodes);
    if (d_tab) hipFree(d_tab);
}

namespace {
std::mutex g_plan_mu;
std::vector<std::shared_ptr<NdPlanDev>> g_plans;     // most recently used last, at most 4 kept alive by the cache
}

int nd_get_plan(helm_op *op, int leaf, int dof, std::shared_ptr<NdPlanDev> *out) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    for (size_t i = 0; i < g_plans.size(); ++i) {
        const NdPlanDev &c = *g_plans[i];
        if (c.device == op->device && c.plan.nz == op->nz && c.plan.nx == op->nx && c.plan.leaf == std::max(2, leaf) && c.plan.dof == dof) {
            std::shared_ptr<NdPlanDev> hit = g_plans[i];
            g_plans.erase(g_plans.begin() + i); g_plans.push_back(hit);
            *out = hit;
            return HELM_OK;
        }
    }
    std::shared_ptr<NdPlanDev> pd(new NdPlanDev());
    pd->device = op->device;
    nd_build_plan(pd->plan, op->nz, op->nx, leaf, dof);
    const NdPlan &P = pd->plan;
    if (2 * P.vregion >= (1LL << 31) || P.total_rows >= (1LL << 31)) HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "direct solver: grid too large for 32-bit row indices");
    HIP_TRY(op, hipMalloc((void **)&pd->d_nodes, P.nodes.size() * sizeof(NdDev)));
    HIP_TRY(op, hipMalloc((void **)&pd->d_tab, (size_t)P.total_rows * sizeof(int4)));
    HIP_TRY(op, hipMemcpyAsync(pd->d_nodes, P.nodes.data(), P.nodes.size() * sizeof(NdDev), hipMemcpyHostToDevice, op->stream));
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        const NdGroup &g = P.groups[gi];
        const int nmax = g.smax + g.mmax;
        for (int j0 = 0; j0 < g.cnt; j0 += 65535) {
            const int nb = std::min(65535, g.cnt - j0);
            hipLaunchKernelGGL(k_nd_build_tab, dim3((nmax + 255) / 256, nb), dim3(256), 0, op->stream, pd->d_nodes, g.first + j0, pd->d_tab, P.nz, P.nx);
        }
    }
    HIP_TRY(op, hipStreamSynchronize(op->stream));      // the host-side node array may go away after this
    g_plans.push_back(pd);
    if (g_plans.size() > 4) g_plans.erase(g_plans.begin());
    *out = pd;
    return HELM_OK;
}

// ---- factorisation ---------------------------------------------------------------------------------------------
void nd_free(NdFactor *f) {
    if (!f) return;
    if (f->d_fac) helm_pool_free(f->pd ? f->pd->device : 0, f->d_fac, (size_t)f->pd->plan.fac_elems * sizeof(cplx));
    delete f;
}

namespace {

// factorisation of one group (tree level x kind) on op->stream
int factor_group(helm_op *op, NdFactor *f, size_t gi, cplx *arenaF, cplx *work, const cplx *planes) {
    const NdPlan &P = f->pd->plan;
    const NdDev *d_nodes = f->pd->d_nodes;
    hipStream_t st = op->stream;
    const cplx one = cmake(1, 0), mone = cmake(-1, 0), zero = cmake(0, 0);
    const NdGroup &g = P.groups[gi];
    const int nmax = g.smax + g.mmax;
    const long long fs = (long long)g.mmax * nmax;              // scratch per front: [F21 | F22]
    const long long s11 = (long long)g.smax * g.smax, s12 = (long long)g.smax * g.mmax;
    cplx *F = arenaF + g.foff;
    cplx *Finv = f->d_fac + g.finv, *G21 = f->d_fac + g.g21, *F12 = f->d_fac + g.f12;
    if (fs > 0) HIP_TRY(op, hipMemsetAsync(F, 0, (size_t)g.cnt * fs * sizeof(cplx), st));
    HIP_TRY(op, hipMemsetAsync(Finv, 0, (size_t)g.cnt * s11 * sizeof(cplx), st));
    if (s12 > 0) HIP_TRY(op, hipMemsetAsync(F12, 0, (size_t)g.cnt * s12 * sizeof(cplx), st));
    for (int j0 = 0; j0 < g.cnt; j0 += 65535) {
        const int nb = std::min(65535, g.cnt - j0);
        hipLaunchKernelGGL(k_nd_assemble, dim3((nmax + 255) / 256, nb), dim3(256), 0, st, d_nodes, g.first + j0, arenaF, f->d_fac, planes, op->nz, op->nx);
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
                hipLaunchKernelGGL(k_nd_extend_add, dim3(gx, nb), dim3(256), shm, st, d_nodes, g.first + j0, slot, arenaF, f->d_fac, op->nz, op->nx, chunk);
            }
    }
    static const int gj_leaf = getenv("HELM_ND_GJ_LEAF") ? atoi(getenv("HELM_ND_GJ_LEAF")) : 0;
    static const int gj_upper = getenv("HELM_ND_GJ_UPPER") ? atoi(getenv("HELM_ND_GJ_UPPER")) : 0;
    // the few huge fronts at the top of the tree are one long chain of single-matrix launches: the wider base block halves it
    static const int gj_top = getenv("HELM_ND_GJ_TOP") ? atoi(getenv("HELM_ND_GJ_TOP")) : 0;
    const int base = g.leaf ? gj_leaf : (g.cnt <= gj_top ? 64 : gj_upper);
    invert(op, Finv, g.smax, s11, g.smax, g.cnt, work, s11, P.dof, base);      // F11 -> F11^-1 where it stays
    if (g.mmax > 0) {
        // G21 = F21 F11^-1 ; F22 -= G21 F12
        gemm(op, g.mmax, g.smax, g.smax, one, F, nmax, fs, Finv, g.smax, s11, zero, G21, g.smax, (long long)g.mmax * g.smax, g.cnt);
        gemm(op, g.mmax, g.mmax, g.smax, mone, G21, g.smax, (long long)g.mmax * g.smax, F12, g.mmax, s12, one, F + g.smax, nmax, fs, g.cnt);
    }
    f->flops += (double)g.cnt * 8.0 * (2.0 * g.smax * g.smax * g.smax + (double)g.smax * g.smax * g.mmax + (double)g.smax * g.mmax * g.mmax);
    return HELM_OK;
}

struct SolveCtx {
    const int4 *tab; cplx *Xt, *arenaV; int nrhs; dim3 rb; int use_idx;
    dim3 rgrid(long long rows) const { return dim3((unsigned)std::min<long long>((rows + rb.y - 1) / rb.y, 1 << 20)); }
};

SolveCtx solve_ctx(const NdFactor *f, cplx *ws, int nrhs) {
    static const int use_idx = getenv("HELM_ND_IDXGEMM") ? atoi(getenv("HELM_ND_IDXGEMM")) : 1;
    const NdPlan &P = f->pd->plan;
    SolveCtx c;
    c.tab = f->pd->d_tab; c.Xt = ws; c.arenaV = ws + (long long)P.dof * P.nz * P.nx * nrhs; c.nrhs = nrhs; c.use_idx = use_idx;
    int lx = 1;
    while (lx < nrhs && lx < 256) lx <<= 1;
    c.rb = dim3(lx, 256 / lx);
    return c;
}

// forward elimination of one group on op->stream
void forward_group(helm_op *op, NdFactor *f, size_t gi, const SolveCtx &c) {
    const NdPlan &P = f->pd->plan;
    const cplx one = cmake(1, 0), mone = cmake(-1, 0), zero = cmake(0, 0);
    const NdGroup &g = P.groups[gi];
    const int nmax = g.smax + g.mmax, nrhs = c.nrhs;
    const long long rows = (long long)g.cnt * nmax;
    cplx *V = c.arenaV + g.voff * nrhs;
    if (c.use_idx && g.leaf && g.mmax > 0 && g.smax <= GB_KIDX) {
        // leaves have no children: the outgoing ring part is -G21 x_S with x_S read straight from Xt
        GemmRows R; R.tabB = c.tab + g.roff; R.offB = 0; R.tab_stride = nmax; R.Bx = c.Xt; R.ldx = nrhs;
        gemm(op, g.mmax, nrhs, g.smax, mone, f->d_fac + g.g21, g.smax, (long long)g.mmax * g.smax, nullptr, 0, 0, zero,
             V + (long long)g.smax * nrhs, nrhs, (long long)nmax * nrhs, g.cnt, &R);
        return;
    }
    hipLaunchKernelGGL(k_nd_fwd_rows, c.rgrid(rows), c.rb, 0, op->stream, c.tab + g.roff, V, c.arenaV, c.Xt, rows, nrhs, g.leaf ? 0 : 1);
    if (g.mmax > 0)     // outgoing ring part: V_B -= G21 V_S
        gemm(op, g.mmax, nrhs, g.smax, mone, f->d_fac + g.g21, g.smax, (long long)g.mmax * g.smax, V, nrhs, (long long)nmax * nrhs, one,
             V + (long long)g.smax * nrhs, nrhs, (long long)nmax * nrhs, g.cnt);
}

// back substitution of one group on op->stream
void backward_group(helm_op *op, NdFactor *f, size_t gk, const SolveCtx &c) {
    const NdPlan &P = f->pd->plan;
    const cplx one = cmake(1, 0), mone = cmake(-1, 0), zero = cmake(0, 0);
    const NdGroup &g = P.groups[gk];
    const int nmax = g.smax + g.mmax, nrhs = c.nrhs;
    const long long rows = (long long)g.cnt * nmax;
    cplx *V = c.arenaV + g.voff * nrhs;
    // the other region is free in this pass: separator results go there
    const long long xs_off = g.voff + ((g.level & 1) ? -P.vregion : P.vregion);
    cplx *XS = c.arenaV + xs_off * nrhs;
    if (c.use_idx && g.mmax > 0 && g.mmax <= GB_KIDX) {
        // lower tree levels (almost all rows): T = y_S - F12 x_B and x_S = F11^-1 T with every Xt row addressed through
        // the row table -- no gather / store pass
        GemmRows R1; R1.tabB = c.tab + g.roff; R1.offB = g.smax; R1.tabCi = c.tab + g.roff; R1.offCi = 0; R1.tab_stride = nmax;
        R1.Bx = c.Xt; R1.Cix = c.Xt; R1.ldx = nrhs;
        gemm(op, g.smax, nrhs, g.mmax, mone, f->d_fac + g.f12, g.mmax, (long long)g.smax * g.mmax, nullptr, 0, 0, one,
             V, nrhs, (long long)g.smax * nrhs, g.cnt, &R1);
        GemmRows R2; R2.tabCo = c.tab + g.roff; R2.offCo = 0; R2.tab_stride = nmax; R2.Cox = c.Xt; R2.ldx = nrhs;
        gemm(op, g.smax, nrhs, g.smax, one, f->d_fac + g.finv, g.smax, (long long)g.smax * g.smax, V, nrhs, (long long)g.smax * nrhs, zero,
             nullptr, 0, 0, g.cnt, &R2);
        return;
    }
    hipLaunchKernelGGL(k_nd_bwd_gather, c.rgrid(rows), c.rb, 0, op->stream, c.tab + g.roff, V, c.Xt, rows, nrhs);
    if (g.mmax > 0)
        gemm(op, g.smax, nrhs, g.mmax, mone, f->d_fac + g.f12, g.mmax, (long long)g.smax * g.mmax, V + (long long)g.smax * nrhs, nrhs, (long long)nmax * nrhs,
             one, V, nrhs, (long long)nmax * nrhs, g.cnt);
    gemm(op, g.smax, nrhs, g.smax, one, f->d_fac + g.finv, g.smax, (long long)g.smax * g.smax, V, nrhs, (long long)nmax * nrhs, zero,
         XS, nrhs, (long long)g.smax * nrhs, g.cnt);
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
    std::vector<hipEvent_t> ev(ng + 2);
    for (auto &e : ev) HIP_TRY(op, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1;
    HIP_TRY(op, hipEventCreate(&t0)); HIP_TRY(op, hipEventCreate(&t1));
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
    for (auto &x : ev) hipEventDestroy(x);
    hipEventDestroy(t0); hipEventDestroy(t1);
    if (rc) return rc;
    if (e != hipSuccess) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: factor + solve failed: %s", hipGetErrorString(e));
    return check_kernels(op, "factor + solve kernels");
}

// dense helpers for other translation units (3-D multigrid: coarsest-level inverse and its application)
int nd_dense_gemm(helm_op *op, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, const cplx *B, int ldb, cplx beta, cplx *C, int ldc) {
    gemm(op, M, Nn, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1);
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
    gemm((helm_op *)nullptr, M, Nn, K, cmake(alpha[0], alpha[1]), dA, K, (long long)M * K, dB, Nn, (long long)K * Nn, cmake(beta[0], beta[1]), dC, Nn, (long long)M * Nn, batch);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(C, dC, nc * 16, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dC);
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
    g_gemm_variant = variant;
    for (int w = 0; w < 2; ++w) gemm((helm_op *)nullptr, M, Nn, K, cmake(1, 0), dA, K, (long long)M * K, dB, Nn, (long long)K * Nn, cmake(0, 0), dC, Nn, (long long)M * Nn, batch);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) gemm((helm_op *)nullptr, M, Nn, K, cmake(1, 0), dA, K, (long long)M * K, dB, Nn, (long long)K * Nn, cmake(0, 0), dC, Nn, (long long)M * Nn, batch);
    hipEventRecord(e1, nullptr);
    hipError_t e = hipEventSynchronize(e1);
    g_gemm_variant = -1;
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = ms / std::max(1, reps);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(dA); hipFree(dB); hipFree(dC);
    return e == hipSuccess ? HELM_OK : HELM_ERR_DEVICE;
}
