// Direct solver, leaf level of the factorisation in one kernel.
#include "nd_internal.hpp"

namespace {

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
#define LEAF_NB 19           // band row: columns i - 9 .. i + 9
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
__global__ __launch_bounds__(128, 4) void k_leaf_factor(const NdDev *nodes, int first, cplx *arenaF, cplx *fac, cplx *g21base, const cplx *planes, int nz, int nx, int *flags, int dbg,
                                                        int nf, int kf) {
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
    // (nf, kf: frequency kf of nf factored together -- this front's slots lie at nf * offset + kf * slot, direct.hpp; 1, 0: on its own)
    cplx *Fcol = fac + (long long)nf * n.finv_off + (long long)kf * smax * nmax + gc;
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
        cplx *G21 = g21base + ((long long)blockIdx.x * nf + kf) * mmax * smax;
        cplx *S = arenaF + (long long)nf * n.foff + (long long)kf * mmax * nmax + smax;
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
__global__ __launch_bounds__(256) void k_leaf_factor_pivoted(const NdDev *nodes, int first, cplx *arenaF, cplx *fac, cplx *g21base, const cplx *planes, int nz, int nx, const int *flags,
                                                             int nf, int kf) {
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
    cplx *Frow = fac + (long long)nf * n.finv_off + (long long)kf * smax * nmax;
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
    cplx *G21 = g21base + ((long long)blockIdx.x * nf + kf) * mmax * smax;
    cplx *S = arenaF + (long long)nf * n.foff + (long long)kf * mmax * nmax + smax;
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

}  // namespace

// (nf, kf: frequency kf of a set of nf -- fac / arenaF / g21b are the SET's bases, the kernels place this frequency's slots among the interleaved ones)
void launch_leaf_factor(hipStream_t st, int smax, int nb, const NdDev *d_nodes, int first, cplx *arenaF, cplx *fac, cplx *g21b, const cplx *planes, int nz, int nx, int *flags, int dbg,
                        int nf, int kf) {
    if (smax <= 49) HELM_LAUNCH(k_leaf_factor<49>, dim3(nb), dim3(128), 0, st, d_nodes, first, arenaF, fac, g21b, planes, nz, nx, flags, dbg, nf, kf);
    else HELM_LAUNCH(k_leaf_factor<64>, dim3(nb), dim3(128), 0, st, d_nodes, first, arenaF, fac, g21b, planes, nz, nx, flags, dbg, nf, kf);
    HELM_LAUNCH(k_leaf_factor_pivoted, dim3(nb), dim3(256), 0, st, d_nodes, first, arenaF, fac, g21b, planes, nz, nx, (const int *)flags, nf, kf);
}
