// Device kernels of the matrix-free Krylov path:
//   * k_stencil   -- batched 9-point complex128 stencil apply, LDS-staged tiles, coefficients
//                    register-blocked over the right-hand-side loop, fused dot-product epilogues
//                    (wave64 shuffle reduction -> LDS -> one partial per workgroup)
//   * k_bicg_* / k_cg_* -- fused vector updates of BiCGSTAB / CGNR with their dot products
//   * k_fin       -- single-workgroup deterministic reduction of the partials + scalar recurrences
//
// The arithmetic replaces the sparse-LU solve behind BaseDiscretization.__mul__
// (zephyr/backend/discretization.py:78-106).  This is HBM-bound work (<= 2.25 flop/B, no MFMA):
// the design rules are 16-B-per-lane coalesced accesses, one pass per vector per kernel, and
// an XCD-aware tile order so that halo rows are served by the XCD's own L2.
#include "helm_internal.hpp"
#include <algorithm>

namespace {

// ------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Reduce NQ doubles per thread over a 256-thread block; result valid in thread 0.
template <int NQ>
__device__ inline void block_sum(double (&v)[NQ], double *smem /* >= 4*NQ doubles */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = wave_sum(v[q]);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) smem[wave * NQ + q] = v[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = (smem[q] + smem[NQ + q]) + (smem[2 * NQ + q] + smem[3 * NQ + q]);
    }
    __syncthreads();
}

__device__ inline bool rhs_active(const RhsScal *scal, int b) { return scal == nullptr || scal[b].status == ST_ACTIVE; }

// Round-robin dispatch puts workgroup b on XCD b % 8; give every XCD one contiguous run of
// tiles (a band of grid rows) so z-/x-neighbouring tiles share an L2.  Bijective for any count.
__device__ inline int xcd_swizzle(int bid, int nblk) {
    const int q = nblk / HELM_NXCD, rem = nblk % HELM_NXCD;
    const int x = bid % HELM_NXCD, k = bid / HELM_NXCD;
    const int start = x * q + (x < rem ? x : rem);
    return start + k;
}

// ------------------------------------------------------------------------------------------
// stencil apply
// ------------------------------------------------------------------------------------------
template <class V>
struct StencilParamsT {
    const V *planes;
    const V *X;
    V *Y;
    const V *W;
    long long ld, N;
    int nz, nx, nrhs, ntx, ntz, nblk;
    const RhsScal *scal;
    double *part;
    const V *dinv;
    double omega_j;
    const int *tiles;
    int planes_tiled;     // 1: planes are stored tile-blocked, [tile][k][row in tile][64] (one contiguous 9*TZ KB chunk per tile)
    // fused multigrid stages (XMODE template parameter)
    int acc, part_stride, part_off;
    V *U;                 // XMODE 1: the smoothed iterate u = omega_j dinv (.) W is also written here
    const V *E;           // XMODE 2: coarse-grid correction, [nrhs][nzc*nxc]; the input is X + P E (bilinear)
    int nzc, nxc;
};
typedef StencilParamsT<cplx> StencilParams;

constexpr int TX = 64;

// XMODE selects how the input tile is produced:
//   0  X is read                                                       (everything else)
//   1  X = omega_j * dinv (.) W, also stored to U        [multigrid: first Jacobi sweep fused into the residual]
//   2  X = X + P E (bilinear prolongation of E)          [multigrid: coarse correction fused into the post-smoothing sweep]
//   3  X is read conjugated                               [direct path: residual of a wavefield already stored as conj(x)]
template <class V, int P, bool SCALED, bool ADJ, int EPI, int XMODE = 0>
__global__ __launch_bounds__(256) void k_stencil_t(StencilParamsT<V> q) {
    constexpr int TZ = 4 * P;
    constexpr int LW = TX + 2;             // tile row length in elements
    constexpr int LR = TZ + 2;             // tile rows
    constexpr int NLOAD = (LR + 3) / 4;    // main-column row loads per wave
    __shared__ __attribute__((aligned(16))) V tile[2][LR * LW];
    __shared__ double red[16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = xcd_swizzle(blockIdx.x, q.nblk);
    if (q.tiles) t = q.tiles[t];
    const int tz = t / q.ntx, tx = t - tz * q.ntx;
    const int z0 = tz * TZ, x0 = tx * TX;
    const int nz = q.nz, nx = q.nx;
    const long long N = q.N;
    const int col = x0 + lane;
    const bool colok = col < nx;

    // ---- coefficients for this thread's P points, kept in registers over the RHS loop ----
    V cf[P][9];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int row = z0 + wave * P + j;
        const bool ok = colok && row < nz;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (SCALED && k == 4) { cf[j][k] = vone<V>(); continue; }
            V v = vzero<V>();
            if (!ADJ) {
                if (q.planes_tiled) v = q.planes[((long long)t * 9 + k) * (TZ * TX) + (wave * P + j) * TX + lane];
                else if (ok) v = q.planes[(long long)k * N + (long long)row * nx + col];
            } else {
                // (A^H x)[i] = sum over neighbours n of conj(A[n,i]) x[n]; the entry A[n,i] sits in
                // plane slot(-dz,-dx) at point n = i + (dz,dx)
                const int dz = k / 3 - 1, dx = k % 3 - 1;
                const int rn = row + dz, cn = col + dx;
                if (ok && rn >= 0 && rn < nz && cn >= 0 && cn < nx)
                    v = cconj(q.planes[(long long)(8 - k) * N + (long long)rn * nx + cn]);
            }
            cf[j][k] = v;
        }
    }

    // ---- RHS loop with register prefetch + double-buffered LDS tile ----
    V pre[NLOAD];
    V prehalo = vzero<V>();
    // value of the (virtual) input vector at grid point (grow, gcol) of right-hand side b
    auto input_at = [&](int b, int grow, int gcol) -> V {
        const long long idx = (long long)grow * nx + gcol;
        if (XMODE == 1) {
            return cmul(cscale(q.dinv[idx], q.omega_j), q.W[(long long)b * q.ld + idx]);
        } else if (XMODE == 2) {
            V v = q.X[(long long)b * q.ld + idx];
            const V *e = q.E + (long long)b * q.nzc * q.nxc;
            const int I = grow >> 1, J = gcol >> 1;
            const bool oi = grow & 1, oj = gcol & 1;
            const double wi0 = oi ? 0.5 : 1.0, wj0 = oj ? 0.5 : 1.0;
            V a = cscale(e[(long long)I * q.nxc + J], wi0 * wj0);
            if (oj && J + 1 < q.nxc) { const V c1 = e[(long long)I * q.nxc + J + 1]; a.x += wi0 * 0.5 * c1.x; a.y += wi0 * 0.5 * c1.y; }
            if (oi && I + 1 < q.nzc) {
                const V c2 = e[(long long)(I + 1) * q.nxc + J]; a.x += 0.5 * wj0 * c2.x; a.y += 0.5 * wj0 * c2.y;
                if (oj && J + 1 < q.nxc) { const V c3 = e[(long long)(I + 1) * q.nxc + J + 1]; a.x += 0.25 * c3.x; a.y += 0.25 * c3.y; }
            }
            return cadd(v, a);
        } else if (XMODE == 3) {
            return cconj(q.X[(long long)b * q.ld + idx]);
        } else {
            return q.X[(long long)b * q.ld + idx];
        }
    };
    auto prefetch = [&](int b) {
#pragma unroll
        for (int l = 0; l < NLOAD; ++l) {
            const int r = wave + 4 * l;                 // tile row
            const int grow = z0 - 1 + r;
            V v = vzero<V>();
            if (r < LR && colok && grow >= 0 && grow < nz) v = input_at(b, grow, col);
            pre[l] = v;
        }
        prehalo = vzero<V>();
        if (tid < 2 * LR) {
            const int r = tid >> 1, side = tid & 1;
            const int grow = z0 - 1 + r, gcol = side ? x0 + TX : x0 - 1;
            if (grow >= 0 && grow < nz && gcol >= 0 && gcol < nx) prehalo = input_at(b, grow, gcol);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int l = 0; l < NLOAD; ++l) {
            const int r = wave + 4 * l;
            if (r < LR) tile[buf][r * LW + 1 + lane] = pre[l];
        }
        if (tid < 2 * LR) {
            const int r = tid >> 1, side = tid & 1;
            tile[buf][r * LW + (side ? TX + 1 : 0)] = prehalo;
        }
    };

    const int bstep = gridDim.y;
    int b = blockIdx.y;
    while (b < q.nrhs && !rhs_active(q.scal, b)) b += bstep;
    if (b < q.nrhs) prefetch(b);
    int buf = 0;
    while (b < q.nrhs) {
        stage(buf);
        int bn = b + bstep;
        while (bn < q.nrhs && !rhs_active(q.scal, bn)) bn += bstep;
        if (bn < q.nrhs) prefetch(bn);
        __syncthreads();

        V acc[P];
#pragma unroll
        for (int j = 0; j < P; ++j) acc[j] = vzero<V>();
        V xc_keep[P];
        const V *trow = &tile[buf][(wave * P) * LW + lane];
#pragma unroll
        for (int rr = 0; rr < P + 2; ++rr) {
            const V xl = trow[rr * LW + 0], xm = trow[rr * LW + 1], xr = trow[rr * LW + 2];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int dzi = rr - j;            // 0,1,2 <-> dz = -1,0,+1
                if (dzi < 0 || dzi > 2) continue;
                cfma(acc[j], cf[j][dzi * 3 + 0], xl);
                if (SCALED && dzi == 1) { acc[j].x += xm.x; acc[j].y += xm.y; }
                else cfma(acc[j], cf[j][dzi * 3 + 1], xm);
                cfma(acc[j], cf[j][dzi * 3 + 2], xr);
                if (dzi == 1) xc_keep[j] = xm;
            }
        }

        double dsum[4] = {0.0, 0.0, 0.0, 0.0};
        V *Yb = q.Y + (long long)b * q.ld;
        const V *Wb = (EPI == EPI_DOT_W || EPI == EPI_RESID || EPI == EPI_JACOBI || EPI == EPI_DOT_WY) ? q.W + (long long)b * q.ld : nullptr;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int row = z0 + wave * P + j;
            if (colok && row < nz) {
                const long long idx = (long long)row * nx + col;
                V y = acc[j];
                if (q.acc) y = cadd(y, Yb[idx]);
                if (EPI == EPI_RESID) {
                    const V w = Wb[idx];
                    y = csub(w, y);
                    dsum[0] += cabs2(y);
                    if (XMODE == 1) q.U[(long long)b * q.ld + idx] = xc_keep[j];
                } else if (EPI == EPI_DOT_W) {
                    const V w = Wb[idx];          // (w, y) = sum conj(w) y
                    dsum[0] += w.x * y.x + w.y * y.y;
                    dsum[1] += w.x * y.y - w.y * y.x;
                } else if (EPI == EPI_DOT_XY) {
                    const V x = xc_keep[j];       // (y, x) = sum conj(y) x ; (y, y)
                    dsum[0] += y.x * x.x + y.y * x.y;
                    dsum[1] += y.x * x.y - y.y * x.x;
                    dsum[2] += cabs2(y);
                } else if (EPI == EPI_DOT_YY) {
                    dsum[0] += cabs2(y);
                } else if (EPI == EPI_DOT_WY) {
                    const V w = Wb[idx];          // (y, w) = sum conj(y) w ; (y, y)
                    dsum[0] += y.x * w.x + y.y * w.y;
                    dsum[1] += y.x * w.y - y.y * w.x;
                    dsum[2] += cabs2(y);
                } else if (EPI == EPI_JACOBI) {
                    const V res = csub(Wb[idx], y);
                    const V d = q.dinv[idx];
                    y = xc_keep[j];
                    cfma(y, cscale(d, q.omega_j), res);
                }
                Yb[idx] = y;
            }
        }
        if (EPI != EPI_NONE && EPI != EPI_JACOBI && sizeof(V) == sizeof(cplx)) {   // single-precision (multigrid) launches need no partials
            block_sum<4>(dsum, red);
            if (tid == 0) {
                double *pp = q.part + ((long long)b * 4) * q.part_stride + q.part_off + blockIdx.x;
                pp[0] = dsum[0];
                pp[(long long)q.part_stride] = dsum[1];
                pp[2LL * q.part_stride] = dsum[2];
                pp[3LL * q.part_stride] = dsum[3];
            }
        }
        buf ^= 1;
        b = bn;
    }
}

// ------------------------------------------------------------------------------------------
// elementwise / vector kernels (grid.x over points with a grid-stride loop, grid.y = rhs)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scale_planes(const cplx *__restrict__ C, cplx *__restrict__ Cs,
                                                      cplx *__restrict__ dinv, long long N, int nblocks, double floor_frac,
                                                      int nplanes, int centre) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    for (int m = 0; m < nblocks; ++m) {
        const cplx *Cm = C + (long long)m * nplanes * N;
        cplx *Sm = Cs + (long long)m * nplanes * N;
        cplx d = Cm[(long long)centre * N + i];
        const bool zero = (d.x == 0.0 && d.y == 0.0);
        if (floor_frac > 0.0 && !zero) {
            // smoother safeguard on multigrid levels: where the diagonal nearly cancels (k h ~ 2: mass term
            // against the Laplacian) keep its phase but not less than floor_frac of the row's absolute sum
            double rows = 0.0;
            for (int k = 0; k < nplanes; ++k) rows += sqrt(cabs2(Cm[(long long)k * N + i]));
            const double ad = sqrt(cabs2(d));
            if (ad < floor_frac * rows) d = cscale(d, floor_frac * rows / ad);
        }
        const cplx di = zero ? cmake(0.0, 0.0) : crecip(d);
        dinv[(long long)m * N + i] = di;
        for (int k = 0; k < nplanes; ++k) Sm[(long long)k * N + i] = (k == centre) ? cmake(1.0, 0.0) : cmul(Cm[(long long)k * N + i], di);
    }
}

// out = scale (.) (premul * rhs - sub)   [scale = dinv or null]
__global__ __launch_bounds__(256) void k_prep_rhs(const cplx *__restrict__ rhs, long long rhs_ld, long long row_off,
                                                  cplx premul, const cplx *__restrict__ sub,
                                                  const cplx *__restrict__ scale, cplx *__restrict__ out, long long N) {
    const int b = blockIdx.y;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        cplx v = cmul(premul, rhs[(long long)b * rhs_ld + row_off + i]);
        if (sub) v = csub(v, sub[(long long)b * N + i]);
        if (scale) v = cmul(scale[i], v);
        out[(long long)b * N + i] = v;
    }
}

// k_prep_rhs with the partial (out, out) of k_norm2 folded in (direct path: one pass over the right-hand sides less)
__global__ __launch_bounds__(256) void k_prep_rhs_norm(const cplx *__restrict__ rhs, long long rhs_ld, long long row_off, cplx premul,
                                                       const cplx *__restrict__ sub, cplx *__restrict__ out, long long N,
                                                       double *__restrict__ part, int nblk) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    double s[1] = {0.0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        cplx v = cmul(premul, rhs[(long long)b * rhs_ld + row_off + i]);
        if (sub) v = csub(v, sub[(long long)b * N + i]);
        out[(long long)b * N + i] = v;
        s[0] += cabs2(v);
    }
    block_sum<1>(s, red);
    if (threadIdx.x == 0) part[((long long)b * 4) * nblk + blockIdx.x] = s[0];
}

// BiCGSTAB / CGNR start: x = 0, r = r0 = bbar, p = v = 0, partial (r, r)
__global__ __launch_bounds__(256) void k_krylov_init(const cplx *__restrict__ bbar, VecPtrs w, long long N,
                                                     double *__restrict__ part, int nblk) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    double s[1] = {0.0};
    const cplx zero = cmake(0.0, 0.0);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const long long g = (long long)b * N + i;
        const cplx v = bbar[g];
        w.x[g] = zero; w.r[g] = v; w.r0[g] = v; w.p[g] = zero; w.v[g] = zero;
        s[0] += cabs2(v);
    }
    block_sum<1>(s, red);
    if (threadIdx.x == 0) part[((long long)b * 4) * nblk + blockIdx.x] = s[0];
}

// p = r + beta (p - omega v)
__global__ __launch_bounds__(256) void k_bicg_p(VecPtrs w, long long N, const RhsScal *__restrict__ scal) {
    const int b = blockIdx.y;
    if (scal[b].status != ST_ACTIVE) return;
    const cplx beta = cmake(scal[b].beta_re, scal[b].beta_im), omega = cmake(scal[b].omega_re, scal[b].omega_im);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const long long g = (long long)b * N + i;
        const cplx r = w.r[g], p = w.p[g], v = w.v[g];
        cplx tmp = csub(p, cmul(omega, v));
        w.p[g] = cadd(r, cmul(beta, tmp));
    }
}

// s = r - alpha v
__global__ __launch_bounds__(256) void k_bicg_s(VecPtrs w, long long N, const RhsScal *__restrict__ scal) {
    const int b = blockIdx.y;
    if (scal[b].status != ST_ACTIVE) return;
    const cplx alpha = cmake(scal[b].alpha_re, scal[b].alpha_im);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const long long g = (long long)b * N + i;
        w.s[g] = csub(w.r[g], cmul(alpha, w.v[g]));
    }
}

// x += alpha p + omega s ; r = s - omega t ; partials (r0, r), (r, r)
__global__ __launch_bounds__(256) void k_bicg_xr(VecPtrs w, const cplx *__restrict__ xp, const cplx *__restrict__ xs, long long N,
                                                 const RhsScal *__restrict__ scal, double *__restrict__ part, int nblk) {
    __shared__ double red[12];
    const int b = blockIdx.y;
    if (scal[b].status != ST_ACTIVE) return;
    const cplx alpha = cmake(scal[b].alpha_re, scal[b].alpha_im), omega = cmake(scal[b].omega_re, scal[b].omega_im);
    double s[3] = {0.0, 0.0, 0.0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const long long g = (long long)b * N + i;
        const cplx sv = w.s[g], tv = w.t[g], pv = xp[g];
        const cplx sx = (xs == w.s) ? sv : xs[g];
        cplx xv = w.x[g];
        cfma(xv, alpha, pv);
        cfma(xv, omega, sx);
        w.x[g] = xv;
        const cplx rv = csub(sv, cmul(omega, tv));
        w.r[g] = rv;
        const cplx r0 = w.r0[g];
        s[0] += r0.x * rv.x + r0.y * rv.y;
        s[1] += r0.x * rv.y - r0.y * rv.x;
        s[2] += cabs2(rv);
    }
    block_sum<3>(s, red);
    if (threadIdx.x == 0) {
        double *pp = part + ((long long)b * 4) * nblk + blockIdx.x;
        pp[0] = s[0]; pp[(long long)nblk] = s[1]; pp[2LL * nblk] = s[2];
    }
}

// r0 = r, p = v = 0 for right-hand sides being (re)started (status == ST_ACTIVE after FIN_RESTART
// is decided by `mask[b]`)
__global__ __launch_bounds__(256) void k_restart_copy(VecPtrs w, long long N, const int *__restrict__ mask) {
    const int b = blockIdx.y;
    if (!mask[b]) return;
    const cplx zero = cmake(0.0, 0.0);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const long long g = (long long)b * N + i;
        w.r0[g] = w.r[g]; w.p[g] = zero; w.v[g] = zero;
    }
}

// CGNR: x += alpha p ; r -= alpha w(v) ; partial (r, r)
__global__ __launch_bounds__(256) void k_cg_xr(VecPtrs w, long long N, const RhsScal *__restrict__ scal,
                                               double *__restrict__ part, int nblk) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    if (scal[b].status != ST_ACTIVE) return;
    const double alpha = scal[b].alpha_re;
    double s[1] = {0.0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const long long g = (long long)b * N + i;
        const cplx pv = w.p[g], wv = w.v[g];
        cplx xv = w.x[g], rv = w.r[g];
        xv.x += alpha * pv.x; xv.y += alpha * pv.y;
        rv.x -= alpha * wv.x; rv.y -= alpha * wv.y;
        w.x[g] = xv; w.r[g] = rv;
        s[0] += cabs2(rv);
    }
    block_sum<1>(s, red);
    if (threadIdx.x == 0) part[((long long)b * 4) * nblk + blockIdx.x] = s[0];
}

// CGNR: p = z(s) + beta p
__global__ __launch_bounds__(256) void k_cg_p(VecPtrs w, long long N, const RhsScal *__restrict__ scal, int first) {
    const int b = blockIdx.y;
    if (scal[b].status != ST_ACTIVE) return;
    const double beta = first ? 0.0 : scal[b].beta_re;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const long long g = (long long)b * N + i;
        const cplx z = w.s[g];
        cplx p = first ? cmake(0.0, 0.0) : w.p[g];
        w.p[g] = cmake(z.x + beta * p.x, z.y + beta * p.y);
    }
}

// partial (a, a)
__global__ __launch_bounds__(256) void k_norm2(const cplx *__restrict__ a, long long N, double *__restrict__ part, int nblk) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    double s[1] = {0.0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
        s[0] += cabs2(a[(long long)b * N + i]);
    block_sum<1>(s, red);
    if (threadIdx.x == 0) part[((long long)b * 4) * nblk + blockIdx.x] = s[0];
}

// out = sign * |in| as a complex number with zero imaginary part (attainable-accuracy estimate of the coupled system)
__global__ __launch_bounds__(256) void k_abs_cplx(const cplx *__restrict__ in, cplx *__restrict__ out, long long n, double sign) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const cplx v = in[i];
        out[i] = cmake(sign * hypot(v.x, v.y), 0.0);
    }
}

// U = conj(x) into a (possibly strided / offset) output
__global__ __launch_bounds__(256) void k_finish(const cplx *__restrict__ x, cplx *__restrict__ U, long long u_ld,
                                                long long row_off, long long N) {
    const int b = blockIdx.y;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
        U[(long long)b * u_ld + row_off + i] = cconj(x[(long long)b * N + i]);
}

__global__ __launch_bounds__(256) void k_finish_ex(const cplx *__restrict__ x, long long x_ld, long long x_off, cplx *__restrict__ U, long long u_ld,
                                                   long long row_off, long long N) {
    const int b = blockIdx.y;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
        U[(long long)b * u_ld + row_off + i] = cconj(x[(long long)b * x_ld + x_off + i]);
}

// out[b*out_ld + out_off + i] = scale[i] * premul * rhs[b*rhs_ld + row_off + i]
__global__ __launch_bounds__(256) void k_prep_rhs_ex(const cplx *__restrict__ rhs, long long rhs_ld, long long row_off, cplx premul,
                                                     const cplx *__restrict__ scale, cplx *__restrict__ out, long long out_ld, long long out_off, long long N) {
    const int b = blockIdx.y;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        cplx v = cmul(premul, rhs[(long long)b * rhs_ld + row_off + i]);
        if (scale) v = cmul(scale[i], v);
        out[(long long)b * out_ld + out_off + i] = v;
    }
}

// Row equilibration of the coupled two-field Eurus system [[M1, M2], [M3, M4]]: every system row is divided by its
// 2-norm (the diagonal of M4 nearly vanishes where eps ~ delta, so Jacobi scaling is useless there).
__global__ __launch_bounds__(256) void k_rowscale_system(const cplx *__restrict__ C, cplx *__restrict__ S, double *__restrict__ rs, long long N) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        double n2 = 0.0;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int k = 0; k < 9; ++k) n2 += cabs2(C[((long long)(2 * row + blk) * 9 + k) * N + i]);
        const double inv = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
        rs[(long long)row * N + i] = inv;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const long long idx = ((long long)(2 * row + blk) * 9 + k) * N + i;
                S[idx] = cscale(C[idx], inv);
            }
    }
}

// out[b*out_ld + out_off + i] = rs[i] * premul * rhs[b*rhs_ld + row_off + i]   (real row scale)
__global__ __launch_bounds__(256) void k_prep_rhs_rs(const cplx *__restrict__ rhs, long long rhs_ld, long long row_off, cplx premul,
                                                     const double *__restrict__ rs, cplx *__restrict__ out, long long out_ld, long long out_off, long long N) {
    const int b = blockIdx.y;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
        out[(long long)b * out_ld + out_off + i] = cscale(cmul(premul, rhs[(long long)b * rhs_ld + row_off + i]), rs[i]);
}

__global__ __launch_bounds__(256) void k_zero(cplx *__restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = cmake(0.0, 0.0);
}

// G[i] += scaler[i] * sum_s UF[s][i] * UB[s][i]      (problem.py:152)
__global__ __launch_bounds__(256) void k_imaging(const cplx *__restrict__ uf, const cplx *__restrict__ ub, int nsrc,
                                                 const cplx *__restrict__ scaler, cplx *__restrict__ g, long long N) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        cplx acc = cmake(0.0, 0.0);
        for (int s = 0; s < nsrc; ++s) cfma(acc, uf[(long long)s * N + i], ub[(long long)s * N + i]);
        cplx gv = g[i];
        cfma(gv, scaler[i], acc);
        g[i] = gv;
    }
}

// ------------------------------------------------------------------------------------------
// finalize: one workgroup per right-hand side sums the per-workgroup partials in a fixed order
// (bitwise reproducible) and advances the scalar recurrences.
// ------------------------------------------------------------------------------------------
__device__ inline void fin_sum(const double *part, int b, int nblk, double (&out)[4], double *smem) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += part[((long long)b * 4 + q) * nblk + i];
    }
    block_sum<4>(v, smem);
    __shared__ double bc[4];
    if (threadIdx.x == 0) { bc[0] = v[0]; bc[1] = v[1]; bc[2] = v[2]; bc[3] = v[3]; }
    __syncthreads();
    out[0] = bc[0]; out[1] = bc[1]; out[2] = bc[2]; out[3] = bc[3];
}

struct FinParams {
    RhsScal *scal;
    const double *part;
    int nblk;
    int which;
    double rtol;
    const int *mask;    // FIN_RESTART: which RHS to touch
    double *aux;        // FIN_NORM: aux[b] = sum
};

__device__ inline bool finite2(cplx a) { return isfinite(a.x) && isfinite(a.y); }

__global__ __launch_bounds__(256) void k_fin(FinParams f) {
    __shared__ double smem[16];
    const int b = blockIdx.x;
    RhsScal *S = f.scal ? f.scal + b : nullptr;
    if (f.which == FIN_NORM || f.which == FIN_NORM2) {
        double v[4];
        fin_sum(f.part, b, f.nblk, v, smem);
        if (threadIdx.x == 0) { f.aux[b] = v[0]; if (f.which == FIN_NORM2) f.aux[gridDim.x + b] = v[1]; }
        return;
    }
    if (f.which == FIN_RESTART) {
        if (!f.mask[b]) {   // right-hand sides parked while others restart resume where they were
            if (threadIdx.x == 0 && S->status == ST_PARKED) S->status = ST_ACTIVE;
            return;
        }
    }
    else if (f.which != FIN_BICG_INIT && S->status != ST_ACTIVE) return;
    double v[4];
    fin_sum(f.part, b, f.nblk, v, smem);
    if (threadIdx.x != 0) return;

    switch (f.which) {
    case FIN_BICG_INIT: {           // v[0] = (b, b) of the scaled system
        S->bb = v[0]; S->rr = v[0];
        S->rho_re = v[0]; S->rho_im = 0.0;
        S->alpha_re = 1.0; S->alpha_im = 0.0; S->omega_re = 1.0; S->omega_im = 0.0;
        S->beta_re = 0.0; S->beta_im = 0.0;
        S->tol2 = f.rtol * f.rtol;
        S->iters = 0;
        S->pad0 = 0;                // restarts
        S->pad1 = 0;
        S->status = (v[0] == 0.0 || !isfinite(v[0])) ? ST_CONVERGED : ST_ACTIVE;
        break;
    }
    case FIN_RESTART: {             // v[0] = (r, r) of the recomputed true residual; r0 := r
        S->rr = v[0];
        S->rho_re = v[0]; S->rho_im = 0.0;
        S->alpha_re = 1.0; S->alpha_im = 0.0; S->omega_re = 1.0; S->omega_im = 0.0;
        S->beta_re = 0.0; S->beta_im = 0.0;
        S->pad0 += 1;
        S->status = (v[0] <= S->tol2 * S->bb) ? ST_CONVERGED : ST_ACTIVE;
        break;
    }
    case FIN_ALPHA: {               // sigma = (r0, v); alpha = rho / sigma
        const cplx sigma = cmake(v[0], v[1]);
        const cplx rho = cmake(S->rho_re, S->rho_im);
        const cplx alpha = cdiv(rho, sigma);
        if (cabs2(sigma) == 0.0 || !finite2(alpha)) { S->status = ST_BREAKDOWN; break; }
        S->alpha_re = alpha.x; S->alpha_im = alpha.y;
        break;
    }
    case FIN_OMEGA: {               // omega = (t, s) / (t, t)
        const double tt = v[2];
        cplx omega = cmake(0.0, 0.0);
        if (tt > 0.0) omega = cmake(v[0] / tt, v[1] / tt);
        if (!finite2(omega)) { S->status = ST_BREAKDOWN; break; }
        S->omega_re = omega.x; S->omega_im = omega.y;
        break;
    }
    case FIN_RHO: {                 // rho' = (r0, r); rr = (r, r); beta = (rho'/rho)(alpha/omega)
        const cplx rhon = cmake(v[0], v[1]);
        const double rr = v[2];
        S->iters += 1;
        S->rr = rr;
        if (!isfinite(rr) || !finite2(rhon)) { S->status = ST_BREAKDOWN; break; }
        if (rr <= S->tol2 * S->bb) { S->status = ST_CONVERGED; break; }
        const cplx rho = cmake(S->rho_re, S->rho_im);
        const cplx alpha = cmake(S->alpha_re, S->alpha_im), omega = cmake(S->omega_re, S->omega_im);
        if (cabs2(omega) == 0.0 || cabs2(rho) == 0.0 || cabs2(rhon) == 0.0) { S->status = ST_BREAKDOWN; break; }
        const cplx beta = cmul(cdiv(rhon, rho), cdiv(alpha, omega));
        if (!finite2(beta)) { S->status = ST_BREAKDOWN; break; }
        S->beta_re = beta.x; S->beta_im = beta.y;
        S->rho_re = rhon.x; S->rho_im = rhon.y;
        break;
    }
    case FIN_CG_INIT: {             // v[0] = (z, z)
        S->rho_re = v[0]; S->rho_im = 0.0;
        if (v[0] == 0.0) S->status = ST_CONVERGED;
        break;
    }
    case FIN_CG_ALPHA: {            // v[0] = (w, w); alpha = gamma / (w, w)
        if (v[0] == 0.0 || !isfinite(v[0])) { S->status = ST_BREAKDOWN; break; }
        S->alpha_re = S->rho_re / v[0]; S->alpha_im = 0.0;
        break;
    }
    case FIN_CG_RR: {               // v[0] = (r, r)
        S->iters += 1;
        S->rr = v[0];
        if (!isfinite(v[0])) { S->status = ST_BREAKDOWN; break; }
        if (v[0] <= S->tol2 * S->bb) S->status = ST_CONVERGED;
        break;
    }
    case FIN_CG_BETA: {             // v[0] = (z, z) new
        if (S->rho_re == 0.0) { S->status = ST_BREAKDOWN; break; }
        S->beta_re = v[0] / S->rho_re; S->beta_im = 0.0;
        S->rho_re = v[0];
        break;
    }
    default: break;
    }
}

inline int vec_blocks(long long N) {
    long long nb = (N + 255) / 256;
    if (nb > 1024) nb = 1024;
    return (int)nb;
}

}  // namespace

// ==========================================================================================
// host launchers
// ==========================================================================================
#ifndef STENCIL_P
#define STENCIL_P 1
#endif

int helm_stencil_tile_rows() { return 4 * STENCIL_P; }

int helm_apply_num_blocks(const helm_op *op) {
    if (op->ny > 0) return helm3d_apply_num_blocks(op);
    const int ntx = (op->nx + TX - 1) / TX, ntz = (op->nz + 4 * STENCIL_P - 1) / (4 * STENCIL_P);
    return ntx * ntz;
}
int helm_vec_num_blocks(const helm_op *op) { return vec_blocks(op->Nv > 0 ? op->Nv : op->N); }

template <int P, bool SCALED, bool ADJ>
static void launch_stencil_epi(hipStream_t st, dim3 grid, const StencilParams &q, int epi) {
    switch (epi) {
    case EPI_NONE: HELM_LAUNCH((k_stencil_t<cplx, P, SCALED, ADJ, EPI_NONE>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_W: HELM_LAUNCH((k_stencil_t<cplx, P, SCALED, ADJ, EPI_DOT_W>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_XY: HELM_LAUNCH((k_stencil_t<cplx, P, SCALED, ADJ, EPI_DOT_XY>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_YY: HELM_LAUNCH((k_stencil_t<cplx, P, SCALED, ADJ, EPI_DOT_YY>), grid, dim3(256), 0, st, q); break;
    case EPI_RESID: HELM_LAUNCH((k_stencil_t<cplx, P, SCALED, ADJ, EPI_RESID>), grid, dim3(256), 0, st, q); break;
    case EPI_JACOBI: HELM_LAUNCH((k_stencil_t<cplx, P, SCALED, ADJ, EPI_JACOBI>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_WY: HELM_LAUNCH((k_stencil_t<cplx, P, SCALED, ADJ, EPI_DOT_WY>), grid, dim3(256), 0, st, q); break;
    }
}

// single-precision launches (multigrid levels): unscaled, forward, EPI_NONE / EPI_RESID / EPI_JACOBI
static int launch_apply_f32(helm_op *op, const ApplyArgs &a) {
    StencilParamsT<cplxf> q;
    q.planes = (const cplxf *)a.planes; q.X = (const cplxf *)a.X; q.Y = (cplxf *)a.Y; q.W = (const cplxf *)a.W; q.ld = a.ld; q.N = op->N;
    q.nz = op->nz; q.nx = op->nx; q.nrhs = a.nrhs;
    q.ntx = (op->nx + TX - 1) / TX; q.ntz = (op->nz + 4 * STENCIL_P - 1) / (4 * STENCIL_P);
    q.nblk = a.tiles ? a.ntiles : q.ntx * q.ntz;
    q.scal = a.scal; q.part = a.part; q.dinv = (const cplxf *)a.dinv; q.omega_j = a.omega_j; q.tiles = a.tiles; q.planes_tiled = 0;
    q.U = (cplxf *)a.U; q.E = (const cplxf *)a.E; q.nzc = a.nzc; q.nxc = a.nxc; q.acc = 0; q.part_stride = q.nblk; q.part_off = 0;
    if (q.nblk < 1) return HELM_OK;
    int split = 1;
    if (q.nblk < 1024) { split = (1024 + q.nblk - 1) / q.nblk; if (split > a.nrhs) split = a.nrhs; if (split < 1) split = 1; }
    dim3 grid(q.nblk, split);
    if (a.xmode == 1 && a.epi == EPI_RESID) HELM_LAUNCH((k_stencil_t<cplxf, STENCIL_P, false, false, EPI_RESID, 1>), grid, dim3(256), 0, op->stream, q);
    else if (a.xmode == 2 && a.epi == EPI_JACOBI) HELM_LAUNCH((k_stencil_t<cplxf, STENCIL_P, false, false, EPI_JACOBI, 2>), grid, dim3(256), 0, op->stream, q);
    else if (a.xmode != 0) HELM_FAIL(op, HELM_ERR_ARG, "unsupported fused stencil mode");
    else switch (a.epi) {
    case EPI_NONE: HELM_LAUNCH((k_stencil_t<cplxf, STENCIL_P, false, false, EPI_NONE>), grid, dim3(256), 0, op->stream, q); break;
    case EPI_RESID: HELM_LAUNCH((k_stencil_t<cplxf, STENCIL_P, false, false, EPI_RESID>), grid, dim3(256), 0, op->stream, q); break;
    case EPI_JACOBI: HELM_LAUNCH((k_stencil_t<cplxf, STENCIL_P, false, false, EPI_JACOBI>), grid, dim3(256), 0, op->stream, q); break;
    default: HELM_FAIL(op, HELM_ERR_ARG, "unsupported single-precision stencil epilogue");
    }
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

int helm_launch_apply(helm_op *op, const ApplyArgs &a) {
    if (op->ny > 0) {           // 3-D operator: own kernel, same profiling bookkeeping
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (op->profiling && a.profile && op->ev_pool.size() < 8192) {
            if (op->ev_used + 2 > op->ev_pool.size())
                if (helm_events_grow(op, 64)) HELM_FAIL(op, HELM_ERR_DEVICE, "hipEventCreate failed");
            e0 = op->ev_pool[op->ev_used]; e1 = op->ev_pool[op->ev_used + 1];
        }
        int rc = helm3d_launch_apply(op, a, e0, e1);
        if (rc) return rc;
        if (e0) {
            const int nact = (a.scal && op->active_hint >= 0 && op->active_hint < a.nrhs) ? op->active_hint : a.nrhs;
            const bool operand = (a.epi == EPI_DOT_W || a.epi == EPI_RESID);
            op->ev_pending.push_back(std::make_pair((int)op->ev_used, (double)op->N * (32.0 * nact + 432.0 + (operand ? 16.0 * nact : 0.0))));
            op->ev_used += 2;
        }
        return HELM_OK;
    }
    if (a.f32) return launch_apply_f32(op, a);
    StencilParams q;
    q.planes = a.planes; q.X = a.X; q.Y = a.Y; q.W = a.W; q.ld = a.ld; q.N = op->N;
    q.nz = op->nz; q.nx = op->nx; q.nrhs = a.nrhs;
    q.ntx = (op->nx + TX - 1) / TX; q.ntz = (op->nz + 4 * STENCIL_P - 1) / (4 * STENCIL_P);
    q.nblk = a.tiles ? a.ntiles : q.ntx * q.ntz;
    q.scal = a.scal; q.part = a.part; q.dinv = a.dinv; q.omega_j = a.omega_j; q.tiles = a.tiles; q.planes_tiled = a.planes_tiled;
    q.U = a.U; q.E = a.E; q.nzc = a.nzc; q.nxc = a.nxc;
    q.acc = a.acc; q.part_stride = a.part_stride > 0 ? a.part_stride : q.nblk; q.part_off = a.part_off;
    if (q.nblk < 1) return HELM_OK;
    int split = 1;
    if (q.nblk < 1024) { split = (1024 + q.nblk - 1) / q.nblk; if (split > a.nrhs) split = a.nrhs; if (split < 1) split = 1; }
    dim3 grid(q.nblk, split);

    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = op->profiling && a.profile;
    if (prof) {
        if (op->ev_used + 2 > op->ev_pool.size() && op->ev_pool.size() >= 8192) { e0 = nullptr; }
        else if (op->ev_used + 2 > op->ev_pool.size()) {
            if (helm_events_grow(op, 64)) HELM_FAIL(op, HELM_ERR_DEVICE, "hipEventCreate failed");
        }
        if (op->ev_used + 2 <= op->ev_pool.size()) {
            e0 = op->ev_pool[op->ev_used]; e1 = op->ev_pool[op->ev_used + 1];
            hipEventRecord(e0, op->stream);
        } else e0 = nullptr;
    }
    if (a.xmode == 1 && a.epi == EPI_RESID && !a.scaled && !a.adjoint) {
        HELM_LAUNCH((k_stencil_t<cplx, STENCIL_P, false, false, EPI_RESID, 1>), grid, dim3(256), 0, op->stream, q);
    } else if (a.xmode == 2 && a.epi == EPI_JACOBI && !a.scaled && !a.adjoint) {
        HELM_LAUNCH((k_stencil_t<cplx, STENCIL_P, false, false, EPI_JACOBI, 2>), grid, dim3(256), 0, op->stream, q);
    } else if (a.xmode == 3 && a.epi == EPI_RESID && !a.scaled && !a.adjoint) {
        HELM_LAUNCH((k_stencil_t<cplx, STENCIL_P, false, false, EPI_RESID, 3>), grid, dim3(256), 0, op->stream, q);
    } else if (a.xmode != 0) {
        HELM_FAIL(op, HELM_ERR_ARG, "unsupported fused stencil mode");
    } else if (a.scaled) {
        if (a.adjoint) launch_stencil_epi<STENCIL_P, true, true>(op->stream, grid, q, a.epi);
        else launch_stencil_epi<STENCIL_P, true, false>(op->stream, grid, q, a.epi);
    } else {
        if (a.adjoint) launch_stencil_epi<STENCIL_P, false, true>(op->stream, grid, q, a.epi);
        else launch_stencil_epi<STENCIL_P, false, false>(op->stream, grid, q, a.epi);
    }
    if (prof && e0) {
        hipEventRecord(e1, op->stream);
        int nact = (a.scal && op->active_hint >= 0 && op->active_hint < a.nrhs) ? op->active_hint : a.nrhs;   // inactive RHS are skipped on the device
        // algorithmic bytes of the launch: the stencil apply N*(32*B + 144) plus, for the fused
        // epilogues that take an operand vector (dot with r0 / s, residual w - Ax), that operand's one read
        const bool operand = (a.epi == EPI_DOT_W || a.epi == EPI_DOT_WY || a.epi == EPI_RESID || a.epi == EPI_JACOBI);
        op->ev_pending.push_back(std::make_pair((int)op->ev_used, (double)op->N * (32.0 * nact + 144.0 + (operand ? 16.0 * nact : 0.0))));
        op->ev_used += 2;
    }
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

int helm_launch_scale_planes(helm_op *op) {
    const int blocks = (int)((op->N + 255) / 256);
    HELM_LAUNCH(k_scale_planes, dim3(blocks), dim3(256), 0, op->stream, op->d_C, op->d_Cs, op->d_dinv, op->N, op->nblocks, op->diag_floor, op->nplanes, op->centre);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

int helm_launch_prep_rhs(helm_op *op, const cplx *dRHS, long long rhs_ld, long long row_off, cplx premul,
                         const cplx *sub, cplx *out, int nrhs) {
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_prep_rhs, grid, dim3(256), 0, op->stream, dRHS, rhs_ld, row_off, premul, sub, (const cplx *)nullptr, out, op->Nv);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

// out = premul*rhs - sub and the partial sums of ||out||^2 (reduce with FIN_NORM over helm_vec_num_blocks partials)
int helm_launch_prep_rhs_norm(helm_op *op, const cplx *dRHS, long long rhs_ld, long long row_off, cplx premul, const cplx *sub, cplx *out, int nrhs) {
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_prep_rhs_norm, grid, dim3(256), 0, op->stream, dRHS, rhs_ld, row_off, premul, sub, out, op->Nv, (double *)op->d_part, (int)grid.x);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

// bbar -> krylov vectors; `sub` etc. were folded into bbar by the caller (prep + scale)
int helm_launch_bicg_init(helm_op *op, int block, const cplx *dRHS, long long rhs_ld, cplx premul, const cplx *sub,
                          VecPtrs w, int nrhs, double rtol) {
    // w.t temporarily receives bbar = dinv * (premul*rhs - sub); the caller copies/keeps it
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_prep_rhs, grid, dim3(256), 0, op->stream, dRHS, rhs_ld, (long long)0, premul, sub,
                       (const cplx *)(op->d_dinv + (long long)block * op->N), w.t, op->Nv);
    HELM_LAUNCH(k_krylov_init, grid, dim3(256), 0, op->stream, (const cplx *)w.t, w, op->Nv, (double *)op->d_part, (int)grid.x);
    FinParams f; f.scal = op->d_scal; f.part = (const double *)op->d_part; f.nblk = grid.x; f.which = FIN_BICG_INIT; f.rtol = rtol; f.mask = nullptr; f.aux = nullptr;
    HELM_LAUNCH(k_fin, dim3(nrhs), dim3(256), 0, op->stream, f);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

// x = 0, r = r0 = bvec, p = v = 0 and the scalar records, for a system whose right-hand side is already formed
int helm_launch_krylov_init(helm_op *op, const cplx *bvec, VecPtrs w, int nrhs, double rtol) {
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_krylov_init, grid, dim3(256), 0, op->stream, bvec, w, op->Nv, (double *)op->d_part, (int)grid.x);
    FinParams f; f.scal = op->d_scal; f.part = (const double *)op->d_part; f.nblk = grid.x; f.which = FIN_BICG_INIT; f.rtol = rtol; f.mask = nullptr; f.aux = nullptr;
    HELM_LAUNCH(k_fin, dim3(nrhs), dim3(256), 0, op->stream, f);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

int helm_launch_bicg_p(helm_op *op, VecPtrs w, int nrhs) {
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_bicg_p, grid, dim3(256), 0, op->stream, w, op->Nv, (const RhsScal *)op->d_scal);
    return HELM_OK;
}
int helm_launch_bicg_s(helm_op *op, VecPtrs w, int nrhs) {
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_bicg_s, grid, dim3(256), 0, op->stream, w, op->Nv, (const RhsScal *)op->d_scal);
    return HELM_OK;
}
int helm_launch_bicg_xr(helm_op *op, VecPtrs w, const cplx *xp, const cplx *xs, int nrhs) {
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_bicg_xr, grid, dim3(256), 0, op->stream, w, xp, xs, op->Nv, (const RhsScal *)op->d_scal, (double *)op->d_part, (int)grid.x);
    return HELM_OK;
}
int helm_launch_cg_xr(helm_op *op, VecPtrs w, int nrhs) {
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_cg_xr, grid, dim3(256), 0, op->stream, w, op->Nv, (const RhsScal *)op->d_scal, (double *)op->d_part, (int)grid.x);
    return HELM_OK;
}
int helm_launch_cg_p(helm_op *op, VecPtrs w, int nrhs, int first) {
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_cg_p, grid, dim3(256), 0, op->stream, w, op->Nv, (const RhsScal *)op->d_scal, first);
    return HELM_OK;
}

int helm_launch_fin(helm_op *op, int which, int nrhs, int nblk_part) {
    FinParams f; f.scal = op->d_scal; f.part = (const double *)op->d_part; f.nblk = nblk_part; f.which = which; f.rtol = 0.0; f.mask = nullptr; f.aux = nullptr;
    HELM_LAUNCH(k_fin, dim3(nrhs), dim3(256), 0, op->stream, f);
    return HELM_OK;
}

int helm_launch_fin_ex(helm_op *op, int which, int nrhs, int nblk_part, const int *mask, double *aux) {
    FinParams f; f.scal = op->d_scal; f.part = (const double *)op->d_part; f.nblk = nblk_part; f.which = which; f.rtol = 0.0; f.mask = mask; f.aux = aux;
    HELM_LAUNCH(k_fin, dim3(nrhs), dim3(256), 0, op->stream, f);
    return HELM_OK;
}

int helm_launch_restart_copy_mask(helm_op *op, VecPtrs w, int nrhs, const int *mask) {
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_restart_copy, grid, dim3(256), 0, op->stream, w, op->Nv, mask);
    return HELM_OK;
}

int helm_launch_norm2(helm_op *op, const cplx *a, int nrhs) {
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_norm2, grid, dim3(256), 0, op->stream, a, op->Nv, (double *)op->d_part, (int)grid.x);
    return HELM_OK;
}

int helm_launch_finish(helm_op *op, const cplx *x, cplx *dU, long long u_ld, int nrhs, long long row_off) {
    dim3 grid(vec_blocks(op->Nv), nrhs);
    HELM_LAUNCH(k_finish, grid, dim3(256), 0, op->stream, x, dU, u_ld, row_off, op->Nv);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

int helm_launch_abs(helm_op *op, const cplx *in, cplx *out, long long n, double sign) {
    HELM_LAUNCH(k_abs_cplx, dim3((unsigned)std::min<long long>((n + 255) / 256, 1 << 20)), dim3(256), 0, op->stream, in, out, n, sign);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

int helm_launch_finish_ex(helm_op *op, const cplx *x, long long x_ld, long long x_off, cplx *dU, long long u_ld, long long row_off, int nrhs) {
    dim3 grid(vec_blocks(op->N), nrhs);
    HELM_LAUNCH(k_finish_ex, grid, dim3(256), 0, op->stream, x, x_ld, x_off, dU, u_ld, row_off, op->N);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

int helm_launch_prep_rhs_ex(helm_op *op, const cplx *dRHS, long long rhs_ld, long long row_off, cplx premul, const cplx *scale,
                            cplx *out, long long out_ld, long long out_off, int nrhs) {
    dim3 grid(vec_blocks(op->N), nrhs);
    HELM_LAUNCH(k_prep_rhs_ex, grid, dim3(256), 0, op->stream, dRHS, rhs_ld, row_off, premul, scale, out, out_ld, out_off, op->N);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

int helm_launch_rowscaled_system(helm_op *op) {
    if (!op->d_S) HIP_TRY(op, hipMalloc(&op->d_S, (size_t)36 * op->N * sizeof(cplx)));
    if (!op->d_rs) HIP_TRY(op, hipMalloc(&op->d_rs, (size_t)2 * op->N * sizeof(double)));
    HELM_LAUNCH(k_rowscale_system, dim3((unsigned)((op->N + 255) / 256)), dim3(256), 0, op->stream, (const cplx *)op->d_C, op->d_S, op->d_rs, op->N);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

int helm_launch_prep_rhs_rs(helm_op *op, const cplx *dRHS, long long rhs_ld, long long row_off, cplx premul, const double *rs,
                            cplx *out, long long out_ld, long long out_off, int nrhs) {
    dim3 grid(vec_blocks(op->N), nrhs);
    HELM_LAUNCH(k_prep_rhs_rs, grid, dim3(256), 0, op->stream, dRHS, rhs_ld, row_off, premul, rs, out, out_ld, out_off, op->N);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

// v[b][i] *= rs[i] in place (row equilibration of a residual before a refinement pass of the coupled system)
__global__ __launch_bounds__(256) void k_rowscale_inplace(cplx *v, const double *__restrict__ rs, long long NV) {
    const int b = blockIdx.y;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < NV; i += (long long)gridDim.x * blockDim.x) {
        cplx *p = v + (long long)b * NV + i;
        *p = cscale(*p, rs[i]);
    }
}
int helm_launch_rowscale_inplace(helm_op *op, cplx *v, const double *rs, long long NV, int nrhs) {
    HELM_LAUNCH(k_rowscale_inplace, dim3(vec_blocks(NV), nrhs), dim3(256), 0, op->stream, v, rs, NV);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

// dense right-hand sides from the triplets of a sparse matrix (no duplicate entries): R[col][row] = val
__global__ __launch_bounds__(256) void k_rhs_from_coo(const long long *__restrict__ row, const int *__restrict__ col, const cplx *__restrict__ val,
                                                      long long nnz, cplx *__restrict__ R, long long rows, int nrhs, int node_major) {
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) {
        if (node_major) R[row[k] * nrhs + col[k]] = val[k];          // the reference's (rows, nrhs) C-order array
        else R[(long long)col[k] * rows + row[k]] = val[k];
    }
}
int helm_launch_rhs_from_coo(helm_op *op, const long long *row, const int *col, const cplx *val, long long nnz, cplx *R, int nrhs, long long rows, int node_major) {
    HIP_TRY(op, hipMemsetAsync(R, 0, (size_t)nrhs * rows * sizeof(cplx), op->stream));
    if (nnz > 0) HELM_LAUNCH(k_rhs_from_coo, dim3((unsigned)std::min<long long>((nnz + 255) / 256, 65535)), dim3(256), 0, op->stream, row, col, val, nnz, R, rows, nrhs, node_major);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

// receiver sampling out[r][s] = sum_k val[k] U[s][col[k]] over the entries k of sparse row r (one thread per (r, s), fixed order)
__global__ __launch_bounds__(256) void k_sample(const cplx *__restrict__ U, int nsrc, long long ld, const long long *__restrict__ rowptr,
                                                const long long *__restrict__ col, const cplx *__restrict__ val, int nrec, cplx *__restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nrec * nsrc) return;
    const int r = (int)(t / nsrc), sidx = (int)(t % nsrc);
    cplx acc = cmake(0.0, 0.0);
    for (long long k = rowptr[r]; k < rowptr[r + 1]; ++k) cfma(acc, val[k], U[(long long)sidx * ld + col[k]]);
    out[t] = acc;
}
int helm_launch_sample(helm_op *op, const cplx *U, int nsrc, long long ld, const long long *rowptr, const long long *col, const cplx *val, int nrec, cplx *out) {
    const long long tot = (long long)nrec * nsrc;
    HELM_LAUNCH(k_sample, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, op->stream, U, nsrc, ld, rowptr, col, val, nrec, out);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

// Gardner's relation, the reference's density default: rho = 310 Re(c)^0.25 (discretization.py:70)
__global__ __launch_bounds__(256) void k_gardner_rho(const cplx *__restrict__ c, double *__restrict__ rho, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        rho[i] = 310.0 * pow(c[i].x, 0.25);
}
int helm_launch_gardner_rho(helm_op *op) {
    HELM_LAUNCH(k_gardner_rho, dim3(vec_blocks(op->N)), dim3(256), 0, op->stream, (const cplx *)op->d_c, op->d_rho, op->N);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

int helm_launch_zero(helm_op *op, cplx *p, long long n) {
    HELM_LAUNCH(k_zero, dim3(vec_blocks(n)), dim3(256), 0, op->stream, p, n);
    return HELM_OK;
}

int helm_launch_imaging(helm_op *op, const cplx *uf, const cplx *ub, int nsrc, const cplx *scaler, cplx *g) {
    HELM_LAUNCH(k_imaging, dim3(vec_blocks(op->N)), dim3(256), 0, op->stream, uf, ub, nsrc, scaler, g, op->N);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}
