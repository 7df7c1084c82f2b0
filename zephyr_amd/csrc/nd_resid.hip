// Direct solver, node-major plumbing around the passes: layout transposes, the right-hand-side preparation, and the TRUE residual
// q' - A x of the returned wavefields evaluated with the 9-point stencil (what makes the direct path's result a checked one).
#include "nd_internal.hpp"

namespace {

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
                if (Uout) Uout[cell * ldu + j] = conj_scaled(oscale, win[o + 1][1]);
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
                                                      int qnorm, cplx *__restrict__ Uout, int ldu, cplx oscale, const unsigned char *__restrict__ qm, int xin_is_u) {
    // xin_is_u: Xin holds u = conj(oscale x) (the back substitution wrote the caller's array itself): the window takes conj(u) = oscale x and q is scaled to
    // match -- r_s = oscale q - A (oscale x) = oscale r, the relative residual is the same number (bit for bit when oscale = 1)
    const double xsg = xin_is_u ? -1.0 : 1.0;
    const bool qsc = xin_is_u && !(oscale.x == 1.0 && oscale.y == 0.0);
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
            win[d][1].y *= xsg; win[d][2].y *= xsg; pre[d].y *= xsg;
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
                v.y *= xsg;
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
                if (qsc) r = cmul(oscale, r);
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
                    const cplx u = conj_scaled(oscale, win[o + 1][1]);
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

__global__ void k_recover_x(const cplx *__restrict__ U, cplx *__restrict__ X, long long n, cplx inv) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        X[i] = cmul(inv, cconj(U[i]));
}

// y += x, or y += conj(x) when y holds the conjugated wavefield  (refinement update), n elements
__global__ void k_axpy_one(cplx *y, const cplx *x, long long n, int conj) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = cadd(y[i], conj ? cconj(x[i]) : x[i]);
}

}  // namespace

void launch_transpose(hipStream_t st, const cplx *in, long long rows, long long cols, cplx *out, int swap, int conj) {
    // (the long dimension rides on gridDim.x: swap = 1 means `cols` is the long one's partner, see k_transpose)
    const dim3 grid = swap ? dim3((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32)) : dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    HELM_LAUNCH(k_transpose, grid, dim3(256), 0, st, in, rows, cols, out, swap, conj);
}

int nd_prep_transpose_norm(helm_op *op, const cplx *rhs, long long rhs_ld, long long row_off, cplx premul, const cplx *sub, cplx *Qt, long long N, int nrhs,
                           double *part, int nblk_cap, int *nblk_out) {
    const int nblk = (int)std::max<long long>(1, std::min<long long>((N + 31) / 32, std::min(nblk_cap, 1024)));
    HELM_LAUNCH(k_prep_transpose_norm, dim3(nblk, (nrhs + 31) / 32), dim3(256), 0, op->stream, rhs, rhs_ld, row_off, premul, sub, Qt, N, nrhs, part, nblk);
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
    const int xin_is_u = ex ? ex->xin_is_u : 0;
    int lx = 64;
    while (lx < ncol && lx < 256) lx <<= 1;
    const int ly = 256 / lx;
    // a thread marches along x over segments of 32 cells of 4 grid rows.  Measured on 1024^2 x 256 (HISTORY.md): rows per thread 1 / 2 / 4 -> 2.52 / 3.71 /
    // 2.23 ms norm-only; with the coefficients staged in LDS (k_resid_nm_lds, full-width batches) 4 / 6 / 8 rows -> 1.57 / 1.63 / 2.73 ms (the wider
    // register window halves the occupancy); wave-uniform scalar loads of the coefficients instead: SGPR spills at 4 rows.
    const int seg = RESID_SEG, rpt = 4;
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
        if (ly == 1)        // full-width batches: four waves share a tile, its coefficients staged in LDS; the wavefield (read by nobody on the GPU) stored nontemporally
            HELM_LAUNCH((k_resid_nm_lds<4, 1>), dim3(nblk), dim3(256, 1), 0, op->stream, planes, op->nz, op->nx, Xin + c0, ldin, Q + (qmap ? 0 : c0), ldq,
                               qmap ? qmap + c0 : nullptr, nc, store, Rout ? Rout + (qmap ? 0 : c0) : nullptr, part + (long long)c0 * 4 * nblk, nblk, ntiles,
                               qnorm, Uout ? Uout + c0 : nullptr, ldu, oscale, c0 == 0 ? qmask : nullptr, xin_is_u);
        else if (xin_is_u) HELM_FAIL(op, HELM_ERR_STATE, "node-major residual: direct output needs the full-width kernel");
        else
            HELM_LAUNCH(k_resid_nm<4>, dim3(nblk), dim3(lx, ly), 0, op->stream, planes, op->nz, op->nx, Xin + c0, ldin, Q + (qmap ? 0 : c0), ldq,
                               qmap ? qmap + c0 : nullptr, nc, store, Rout ? Rout + (qmap ? 0 : c0) : nullptr, part + (long long)c0 * 4 * nblk, nblk, seg, ntiles,
                               qnorm, Uout ? Uout + c0 : nullptr, ldu, oscale);
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

// Xt = conj(U) / oscale (direct output: a refinement pass needs x of every cell back where the passes expect it)
int nd_recover_x(helm_op *op, const cplx *U, cplx *Xt, long long elems, cplx oscale) {
    const double d = oscale.x * oscale.x + oscale.y * oscale.y;
    const cplx inv = cmake(oscale.x / d, -oscale.y / d);
    HELM_LAUNCH(k_recover_x, dim3((unsigned)std::min<long long>((elems + 255) / 256, 65535)), dim3(256), 0, op->stream, U, Xt, elems, inv);
    return check_kernels(op, "recovering x from the wavefield array");
}

int nd_scatter_add_cols(helm_op *op, cplx *Xt, int ldq, const int *d_cols, int k, const cplx *Dp, long long N) {
    HELM_LAUNCH(k_scatter_add_cols, dim3((unsigned)std::min<long long>((N * k + 255) / 256, 1 << 20)), dim3(256), 0, op->stream, Xt, ldq, d_cols, k, Dp, N);
    return check_kernels(op, "column scatter");
}

int nd_pack_cols(helm_op *op, const cplx *Qt, int ldq, const int *d_cols, int k, cplx *Rp, long long N) {
    HELM_LAUNCH(k_pack_cols, dim3((unsigned)std::min<long long>((N * k + 255) / 256, 1 << 20)), dim3(256), 0, op->stream, Qt, ldq, d_cols, k, Rp, N);
    return check_kernels(op, "column packing");
}

// out[c][r] = in[r][c] for an (rows x cols) array (layout conversions at the C ABI: the reference's (N, nrhs) arrays <-> one right-hand side per row)
int nd_transpose(helm_op *op, const cplx *in, long long rows, long long cols, cplx *out) {
    const bool swap = rows > cols;            // the long dimension rides on gridDim.x
    dim3 grid = swap ? dim3((unsigned)((rows + 31) / 32), (unsigned)((cols + 31) / 32)) : dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    HELM_LAUNCH(k_transpose, grid, dim3(256), 0, op->stream, in, rows, cols, out, swap ? 1 : 0, 0);
    return check_kernels(op, "transpose");
}

// Xt (cells x nrhs) -> U (nrhs x N), conjugated on request
int nd_transpose_out(helm_op *op, const cplx *Xt, long long N, int nrhs, cplx *U, int conj) {
    HELM_LAUNCH(k_transpose, dim3((unsigned)((N + 31) / 32), (nrhs + 31) / 32), dim3(256), 0, op->stream, Xt, N, (long long)nrhs, U, 1, conj);
    return check_kernels(op, "transpose out");
}

int nd_axpy_one(helm_op *op, cplx *y, const cplx *x, long long n, int conj) {
    HELM_LAUNCH(k_axpy_one, dim3((unsigned)std::min<long long>((n + 255) / 256, 65535)), dim3(256), 0, op->stream, y, x, n, conj);
    return HELM_OK;
}

