// 3-D 27-point Helmholtz operator (BASELINE config 5).  The reference has no 3-D discretisation
// (zephyr/backend/base.py:20,36-40 only reserves `ny`; source.py:43-44 raises NotImplementedError), so the
// operator is defined by this project -- see oracle/helm3d_oracle.py for the formula and its analytic check.
//
//   C_o = bbar_o (Lx(ox) m(oy) m(oz)/dx^2 + m(ox) Ly(oy) m(oz)/dy^2 + m(ox) m(oy) Lz(oz)/dz^2)
//         + K_{p+o} (a [o==0] + (1-a) m(ox) m(oy) m(oz)),   slot k = 9 (oz+1) + 3 (oy+1) + (ox+1)
//
// Kernels: k_assemble_3d (model -> 27 planes) and k_stencil3 (batched apply: one 64 x 4 tile of one z-plane per
// workgroup, the three z-planes of the input staged in LDS with halo, the thread's 27 coefficients held in
// registers over the right-hand-side loop, same fused epilogues as the 2-D kernel).  Everything else (Krylov
// vector kernels, finalize kernels, drivers) is shared with the 2-D path.
#include "helm_internal.hpp"
#include <complex>

namespace {

__device__ inline int clampi3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// one coefficient of the 27-point operator from what it depends on -- used by the assembly kernel (planes) AND by the on-the-fly apply, with the
// floating-point contraction pinned so that both produce the same bits whatever surrounds the call
//   out = bbar (lx' + ly' + lz') + K mass,  lx' = Lx[ox] my mz / dx^2 ..., bbar = (b_i + b_j) / 2, mass = [o == 0] blend + (1 - blend) mx my mz
__device__ __forceinline__ cplx coeff3(int ox, int oy, int oz, cplx Lxo, cplx Lyo, cplx Lzo, double idx2, double idy2, double idz2,
                                       double b0, double bj, cplx Knb, double blend) {
#pragma clang fp contract(off)
    const double bbar = (b0 + bj) / 2.0;
    const double mx = ox == 0 ? 2.0 / 3.0 : 1.0 / 6.0, my = oy == 0 ? 2.0 / 3.0 : 1.0 / 6.0, mz = oz == 0 ? 2.0 / 3.0 : 1.0 / 6.0;
    const double sx = my * mz * idx2, sy = mx * mz * idy2, sz = mx * my * idz2;
    const double mass = ((ox == 0 && oy == 0 && oz == 0) ? blend : 0.0) + (1.0 - blend) * mx * my * mz;
    const double lre = (Lxo.x * sx + Lyo.x * sy) + Lzo.x * sz, lim = (Lxo.y * sx + Lyo.y * sy) + Lzo.y * sz;
    return cmake(lre * bbar + Knb.x * mass, lim * bbar + Knb.y * mass);
}

// K = om^2 / (rho c^2) of one point (Smith's division, as cdiv), contraction pinned for the same reason
__device__ __forceinline__ cplx kfield3(cplx om2, cplx cj, double rj) {
#pragma clang fp contract(off)
    const double br = (cj.x * cj.x - cj.y * cj.y) * rj, bi = (cj.x * cj.y + cj.y * cj.x) * rj;
    if (fabs(br) >= fabs(bi)) {
        const double r = bi / br, d = br + bi * r;
        return cmake((om2.x + om2.y * r) / d, (om2.y - om2.x * r) / d);
    }
    const double r = br / bi, d = br * r + bi;
    return cmake((om2.x * r + om2.y) / d, (om2.y * r - om2.x) / d);
}

struct Asm3Params {
    int nz, ny, nx;
    double dx, dy, dz;
    cplx om;
    double blend;
};

// Lt: [axis][3][n] laid out as Lx(-1)[nx], Lx(0)[nx], Lx(+1)[nx], Ly..., Lz...
__global__ __launch_bounds__(256) void k_assemble_3d(Asm3Params P, const cplx *__restrict__ c, const double *__restrict__ rho,
                                                     const cplx *__restrict__ Lx, const cplx *__restrict__ Ly, const cplx *__restrict__ Lz,
                                                     cplx *__restrict__ C) {
    const long long N = (long long)P.nz * P.ny * P.nx;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int ix = (int)(i % P.nx), iy = (int)((i / P.nx) % P.ny), iz = (int)(i / ((long long)P.nx * P.ny));
    const bool edge = ix == 0 || ix == P.nx - 1 || iy == 0 || iy == P.ny - 1 || iz == 0 || iz == P.nz - 1;
    const double b0 = 1.0 / rho[i];
    const cplx om2 = cmul(P.om, P.om);
    const double idx2 = 1.0 / (P.dx * P.dx), idy2 = 1.0 / (P.dy * P.dy), idz2 = 1.0 / (P.dz * P.dz);
#pragma unroll 1
    for (int oz = -1; oz <= 1; ++oz)
#pragma unroll 1
        for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
            for (int ox = -1; ox <= 1; ++ox) {
                const int k = 9 * (oz + 1) + 3 * (oy + 1) + (ox + 1);
                cplx out;
                if (edge) out = (k == 13) ? cmake(1.0, 0.0) : cmake(0.0, 0.0);
                else {
                    const int jz = clampi3(iz + oz, 0, P.nz - 1), jy = clampi3(iy + oy, 0, P.ny - 1), jx = clampi3(ix + ox, 0, P.nx - 1);
                    const long long j = ((long long)jz * P.ny + jy) * P.nx + jx;
                    const double rj = rho[j];
                    const cplx cj = c[j];
                    const cplx Knb = kfield3(om2, cj, rj);
                    out = coeff3(ox, oy, oz, Lx[(long long)(ox + 1) * P.nx + ix], Ly[(long long)(oy + 1) * P.ny + iy], Lz[(long long)(oz + 1) * P.nz + iz],
                                 idx2, idy2, idz2, b0, 1.0 / rj, Knb, P.blend);
                }
                C[(long long)k * N + i] = out;
            }
}

// K = om^2 / (rho c^2) and b = 1 / rho per point, with the very expressions k_assemble_3d uses for a neighbour: the on-the-fly kernel below then rebuilds
// every coefficient bit for bit
__global__ __launch_bounds__(256) void k_kb_3d(Asm3Params P, const cplx *__restrict__ c, const double *__restrict__ rho, cplx *__restrict__ K, double *__restrict__ b) {
    const long long N = (long long)P.nz * P.ny * P.nx;
    const cplx om2 = cmul(P.om, P.om);
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (long long)gridDim.x * blockDim.x) {
        const double rj = rho[j];
        const cplx cj = c[j];
        K[j] = kfield3(om2, cj, rj);
        b[j] = 1.0 / rj;
    }
}

// ---- batched 27-point apply -----------------------------------------------------------------------
struct Stencil3Params {
    const cplx *planes;      // 27 planes, stride N
    const void *X;           // complex128, or complex64 in the mixed-precision instantiations (the vectors of the multigrid cycle's finest level, mg3d.hip)
    void *Y;
    const void *W;
    long long ld, N;
    int nz, ny, nx, nrhs, ntx, nty, nblk;
    int zfast;               // tile order: 1 = z fastest (workgroups that run together share their z-halo planes in the XCD's L2)
    const RhsScal *scal;
    double *part;
    const cplx *dinv;        // EPI_JACOBI: 1 / diagonal
    double omega_j;
    // OTF: the coefficients are rebuilt from K, b (per point) and the per-axis factor tables instead of being read from the 27 planes
    const cplx *K3; const double *b3; const cplx *Lx, *Ly, *Lz;
    double idx2, idy2, idz2, blend;
    int npart;               // partial sums per right-hand side and quantity the caller's reduction reads (>= the workgroups of this launch: the rest is zeroed)
};

__device__ inline double wave_sum3(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ inline int xcd_swizzle3(int bid, int nblk) {
    const int q = nblk / HELM_NXCD, rem = nblk % HELM_NXCD;
    const int x = bid % HELM_NXCD, k = bid / HELM_NXCD;
    return x * q + (x < rem ? x : rem) + k;
}

constexpr int T3X = 64, T3Y = 4;

// OTF (round 5): the thread's 27 coefficients are not read from the stored planes (432 B per point: 46 % of the launch's bytes at 16 right-hand sides, and
// SURVEY.md 7 hard-part 6 calls rebuilding them mandatory) but rebuilt once, before the right-hand-side loop, from K = om^2 / (rho c^2) and b = 1 / rho of the
// 3 x 3 x 3 neighbourhood -- staged through the tile's own LDS buffers, 24 B per point with halo -- and the three 1-D factor tables: the same expressions in
// the same order as k_assemble_3d, so the apply is bit for bit the stored-plane apply.  Amortised over the right-hand sides of the launch (helm3d_launch_apply
// takes this path from 4 right-hand sides up); the coarse levels that carry a Galerkin operator keep their stored planes.
// TX / TW / TY (round 6): element types of X, W and Y -- cplx, or cplxf for vectors the multigrid cycle keeps in single precision.  With X in complex64 the 27
// multiply-adds run in single precision too (coefficients built in fp64, rounded once per thread; the residual / sweep epilogue in fp64): a first version that
// converted every staged value to fp64 on its way out of LDS was 8 % SLOWER than the all-fp64 kernel -- 54 conversions per output on top of 108 fp64
// multiply-adds -- where this one halves the tile staging, the coefficient registers (108 -> 54) and the multiply-add cost.
__device__ __forceinline__ cplx cvt64(cplx a) { return a; }
__device__ __forceinline__ cplx cvt64(cplxf a) { return to_f64(a); }
template <bool SCALED, int EPI, bool OTF, class TX, class TW, class TY>
__device__ __forceinline__ void stencil3_body(const Stencil3Params &q) {
    constexpr int LW = T3X + 2, LH = T3Y + 2;          // one staged plane: LH rows of LW
    constexpr int PLANE = LW * LH;                      // 396 elements
    constexpr int NEL = 3 * PLANE;                      // three z-planes
    constexpr int NLOAD = (NEL + 255) / 256;            // elements staged per thread
    constexpr size_t TILEB = 2 * (size_t)NEL * sizeof(TX), OTFB = OTF ? (size_t)NEL * (sizeof(cplx) + sizeof(double)) : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TILEB > OTFB ? TILEB : OTFB];
    TX *const tile0 = reinterpret_cast<TX *>(smem);     // tile[buf] = tile0 + buf * NEL
    __shared__ double red[16];

    const int tid = threadIdx.x, lane = tid & 63, wy = tid >> 6;
    const int t = xcd_swizzle3(blockIdx.x, q.nblk);
    // Tile order.  Every workgroup loops over the right-hand sides of its tile, so the three z-planes it stages are shared with the
    // tiles above and below only if THOSE run at the same time: with x fastest the same (x, y) tile of the next plane comes a whole
    // plane of tiles later and, at 8-16 right-hand sides, its halo planes have long left the 4 MB L2 (HBM traffic up to 3x the
    // input); with z fastest the ~100 workgroups an XCD has in flight are consecutive planes of one (x, y) column.
    int tx, ty, iz;
    if (q.zfast) { iz = t % q.nz; tx = (t / q.nz) % q.ntx; ty = t / (q.nz * q.ntx); }
    else { tx = t % q.ntx; ty = (t / q.ntx) % q.nty; iz = t / (q.ntx * q.nty); }
    const int x0 = tx * T3X, y0 = ty * T3Y;
    const int nz = q.nz, ny = q.ny, nx = q.nx;
    const long long N = q.N;
    const int col = x0 + lane, row = y0 + wy;
    const bool ok = col < nx && row < ny;
    const long long idx = ((long long)iz * ny + row) * nx + col;

    using TA = TX;                                       // the type the stencil sum is formed in: that of the staged values
    TA cf[27];
    if (!OTF) {
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            if (SCALED && k == 13) cf[k] = vone<TA>();
            else cf[k] = ok ? vfrom<TA>(q.planes[(long long)k * N + idx]) : vzero<TA>();
        }
    } else {
        // K of the three planes with halo into tile[0], b into tile[1] (as doubles), then every thread reads its 27 neighbours
        cplx *Kt = reinterpret_cast<cplx *>(smem);
        double *bt = reinterpret_cast<double *>(smem + (size_t)NEL * sizeof(cplx));
        for (int e = tid; e < NEL; e += 256) {
            const int p = e / PLANE, rem = e - p * PLANE, r = rem / LW, cc = rem - r * LW;
            const int gz = iz - 1 + p, gy = y0 - 1 + r, gx = x0 - 1 + cc;
            cplx kv = cmake(0.0, 0.0); double bv = 0.0;
            if (gz >= 0 && gz < nz && gy >= 0 && gy < ny && gx >= 0 && gx < nx) { const long long j = ((long long)gz * ny + gy) * nx + gx; kv = q.K3[j]; bv = q.b3[j]; }
            Kt[e] = kv; bt[e] = bv;
        }
        __syncthreads();
        const bool edge = !ok || col == 0 || col == nx - 1 || row == 0 || row == ny - 1 || iz == 0 || iz == nz - 1;
        const int ce = (wy + 1) * LW + lane + 1;                    // this thread's own cell in a staged plane
        const double b0 = bt[PLANE + ce];
        cplx lxv[3], lyv[3], lzv[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            lxv[o] = ok ? q.Lx[(long long)o * nx + col] : cmake(0.0, 0.0);
            lyv[o] = ok ? q.Ly[(long long)o * ny + row] : cmake(0.0, 0.0);
            lzv[o] = q.Lz[(long long)o * nz + iz];
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    const int k = 9 * p + 3 * r + cc;
                    cplx out;
                    if (edge) out = (k == 13 && ok) ? cmake(1.0, 0.0) : cmake(0.0, 0.0);
                    else {
                        const int e = p * PLANE + (wy + r) * LW + lane + cc;
                        out = coeff3(cc - 1, r - 1, p - 1, lxv[cc], lyv[r], lzv[p], q.idx2, q.idy2, q.idz2, b0, bt[e], Kt[e], q.blend);
                    }
                    cf[k] = vfrom<TA>(out);
                }
        __syncthreads();                                            // the tile buffers go back to the right-hand sides
    }

    // staging map: element e of the 3-plane tile -> (plane p, tile row r, tile column cc)
    TX pre[NLOAD];
    auto prefetch = [&](int b) {
        const TX *Xb = static_cast<const TX *>(q.X) + (long long)b * q.ld;
#pragma unroll
        for (int l = 0; l < NLOAD; ++l) {
            const int e = tid + 256 * l;
            TX v = vzero<TX>();
            if (e < NEL) {
                const int p = e / PLANE, rem = e - p * PLANE, r = rem / LW, cc = rem - r * LW;
                const int gz = iz - 1 + p, gy = y0 - 1 + r, gx = x0 - 1 + cc;
                if (gz >= 0 && gz < nz && gy >= 0 && gy < ny && gx >= 0 && gx < nx) v = Xb[((long long)gz * ny + gy) * nx + gx];
            }
            pre[l] = v;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int l = 0; l < NLOAD; ++l) {
            const int e = tid + 256 * l;
            if (e < NEL) tile0[buf * NEL + e] = pre[l];
        }
    };

    const int bstep = gridDim.y;
    int b = blockIdx.y;
    auto active = [&](int bb) { return q.scal == nullptr || q.scal[bb].status == ST_ACTIVE; };
    while (b < q.nrhs && !active(b)) b += bstep;
    if (b < q.nrhs) prefetch(b);
    int buf = 0;
    while (b < q.nrhs) {
        stage(buf);
        int bn = b + bstep;
        while (bn < q.nrhs && !active(bn)) bn += bstep;
        if (bn < q.nrhs) prefetch(bn);
        __syncthreads();

        TA acc = vzero<TA>(), xc0 = vzero<TA>();
        const TX *tb = tile0 + buf * NEL + wy * LW + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    const int k = 9 * p + 3 * r + cc;
                    const TA xv = tb[p * PLANE + r * LW + cc];
                    if (SCALED && k == 13) { acc.x += xv.x; acc.y += xv.y; }
                    else cfma(acc, cf[k], xv);
                    if (k == 13) xc0 = xv;
                }
        }
        double dsum[4] = {0.0, 0.0, 0.0, 0.0};
        if (ok) {
            cplx y = cvt64(acc);
            const cplx xc = cvt64(xc0);
            const long long g = (long long)b * q.ld + idx;
            const TW *Wp = static_cast<const TW *>(q.W);
            if (EPI == EPI_RESID) { const cplx w = cvt64(Wp[g]); y = csub(w, y); dsum[0] += cabs2(y); }
            else if (EPI == EPI_DOT_W) { const cplx w = cvt64(Wp[g]); dsum[0] += w.x * y.x + w.y * y.y; dsum[1] += w.x * y.y - w.y * y.x; }
            else if (EPI == EPI_DOT_XY) { dsum[0] += y.x * xc.x + y.y * xc.y; dsum[1] += y.x * xc.y - y.y * xc.x; dsum[2] += cabs2(y); }
            else if (EPI == EPI_DOT_YY) { dsum[0] += cabs2(y); }
            else if (EPI == EPI_DOT_WY) { const cplx w = cvt64(Wp[g]); dsum[0] += y.x * w.x + y.y * w.y; dsum[1] += y.x * w.y - y.y * w.x; dsum[2] += cabs2(y); }
            else if (EPI == EPI_JACOBI) { const cplx res = csub(cvt64(Wp[g]), y); y = xc; cfma(y, cscale(q.dinv[idx], q.omega_j), res); }
            static_cast<TY *>(q.Y)[g] = vfrom<TY>(y);
        }
        if (EPI != EPI_NONE && EPI != EPI_JACOBI) {
            const int wave = tid >> 6;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) dsum[qq] = wave_sum3(dsum[qq]);
            if (lane == 0) { red[wave * 4 + 0] = dsum[0]; red[wave * 4 + 1] = dsum[1]; red[wave * 4 + 2] = dsum[2]; red[wave * 4 + 3] = dsum[3]; }
            __syncthreads();
            if (tid == 0) {
                double *pp = q.part + ((long long)b * 4) * q.npart + blockIdx.x;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) pp[(long long)qq * q.npart] = (red[qq] + red[4 + qq]) + (red[8 + qq] + red[12 + qq]);
                if (q.nblk + (int)blockIdx.x < q.npart)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) pp[(long long)qq * q.npart + q.nblk] = 0.0;
            }
            __syncthreads();
        }
        buf ^= 1;
        b = bn;
    }
}

template <bool SCALED, int EPI, bool OTF = false, class TX = cplx, class TW = cplx, class TY = cplx>
__global__ __launch_bounds__(256) void k_stencil3(Stencil3Params q) { stencil3_body<SCALED, EPI, OTF, TX, TW, TY>(q); }
// the single-precision instantiations (X complex64, W complex128).  (Bounded to three waves per SIMD -- __launch_bounds__(256, 3) -- the coefficient build spills
// 68-100 bytes per lane and config 5 loses the 0.02 s per frequency these kernels gain; left to the register allocator: 156-210 registers.)
template <int EPI, bool OTF, class TY>
__global__ __launch_bounds__(256) void k_stencil3_lp(Stencil3Params q) { stencil3_body<false, EPI, OTF, cplxf, cplx, TY>(q); }

// (Round 5 measured a lane-shifted form of this kernel -- a thread reads only the nine values of its own x column from LDS and forms three partial sums, two
// of which cross to the neighbour lanes as v_mov_b32_dpp wave_shl / wave_shr values; lanes 0 and 63 as halo lanes, 62 outputs per wave -- 9 LDS reads per output
// instead of 27: 2426 us at 16 right-hand sides against 2188 us for the kernel above, 1493 against 1348 at 8.  A third of the LDS traffic bought 10 % per tile,
// the 62-wide tiling costs 25 % more tiles at nx = 256: the launch is bound by the latency of staging 3 x 396 elements per 256 outputs with one right-hand side
// of prefetch in flight at two workgroups per compute unit (the 27 coefficients are 108 of the 203-210 VGPRs), not by LDS reads.  Removed; HISTORY.md.)

void profile3(int n, int npml, double h, double cpml, std::complex<double> om, std::vector<cplx> &Lt) {
    // padded stretch profile xi and the three Laplacian factor arrays L(-1), L(0), L(+1)
    std::vector<double> g(n, 0.0);
    const double L = h * (npml - 1);
    for (int k = 0; k < npml && k < n; ++k) g[k] = cpml * cos((M_PI / 2) * (k * h / L));
    for (int k = 0; k < npml && k < n; ++k) g[n - npml + k] = cpml * cos((M_PI / 2) * ((npml - 1 - k) * h / L));
    std::vector<std::complex<double>> xi(n + 2);
    for (int p = 0; p < n + 2; ++p) {
        const int k = p == 0 ? 0 : (p == n + 1 ? n - 1 : p - 1);
        xi[p] = 1.0 - std::complex<double>(0.0, g[k]) / om;
    }
    Lt.resize((size_t)3 * n);
    for (int i = 0; i < n; ++i) {
        const std::complex<double> c = xi[i + 1];
        const std::complex<double> lm = 1.0 / (c * (c + xi[i]) / 2.0), lp = 1.0 / (c * (c + xi[i + 2]) / 2.0);
        const std::complex<double> l0 = -(lm + lp);
        Lt[i] = cmake(lm.real(), lm.imag());
        Lt[(size_t)n + i] = cmake(l0.real(), l0.imag());
        Lt[(size_t)2 * n + i] = cmake(lp.real(), lp.imag());
    }
}

}  // namespace

int helm3d_launch_assemble(helm_op *op, double freq_re, double freq_im, double tau, double cPML) {
    if (op->nPML < 2 || op->nPML > op->nx || op->nPML > op->ny || op->nPML > op->nz) HELM_FAIL(op, HELM_ERR_ARG, "nPML out of range for the grid");
    std::complex<double> om(2.0 * M_PI * freq_re, 2.0 * M_PI * freq_im);
    if (std::isfinite(tau) && tau != 0.0) om -= std::complex<double>(0.0, 1.0 / tau);
    std::vector<cplx> Lx, Ly, Lz;
    profile3(op->nx, op->nPML, op->dx, cPML, om, Lx);
    profile3(op->ny, op->nPML, op->dy, cPML, om, Ly);
    profile3(op->nz, op->nPML, op->dz, cPML, om, Lz);
    std::vector<cplx> all(Lx);
    all.insert(all.end(), Ly.begin(), Ly.end());
    all.insert(all.end(), Lz.begin(), Lz.end());
    // levels of the layer-preserving multigrid hierarchy (mg3d.hip) bring their own factors: non-uniform node spacing, 1/h^2 included
    const bool over = op->lap_override.size() == all.size();
    if (over) all = op->lap_override;
    // (the factor tables stay on the operator: the on-the-fly apply reads them at every launch)
    op->otf3 = false;
    if (op->d_L3 && op->l3_elems != all.size()) { helm_pool_free(op->device, op->d_L3, op->l3_elems * sizeof(cplx)); op->d_L3 = nullptr; op->l3_elems = 0; }
    if (!op->d_L3) { op->d_L3 = (cplx *)helm_pool_alloc(op->device, all.size() * sizeof(cplx)); op->l3_elems = op->d_L3 ? all.size() : 0; }
    cplx *d_L = op->d_L3;
    if (!d_L) HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc of the stretch profiles failed");
    HIP_TRY(op, hipMemcpyAsync(d_L, all.data(), all.size() * sizeof(cplx), hipMemcpyHostToDevice, op->stream));
    Asm3Params P;
    P.nz = op->nz; P.ny = op->ny; P.nx = op->nx; P.dx = op->dx; P.dy = op->dy; P.dz = op->dz;
    if (over) P.dx = P.dy = P.dz = 1.0;
    P.om = cmake(om.real(), om.imag()); P.blend = 0.5;
    const int blocks = (int)((op->N + 255) / 256);
    HELM_LAUNCH(k_assemble_3d, dim3(blocks), dim3(256), 0, op->stream, P, (const cplx *)op->d_c, (const double *)op->d_rho,
                       (const cplx *)d_L, (const cplx *)(d_L + 3 * (size_t)op->nx), (const cplx *)(d_L + 3 * (size_t)op->nx + 3 * (size_t)op->ny), op->d_C);
    HIP_TRY(op, hipGetLastError());
    // K and b for the on-the-fly apply (24 B per point; a handle that cannot have them simply keeps reading its planes)
    if (!op->d_K3) op->d_K3 = (cplx *)helm_pool_alloc(op->device, (size_t)op->N * sizeof(cplx));
    if (!op->d_b3) op->d_b3 = (double *)helm_pool_alloc(op->device, (size_t)op->N * sizeof(double));
    if (op->d_K3 && op->d_b3) {
        HELM_LAUNCH(k_kb_3d, dim3((unsigned)std::min<long long>(blocks, 65535)), dim3(256), 0, op->stream, P, (const cplx *)op->d_c, (const double *)op->d_rho, op->d_K3, op->d_b3);
        HIP_TRY(op, hipGetLastError());
        op->otf_idx2 = 1.0 / (P.dx * P.dx); op->otf_idy2 = 1.0 / (P.dy * P.dy); op->otf_idz2 = 1.0 / (P.dz * P.dz); op->otf_blend = P.blend;
        op->otf3 = true;
    }
    HIP_TRY(op, hipStreamSynchronize(op->stream));
    return HELM_OK;
}

int helm3d_apply_num_blocks(const helm_op *op) {
    return ((op->nx + T3X - 1) / T3X) * ((op->ny + T3Y - 1) / T3Y) * op->nz;
}

static void launch3_otf(hipStream_t st, dim3 grid, const Stencil3Params &q, int epi) {
    switch (epi) {
    case EPI_NONE: HELM_LAUNCH((k_stencil3<false, EPI_NONE, true>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_W: HELM_LAUNCH((k_stencil3<false, EPI_DOT_W, true>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_XY: HELM_LAUNCH((k_stencil3<false, EPI_DOT_XY, true>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_YY: HELM_LAUNCH((k_stencil3<false, EPI_DOT_YY, true>), grid, dim3(256), 0, st, q); break;
    case EPI_RESID: HELM_LAUNCH((k_stencil3<false, EPI_RESID, true>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_WY: HELM_LAUNCH((k_stencil3<false, EPI_DOT_WY, true>), grid, dim3(256), 0, st, q); break;
    case EPI_JACOBI: HELM_LAUNCH((k_stencil3<false, EPI_JACOBI, true>), grid, dim3(256), 0, st, q); break;
    default: break;
    }
}

template <bool SCALED>
static void launch3_epi(hipStream_t st, dim3 grid, const Stencil3Params &q, int epi) {
    switch (epi) {
    case EPI_NONE: HELM_LAUNCH((k_stencil3<SCALED, EPI_NONE>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_W: HELM_LAUNCH((k_stencil3<SCALED, EPI_DOT_W>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_XY: HELM_LAUNCH((k_stencil3<SCALED, EPI_DOT_XY>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_YY: HELM_LAUNCH((k_stencil3<SCALED, EPI_DOT_YY>), grid, dim3(256), 0, st, q); break;
    case EPI_RESID: HELM_LAUNCH((k_stencil3<SCALED, EPI_RESID>), grid, dim3(256), 0, st, q); break;
    case EPI_DOT_WY: HELM_LAUNCH((k_stencil3<SCALED, EPI_DOT_WY>), grid, dim3(256), 0, st, q); break;
    case EPI_JACOBI: HELM_LAUNCH((k_stencil3<SCALED, EPI_JACOBI>), grid, dim3(256), 0, st, q); break;
    default: break;
    }
}

// mixed precision (ApplyArgs::x32 / y32): X in complex64, W in complex128, Y in complex64 or complex128 -- the sweeps and the residual of the multigrid cycle's
// finest level (mg3d.hip); unscaled planes only
template <bool OTF>
static bool launch3_mixed(hipStream_t st, dim3 grid, const Stencil3Params &q, int epi, int y32) {
    if (epi == EPI_RESID && y32) { HELM_LAUNCH((k_stencil3_lp<EPI_RESID, OTF, cplxf>), grid, dim3(256), 0, st, q); return true; }
    if (epi == EPI_JACOBI && y32) { HELM_LAUNCH((k_stencil3_lp<EPI_JACOBI, OTF, cplxf>), grid, dim3(256), 0, st, q); return true; }
    if (epi == EPI_JACOBI && !y32) { HELM_LAUNCH((k_stencil3_lp<EPI_JACOBI, OTF, cplx>), grid, dim3(256), 0, st, q); return true; }
    return false;
}

int helm3d_launch_apply(helm_op *op, const ApplyArgs &a, hipEvent_t e0, hipEvent_t e1) {
    if (a.adjoint) HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "adjoint apply is not available for the 3-D operator");
    Stencil3Params q;
    q.planes = a.planes; q.X = a.X; q.Y = a.Y; q.W = a.W; q.ld = a.ld; q.N = op->N;
    q.nz = op->nz; q.ny = op->ny; q.nx = op->nx; q.nrhs = a.nrhs;
    q.ntx = (op->nx + T3X - 1) / T3X; q.nty = (op->ny + T3Y - 1) / T3Y; q.nblk = q.ntx * q.nty * op->nz;
    q.scal = a.scal; q.part = a.part; q.dinv = a.dinv; q.omega_j = a.omega_j;
    q.zfast = 1;
    int split = 1;
    if (q.nblk < 2048) { split = (2048 + q.nblk - 1) / q.nblk; if (split > a.nrhs) split = a.nrhs; if (split < 1) split = 1; }
    dim3 grid(q.nblk, split);
    // coefficients on the fly: this operator's own planes (not a caller's override), unscaled, enough right-hand sides per workgroup to amortise the rebuild
    const int per_wg = (a.nrhs + split - 1) / split;
    // mg3_otf: 0 stored planes; 1 on the fly where it pays (from 4 right-hand sides per workgroup up); 2 wherever it is possible (tests)
    const int otf_mode = helm_tuning_now().mg3_otf;
    const bool otf = op->otf3 && a.planes == op->d_C && !a.scaled && (otf_mode == 2 || (otf_mode == 1 && per_wg >= 4));
    q.K3 = op->d_K3; q.b3 = op->d_b3; q.Lx = op->d_L3; q.Ly = op->d_L3 ? op->d_L3 + 3 * (size_t)op->nx : nullptr;
    q.Lz = op->d_L3 ? op->d_L3 + 3 * (size_t)op->nx + 3 * (size_t)op->ny : nullptr;
    q.idx2 = op->otf_idx2; q.idy2 = op->otf_idy2; q.idz2 = op->otf_idz2; q.blend = op->otf_blend;
    q.npart = helm3d_apply_num_blocks(op);
    if (e0) hipEventRecord(e0, op->stream);
    if (a.x32) {
        if (a.scaled || a.w32 || !(otf ? launch3_mixed<true>(op->stream, grid, q, a.epi, a.y32) : launch3_mixed<false>(op->stream, grid, q, a.epi, a.y32)))
            HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "3-D apply: this combination of vector precisions has no kernel (X complex64 with W complex128: residual and Jacobi sweep only)");
    }
    else if (otf) launch3_otf(op->stream, grid, q, a.epi);
    else if (a.scaled) launch3_epi<true>(op->stream, grid, q, a.epi);
    else launch3_epi<false>(op->stream, grid, q, a.epi);
    if (e1) hipEventRecord(e1, op->stream);
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}
