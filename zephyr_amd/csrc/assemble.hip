// Coefficient assembly on the GPU: model (c, rho[, theta, eps, delta]) -> 9 (MiniZephyr) or
// 4 x 9 (Eurus) complex planes resident in HBM, bug-compatible with the reference.
//
//   MiniZephyr: zephyr/backend/minizephyr.py:40-298   (OMEGA-style 9-point star, Roecker PML)
//   Eurus:      zephyr/backend/eurus.py:28-485        (Operto 2009 mixed-grid TTI, C-PML)
//
// One thread per grid point; everything is elementwise on (c, rho) with clamped ("edge"
// padded) neighbour reads, so the kernel is a coalesced streaming pass: it runs once per
// frequency and is not on the per-iteration path.
#include "helm_internal.hpp"

namespace {

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__host__ __device__ inline int slot(int dz, int dx) { return 3 * (dz + 1) + (dx + 1); }

struct MzParams {
    int nz, nx;
    double dx, dz, aky;
    cplx om;          // damped angular frequency
    double fx, fz;    // PML factors 3 ln(1/1e-3) / (2 L^3)
    int fs0, fs1, fs2, fs3;
};

// MiniZephyr: minizephyr.py:98-133 (PML), :169-202 (buoyancy / K), :219-243 (star), :269-298 (edges)
__global__ __launch_bounds__(256) void k_assemble_mz(MzParams P, const cplx *__restrict__ c,
                                                     const double *__restrict__ rho,
                                                     const double *__restrict__ distx, const double *__restrict__ sgnx,
                                                     const double *__restrict__ distz, const double *__restrict__ sgnz,
                                                     cplx *__restrict__ C) {
    const long long N = (long long)P.nz * P.nx;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int iz = (int)(i / P.nx), ix = (int)(i % P.nx);

    const double ac = 0.5461, bc = 0.4539, cc = 0.6248, dc = 0.09381, ec = 0.000001297;
    const double dxx = P.dx * P.dx, dzz = P.dz * P.dz, dxz = (dxx + dzz) / 2.0, dd = sqrt(dxz);
    const cplx iom = cmake(-P.om.y, P.om.x);          // 1j * om
    const cplx om2 = cmul(P.om, P.om);
    const double aky2 = P.aky * P.aky;

    const cplx c0 = c[i];
    // PML stretch factors from the local (unpadded) velocity
    const double dX = distx[ix], dZ = distz[iz];
    cplx denx = cadd(cscale(cscale(c0, P.fx), dX * dX), iom);
    cplx r1x = cdiv(iom, denx);
    cplx X = cmul(r1x, r1x);
    cplx px = cdiv(cmul(cscale(X, sgnx[ix]), cscale(cscale(c0, 2.0 * P.fx), dX)), denx);
    cplx denz = cadd(cscale(cscale(c0, P.fz), dZ * dZ), iom);
    cplx r1z = cdiv(iom, denz);
    cplx Z = cmul(r1z, r1z);
    cplx pz = cdiv(cmul(cscale(Z, sgnz[iz]), cscale(cscale(c0, 2.0 * P.fz), dZ)), denz);

    // neighbour properties with edge padding
    double bn[9];
    cplx Kn[9];
    const double b0 = 1.0 / rho[i];
#pragma unroll
    for (int sz = -1; sz <= 1; ++sz)
#pragma unroll
        for (int sx = -1; sx <= 1; ++sx) {
            const int jz = clampi(iz + sz, 0, P.nz - 1), jx = clampi(ix + sx, 0, P.nx - 1);
            const long long j = (long long)jz * P.nx + jx;
            const double r = rho[j];
            const cplx cj = c[j];
            bn[slot(sz, sx)] = (b0 + 1.0 / r) / 2.0;
            cplx k = cdiv(om2, cmul(cj, cj));
            k.x -= aky2;
            Kn[slot(sz, sx)] = cmake(k.x / r, k.y / r);
        }

    cplx out[9];
    const cplx ZpX = cadd(Z, X), ZmX = csub(Z, X), XmZ = csub(X, Z);
    // corners
#pragma unroll
    for (int sz = -1; sz <= 1; sz += 2)
#pragma unroll
        for (int sx = -1; sx <= 1; sx += 2) {
            const int k = slot(sz, sx);
            cplx t = cadd(cscale(ZpX, 1.0 / (4 * dxz)),
                          cscale(cadd(cscale(pz, (double)sz), cscale(px, (double)sx)), 1.0 / (4 * dd)));
            out[k] = cadd(cscale(Kn[k], ec), cscale(t, bc * bn[k]));
        }
    // vertical neighbours
#pragma unroll
    for (int sz = -1; sz <= 1; sz += 2) {
        const int k = slot(sz, 0);
        cplx t1 = cscale(cadd(cscale(Z, 1.0 / P.dz), cscale(pz, sz / 2.0)), ac * bn[k] / P.dz);
        cplx t2 = cscale(ZmX, bc * (bn[slot(sz, 1)] + bn[slot(sz, -1)]) / (4 * dxz));
        out[k] = cadd(cscale(Kn[k], dc), cadd(t1, t2));
    }
    // horizontal neighbours
#pragma unroll
    for (int sx = -1; sx <= 1; sx += 2) {
        const int k = slot(0, sx);
        cplx t1 = cscale(cadd(cscale(X, 1.0 / P.dx), cscale(px, sx / 2.0)), ac * bn[k] / P.dx);
        cplx t2 = cscale(XmZ, bc * (bn[slot(1, sx)] + bn[slot(-1, sx)]) / (4 * dxz));
        out[k] = cadd(cscale(Kn[k], dc), cadd(t1, t2));
    }
    // centre
    {
        const double bW = bn[slot(0, -1)], bE = bn[slot(0, 1)], bS = bn[slot(-1, 0)], bNn = bn[slot(1, 0)];
        const double bSW = bn[slot(-1, -1)], bNE = bn[slot(1, 1)], bSE = bn[slot(-1, 1)], bNW = bn[slot(1, -1)];
        cplx a1 = cscale(px, (bW - bE) / (2 * P.dx));
        cplx a2 = cscale(pz, (bS - bNn) / (2 * P.dz));
        cplx a3 = cscale(X, (bW + bE) / dxx);
        cplx a4 = cscale(Z, (bS + bNn) / dzz);
        cplx aterm = csub(csub(cadd(a1, a2), a3), a4);
        cplx b1 = cscale(cadd(cscale(cadd(px, pz), (bSW - bNE)), cscale(csub(pz, px), (bSE - bNW))), 1.0 / (4 * dd));
        cplx b2 = cscale(cadd(X, Z), (bSW + bNE + bNW + bSE) / (4 * dxz));
        cplx bterm = csub(b1, b2);
        out[4] = cadd(cscale(Kn[4], cc), cadd(cscale(aterm, ac), cscale(bterm, bc)));
    }

    // boundary rows: +/- identity, write order left, right, iz=0, iz=nz-1 (later wins)
    int edge = -1;
    if (ix == 0) edge = 3;
    if (ix == P.nx - 1) edge = 1;
    if (iz == 0) edge = 0;
    if (iz == P.nz - 1) edge = 2;
    if (edge >= 0) {
        const int f = edge == 0 ? P.fs0 : edge == 1 ? P.fs1 : edge == 2 ? P.fs2 : P.fs3;
#pragma unroll
        for (int k = 0; k < 9; ++k) out[k] = cmake(0.0, 0.0);
        out[4] = cmake(f ? -1.0 : 1.0, 0.0);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) C[(long long)k * N + i] = out[k];
}

struct EuParams {
    int nz, nx;
    double dx, dz;
    cplx om;
    int aniso;
    int nblk_out;     // 4, or 1 to build only M1
};

// Eurus: eurus.py:140-168 (PML averages), :170-226 (buoyancy squares/lines), :229-269 (mass),
// :279-295 (TTI fields), :300-427 (the nine entries), :479-485 (edges), :117-127 (z-flipped slots)
__global__ __launch_bounds__(256) void k_assemble_eurus(EuParams P, const cplx *__restrict__ c,
                                                        const double *__restrict__ rho,
                                                        const double *__restrict__ theta, const double *__restrict__ eps,
                                                        const double *__restrict__ delta,
                                                        const cplx *__restrict__ xix, const cplx *__restrict__ xiz,
                                                        cplx *__restrict__ C) {
    const long long N = (long long)P.nz * P.nx;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int iz = (int)(i / P.nx), ix = (int)(i % P.nx);
    const double dxx = P.dx * P.dx, dzz = P.dz * P.dz;

    // padded 1-D PML profiles: index p = i+1
    const cplx xl = xix[ix], xc = xix[ix + 1], xr = xix[ix + 2];
    const cplx zl = xiz[iz], zc = xiz[iz + 1], zr = xiz[iz + 2];
    const cplx xM = cscale(cadd(xl, xc), 0.5), xP = cscale(cadd(xc, xr), 0.5);
    const cplx zM = cscale(cadd(zl, zc), 0.5), zP = cscale(cadd(zc, zr), 0.5);
    const cplx Lx4 = crecip(cscale(xc, 4 * dxx)), Lx = crecip(cscale(xc, dxx));
    const cplx Lz4 = crecip(cscale(zc, 4 * dzz)), Lz = crecip(cscale(zc, dzz));

    double b[9];
    cplx K[9];
    const cplx om2 = cmul(P.om, P.om);
#pragma unroll
    for (int sz = -1; sz <= 1; ++sz)
#pragma unroll
        for (int sx = -1; sx <= 1; ++sx) {
            const int jz = clampi(iz + sz, 0, P.nz - 1), jx = clampi(ix + sx, 0, P.nx - 1);
            const long long j = (long long)jz * P.nx + jx;
            const double r = rho[j];
            const cplx cj = c[j];
            b[slot(sz, sx)] = 1.0 / r;
            K[slot(sz, sx)] = cdiv(om2, cscale(cmul(cj, cj), r));
        }
    // property-space names: m = index-1, p = index+1; first letter z, second x
    const double bmm = b[0], bm0 = b[1], bmp = b[2], b0m = b[3], b00 = b[4], b0p = b[5], bpm = b[6], bp0 = b[7], bpp = b[8];
    const double sq1 = (bpm + bp0 + b0m + b00) / 4, sq2 = (bp0 + bpp + b00 + b0p) / 4;
    const double sq3 = (b0m + b00 + bmm + bm0) / 4, sq4 = (b00 + b0p + bm0 + bmp) / 4;
    const cplx rxM = crecip(xM), rxP = crecip(xP), rzM = crecip(zM), rzP = crecip(zP), rxC = crecip(xc), rzC = crecip(zc);
    // NB: the reference divides (b / xi); multiply by the reciprocal differs by <= 1 ulp-level rounding
    const cplx s1x = cdiv(cmake(sq1, 0), xM), s2x = cdiv(cmake(sq2, 0), xP), s3x = cdiv(cmake(sq3, 0), xM), s4x = cdiv(cmake(sq4, 0), xP);
    const cplx s1z = cdiv(cmake(sq1, 0), zM), s2z = cdiv(cmake(sq2, 0), zM), s3z = cdiv(cmake(sq3, 0), zP), s4z = cdiv(cmake(sq4, 0), zP);
    (void)rxM; (void)rxP; (void)rzM; (void)rzP; (void)rxC; (void)rzC;
    const cplx ln1 = cdiv(cmake((bp0 + b00) / 2, 0), zM), ln2 = cdiv(cmake((b0m + b00) / 2, 0), xM);
    const cplx ln3 = cdiv(cmake((b00 + b0p) / 2, 0), xP), ln4 = cdiv(cmake((b00 + bm0) / 2, 0), zP);
    const cplx ln1c = cdiv(cmake((bp0 + b00) / 2, 0), xc), ln2c = cdiv(cmake((b0m + b00) / 2, 0), zc);
    const cplx ln3c = cdiv(cmake((b00 + b0p) / 2, 0), zc), ln4c = cdiv(cmake((b00 + bm0) / 2, 0), xc);

    const double wm1 = 0.6287326, wm2r = 0.3712667;
    const double wm2 = 0.25 * wm2r, wm3 = 0.25 * (1.0 - wm1 - wm2r), w1 = 0.4382634, w1c = 1.0 - 0.4382634;
    const cplx Kmm = cscale(K[0], wm3), Km0 = cscale(K[1], wm2), Kmp = cscale(K[2], wm3);
    const cplx K0m = cscale(K[3], wm2), K00 = cscale(K[4], wm1), K0p = cscale(K[5], wm2);
    const cplx Kpm = cscale(K[6], wm3), Kp0 = cscale(K[7], wm2), Kpp = cscale(K[8], wm3);

    double th = 0, ep = 0, de = 0;
    if (P.aniso) { th = theta ? theta[i] : 0.0; ep = eps ? eps[i] : 0.0; de = delta ? delta[i] : 0.0; }
    const double ct = cos(th), st = sin(th), s2t = sin(2.0 * th);
    const double ct2 = ct * ct, st2 = st * st;
    const double Ax = 1.0 + (2.0 * de) * ct2, Bx = (-1.0 * de) * s2t, Cx = (1.0 + (2.0 * de)) * ct2;
    const double Dx = (-0.5 * (1.0 + (2.0 * de))) * s2t, Ex = (2.0 * (ep - de)) * ct2, Fx = (-1.0 * (ep - de)) * s2t;
    const double Bz = 1.0 + (2.0 * de) * st2, Dz = (1.0 + (2.0 * de)) * st2, Fz = (2.0 * (ep - de)) * st2;
    const double massv[4] = {1.0, 0.0, 0.0, 1.0};
    const double c1xv[4] = {Ax, Cx, Ex, Ex}, c1zv[4] = {Bx, Dx, Fx, Fx};
    const double c2xv[4] = {Bx, Dx, Fx, Fx}, c2zv[4] = {Bz, Dz, Fz, Fz};

    const bool edge = (ix == 0) || (ix == P.nx - 1) || (iz == 0) || (iz == P.nz - 1);

#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m >= P.nblk_out) break;
        const double mass = massv[m], c1x = c1xv[m], c1z = c1zv[m], c2x = c2xv[m], c2z = c2zv[m];
        const cplx ax = cscale(Lx4, c1x), bx = cscale(Lx4, c2x), az = cscale(Lz4, c1z), bz = cscale(Lz4, c2z);
        const cplx axf = cscale(Lx, c1x), bzf = cscale(Lz, c2z);
        cplx GG, HH, II, DD, EE, FF, AA, BB, CC, t, u;
        // GG
        t = csub(csub(cmul(ax, s3x), cmul(bx, s3z)), cmul(az, s3x)); t = cadd(t, cmul(bz, s3z));
        u = cneg(cadd(cmul(bx, ln2c), cmul(az, ln4c)));
        GG = cadd(cscale(Kmm, mass), cadd(cscale(t, w1), cscale(u, w1c)));
        // HH
        t = cadd(cadd(cmul(ax, cneg(cadd(s3x, s4x))), cmul(bx, csub(s4z, s3z))),
                 cadd(cmul(az, csub(s3x, s4x)), cmul(bz, cadd(s3z, s4z))));
        u = cadd(cmul(bx, csub(ln3c, ln2c)), cmul(bzf, ln4));
        HH = cadd(cscale(Km0, mass), cadd(cscale(t, w1), cscale(u, w1c)));
        // II
        t = cadd(cadd(cmul(ax, s4x), cmul(bx, s4z)), cadd(cmul(az, s4x), cmul(bz, s4z)));
        u = cadd(cmul(bx, ln3c), cmul(az, ln4c));
        II = cadd(cscale(Kmp, mass), cadd(cscale(t, w1), cscale(u, w1c)));
        // DD
        t = cadd(cadd(cmul(ax, cadd(s3x, s1x)), cmul(bx, csub(s3z, s1z))),
                 cadd(cmul(az, csub(s1x, s3x)), cmul(bz, cneg(cadd(s3z, s1z)))));
        u = cadd(cmul(axf, ln2), cmul(az, csub(ln1c, ln4c)));
        DD = cadd(cscale(K0m, mass), cadd(cscale(t, w1), cscale(u, w1c)));
        // EE
        t = cadd(cadd(cmul(cneg(ax), cadd(cadd(s1x, s2x), cadd(s3x, s4x))),
                      cmul(bx, csub(cadd(s2z, s3z), cadd(s1z, s4z)))),
                 cadd(cmul(az, csub(cadd(s2x, s3x), cadd(s1x, s4x))),
                      cmul(cneg(bz), cadd(cadd(s1z, s2z), cadd(s3z, s4z)))));
        u = cadd(cmul(axf, cneg(cadd(ln2, ln3))), cmul(bzf, cneg(cadd(ln1, ln4))));
        EE = cadd(cscale(K00, mass), cadd(cscale(t, w1), cscale(u, w1c)));
        // FF
        t = cadd(cadd(cmul(ax, cadd(s2x, s4x)), cmul(bx, csub(s2z, s4z))),
                 cadd(cmul(az, csub(s4x, s2x)), cmul(bz, cneg(cadd(s2z, s4z)))));
        u = cadd(cmul(axf, ln3), cmul(az, csub(ln4c, ln1c)));
        FF = cadd(cscale(K0p, mass), cadd(cscale(t, w1), cscale(u, w1c)));
        // AA
        t = cadd(cadd(cmul(ax, s1x), cmul(bx, s1z)), cadd(cmul(az, s1x), cmul(bz, s1z)));
        u = cadd(cmul(bx, ln2c), cmul(az, ln1c));
        AA = cadd(cscale(Kpm, mass), cadd(cscale(t, w1), cscale(u, w1c)));
        // BB
        t = cadd(cadd(cmul(ax, cneg(cadd(s2x, s1x))), cmul(bx, csub(s1z, s2z))),
                 cadd(cmul(az, csub(s2x, s1x)), cmul(bz, cadd(s2z, s1z))));
        u = cadd(cmul(bx, csub(ln2c, ln3c)), cmul(bzf, ln1));
        BB = cadd(cscale(Kp0, mass), cadd(cscale(t, w1), cscale(u, w1c)));
        // CC
        t = csub(csub(cmul(ax, s2x), cmul(bx, s2z)), cmul(az, s2x)); t = cadd(t, cmul(bz, s2z));
        u = cneg(cadd(cmul(bx, ln3c), cmul(az, ln1c)));
        CC = cadd(cscale(Kpp, mass), cadd(cscale(t, w1), cscale(u, w1c)));

        const cplx zero = cmake(0.0, 0.0);
        cplx *Cm = C + (long long)m * 9 * N;
        // matrix slots with the reference's mord = (-nx, +1): property row index-1 couples to u(iz+1)
        Cm[(long long)slot(1, -1) * N + i] = edge ? zero : GG;
        Cm[(long long)slot(1, 0) * N + i] = edge ? zero : HH;
        Cm[(long long)slot(1, 1) * N + i] = edge ? zero : II;
        Cm[(long long)slot(0, -1) * N + i] = edge ? zero : DD;
        Cm[(long long)slot(0, 0) * N + i] = EE;
        Cm[(long long)slot(0, 1) * N + i] = edge ? zero : FF;
        Cm[(long long)slot(-1, -1) * N + i] = edge ? zero : AA;
        Cm[(long long)slot(-1, 0) * N + i] = edge ? zero : BB;
        Cm[(long long)slot(-1, 1) * N + i] = edge ? zero : CC;
    }
}

// 1-D profile builders (host).  MiniZephyr: minizephyr.py:98-118,126-127 (assignment order kept).
void mz_profiles(int n, int npml, double h, bool fs_low, bool fs_high, std::vector<double> &dist, std::vector<double> &sgn) {
    dist.assign(n, 0.0);
    sgn.assign(n, 0.0);
    // numpy slice semantics a[:npml] / a[-npml:] clip to the array
    const int lo_len = npml < n ? npml : n;
    for (int k = 0; k < lo_len; ++k) dist[k] = (double)(npml - k) * h;
    const int hi_start = n - npml > 0 ? n - npml : 0;
    // a[-npml:] = arange(1..npml): when npml > n numpy would raise; we keep the tail-aligned values
    for (int k = hi_start; k < n; ++k) dist[k] = (double)(k - (n - npml) + 1) * h;
    if (!fs_high) for (int k = hi_start; k < n; ++k) sgn[k] = -1.0;
    if (!fs_low) for (int k = 0; k < lo_len; ++k) sgn[k] = 1.0;
}

}  // namespace

int helm_launch_assemble(helm_op *op, double freq_re, double freq_im, double tau, double ky, double cPML) {
    const int nz = op->nz, nx = op->nx;
    const long long N = op->N;
    // omega~ = 2 pi f - i / tau      (discretization.py:38-41)
    const double twopi = 2.0 * M_PI;
    cplx om = cmake(twopi * freq_re, twopi * freq_im);
    if (std::isfinite(tau) && tau != 0.0) om.y -= 1.0 / tau;
    const int threads = 256;
    const int blocks = (int)((N + threads - 1) / threads);

    if (op->variant == HELM_MINIZEPHYR) {
        if (op->nPML < 2) HELM_FAIL(op, HELM_ERR_ARG, "nPML must be >= 2 (PML length dx*(nPML-1) is a divisor)");
        if (op->nPML > nx || op->nPML > nz) HELM_FAIL(op, HELM_ERR_ARG, "nPML larger than the grid (reference raises a broadcast ValueError)");
        std::vector<double> distx, sgnx, distz, sgnz;
        mz_profiles(nx, op->nPML, op->dx, op->fs[3], op->fs[1], distx, sgnx);
        mz_profiles(nz, op->nPML, op->dz, op->fs[0], op->fs[2], distz, sgnz);
        double *d_prof = nullptr;
        const size_t bytes = sizeof(double) * (2 * (size_t)nx + 2 * (size_t)nz);
        d_prof = (double *)helm_pool_alloc(op->device, bytes);          // (pool: hipMalloc / hipFree per assembly would wait for every stream of the device)
        if (!d_prof) HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc of the PML profiles failed");
        std::vector<double> h(2 * (size_t)nx + 2 * (size_t)nz);
        std::copy(distx.begin(), distx.end(), h.begin());
        std::copy(sgnx.begin(), sgnx.end(), h.begin() + nx);
        std::copy(distz.begin(), distz.end(), h.begin() + 2 * nx);
        std::copy(sgnz.begin(), sgnz.end(), h.begin() + 2 * nx + nz);
        HIP_TRY(op, hipMemcpyAsync(d_prof, h.data(), bytes, hipMemcpyHostToDevice, op->stream));
        MzParams P;
        P.nz = nz; P.nx = nx; P.dx = op->dx; P.dz = op->dz; P.aky = twopi * ky; P.om = om;
        const double lx = op->dx * (op->nPML - 1), lz = op->dz * (op->nPML - 1);
        P.fx = op->pml_scale * 3.0 * log(1.0 / 1e-3) / (2.0 * lx * lx * lx);
        P.fz = op->pml_scale * 3.0 * log(1.0 / 1e-3) / (2.0 * lz * lz * lz);
        P.fs0 = op->fs[0]; P.fs1 = op->fs[1]; P.fs2 = op->fs[2]; P.fs3 = op->fs[3];
        HELM_LAUNCH(k_assemble_mz, dim3(blocks), dim3(threads), 0, op->stream, P, op->d_c, op->d_rho,
                           d_prof, d_prof + nx, d_prof + 2 * nx, d_prof + 2 * nx + nz, op->d_C);
        HIP_TRY(op, hipGetLastError());
        HIP_TRY(op, hipStreamSynchronize(op->stream));
        helm_pool_free(op->device, d_prof, bytes);
        op->block_zero[0] = false;
    } else {
        if (op->nPML < 2) HELM_FAIL(op, HELM_ERR_ARG, "nPML must be >= 2");
        if (op->nPML > nx || op->nPML > nz) HELM_FAIL(op, HELM_ERR_ARG, "nPML larger than the grid");
        // gamma profiles, eurus.py:77-97.  np.arange(0, L+h, h) has ceil((L+h)/h) entries; the
        // reference breaks (broadcast ValueError) when float rounding makes that != nPML.
        auto build = [&](int n, double h, std::vector<cplx> &xi) -> int {
            const double L = h * (op->nPML - 1);
            const int len = (int)ceil((L + h) / h);
            if (len != op->nPML && !op->block0_only) return HELM_ERR_PML;   // preconditioner levels are free of the hazard
            std::vector<double> g(n, 0.0);
            for (int k = 0; k < op->nPML; ++k) g[k] = cPML * cos((M_PI / 2) * ((0.0 + k * h) / L));
            for (int k = 0; k < op->nPML; ++k) g[n - op->nPML + k] = cPML * cos((M_PI / 2) * ((0.0 + (op->nPML - 1 - k) * h) / L));
            xi.resize(n + 2);
            for (int p = 0; p < n + 2; ++p) {
                const int k = p == 0 ? 0 : (p == n + 1 ? n - 1 : p - 1);
                // xi = 1 - (1j*gamma)/om
                cplx q = cdiv(cmake(0.0, g[k]), om);
                xi[p] = cmake(1.0 - q.x, -q.y);
            }
            return 0;
        };
        std::vector<cplx> xix, xiz;
        if (build(nx, op->dx, xix) || build(nz, op->dz, xiz))
            HELM_FAIL(op, HELM_ERR_PML, "np.arange(0, L+h, h) length != nPML for this dx/dz/nPML (reference raises ValueError, eurus.py:84-91)");
        cplx *d_xi = nullptr;
        const size_t bytes = sizeof(cplx) * ((size_t)nx + nz + 4);
        d_xi = (cplx *)helm_pool_alloc(op->device, bytes);
        if (!d_xi) HELM_FAIL(op, HELM_ERR_DEVICE, "hipMalloc of the PML profiles failed");
        std::vector<cplx> h(xix);
        h.insert(h.end(), xiz.begin(), xiz.end());
        HIP_TRY(op, hipMemcpyAsync(d_xi, h.data(), bytes, hipMemcpyHostToDevice, op->stream));
        EuParams P;
        P.nz = nz; P.nx = nx; P.dx = op->dx; P.dz = op->dz; P.om = om; P.aniso = op->aniso ? 1 : 0;
        P.nblk_out = op->block0_only ? 1 : (op->asm_nblk == 1 ? 1 : 4);
        HELM_LAUNCH(k_assemble_eurus, dim3(blocks), dim3(threads), 0, op->stream, P, op->d_c, op->d_rho,
                           op->d_theta, op->d_eps, op->d_delta, d_xi, d_xi + nx + 2, op->d_C);
        HIP_TRY(op, hipGetLastError());
        HIP_TRY(op, hipStreamSynchronize(op->stream));
        helm_pool_free(op->device, d_xi, bytes);
        op->blocks_ready = P.nblk_out;
    }
    return HELM_OK;
}
