// Multigrid preconditioner for the 3-D 27-point operator: one V(1,1) cycle on the complex-shifted operator
// (1/tau_M = 1/tau + omega beta / 2) with a weak absorbing layer (cPML_M), rediscretised on coarser grids (model by
// injection, spacing doubled, layer thickness halved), damped Jacobi smoothing through the 3-D stencil kernel, full-weighting
// restriction / trilinear prolongation, dense inverse on the coarsest grid (dense kernels of direct.hip).
//
// Why this recipe: numpy experiments on the oracle's 3-D matrices -- unlike in 2-D, the mismatch
// between the weak layer of the preconditioner and the true layer of the operator costs little in 3-D, so no line
// relaxation is needed; the cycle right-preconditions the same BiCGSTAB as in 2-D (capi.hip).
#include "helm_internal.hpp"
#include "direct.hpp"
#include <algorithm>
#include <complex>

struct Mg3Level {
    helm_op *op = nullptr;
    int nz = 0, ny = 0, nx = 0;
    long long N = 0;
    cplx *u = nullptr, *f = nullptr, *r = nullptr, *t = nullptr;      // [batch][N]
};

struct Mg3Precond {
    std::vector<Mg3Level> lv;
    cplx *cinvT = nullptr;        // transposed dense inverse of the coarsest operator
    int nc = 0, batch = 0;
    double omega_j = 0.8, beta = 0.6, cpml_m = 30.0;
    int nu1 = 1, nu2 = 1, min_n = 8;
};

namespace {

double envd(const char *n, double d) { const char *v = getenv(n); return v ? atof(v) : d; }
int envi(const char *n, int d) { const char *v = getenv(n); return v ? atoi(v) : d; }

__global__ void k3_jac0(const cplx *__restrict__ f, const cplx *__restrict__ dinv, cplx *__restrict__ u, long long N, double w) {
    const cplx *fb = f + (long long)blockIdx.y * N; cplx *ub = u + (long long)blockIdx.y * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
        ub[i] = cmul(cscale(dinv[i], w), fb[i]);
}

// coarse = R fine, R = P^T / 8 (full weighting; weights 1, 1/2, 1/4, 1/8 by distance class, fine points outside the grid skipped)
__global__ void k3_restrict(const cplx *__restrict__ fine, cplx *__restrict__ coarse, int nz, int ny, int nx, int nzc, int nyc, int nxc) {
    const long long Nf = (long long)nz * ny * nx, Nc = (long long)nzc * nyc * nxc;
    const cplx *fb = fine + (long long)blockIdx.y * Nf; cplx *cb = coarse + (long long)blockIdx.y * Nc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < Nc; i += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(i % nxc), Y = (int)((i / nxc) % nyc), Z = (int)(i / ((long long)nxc * nyc));
        cplx acc = cmake(0.0, 0.0);
        for (int dz = -1; dz <= 1; ++dz) { const int z = 2 * Z + dz; if (z < 0 || z >= nz) continue;
            for (int dy = -1; dy <= 1; ++dy) { const int y = 2 * Y + dy; if (y < 0 || y >= ny) continue;
                for (int dx = -1; dx <= 1; ++dx) { const int x = 2 * X + dx; if (x < 0 || x >= nx) continue;
                    const double w = (dz ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) * (dx ? 0.5 : 1.0) * 0.125;
                    const cplx v = fb[((long long)z * ny + y) * nx + x];
                    acc.x += w * v.x; acc.y += w * v.y;
                } } }
        cb[i] = acc;
    }
}

// fine += P coarse (trilinear)
__global__ void k3_prolong_add(const cplx *__restrict__ coarse, cplx *__restrict__ fine, int nz, int ny, int nx, int nzc, int nyc, int nxc) {
    const long long Nf = (long long)nz * ny * nx, Nc = (long long)nzc * nyc * nxc;
    cplx *fb = fine + (long long)blockIdx.y * Nf; const cplx *cb = coarse + (long long)blockIdx.y * Nc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < Nf; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((long long)nx * ny));
        const int X = x >> 1, Y = y >> 1, Z = z >> 1;
        const int ox = x & 1, oy = y & 1, oz = z & 1;
        cplx acc = cmake(0.0, 0.0);
        for (int a = 0; a <= oz; ++a) { const int ZZ = Z + a; if (ZZ >= nzc) continue;
            for (int b = 0; b <= oy; ++b) { const int YY = Y + b; if (YY >= nyc) continue;
                for (int c = 0; c <= ox; ++c) { const int XX = X + c; if (XX >= nxc) continue;
                    const double w = (oz ? 0.5 : 1.0) * (oy ? 0.5 : 1.0) * (ox ? 0.5 : 1.0);
                    const cplx v = cb[((long long)ZZ * nyc + YY) * nxc + XX];
                    acc.x += w * v.x; acc.y += w * v.y;
                } } }
        fb[i] = cadd(fb[i], acc);
    }
}

// dense row-major matrix of the 27-plane operator (coarsest level)
__global__ void k3_dense(const cplx *__restrict__ planes, cplx *__restrict__ A, int nz, int ny, int nx) {
    const long long N = (long long)nz * ny * nx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((long long)nx * ny));
        for (int k = 0; k < 27; ++k) {
            const int z2 = z + k / 9 - 1, y2 = y + (k / 3) % 3 - 1, x2 = x + k % 3 - 1;
            if (z2 < 0 || z2 >= nz || y2 < 0 || y2 >= ny || x2 < 0 || x2 >= nx) continue;
            A[i * N + ((long long)z2 * ny + y2) * nx + x2] = planes[(long long)k * N + i];
        }
    }
}

__global__ void k3_transpose_sq(const cplx *__restrict__ A, cplx *__restrict__ T, int n) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * blockDim.x)
        T[(e % n) * n + e / n] = A[e];
}

inline dim3 vgrid(long long N, int nrhs) { return dim3((unsigned)std::min<long long>((N + 255) / 256, 16384), nrhs); }

template <typename T> std::vector<T> inject3(const std::vector<T> &a, int nz, int ny, int nx) {
    const int nzc = (nz + 1) / 2, nyc = (ny + 1) / 2, nxc = (nx + 1) / 2;
    std::vector<T> out((size_t)nzc * nyc * nxc);
    for (int Z = 0; Z < nzc; ++Z) for (int Y = 0; Y < nyc; ++Y) for (int X = 0; X < nxc; ++X)
        out[((size_t)Z * nyc + Y) * nxc + X] = a[((size_t)(2 * Z) * ny + 2 * Y) * nx + 2 * X];
    return out;
}

int level_apply(helm_op *top, Mg3Level &L, const cplx *x, cplx *y, const cplx *w, int nrhs, int epi, double omega_j) {
    ApplyArgs a;
    a.planes = L.op->d_C; a.X = x; a.Y = y; a.W = w; a.ld = L.N; a.nrhs = nrhs; a.epi = epi; a.scaled = 0; a.adjoint = 0;
    a.scal = nullptr; a.part = (double *)top->d_part; a.dinv = L.op->d_dinv; a.omega_j = omega_j; a.profile = 0;
    return helm_launch_apply(L.op, a);
}

// final_out (level 0 only): where the last post-smoothing sweep writes the result -- the caller's output vector, no copy
int cycle(helm_op *op, Mg3Precond *P, size_t l, int nrhs, cplx *final_out = nullptr) {
    Mg3Level &L = P->lv[l];
    hipStream_t st = op->stream;
    if (l + 1 == P->lv.size()) {        // coarsest: u = Cinv f, stored as U = F Cinv^T
        return nd_dense_gemm(op, nrhs, P->nc, P->nc, cmake(1, 0), L.f, P->nc, P->cinvT, P->nc, cmake(0, 0), L.u, P->nc);
    }
    Mg3Level &C = P->lv[l + 1];
    hipLaunchKernelGGL(k3_jac0, vgrid(L.N, nrhs), dim3(256), 0, st, L.f, L.op->d_dinv, L.u, L.N, P->omega_j);
    int rc;
    for (int s = 1; s < P->nu1; ++s) {
        rc = level_apply(op, L, L.u, L.t, L.f, nrhs, EPI_JACOBI, P->omega_j); if (rc) return rc;
        std::swap(L.u, L.t);
    }
    rc = level_apply(op, L, L.u, L.r, L.f, nrhs, EPI_RESID, 0.0); if (rc) return rc;
    hipLaunchKernelGGL(k3_restrict, vgrid(C.N, nrhs), dim3(256), 0, st, L.r, C.f, L.nz, L.ny, L.nx, C.nz, C.ny, C.nx);
    rc = cycle(op, P, l + 1, nrhs); if (rc) return rc;
    hipLaunchKernelGGL(k3_prolong_add, vgrid(L.N, nrhs), dim3(256), 0, st, C.u, L.u, L.nz, L.ny, L.nx, C.nz, C.ny, C.nx);
    for (int s = 0; s < P->nu2; ++s) {
        if (final_out && s == P->nu2 - 1) return level_apply(op, L, L.u, final_out, L.f, nrhs, EPI_JACOBI, P->omega_j);
        rc = level_apply(op, L, L.u, L.t, L.f, nrhs, EPI_JACOBI, P->omega_j); if (rc) return rc;
        std::swap(L.u, L.t);
    }
    if (final_out) HIP_TRY(op, hipMemcpyAsync(final_out, L.u, (size_t)nrhs * L.N * sizeof(cplx), hipMemcpyDeviceToDevice, st));
    return HELM_OK;
}

}  // namespace

void mg3_destroy(helm_op *op) {
    Mg3Precond *P = op->mg3;
    if (!P) return;
    for (Mg3Level &L : P->lv) {
        hipFree(L.u); hipFree(L.f); hipFree(L.r); hipFree(L.t);
        if (L.op) { L.op->own_stream = false; L.op->stream = nullptr; helm_destroy(L.op); }
    }
    hipFree(P->cinvT);
    delete P;
    op->mg3 = nullptr;
}

int mg3_setup(helm_op *op, int batch) {
    if (op->mg3 && op->mg3->batch >= batch) return HELM_OK;
    if (op->mg3) mg3_destroy(op);
    int rc = helm_ensure_host_model(op);
    if (rc) return rc;
    Mg3Precond *P = new Mg3Precond();
    op->mg3 = P;
    P->batch = batch;
    // Jacobi damping: measured at 256 x 256 x 128, 4 sources (tools/sweep3d.sh): 0.8 / 0.9 / 1.0 / 1.1 -> 10.9 / 9.6 / 10.0 / 16.4 s at 3 Hz and
    // 7.4 / 7.0 / 6.6 / 8.9 s at 5 Hz
    P->omega_j = envd("HELM_MG3_OMEGA", 0.9);
    P->nu1 = envi("HELM_MG3_NU1", 1); P->nu2 = envi("HELM_MG3_NU2", 1); P->min_n = envi("HELM_MG3_MIN_N", 8);
    const double omega = 2.0 * M_PI * std::abs(std::complex<double>(op->a_freq_re, op->a_freq_im));
    // shift: 0.6 at 10 grid points per wavelength, growing with the square of the oversampling up to 8 -- measured at
    // 256 x 256 x 128, 40-100 points per wavelength: beta 0.6 / 3 / 6 / 12 -> 26 / 16 / 14 / 14 s per 4 sources at 3 Hz
    double cmin = 1e300;
    for (const cplx &cv : op->h_c) cmin = std::min(cmin, cv.x);
    const double hmax = std::max(op->dx, std::max(op->dy, op->dz));
    const double ppw = omega > 0 ? cmin / (omega / (2.0 * M_PI) * hmax) : 10.0;
    const double over = std::max(1.0, ppw / 10.0);
    P->beta = envd("HELM_MG3_BETA", std::min(8.0, 0.6 * over * over));
    double inv_tau = omega * P->beta / 2.0;
    if (std::isfinite(op->a_tau) && op->a_tau != 0.0) inv_tau += 1.0 / op->a_tau;
    const double tauM = 1.0 / inv_tau;
    // layer of the preconditioner: gamma / omega = 2 (measured at 256 x 256 x 128 with the shift above: 0.2 omega -> 26 / 14 / 8.8 s
    // per 4 sources at 2 / 3 / 5 Hz, 2 omega -> 22 / 10.5 / 7.4 s, 5 omega worse again, the true layer (300) does not converge;
    // with the small shift beta = 0.6 only gamma / omega <= 0.4 was stable)
    P->cpml_m = envd("HELM_MG3_CPML", 2.0 * omega);
    const double cpml = std::min(P->cpml_m, op->a_cpml > 0 ? op->a_cpml : P->cpml_m);
    std::vector<cplx> c = op->h_c;
    std::vector<double> rho = op->h_rho;
    int nz = op->nz, ny = op->ny, nx = op->nx, npml = op->nPML;
    double dx = op->dx, dy = op->dy, dz = op->dz;
    auto fail = [&](int code, const char *msg) { helm_set_error(op, msg); mg3_destroy(op); return code; };
    while (true) {
        Mg3Level L;
        L.nz = nz; L.ny = ny; L.nx = nx; L.N = (long long)nz * ny * nx;
        L.op = helm_create3d(op->device, nz, ny, nx, dx, dy, dz, npml);
        if (!L.op) return fail(HELM_ERR_DEVICE, helm_last_error(nullptr));
        P->lv.push_back(L);
        Mg3Level &Lr = P->lv.back();
        if (helm_set_stream(Lr.op, op->stream)) return fail(HELM_ERR_DEVICE, "3-D multigrid: cannot share the stream");
        rc = helm_set_model(Lr.op, (const double *)c.data(), rho.data(), nullptr, nullptr, nullptr);
        if (!rc) rc = helm_assemble(Lr.op, op->a_freq_re, op->a_freq_im, tauM, 0.0, cpml);
        if (!rc) rc = helm_ensure_scaled(Lr.op);
        if (rc) return fail(rc, helm_last_error(Lr.op));
        const size_t vb = (size_t)batch * Lr.N * sizeof(cplx);
        if (hipMalloc((void **)&Lr.u, vb) != hipSuccess || hipMalloc((void **)&Lr.f, vb) != hipSuccess ||
            hipMalloc((void **)&Lr.r, vb) != hipSuccess || hipMalloc((void **)&Lr.t, vb) != hipSuccess)
            return fail(HELM_ERR_DEVICE, "3-D multigrid: level vectors do not fit");
        const int nzc = (nz + 1) / 2, nyc = (ny + 1) / 2, nxc = (nx + 1) / 2;
        const int npmlc = std::max((npml - 1) / 2 + 1, 2);
        const long long Nc = (long long)nzc * nyc * nxc;
        if (std::min(nz, std::min(ny, nx)) <= P->min_n || std::min(nzc, std::min(nyc, nxc)) < 2 * npmlc + 2 || Lr.N <= 4096 || Nc < 27) break;
        c = inject3(c, nz, ny, nx); rho = inject3(rho, nz, ny, nx);
        nz = nzc; ny = nyc; nx = nxc; dx *= 2; dy *= 2; dz *= 2; npml = npmlc;
    }
    // coarsest level: dense inverse
    Mg3Level &Lc = P->lv.back();
    if (Lc.N > 8192) return fail(HELM_ERR_UNSUPPORTED, "3-D multigrid: coarsest grid too large for a dense inverse");
    P->nc = (int)Lc.N;
    cplx *A = nullptr, *W = nullptr;
    const size_t mb = (size_t)P->nc * P->nc * sizeof(cplx);
    if (hipMalloc((void **)&A, mb) != hipSuccess || hipMalloc((void **)&W, mb) != hipSuccess || hipMalloc((void **)&P->cinvT, mb) != hipSuccess) {
        hipFree(A); hipFree(W); return fail(HELM_ERR_DEVICE, "3-D multigrid: coarsest inverse does not fit");
    }
    hipMemsetAsync(A, 0, mb, op->stream);
    hipLaunchKernelGGL(k3_dense, dim3((unsigned)((Lc.N + 255) / 256)), dim3(256), 0, op->stream, (const cplx *)Lc.op->d_C, A, Lc.nz, Lc.ny, Lc.nx);
    rc = nd_dense_inverse(op, A, P->nc, W);
    if (!rc) hipLaunchKernelGGL(k3_transpose_sq, dim3(4096), dim3(256), 0, op->stream, (const cplx *)A, P->cinvT, P->nc);
    hipStreamSynchronize(op->stream);
    hipFree(A); hipFree(W);
    if (rc) return fail(rc, "3-D multigrid: coarsest inverse failed");
    return HELM_OK;
}

// out[b] = M^-1 in[b]
int mg3_apply(helm_op *op, const cplx *in, cplx *out, int nrhs) {
    Mg3Precond *P = op->mg3;
    if (!P || nrhs > P->batch) HELM_FAIL(op, HELM_ERR_STATE, "3-D multigrid preconditioner not set up");
    Mg3Level &L0 = P->lv[0];
    // the finest level reads its right-hand side in place and its last sweep writes straight into `out` (these two device copies
    // of batch x 128 MB were 9 % of the GPU time of a config-5 solve)
    cplx *own_f = L0.f;
    if (P->lv.size() > 1) L0.f = const_cast<cplx *>(in);
    else HIP_TRY(op, hipMemcpyAsync(L0.f, in, (size_t)nrhs * L0.N * sizeof(cplx), hipMemcpyDeviceToDevice, op->stream));
    const int rc = cycle(op, P, 0, nrhs, P->lv.size() > 1 ? out : nullptr);
    L0.f = own_f;
    if (rc) return rc;
    if (P->lv.size() == 1) HIP_TRY(op, hipMemcpyAsync(out, L0.u, (size_t)nrhs * L0.N * sizeof(cplx), hipMemcpyDeviceToDevice, op->stream));
    return HELM_OK;
}
