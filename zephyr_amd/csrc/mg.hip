// Preconditioner for the Krylov solve: complex-shifted-Laplacian multigrid + PML strip relaxation.
//
//   M^-1 r  =  strip_relax( Vcycle(r) )
//
// * Vcycle: geometric multigrid (rediscretised coarse levels, damped-Jacobi smoothing, full
//   weighting / bilinear transfers, dense inverse on the coarsest grid) on the operator with a
//   complex shift (w~^2 -> w^2 (1 - i beta), implemented through the reference's own Laplace
//   damping `tau`, discretization.py:33-41) and a WEAK absorbing layer (small cPML), for which
//   point smoothing works.
// * strip_relax: a few line-Jacobi sweeps with the shifted operator carrying the TRUE PML, on
//   the frame of width nPML+2 around the grid (tridiagonal solves along the strongly coupled
//   direction).  The reference's default C-PML (cPML = 1e3, eurus.py:500-504) stretches the
//   outer cells by |xi| = 16..80: point-smoothed multigrid diverges on it, and a preconditioner
//   with a different layer is useless (eigenvalues ~ xi_M^2 / xi_A^2); correcting the weak-layer
//   cycle with true-operator strip solves restores the quality of the exact shifted inverse.
//
// Everything is batched over the right-hand sides of the Krylov batch and matrix-free (the same
// k_stencil kernel on every level); nothing here changes A or the result -- it only reduces the
// iteration count of the outer BiCGSTAB (tens of thousands -> hundreds).
#include "helm_internal.hpp"
#include <cstdlib>
#include <complex>
#include <algorithm>

extern "C" helm_op *helm_create(int device, int variant, int nz, int nx, double dx, double dz, int nPML, const int *freeSurf);

namespace {

struct MgLevel {
    helm_op *op = nullptr;
    cplx *u = nullptr, *f = nullptr, *r = nullptr, *t = nullptr, *f2 = nullptr, *g = nullptr;   // [batch][N_l]
};

}  // namespace

struct MgPrecond {
    std::vector<MgLevel> lv;
    cplx *d_cinvT = nullptr; int nc = 0;        // dense inverse of the coarsest operator, transposed
    helm_op *sop = nullptr;                     // fine grid, true PML, shifted: strip relaxation operator
    int W = 12;
    int *d_tiles = nullptr; int ntiles = 0;
    cplx *zl_m = nullptr, *zl_c = nullptr, *zl_a = nullptr;      // z-line factors [2W][nz]
    cplx *xl_m = nullptr, *xl_c = nullptr, *xl_a = nullptr;      // x-line factors [2W][nx-2W]
    cplx *strip_r = nullptr;                    // [batch][N]
    int batch = 0;
    double omega_j = 0.8, beta = 0.5, cpml_m = 30.0, wstrip = 1.0;
    int nu1 = 1, nu2 = 1, sweeps = 4, min_n = 16;
    int fdepth = 0;          // levels < fdepth revisit their coarse level a second time (truncated F-cycle); 0 = V-cycle
};

namespace {

double env_double(const char *name, double dflt) { const char *v = getenv(name); return v ? atof(v) : dflt; }
int env_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }

__device__ inline bool active(const RhsScal *scal, int b) { return scal == nullptr || scal[b].status == ST_ACTIVE; }

// u = omega_j * dinv * f      (first smoothing sweep from a zero initial guess)
__global__ __launch_bounds__(256) void k_jac0(const cplx *__restrict__ dinv, const cplx *__restrict__ f, cplx *__restrict__ u,
                                              long long N, double omega_j, const RhsScal *scal) {
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
        u[(long long)b * N + i] = cmul(cscale(dinv[i], omega_j), f[(long long)b * N + i]);
}

// u += e
__global__ __launch_bounds__(256) void k_axpy1(const cplx *__restrict__ e, cplx *__restrict__ u, long long N, const RhsScal *scal) {
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const long long g = (long long)b * N + i;
        u[g] = cadd(u[g], e[g]);
    }
}

// full weighting: fc[I,J] = (1/16) sum_{di,dj} w(di) w(dj) r[2I+di, 2J+dj], w = (1,2,1), fine points outside skipped
__global__ __launch_bounds__(256) void k_restrict(const cplx *__restrict__ r, cplx *__restrict__ fc, int nzf, int nxf, int nzc, int nxc,
                                                  const RhsScal *scal) {
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    const long long Nf = (long long)nzf * nxf, Nc = (long long)nzc * nxc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < Nc; i += (long long)gridDim.x * blockDim.x) {
        const int I = (int)(i / nxc), J = (int)(i % nxc);
        cplx acc = cmake(0.0, 0.0);
#pragma unroll
        for (int di = -1; di <= 1; ++di) {
            const int fi = 2 * I + di;
            if (fi < 0 || fi >= nzf) continue;
#pragma unroll
            for (int dj = -1; dj <= 1; ++dj) {
                const int fj = 2 * J + dj;
                if (fj < 0 || fj >= nxf) continue;
                const double w = (di == 0 ? 2.0 : 1.0) * (dj == 0 ? 2.0 : 1.0) / 16.0;
                const cplx v = r[(long long)b * Nf + (long long)fi * nxf + fj];
                acc.x += w * v.x; acc.y += w * v.y;
            }
        }
        fc[(long long)b * Nc + i] = acc;
    }
}

// bilinear prolongation and correction: u += P ec
__global__ __launch_bounds__(256) void k_prolong_add(const cplx *__restrict__ ec, cplx *__restrict__ u, int nzf, int nxf, int nzc, int nxc,
                                                     const RhsScal *scal) {
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    const long long Nf = (long long)nzf * nxf, Nc = (long long)nzc * nxc;
    const cplx *e = ec + (long long)b * Nc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < Nf; i += (long long)gridDim.x * blockDim.x) {
        const int fi = (int)(i / nxf), fj = (int)(i % nxf);
        const int I = fi >> 1, J = fj >> 1;
        const bool oi = fi & 1, oj = fj & 1;
        const double wi0 = oi ? 0.5 : 1.0, wj0 = oj ? 0.5 : 1.0;
        cplx acc = cscale(e[(long long)I * nxc + J], wi0 * wj0);
        if (oj && J + 1 < nxc) { const cplx v = e[(long long)I * nxc + J + 1]; acc.x += wi0 * 0.5 * v.x; acc.y += wi0 * 0.5 * v.y; }
        if (oi && I + 1 < nzc) {
            const cplx v = e[(long long)(I + 1) * nxc + J]; acc.x += 0.5 * wj0 * v.x; acc.y += 0.5 * wj0 * v.y;
            if (oj && J + 1 < nxc) { const cplx v2 = e[(long long)(I + 1) * nxc + J + 1]; acc.x += 0.25 * v2.x; acc.y += 0.25 * v2.y; }
        }
        cplx *up = u + (long long)b * Nf + i;
        *up = cadd(*up, acc);
    }
}

// coarsest grid: u = Ainv f with the dense inverse stored transposed (coalesced over rows)
__global__ __launch_bounds__(256) void k_coarse_dense(const cplx *__restrict__ invT, const cplx *__restrict__ f, cplx *__restrict__ u,
                                                      int nc, const RhsScal *scal) {
    extern __shared__ cplx fsh[];
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    for (int j = threadIdx.x; j < nc; j += blockDim.x) fsh[j] = f[(long long)b * nc + j];
    __syncthreads();
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nc) return;
    cplx acc = cmake(0.0, 0.0);
    for (int j = 0; j < nc; ++j) cfma(acc, invT[(long long)j * nc + row], fsh[j]);
    u[(long long)b * nc + row] = acc;
}

// ---- strip line relaxation --------------------------------------------------------------------------
// z-lines: columns ix in [0,W) u [nx-W,nx), all iz.  line li -> ix.
__device__ inline int zline_ix(int li, int W, int nx) { return li < W ? li : nx - 2 * W + li; }
// x-lines: rows iz in [0,W) u [nz-W,nz), ix in [W, nx-W).  line li -> iz.
__device__ inline int xline_iz(int li, int W, int nz) { return li < W ? li : nz - 2 * W + li; }

// Thomas factors of every line: m_i = 1/(b_i - a_i c'_{i-1}), c'_i = c_i m_i
__global__ void k_line_factor(const cplx *__restrict__ C, int nz, int nx, int W, int zdir, cplx *__restrict__ m, cplx *__restrict__ cp, cplx *__restrict__ af) {
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= 2 * W) return;
    const long long N = (long long)nz * nx;
    const int len = zdir ? nz : nx - 2 * W;
    const cplx *Ca = C + (long long)(zdir ? 1 : 3) * N;     // slot(-1,0) or slot(0,-1)
    const cplx *Cb = C + 4LL * N;
    const cplx *Cc = C + (long long)(zdir ? 7 : 5) * N;     // slot(+1,0) or slot(0,+1)
    cplx cprev = cmake(0.0, 0.0);
    for (int i = 0; i < len; ++i) {
        const long long idx = zdir ? (long long)i * nx + zline_ix(li, W, nx) : (long long)xline_iz(li, W, nz) * nx + (W + i);
        const cplx a = (i == 0) ? cmake(0.0, 0.0) : Ca[idx];
        const cplx den = csub(Cb[idx], cmul(a, cprev));
        const cplx mi = crecip(den);
        const cplx c = (i == len - 1) ? cmake(0.0, 0.0) : Cc[idx];
        cprev = cmul(c, mi);
        m[(long long)li * len + i] = mi;
        cp[(long long)li * len + i] = cprev;
        af[(long long)li * len + i] = cneg(cmul(a, mi));
    }
}

// Solve every strip line for the residual r and add the correction: u += wstrip * T^-1 r.
// One wave per (line, right-hand side).  Both Thomas sweeps are first-order linear recurrences
//     forward : y_i = af_i y_{i-1} + m_i r_i          (af_i = -a_i m_i)
//     backward: x_i = y_i - cp_i x_{i+1}
// evaluated as segmented scans of affine maps: every lane owns SEG consecutive points (local
// sequential pass), the 64 segment maps are combined with a shuffle scan, and a second local pass
// applies the incoming value -- ~2*SEG + 6 dependent steps instead of 2*len.
constexpr int LSEG = 16;

struct Affine { cplx A, B; };     // x_out = A x_in + B
__device__ inline Affine compose(const Affine &second, const Affine &first) {   // second after first
    Affine r; r.A = cmul(second.A, first.A); r.B = cadd(cmul(second.A, first.B), second.B); return r;
}
__device__ inline cplx shfl_c(cplx v, int src) { cplx r; r.x = __shfl(v.x, src, 64); r.y = __shfl(v.y, src, 64); return r; }

struct LineFactors { const cplx *m, *cp, *af; };

__global__ __launch_bounds__(64) void k_line_solve(int nz, int nx, int W, LineFactors zf, LineFactors xf,
                                                   const cplx *__restrict__ r, cplx *__restrict__ u, double wstrip,
                                                   const RhsScal *scal) {
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    const int zdir = blockIdx.x < 2 * W ? 1 : 0;          // first 2W workgroups: z-lines, next 2W: x-lines
    const int li = zdir ? blockIdx.x : blockIdx.x - 2 * W, lane = threadIdx.x;
    const cplx *m = zdir ? zf.m : xf.m, *cp = zdir ? zf.cp : xf.cp, *af = zdir ? zf.af : xf.af;
    const long long N = (long long)nz * nx;
    const int len = zdir ? nz : nx - 2 * W;
    const cplx *rb = r + (long long)b * N;
    cplx *ub = u + (long long)b * N;
    const long long base = zdir ? (long long)zline_ix(li, W, nx) : (long long)xline_iz(li, W, nz) * nx + W;
    const long long stride = zdir ? nx : 1;
    const cplx *mm = m + (long long)li * len, *cc = cp + (long long)li * len, *aa = af + (long long)li * len;
    const cplx zero = cmake(0.0, 0.0), one = cmake(1.0, 0.0);

    const int chunk = 64 * LSEG;
    const int nchunk = (len + chunk - 1) / chunk;
    // ---- forward sweep, chunk by chunk; y kept in registers only for the last chunk, so it is staged through u's
    //      companion buffer r (each point is read and written by the same lane) ----
    cplx carry = zero;
    cplx *rw = const_cast<cplx *>(rb);
    for (int c = 0; c < nchunk; ++c) {
        const int i0 = c * chunk + lane * LSEG;
        cplx y[LSEG], a_[LSEG];
        Affine seg; seg.A = one; seg.B = zero;
#pragma unroll
        for (int k = 0; k < LSEG; ++k) {
            const int i = i0 + k;
            cplx d = zero, mi = zero, ai = zero;
            if (i < len) { d = rb[base + (long long)i * stride]; mi = mm[i]; ai = aa[i]; }
            else { ai = one; }                       // identity map beyond the end
            a_[k] = ai;
            y[k] = (i < len) ? cmul(d, mi) : zero;   // beta_i
        }
#pragma unroll
        for (int k = 0; k < LSEG; ++k) { seg.B = cadd(cmul(a_[k], seg.B), y[k]); seg.A = cmul(a_[k], seg.A); }
        // inclusive scan of the segment maps over the wave
        Affine inc = seg;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            Affine prev; prev.A = shfl_c(inc.A, lane - off < 0 ? 0 : lane - off); prev.B = shfl_c(inc.B, lane - off < 0 ? 0 : lane - off);
            if (lane >= off) inc = compose(inc, prev);
        }
        // incoming value of this lane's segment: exclusive prefix applied to the carry
        Affine exc; exc.A = shfl_c(inc.A, lane == 0 ? 0 : lane - 1); exc.B = shfl_c(inc.B, lane == 0 ? 0 : lane - 1);
        cplx xin = (lane == 0) ? carry : cadd(cmul(exc.A, carry), exc.B);
#pragma unroll
        for (int k = 0; k < LSEG; ++k) {
            xin = cadd(cmul(a_[k], xin), y[k]);
            const int i = i0 + k;
            if (i < len) rw[base + (long long)i * stride] = xin;
        }
        // carry for the next chunk = value after the whole chunk
        const cplx lastA = shfl_c(inc.A, 63), lastB = shfl_c(inc.B, 63);
        carry = cadd(cmul(lastA, carry), lastB);
    }
    // ---- backward sweep: x_i = y_i - cp_i x_{i+1}, lanes own the same segments, scanned from the high end ----
    carry = zero;
    for (int c = nchunk - 1; c >= 0; --c) {
        const int i0 = c * chunk + lane * LSEG;
        cplx y[LSEG], a_[LSEG];
#pragma unroll
        for (int k = 0; k < LSEG; ++k) {
            const int i = i0 + k;
            if (i < len) { y[k] = rw[base + (long long)i * stride]; a_[k] = cneg(cc[i]); }
            else { y[k] = zero; a_[k] = one; }
        }
        Affine seg; seg.A = one; seg.B = zero;
#pragma unroll
        for (int k = LSEG - 1; k >= 0; --k) { seg.B = cadd(cmul(a_[k], seg.B), y[k]); seg.A = cmul(a_[k], seg.A); }
        Affine inc = seg;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int src = lane + off > 63 ? 63 : lane + off;
            Affine nxt; nxt.A = shfl_c(inc.A, src); nxt.B = shfl_c(inc.B, src);
            if (lane + off <= 63) inc = compose(inc, nxt);
        }
        const int srcx = lane == 63 ? 63 : lane + 1;
        Affine exc; exc.A = shfl_c(inc.A, srcx); exc.B = shfl_c(inc.B, srcx);
        cplx xin = (lane == 63) ? carry : cadd(cmul(exc.A, carry), exc.B);
#pragma unroll
        for (int k = LSEG - 1; k >= 0; --k) {
            xin = cadd(cmul(a_[k], xin), y[k]);
            const int i = i0 + k;
            if (i < len) {
                cplx uv = ub[base + (long long)i * stride];
                uv.x += wstrip * xin.x; uv.y += wstrip * xin.y;
                ub[base + (long long)i * stride] = uv;
            }
        }
        const cplx firstA = shfl_c(inc.A, 0), firstB = shfl_c(inc.B, 0);
        carry = cadd(cmul(firstA, carry), firstB);
    }
}

inline int vblocks(long long N) { long long nb = (N + 255) / 256; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1; return (int)nb; }

// dense complex inverse by Gauss-Jordan with partial pivoting (host, n <= ~1000)
bool invert_dense(std::vector<std::complex<double>> &A, int n) {
    std::vector<std::complex<double>> I((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) I[(size_t)i * n + i] = 1.0;
    for (int col = 0; col < n; ++col) {
        int piv = col; double best = std::abs(A[(size_t)col * n + col]);
        for (int r = col + 1; r < n; ++r) { const double v = std::abs(A[(size_t)r * n + col]); if (v > best) { best = v; piv = r; } }
        if (best == 0.0) return false;
        if (piv != col) for (int j = 0; j < n; ++j) { std::swap(A[(size_t)piv * n + j], A[(size_t)col * n + j]); std::swap(I[(size_t)piv * n + j], I[(size_t)col * n + j]); }
        const std::complex<double> d = 1.0 / A[(size_t)col * n + col];
        for (int j = 0; j < n; ++j) { A[(size_t)col * n + j] *= d; I[(size_t)col * n + j] *= d; }
        for (int r = 0; r < n; ++r) {
            if (r == col) continue;
            const std::complex<double> f = A[(size_t)r * n + col];
            if (f == 0.0) continue;
            std::complex<double> *Ar = &A[(size_t)r * n], *Ac = &A[(size_t)col * n], *Ir = &I[(size_t)r * n], *Ic = &I[(size_t)col * n];
            for (int j = 0; j < n; ++j) { Ar[j] -= f * Ac[j]; Ir[j] -= f * Ic[j]; }
        }
    }
    A.swap(I);
    return true;
}

template <typename T>
std::vector<T> inject(const std::vector<T> &a, int nz, int nx) {
    const int nzc = (nz + 1) / 2, nxc = (nx + 1) / 2;
    std::vector<T> out((size_t)nzc * nxc);
    for (int I = 0; I < nzc; ++I) for (int J = 0; J < nxc; ++J) out[(size_t)I * nxc + J] = a[(size_t)(2 * I) * nx + 2 * J];
    return out;
}

int assemble_child(helm_op *parent, helm_op *child, const std::vector<cplx> &c, const std::vector<double> &rho,
                   const std::vector<double> &th, const std::vector<double> &ep, const std::vector<double> &de,
                   double tau, double cpml) {
    int rc = helm_set_stream(child, parent->stream);
    if (rc) return rc;
    rc = helm_set_model(child, (const double *)c.data(), rho.data(), th.empty() ? nullptr : th.data(), ep.empty() ? nullptr : ep.data(),
                        de.empty() ? nullptr : de.data());
    if (rc) return rc;
    return helm_assemble(child, parent->a_freq_re, parent->a_freq_im, tau, parent->a_ky, cpml);
}

}  // namespace

void mg_destroy(helm_op *op) {
    MgPrecond *P = op->mg;
    if (!P) return;
    hipSetDevice(op->device);
    if (op->stream) hipStreamSynchronize(op->stream);
    for (MgLevel &L : P->lv) {
        if (L.op) { L.op->own_stream = false; L.op->stream = nullptr; helm_destroy(L.op); }
        hipFree(L.u); hipFree(L.f); hipFree(L.r); hipFree(L.t); hipFree(L.f2); hipFree(L.g);
    }
    if (P->sop) { P->sop->own_stream = false; P->sop->stream = nullptr; helm_destroy(P->sop); }
    hipFree(P->d_cinvT); hipFree(P->d_tiles); hipFree(P->zl_m); hipFree(P->zl_c); hipFree(P->xl_m); hipFree(P->xl_c); hipFree(P->zl_a); hipFree(P->xl_a); hipFree(P->strip_r);
    delete P;
    op->mg = nullptr;
}

#define MG_TRY(call) do { int _rc = (call); if (_rc) { helm_set_error(op, helm_last_error(nullptr)); mg_destroy(op); return _rc; } } while (0)
#define MG_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { char _b[256]; snprintf(_b, sizeof(_b), "%s failed: %s", #call, hipGetErrorString(_e)); \
    helm_set_error(op, _b); mg_destroy(op); return HELM_ERR_DEVICE; } } while (0)

int mg_setup(helm_op *op, int batch) {
    if (op->mg && op->mg->batch >= batch) return HELM_OK;
    if (op->mg) mg_destroy(op);
    if (op->h_c.empty()) HELM_FAIL(op, HELM_ERR_STATE, "model not set");
    MgPrecond *P = new MgPrecond();
    op->mg = P;
    P->batch = batch;
    P->beta = env_double("HELM_MG_BETA", 0.5);
    P->omega_j = env_double("HELM_MG_OMEGA", 0.8);
    P->cpml_m = env_double("HELM_MG_CPML", 30.0);
    P->sweeps = env_int("HELM_MG_SWEEPS", 4);
    P->wstrip = env_double("HELM_MG_WSTRIP", 1.0);
    P->nu1 = env_int("HELM_MG_NU1", 1);
    P->nu2 = env_int("HELM_MG_NU2", 1);
    P->min_n = env_int("HELM_MG_MIN_N", 16);
    P->fdepth = env_int("HELM_MG_FDEPTH", 3);
    P->W = op->nPML + 2;
    if (2 * P->W + 2 > op->nx || 2 * P->W + 2 > op->nz) P->sweeps = 0;      // grid too small for a frame: plain cycle

    // shifted damping: 1/tau_M = 1/tau_A + omega * beta / 2   (w~ = w (1 - i beta / 2)  =>  w~^2 ~ w^2 (1 - i beta))
    const double omega = 2.0 * M_PI * std::abs(std::complex<double>(op->a_freq_re, op->a_freq_im));
    double inv_tau = omega * P->beta / 2.0;
    if (std::isfinite(op->a_tau) && op->a_tau != 0.0) inv_tau += 1.0 / op->a_tau;
    const double tauM = 1.0 / inv_tau;
    const double cpml_weak = op->variant == HELM_EURUS ? std::min(P->cpml_m, op->a_cpml) : 0.0;
    const double mz_weak = env_double("HELM_MG_MZ_PMLSCALE", 0.02);

    // ---- levels ----
    std::vector<cplx> c = op->h_c;
    std::vector<double> rho = op->h_rho, th = op->h_theta, ep = op->h_eps, de = op->h_delta;
    int nz = op->nz, nx = op->nx, npml = op->nPML;
    double dx = op->dx, dz = op->dz;
    while (true) {
        MgLevel L;
        L.op = helm_create(op->device, op->variant, nz, nx, dx, dz, -npml, op->fs);
        if (!L.op) { helm_set_error(op, helm_last_error(nullptr)); mg_destroy(op); return HELM_ERR_DEVICE; }
        P->lv.push_back(L);
        if (op->variant == HELM_MINIZEPHYR) L.op->pml_scale = mz_weak;
        L.op->diag_floor = env_double("HELM_MG_DIAGFLOOR", 0.5);
        P->lv.back().op = L.op;
        int rc = assemble_child(op, L.op, c, rho, th, ep, de, tauM, cpml_weak);
        if (rc) { helm_set_error(op, helm_last_error(L.op)); mg_destroy(op); return rc; }
        const size_t bytes = (size_t)batch * nz * nx * sizeof(cplx);
        MgLevel &R = P->lv.back();
        if (P->lv.size() > 1) { MG_HIP(hipMalloc(&R.u, bytes)); MG_HIP(hipMalloc(&R.f, bytes)); MG_HIP(hipMalloc(&R.f2, bytes)); MG_HIP(hipMalloc(&R.g, bytes)); }
        MG_HIP(hipMalloc(&R.r, bytes)); MG_HIP(hipMalloc(&R.t, bytes));
        const int nzc = (nz + 1) / 2, nxc = (nx + 1) / 2;
        const int npmlc = std::max((npml - 1) / 2 + 1, 2);
        if (std::min(nz, nx) <= P->min_n || nzc < 2 * npmlc + 3 || nxc < 2 * npmlc + 3 || (long long)nz * nx <= 64) break;
        c = inject(c, nz, nx); rho = inject(rho, nz, nx);
        if (!th.empty()) th = inject(th, nz, nx);
        if (!ep.empty()) ep = inject(ep, nz, nx);
        if (!de.empty()) de = inject(de, nz, nx);
        nz = nzc; nx = nxc; dx *= 2; dz *= 2; npml = npmlc;
    }
    // ---- coarsest: dense inverse on the host ----
    {
        helm_op *co = P->lv.back().op;
        const int n = (int)co->N;
        if (n > 4096) HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "coarsest multigrid level too large (%d unknowns)", n);
        std::vector<cplx> planes((size_t)9 * n);
        MG_HIP(hipMemcpy(planes.data(), co->d_C, planes.size() * sizeof(cplx), hipMemcpyDeviceToHost));
        std::vector<std::complex<double>> A((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) {
            const int iz = i / co->nx, ix = i % co->nx;
            for (int k = 0; k < 9; ++k) {
                const int jz = iz + k / 3 - 1, jx = ix + k % 3 - 1;
                if (jz < 0 || jz >= co->nz || jx < 0 || jx >= co->nx) continue;
                const cplx v = planes[(size_t)k * n + i];
                A[(size_t)i * n + (size_t)jz * co->nx + jx] = std::complex<double>(v.x, v.y);
            }
        }
        if (!invert_dense(A, n)) { mg_destroy(op); HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "singular coarsest-level operator"); }
        std::vector<cplx> invT((size_t)n * n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) invT[(size_t)j * n + i] = cmake(A[(size_t)i * n + j].real(), A[(size_t)i * n + j].imag());
        MG_HIP(hipMalloc(&P->d_cinvT, invT.size() * sizeof(cplx)));
        MG_HIP(hipMemcpy(P->d_cinvT, invT.data(), invT.size() * sizeof(cplx), hipMemcpyHostToDevice));
        P->nc = n;
    }
    // ---- strip relaxation operator (true PML, shifted) ----
    if (P->sweeps > 0) {
        P->sop = helm_create(op->device, op->variant, op->nz, op->nx, op->dx, op->dz, -op->nPML, op->fs);
        if (!P->sop) { helm_set_error(op, helm_last_error(nullptr)); mg_destroy(op); return HELM_ERR_DEVICE; }
        int rc = assemble_child(op, P->sop, op->h_c, op->h_rho, op->h_theta, op->h_eps, op->h_delta, tauM, op->a_cpml);
        if (rc) { helm_set_error(op, helm_last_error(P->sop)); mg_destroy(op); return rc; }
        const int W = P->W;
        // stencil tiles (64 x 8) that touch the frame
        const int TZ = 8, ntx = (op->nx + 63) / 64, ntz = (op->nz + TZ - 1) / TZ;
        std::vector<int> tiles;
        for (int tz = 0; tz < ntz; ++tz) for (int tx = 0; tx < ntx; ++tx) {
            const int z0 = tz * TZ, z1 = std::min(z0 + TZ, op->nz) - 1, x0 = tx * 64, x1 = std::min(x0 + 64, op->nx) - 1;
            if (z0 < W || z1 >= op->nz - W || x0 < W || x1 >= op->nx - W) tiles.push_back(tz * ntx + tx);
        }
        P->ntiles = (int)tiles.size();
        MG_HIP(hipMalloc(&P->d_tiles, tiles.size() * sizeof(int)));
        MG_HIP(hipMemcpy(P->d_tiles, tiles.data(), tiles.size() * sizeof(int), hipMemcpyHostToDevice));
        const size_t zl = (size_t)2 * W * op->nz, xl = (size_t)2 * W * (op->nx - 2 * W);
        MG_HIP(hipMalloc(&P->zl_m, zl * sizeof(cplx))); MG_HIP(hipMalloc(&P->zl_c, zl * sizeof(cplx)));
        MG_HIP(hipMalloc(&P->xl_m, xl * sizeof(cplx))); MG_HIP(hipMalloc(&P->xl_c, xl * sizeof(cplx)));
        MG_HIP(hipMalloc(&P->zl_a, zl * sizeof(cplx))); MG_HIP(hipMalloc(&P->xl_a, xl * sizeof(cplx)));
        MG_HIP(hipMalloc(&P->strip_r, (size_t)batch * op->N * sizeof(cplx)));
        MG_HIP(hipMemsetAsync(P->strip_r, 0, (size_t)batch * op->N * sizeof(cplx), op->stream));
        hipLaunchKernelGGL(k_line_factor, dim3((2 * W + 63) / 64), dim3(64), 0, op->stream, (const cplx *)P->sop->d_C, op->nz, op->nx, W, 1, P->zl_m, P->zl_c, P->zl_a);
        hipLaunchKernelGGL(k_line_factor, dim3((2 * W + 63) / 64), dim3(64), 0, op->stream, (const cplx *)P->sop->d_C, op->nz, op->nx, W, 0, P->xl_m, P->xl_c, P->xl_a);
        MG_HIP(hipGetLastError());
    }
    MG_HIP(hipStreamSynchronize(op->stream));
    return HELM_OK;
}

namespace {

int smooth_sweeps(helm_op *op, MgLevel &L, cplx *&u, cplx *&alt, const cplx *f, int n, int nrhs, double omega_j, const RhsScal *scal) {
    for (int k = 0; k < n; ++k) {
        ApplyArgs a;
        a.planes = L.op->d_C; a.X = u; a.Y = alt; a.W = f; a.ld = L.op->N; a.nrhs = nrhs; a.epi = EPI_JACOBI;
        a.scal = scal; a.dinv = L.op->d_dinv; a.omega_j = omega_j; a.profile = 0; a.part = (double *)op->d_part;
        int rc = helm_launch_apply(L.op, a);
        if (rc) return rc;
        std::swap(u, alt);
    }
    return HELM_OK;
}

// One multigrid cycle on level l: u_out = approx M_l^-1 f.  fmode: levels < P->fdepth visit their coarse
// level a second time (with the coarse residual), the second visit being a plain V-cycle (truncated F-cycle).
int vcycle(helm_op *op, MgPrecond *P, int l, const cplx *f, cplx *u_out, int nrhs, const RhsScal *scal, bool fmode) {
    MgLevel &L = P->lv[l];
    helm_op *lo = L.op;
    const long long N = lo->N;
    hipStream_t st = op->stream;
    if (l == (int)P->lv.size() - 1) {
        dim3 grid((P->nc + 255) / 256, nrhs);
        hipLaunchKernelGGL(k_coarse_dense, grid, dim3(256), (size_t)P->nc * sizeof(cplx), st, (const cplx *)P->d_cinvT, f, u_out, P->nc, scal);
        return HELM_OK;
    }
    // Jacobi sweeps ping-pong between two buffers; start so that the last sweep lands in u_out
    const int pingpongs = (P->nu1 - 1) + P->nu2;
    cplx *u = (pingpongs & 1) ? L.t : u_out, *alt = (pingpongs & 1) ? u_out : L.t;
    dim3 vg(vblocks(N), nrhs);
    hipLaunchKernelGGL(k_jac0, vg, dim3(256), 0, st, (const cplx *)lo->d_dinv, f, u, N, P->omega_j, scal);
    int rc = smooth_sweeps(op, L, u, alt, f, P->nu1 - 1, nrhs, P->omega_j, scal);
    if (rc) return rc;
    auto residual = [&](helm_op *o, const cplx *x, const cplx *rhs, cplx *out) -> int {
        ApplyArgs a;
        a.planes = o->d_C; a.X = x; a.Y = out; a.W = rhs; a.ld = o->N; a.nrhs = nrhs; a.epi = EPI_RESID; a.scal = scal; a.profile = 0;
        a.part = (double *)op->d_part;
        return helm_launch_apply(o, a);
    };
    rc = residual(lo, u, f, L.r);
    if (rc) return rc;
    MgLevel &C = P->lv[l + 1];
    dim3 cg(vblocks(C.op->N), nrhs);
    hipLaunchKernelGGL(k_restrict, cg, dim3(256), 0, st, (const cplx *)L.r, C.f, lo->nz, lo->nx, C.op->nz, C.op->nx, scal);
    rc = vcycle(op, P, l + 1, C.f, C.u, nrhs, scal, fmode);
    if (rc) return rc;
    if (fmode && l < P->fdepth && l + 1 < (int)P->lv.size() - 1) {
        // second visit of the coarse level: e += V(f_c - M_c e), with its own right-hand-side / result buffers
        rc = residual(C.op, C.u, C.f, C.g);
        if (rc) return rc;
        rc = vcycle(op, P, l + 1, C.g, C.f2, nrhs, scal, false);
        if (rc) return rc;
        hipLaunchKernelGGL(k_axpy1, cg, dim3(256), 0, st, (const cplx *)C.f2, C.u, C.op->N, scal);
    }
    hipLaunchKernelGGL(k_prolong_add, vg, dim3(256), 0, st, (const cplx *)C.u, u, lo->nz, lo->nx, C.op->nz, C.op->nx, scal);
    rc = smooth_sweeps(op, L, u, alt, f, P->nu2, nrhs, P->omega_j, scal);
    if (rc) return rc;
    if (u != u_out) HELM_FAIL(op, HELM_ERR_STATE, "multigrid ping-pong parity error");
    return HELM_OK;
}

}  // namespace

int mg_apply(helm_op *op, const cplx *in, cplx *out, int nrhs, const RhsScal *scal) {
    MgPrecond *P = op->mg;
    if (!P) HELM_FAIL(op, HELM_ERR_STATE, "preconditioner not built");
    if (nrhs > P->batch) HELM_FAIL(op, HELM_ERR_ARG, "preconditioner batch too small");
    int rc = vcycle(op, P, 0, in, out, nrhs, scal, P->fdepth > 0);
    if (rc) return rc;
    const int W = P->W;
    for (int k = 0; k < P->sweeps; ++k) {
        ApplyArgs a;
        a.planes = P->sop->d_C; a.X = out; a.Y = P->strip_r; a.W = in; a.ld = op->N; a.nrhs = nrhs; a.epi = EPI_RESID; a.scal = scal;
        a.tiles = P->d_tiles; a.ntiles = P->ntiles; a.profile = 0; a.part = (double *)op->d_part;
        rc = helm_launch_apply(P->sop, a);
        if (rc) return rc;
        dim3 lg(4 * W, nrhs);
        LineFactors zf = {P->zl_m, P->zl_c, P->zl_a}, xf = {P->xl_m, P->xl_c, P->xl_a};
        hipLaunchKernelGGL(k_line_solve, lg, dim3(64), 0, op->stream, op->nz, op->nx, W, zf, xf,
                           (const cplx *)P->strip_r, out, P->wstrip, scal);
    }
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}
