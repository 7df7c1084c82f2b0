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
    helm_op *op = nullptr;                  // operator of this level (double-precision planes from the assembly kernels)
    void *C = nullptr, *dinv = nullptr;     // planes / inverse diagonal in the preconditioner's storage precision
    bool own_planes = false;                // C / dinv are single-precision copies owned by the level
    void *u = nullptr, *f = nullptr, *r = nullptr, *t = nullptr, *f2 = nullptr, *g = nullptr;   // [batch][N_l]
};

}  // namespace

struct MgPrecond {
    std::vector<MgLevel> lv;
    bool f32 = true;                            // single-precision storage/arithmetic inside the preconditioner
    void *d_cinvT = nullptr; int nc = 0;        // dense inverse of the coarsest operator, transposed
    helm_op *sop = nullptr;                     // fine grid, true PML, shifted: strip relaxation operator
    void *sC = nullptr;                         // its planes in the storage precision
    int W = 12;
    int *d_tiles = nullptr; int ntiles = 0;
    void *zl[3] = {nullptr, nullptr, nullptr};  // z-line factors m, cp, af  [2W][nz]
    void *xl[3] = {nullptr, nullptr, nullptr};  // x-line factors            [2W][nx-2W]
    void *strip_r = nullptr;                    // [batch][N]
    int batch = 0;
    double omega_j = 0.8, beta = 0.5, cpml_m = 30.0, wstrip = 1.0;
    int nu1 = 1, nu2 = 1, sweeps = 4, min_n = 16;
    bool fuse = true;        // fuse jac0 into the residual and the prolongation into the post-smoothing sweep
    int fdepth = 0;          // levels < fdepth revisit their coarse level a second time (truncated F-cycle); 0 = V-cycle
};

namespace {


__device__ inline bool active(const RhsScal *scal, int b) { return scal == nullptr || scal[b].status == ST_ACTIVE; }

// u = omega_j * dinv * f      (first smoothing sweep from a zero initial guess)
template <class V>
__global__ __launch_bounds__(256) void k_jac0(const V *__restrict__ dinv, const V *__restrict__ f, V *__restrict__ u,
                                              long long N, double omega_j, const RhsScal *scal) {
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
        u[(long long)b * N + i] = cmul(cscale(dinv[i], omega_j), f[(long long)b * N + i]);
}

// u += e
template <class V>
__global__ __launch_bounds__(256) void k_axpy1(const V *__restrict__ e, V *__restrict__ u, long long N, const RhsScal *scal) {
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const long long g = (long long)b * N + i;
        u[g] = cadd(u[g], e[g]);
    }
}

// full weighting: fc[I,J] = (1/16) sum_{di,dj} w(di) w(dj) r[2I+di, 2J+dj], w = (1,2,1), fine points outside skipped
template <class V>
__global__ __launch_bounds__(256) void k_restrict(const V *__restrict__ r, V *__restrict__ fc, int nzf, int nxf, int nzc, int nxc,
                                                  const RhsScal *scal) {
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    const long long Nf = (long long)nzf * nxf, Nc = (long long)nzc * nxc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < Nc; i += (long long)gridDim.x * blockDim.x) {
        const int I = (int)(i / nxc), J = (int)(i % nxc);
        V acc = vzero<V>();
#pragma unroll
        for (int di = -1; di <= 1; ++di) {
            const int fi = 2 * I + di;
            if (fi < 0 || fi >= nzf) continue;
#pragma unroll
            for (int dj = -1; dj <= 1; ++dj) {
                const int fj = 2 * J + dj;
                if (fj < 0 || fj >= nxf) continue;
                const double w = (di == 0 ? 2.0 : 1.0) * (dj == 0 ? 2.0 : 1.0) / 16.0;
                const V v = r[(long long)b * Nf + (long long)fi * nxf + fj];
                acc.x += w * v.x; acc.y += w * v.y;
            }
        }
        fc[(long long)b * Nc + i] = acc;
    }
}

// bilinear prolongation and correction: u += P ec
template <class V>
__global__ __launch_bounds__(256) void k_prolong_add(const V *__restrict__ ec, V *__restrict__ u, int nzf, int nxf, int nzc, int nxc,
                                                     const RhsScal *scal) {
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    const long long Nf = (long long)nzf * nxf, Nc = (long long)nzc * nxc;
    const V *e = ec + (long long)b * Nc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < Nf; i += (long long)gridDim.x * blockDim.x) {
        const int fi = (int)(i / nxf), fj = (int)(i % nxf);
        const int I = fi >> 1, J = fj >> 1;
        const bool oi = fi & 1, oj = fj & 1;
        const double wi0 = oi ? 0.5 : 1.0, wj0 = oj ? 0.5 : 1.0;
        V acc = cscale(e[(long long)I * nxc + J], wi0 * wj0);
        if (oj && J + 1 < nxc) { const V v = e[(long long)I * nxc + J + 1]; acc.x += wi0 * 0.5 * v.x; acc.y += wi0 * 0.5 * v.y; }
        if (oi && I + 1 < nzc) {
            const V v = e[(long long)(I + 1) * nxc + J]; acc.x += 0.5 * wj0 * v.x; acc.y += 0.5 * wj0 * v.y;
            if (oj && J + 1 < nxc) { const V v2 = e[(long long)(I + 1) * nxc + J + 1]; acc.x += 0.25 * v2.x; acc.y += 0.25 * v2.y; }
        }
        V *up = u + (long long)b * Nf + i;
        *up = cadd(*up, acc);
    }
}

// coarsest grid: u = Ainv f with the dense inverse stored transposed (coalesced over rows)
template <class V>
__global__ __launch_bounds__(256) void k_coarse_dense(const V *__restrict__ invT, const V *__restrict__ f, V *__restrict__ u,
                                                      int nc, const RhsScal *scal) {
    extern __shared__ double fsh_raw[];
    V *fsh = reinterpret_cast<V *>(fsh_raw);
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    for (int j = threadIdx.x; j < nc; j += blockDim.x) fsh[j] = f[(long long)b * nc + j];
    __syncthreads();
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nc) return;
    V acc = vzero<V>();
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

template <class V> struct Affine { V A, B; };     // x_out = A x_in + B
template <class V> __device__ inline Affine<V> compose(const Affine<V> &second, const Affine<V> &first) {   // second after first
    Affine<V> r; r.A = cmul(second.A, first.A); r.B = cadd(cmul(second.A, first.B), second.B); return r;
}
template <class V> __device__ inline V shfl_c(V v, int src) { V r; r.x = __shfl(v.x, src, 64); r.y = __shfl(v.y, src, 64); return r; }

template <class V> struct LineFactors { const V *m, *cp, *af; };

template <class V>
__global__ __launch_bounds__(64) void k_line_solve(int nz, int nx, int W, LineFactors<V> zf, LineFactors<V> xf,
                                                   const V *__restrict__ r, V *__restrict__ u, double wstrip,
                                                   const RhsScal *scal) {
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    const int zdir = blockIdx.x < 2 * W ? 1 : 0;          // first 2W workgroups: z-lines, next 2W: x-lines
    const int li = zdir ? blockIdx.x : blockIdx.x - 2 * W, lane = threadIdx.x;
    const V *m = zdir ? zf.m : xf.m, *cp = zdir ? zf.cp : xf.cp, *af = zdir ? zf.af : xf.af;
    const long long N = (long long)nz * nx;
    const int len = zdir ? nz : nx - 2 * W;
    const V *rb = r + (long long)b * N;
    V *ub = u + (long long)b * N;
    const long long base = zdir ? (long long)zline_ix(li, W, nx) : (long long)xline_iz(li, W, nz) * nx + W;
    const long long stride = zdir ? nx : 1;
    const V *mm = m + (long long)li * len, *cc = cp + (long long)li * len, *aa = af + (long long)li * len;
    const V zero = vzero<V>(), one = vone<V>();

    const int chunk = 64 * LSEG;
    const int nchunk = (len + chunk - 1) / chunk;
    // ---- forward sweep, chunk by chunk; y kept in registers only for the last chunk, so it is staged through u's
    //      companion buffer r (each point is read and written by the same lane) ----
    V carry = zero;
    V *rw = const_cast<V *>(rb);
    for (int c = 0; c < nchunk; ++c) {
        const int i0 = c * chunk + lane * LSEG;
        V y[LSEG], a_[LSEG];
        Affine<V> seg; seg.A = one; seg.B = zero;
#pragma unroll
        for (int k = 0; k < LSEG; ++k) {
            const int i = i0 + k;
            V d = zero, mi = zero, ai = zero;
            if (i < len) { d = rb[base + (long long)i * stride]; mi = mm[i]; ai = aa[i]; }
            else { ai = one; }                       // identity map beyond the end
            a_[k] = ai;
            y[k] = (i < len) ? cmul(d, mi) : zero;   // beta_i
        }
#pragma unroll
        for (int k = 0; k < LSEG; ++k) { seg.B = cadd(cmul(a_[k], seg.B), y[k]); seg.A = cmul(a_[k], seg.A); }
        // inclusive scan of the segment maps over the wave
        Affine<V> inc = seg;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            Affine<V> prev; prev.A = shfl_c(inc.A, lane - off < 0 ? 0 : lane - off); prev.B = shfl_c(inc.B, lane - off < 0 ? 0 : lane - off);
            if (lane >= off) inc = compose(inc, prev);
        }
        // incoming value of this lane's segment: exclusive prefix applied to the carry
        Affine<V> exc; exc.A = shfl_c(inc.A, lane == 0 ? 0 : lane - 1); exc.B = shfl_c(inc.B, lane == 0 ? 0 : lane - 1);
        V xin = (lane == 0) ? carry : cadd(cmul(exc.A, carry), exc.B);
#pragma unroll
        for (int k = 0; k < LSEG; ++k) {
            xin = cadd(cmul(a_[k], xin), y[k]);
            const int i = i0 + k;
            if (i < len) rw[base + (long long)i * stride] = xin;
        }
        // carry for the next chunk = value after the whole chunk
        const V lastA = shfl_c(inc.A, 63), lastB = shfl_c(inc.B, 63);
        carry = cadd(cmul(lastA, carry), lastB);
    }
    // ---- backward sweep: x_i = y_i - cp_i x_{i+1}, lanes own the same segments, scanned from the high end ----
    carry = zero;
    for (int c = nchunk - 1; c >= 0; --c) {
        const int i0 = c * chunk + lane * LSEG;
        V y[LSEG], a_[LSEG];
#pragma unroll
        for (int k = 0; k < LSEG; ++k) {
            const int i = i0 + k;
            if (i < len) { y[k] = rw[base + (long long)i * stride]; a_[k] = cneg(cc[i]); }
            else { y[k] = zero; a_[k] = one; }
        }
        Affine<V> seg; seg.A = one; seg.B = zero;
#pragma unroll
        for (int k = LSEG - 1; k >= 0; --k) { seg.B = cadd(cmul(a_[k], seg.B), y[k]); seg.A = cmul(a_[k], seg.A); }
        Affine<V> inc = seg;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int src = lane + off > 63 ? 63 : lane + off;
            Affine<V> nxt; nxt.A = shfl_c(inc.A, src); nxt.B = shfl_c(inc.B, src);
            if (lane + off <= 63) inc = compose(inc, nxt);
        }
        const int srcx = lane == 63 ? 63 : lane + 1;
        Affine<V> exc; exc.A = shfl_c(inc.A, srcx); exc.B = shfl_c(inc.B, srcx);
        V xin = (lane == 63) ? carry : cadd(cmul(exc.A, carry), exc.B);
#pragma unroll
        for (int k = LSEG - 1; k >= 0; --k) {
            xin = cadd(cmul(a_[k], xin), y[k]);
            const int i = i0 + k;
            if (i < len) {
                V uv = ub[base + (long long)i * stride];
                uv.x += wstrip * xin.x; uv.y += wstrip * xin.y;
                ub[base + (long long)i * stride] = uv;
            }
        }
        const V firstA = shfl_c(inc.A, 0), firstB = shfl_c(inc.B, 0);
        carry = cadd(cmul(firstA, carry), firstB);
    }
}

inline int vblocks(long long N) { long long nb = (N + 255) / 256; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1; return (int)nb; }

// out = (V) in, elementwise (precision conversion of vectors / planes)
template <class VO, class VI>
__global__ __launch_bounds__(256) void k_convert(const VI *__restrict__ in, VO *__restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        VO o; o.x = in[i].x; o.y = in[i].y; out[i] = o;
    }
}
// batched variant honouring the active mask: out[b][i] = in[b][i]
template <class VO, class VI>
__global__ __launch_bounds__(256) void k_convert_rhs(const VI *__restrict__ in, VO *__restrict__ out, long long N, const RhsScal *scal) {
    const int b = blockIdx.y;
    if (!active(scal, b)) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        VO o; const VI v = in[(long long)b * N + i]; o.x = v.x; o.y = v.y; out[(long long)b * N + i] = o;
    }
}

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

void destroy_child(helm_op *&c) {
    if (!c) return;
    c->own_stream = false; c->stream = nullptr;
    helm_destroy(c);
    c = nullptr;
}

}  // namespace

void mg_destroy(helm_op *op) {
    if (op->mg3) mg3_destroy(op);
    MgPrecond *P = op->mg;
    if (!P) return;
    hipSetDevice(op->device);
    if (op->stream) hipStreamSynchronize(op->stream);
    for (MgLevel &L : P->lv) {
        if (L.own_planes) { hipFree(L.C); hipFree(L.dinv); }
        destroy_child(L.op);
        hipFree(L.u); hipFree(L.f); hipFree(L.r); hipFree(L.t); hipFree(L.f2); hipFree(L.g);
    }
    destroy_child(P->sop);
    hipFree(P->d_cinvT); hipFree(P->d_tiles); hipFree(P->strip_r);
    for (int k = 0; k < 3; ++k) { hipFree(P->zl[k]); hipFree(P->xl[k]); }
    delete P;
    op->mg = nullptr;
}

#define MG_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { char _b[256]; snprintf(_b, sizeof(_b), "%s failed: %s", #call, hipGetErrorString(_e)); \
    helm_set_error(op, _b); mg_destroy(op); return HELM_ERR_DEVICE; } } while (0)

namespace {

// Build everything that depends on the storage type V.
template <class V>
int setup_typed(helm_op *op, MgPrecond *P) {
    const int batch = P->batch;
    hipStream_t st = op->stream;
    constexpr bool F32 = sizeof(V) == sizeof(cplxf);
    // per-level planes / vectors
    for (size_t l = 0; l < P->lv.size(); ++l) {
        MgLevel &L = P->lv[l];
        const long long N = L.op->N;
        if (F32) {
            MG_HIP(hipMalloc(&L.C, (size_t)9 * N * sizeof(V))); MG_HIP(hipMalloc(&L.dinv, (size_t)N * sizeof(V)));
            L.own_planes = true;
            HELM_LAUNCH((k_convert<V, cplx>), dim3(vblocks(9 * N)), dim3(256), 0, st, (const cplx *)L.op->d_C, (V *)L.C, 9 * N);
            HELM_LAUNCH((k_convert<V, cplx>), dim3(vblocks(N)), dim3(256), 0, st, (const cplx *)L.op->d_dinv, (V *)L.dinv, N);
        } else { L.C = L.op->d_C; L.dinv = L.op->d_dinv; }
        const size_t bytes = (size_t)batch * N * sizeof(V);
        if (l > 0 || F32) { MG_HIP(hipMalloc(&L.u, bytes)); MG_HIP(hipMalloc(&L.f, bytes)); }
        if (l > 0) { MG_HIP(hipMalloc(&L.f2, bytes)); MG_HIP(hipMalloc(&L.g, bytes)); }
        MG_HIP(hipMalloc(&L.r, bytes)); MG_HIP(hipMalloc(&L.t, bytes));
    }
    // coarsest: dense inverse on the host (double), stored transposed in V
    {
        helm_op *co = P->lv.back().op;
        const int n = (int)co->N;
        if (n > 4096) { mg_destroy(op); HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "coarsest multigrid level too large (%d unknowns)", n); }
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
        std::vector<V> invT((size_t)n * n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { V v; v.x = A[(size_t)i * n + j].real(); v.y = A[(size_t)i * n + j].imag(); invT[(size_t)j * n + i] = v; }
        MG_HIP(hipMalloc(&P->d_cinvT, invT.size() * sizeof(V)));
        MG_HIP(hipMemcpy(P->d_cinvT, invT.data(), invT.size() * sizeof(V), hipMemcpyHostToDevice));
        P->nc = n;
    }
    // strip relaxation (always double precision: the true C-PML makes the line systems too ill-conditioned
    // for single precision -- measured +50 % outer iterations): line factors and residual buffer
    if (P->sweeps > 0) {
        const int W = P->W;
        const long long N = op->N;
        P->sC = P->sop->d_C;
        const size_t zl = (size_t)2 * W * op->nz, xl = (size_t)2 * W * (op->nx - 2 * W);
        for (int k = 0; k < 3; ++k) { MG_HIP(hipMalloc(&P->zl[k], zl * sizeof(cplx))); MG_HIP(hipMalloc(&P->xl[k], xl * sizeof(cplx))); }
        HELM_LAUNCH(k_line_factor, dim3((2 * W + 63) / 64), dim3(64), 0, st, (const cplx *)P->sop->d_C, op->nz, op->nx, W, 1,
                           (cplx *)P->zl[0], (cplx *)P->zl[1], (cplx *)P->zl[2]);
        HELM_LAUNCH(k_line_factor, dim3((2 * W + 63) / 64), dim3(64), 0, st, (const cplx *)P->sop->d_C, op->nz, op->nx, W, 0,
                           (cplx *)P->xl[0], (cplx *)P->xl[1], (cplx *)P->xl[2]);
        MG_HIP(hipMalloc(&P->strip_r, (size_t)batch * N * sizeof(cplx)));
        MG_HIP(hipMemsetAsync(P->strip_r, 0, (size_t)batch * N * sizeof(cplx), st));
    }
    MG_HIP(hipGetLastError());
    MG_HIP(hipStreamSynchronize(st));
    return HELM_OK;
}

}  // namespace

int mg_setup(helm_op *op, int batch) {
    if (op->ny > 0) return mg3_setup(op, batch);
    if (op->mg && op->mg->batch >= batch) return HELM_OK;
    if (op->mg) mg_destroy(op);
    { const int rch = helm_ensure_host_model(op); if (rch) return rch; }
    MgPrecond *P = new MgPrecond();
    op->mg = P;
    P->batch = batch;
    P->f32 = false;   // single-precision cycle: measured +10-25 % iterations, no net gain -> off
    P->beta = 0.6;   // 0.42 diverges on the 1024^2 model, 0.5-0.7 equivalent: keep a margin
    P->omega_j = 0.8;
    P->cpml_m = 30.0;
    P->sweeps = 4;
    P->wstrip = 1.0;
    P->nu1 = 1;
    P->nu2 = 1;
    P->min_n = 16;
    P->fdepth = 3;
    P->fuse = false;   // measured: no gain (extra address math in the tile load offsets the saved pass)
    P->W = op->nPML + 2;
    if (2 * P->W + 2 > op->nx || 2 * P->W + 2 > op->nz) P->sweeps = 0;      // grid too small for a frame: plain cycle

    // shifted damping: 1/tau_M = 1/tau_A + omega * beta / 2   (w~ = w (1 - i beta / 2)  =>  w~^2 ~ w^2 (1 - i beta))
    const double omega = 2.0 * M_PI * std::abs(std::complex<double>(op->a_freq_re, op->a_freq_im));
    double inv_tau = omega * P->beta / 2.0;
    if (std::isfinite(op->a_tau) && op->a_tau != 0.0) inv_tau += 1.0 / op->a_tau;
    const double tauM = 1.0 / inv_tau;
    const double cpml_weak = op->variant == HELM_EURUS ? std::min(P->cpml_m, op->a_cpml) : 0.0;
    const double mz_weak = 0.1;

    // ---- levels (operators assembled in double precision by the regular assembly kernels) ----
    std::vector<cplx> c = op->h_c;
    std::vector<double> rho = op->h_rho, th = op->h_theta, ep = op->h_eps, de = op->h_delta;
    int nz = op->nz, nx = op->nx, npml = op->nPML;
    double dx = op->dx, dz = op->dz;
    while (true) {
        MgLevel L;
        L.op = helm_create(op->device, op->variant, nz, nx, dx, dz, -npml, op->fs);
        if (!L.op) { helm_set_error(op, helm_last_error(nullptr)); mg_destroy(op); return HELM_ERR_DEVICE; }
        if (op->variant == HELM_MINIZEPHYR) L.op->pml_scale = mz_weak;
        L.op->diag_floor = 0.5;
        P->lv.push_back(L);
        int rc = assemble_child(op, L.op, c, rho, th, ep, de, tauM, cpml_weak);
        if (rc) { helm_set_error(op, helm_last_error(L.op)); mg_destroy(op); return rc; }
        const int nzc = (nz + 1) / 2, nxc = (nx + 1) / 2;
        const int npmlc = std::max((npml - 1) / 2 + 1, 2);
        if (std::min(nz, nx) <= P->min_n || nzc < 2 * npmlc + 3 || nxc < 2 * npmlc + 3 || (long long)nz * nx <= 64) break;
        c = inject(c, nz, nx); rho = inject(rho, nz, nx);
        if (!th.empty()) th = inject(th, nz, nx);
        if (!ep.empty()) ep = inject(ep, nz, nx);
        if (!de.empty()) de = inject(de, nz, nx);
        nz = nzc; nx = nxc; dx *= 2; dz *= 2; npml = npmlc;
    }
    if (P->sweeps > 0) {
        P->sop = helm_create(op->device, op->variant, op->nz, op->nx, op->dx, op->dz, -op->nPML, op->fs);
        if (!P->sop) { helm_set_error(op, helm_last_error(nullptr)); mg_destroy(op); return HELM_ERR_DEVICE; }
        int rc = assemble_child(op, P->sop, op->h_c, op->h_rho, op->h_theta, op->h_eps, op->h_delta, tauM, op->a_cpml);
        if (rc) { helm_set_error(op, helm_last_error(P->sop)); mg_destroy(op); return rc; }
        // stencil tiles (64 x helm_stencil_tile_rows()) that touch the frame
        const int W = P->W, TZ = helm_stencil_tile_rows(), ntx = (op->nx + 63) / 64, ntz = (op->nz + TZ - 1) / TZ;
        std::vector<int> tiles;
        for (int tz = 0; tz < ntz; ++tz) for (int tx = 0; tx < ntx; ++tx) {
            const int z0 = tz * TZ, z1 = std::min(z0 + TZ, op->nz) - 1, x0 = tx * 64, x1 = std::min(x0 + 64, op->nx) - 1;
            if (z0 < W || z1 >= op->nz - W || x0 < W || x1 >= op->nx - W) tiles.push_back(tz * ntx + tx);
        }
        P->ntiles = (int)tiles.size();
        MG_HIP(hipMalloc(&P->d_tiles, tiles.size() * sizeof(int)));
        MG_HIP(hipMemcpy(P->d_tiles, tiles.data(), tiles.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    return P->f32 ? setup_typed<cplxf>(op, P) : setup_typed<cplx>(op, P);
}

namespace {

template <class V>
int stencil_level(helm_op *op, helm_op *lo, const void *planes, const V *x, V *y, const V *w, int nrhs, int epi, const RhsScal *scal,
                  const void *dinv = nullptr, double omega_j = 0.0, const int *tiles = nullptr, int ntiles = 0,
                  int xmode = 0, V *uout = nullptr, const V *ecoarse = nullptr, int nzc = 0, int nxc = 0) {
    ApplyArgs a;
    a.xmode = xmode; a.U = (cplx *)uout; a.E = (const cplx *)ecoarse; a.nzc = nzc; a.nxc = nxc;
    a.planes = (const cplx *)planes; a.X = (const cplx *)x; a.Y = (cplx *)y; a.W = (const cplx *)w; a.ld = lo->N; a.nrhs = nrhs; a.epi = epi;
    a.scal = scal; a.dinv = (const cplx *)dinv; a.omega_j = omega_j; a.profile = 0; a.part = (double *)op->d_part;
    a.tiles = tiles; a.ntiles = ntiles; a.f32 = sizeof(V) == sizeof(cplxf) ? 1 : 0;
    return helm_launch_apply(lo, a);
}

// One multigrid cycle on level l: u_out = approx M_l^-1 f.  fmode: levels < P->fdepth visit their coarse
// level a second time (with the coarse residual), the second visit being a plain V-cycle (truncated F-cycle).
template <class V>
int vcycle(helm_op *op, MgPrecond *P, int l, const V *f, V *u_out, int nrhs, const RhsScal *scal, bool fmode) {
    MgLevel &L = P->lv[l];
    helm_op *lo = L.op;
    const long long N = lo->N;
    hipStream_t st = op->stream;
    if (l == (int)P->lv.size() - 1) {
        dim3 grid((P->nc + 255) / 256, nrhs);
        HELM_LAUNCH((k_coarse_dense<V>), grid, dim3(256), (size_t)P->nc * sizeof(V), st, (const V *)P->d_cinvT, f, u_out, P->nc, scal);
        return HELM_OK;
    }
    // Jacobi sweeps ping-pong between two buffers; start so that the last sweep lands in u_out
    const int pingpongs = (P->nu1 - 1) + P->nu2;
    V *u = (pingpongs & 1) ? (V *)L.t : u_out, *alt = (pingpongs & 1) ? u_out : (V *)L.t;
    dim3 vg(vblocks(N), nrhs);
    int rc;
    if (P->fuse && P->nu1 == 1) {
        // first sweep from zero fused into the residual: u = w D^-1 f on the fly, r = f - M u, u stored
        rc = stencil_level<V>(op, lo, L.C, (const V *)nullptr, (V *)L.r, f, nrhs, EPI_RESID, scal, L.dinv, P->omega_j, nullptr, 0, 1, u);
        if (rc) return rc;
    } else {
        HELM_LAUNCH((k_jac0<V>), vg, dim3(256), 0, st, (const V *)L.dinv, f, u, N, P->omega_j, scal);
        for (int k = 0; k < P->nu1 - 1; ++k) {
            rc = stencil_level<V>(op, lo, L.C, u, alt, f, nrhs, EPI_JACOBI, scal, L.dinv, P->omega_j);
            if (rc) return rc;
            std::swap(u, alt);
        }
        rc = stencil_level<V>(op, lo, L.C, u, (V *)L.r, f, nrhs, EPI_RESID, scal);
        if (rc) return rc;
    }
    MgLevel &C = P->lv[l + 1];
    dim3 cg(vblocks(C.op->N), nrhs);
    HELM_LAUNCH((k_restrict<V>), cg, dim3(256), 0, st, (const V *)L.r, (V *)C.f, lo->nz, lo->nx, C.op->nz, C.op->nx, scal);
    rc = vcycle<V>(op, P, l + 1, (const V *)C.f, (V *)C.u, nrhs, scal, fmode);
    if (rc) return rc;
    if (fmode && l < P->fdepth && l + 1 < (int)P->lv.size() - 1) {
        // second visit of the coarse level: e += V(f_c - M_c e), with its own right-hand-side / result buffers
        rc = stencil_level<V>(op, C.op, C.C, (const V *)C.u, (V *)C.g, (const V *)C.f, nrhs, EPI_RESID, scal);
        if (rc) return rc;
        rc = vcycle<V>(op, P, l + 1, (const V *)C.g, (V *)C.f2, nrhs, scal, false);
        if (rc) return rc;
        HELM_LAUNCH((k_axpy1<V>), cg, dim3(256), 0, st, (const V *)C.f2, (V *)C.u, C.op->N, scal);
    }
    int k0 = 0;
    if (P->fuse && P->nu2 >= 1) {
        // coarse correction fused into the first post-smoothing sweep: input = u + P e
        rc = stencil_level<V>(op, lo, L.C, u, alt, f, nrhs, EPI_JACOBI, scal, L.dinv, P->omega_j, nullptr, 0, 2, nullptr, (const V *)C.u, C.op->nz, C.op->nx);
        if (rc) return rc;
        std::swap(u, alt);
        k0 = 1;
    } else {
        HELM_LAUNCH((k_prolong_add<V>), vg, dim3(256), 0, st, (const V *)C.u, u, lo->nz, lo->nx, C.op->nz, C.op->nx, scal);
    }
    for (int k = k0; k < P->nu2; ++k) {
        rc = stencil_level<V>(op, lo, L.C, u, alt, f, nrhs, EPI_JACOBI, scal, L.dinv, P->omega_j);
        if (rc) return rc;
        std::swap(u, alt);
    }
    if (u != u_out) HELM_FAIL(op, HELM_ERR_STATE, "multigrid ping-pong parity error");
    return HELM_OK;
}

template <class V>
int apply_typed(helm_op *op, MgPrecond *P, const cplx *in, cplx *out, int nrhs, const RhsScal *scal) {
    constexpr bool F32 = sizeof(V) == sizeof(cplxf);
    hipStream_t st = op->stream;
    const long long N = op->N;
    MgLevel &L0 = P->lv[0];
    dim3 vg(vblocks(N), nrhs);
    int rc;
    if (F32) {      // cycle in single precision between two conversions
        HELM_LAUNCH((k_convert_rhs<V, cplx>), vg, dim3(256), 0, st, in, (V *)L0.f, N, scal);
        rc = vcycle<V>(op, P, 0, (const V *)L0.f, (V *)L0.u, nrhs, scal, P->fdepth > 0);
        if (rc) return rc;
        HELM_LAUNCH((k_convert_rhs<cplx, V>), vg, dim3(256), 0, st, (const V *)L0.u, out, N, scal);
    } else {
        rc = vcycle<V>(op, P, 0, (const V *)in, (V *)out, nrhs, scal, P->fdepth > 0);
        if (rc) return rc;
    }
    // strip relaxation with the true-PML shifted operator, double precision
    const int W = P->W;
    for (int k = 0; k < P->sweeps; ++k) {
        rc = stencil_level<cplx>(op, P->sop, P->sC, out, (cplx *)P->strip_r, in, nrhs, EPI_RESID, scal, nullptr, 0.0, P->d_tiles, P->ntiles);
        if (rc) return rc;
        dim3 lg(4 * W, nrhs);
        LineFactors<cplx> zf = {(const cplx *)P->zl[0], (const cplx *)P->zl[1], (const cplx *)P->zl[2]};
        LineFactors<cplx> xf = {(const cplx *)P->xl[0], (const cplx *)P->xl[1], (const cplx *)P->xl[2]};
        HELM_LAUNCH((k_line_solve<cplx>), lg, dim3(64), 0, st, op->nz, op->nx, W, zf, xf, (const cplx *)P->strip_r, out, P->wstrip, scal);
    }
    HIP_TRY(op, hipGetLastError());
    return HELM_OK;
}

}  // namespace

int mg_apply(helm_op *op, const cplx *in, cplx *out, int nrhs, const RhsScal *scal) {
    if (op->ny > 0) return mg3_apply(op, in, out, nrhs);      // (the 3-D cycle runs on every right-hand side of the batch)
    MgPrecond *P = op->mg;
    if (!P) HELM_FAIL(op, HELM_ERR_STATE, "preconditioner not built");
    if (nrhs > P->batch) HELM_FAIL(op, HELM_ERR_ARG, "preconditioner batch too small");
    return P->f32 ? apply_typed<cplxf>(op, P, in, out, nrhs, scal) : apply_typed<cplx>(op, P, in, out, nrhs, scal);
}
