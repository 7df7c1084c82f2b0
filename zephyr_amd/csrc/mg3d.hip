// Multigrid preconditioners for the 3-D 27-point operator (right-preconditioned BiCGSTAB of capi.hip).
//
// 1. Standard cycle (first half of this file; grids below 20 points per wavelength, fallback): one V(1,1) cycle on the
//    complex-shifted operator (1/tau_M = 1/tau + omega beta / 2) with a weak absorbing layer (cPML_M), rediscretised on coarser
//    grids (model by injection, spacing doubled, layer thickness halved), damped Jacobi smoothing through the 3-D stencil
//    kernel, full-weighting restriction / trilinear prolongation, dense inverse on the coarsest grid (dense kernels of direct.hip).
// 2. Layer-preserving hierarchy (second half; oversampled grids such as BASELINE config 5): coarse grids keep every node of the
//    absorbing layers, l1-Jacobi, true layer with a small shift, block-tridiagonal direct solve of the level that still has
//    ~10 points per wavelength.  DESIGN.md section 5.3 has the measurements behind each of these choices.
#include "helm_internal.hpp"
#include <chrono>
#include <map>
#include <tuple>
#include <mutex>
#include "direct.hpp"
#include <algorithm>
#include <complex>
#include <cstring>
#include <memory>

struct Mg3Level {
    helm_op *op = nullptr;
    int nz = 0, ny = 0, nx = 0;
    long long N = 0;
    cplx *u = nullptr, *f = nullptr, *r = nullptr, *t = nullptr;      // [batch][N]
    size_t vbytes = 0;
};

struct Mg3Keep;
struct Mg3Precond {
    std::vector<Mg3Level> lv;
    Mg3Keep *keep = nullptr;      // layer-preserving hierarchy (below) instead of the standard one
    int kept_levels = 0; double ppw_direct = 0.0;     // (layer-preserving) coarsenings above the directly solved level and its points per wavelength: the class its iteration counts are booked under
    cplx *cinvT = nullptr;        // transposed dense inverse of the coarsest operator
    int nc = 0, batch = 0;
    double omega_j = 0.8, beta = 0.6, cpml_m = 30.0;
    int nu1 = 1, nu2 = 1, min_n = 8;
    bool fine32 = false;          // (layer-preserving cycle) the finest level's work vectors u, t, r hold complex64 -- see cycle_keep
};

// ---- what the layer-preserving cycle has actually needed: iterations per right-hand side, by class -----------------------------------------
// The depth decision of mg3_setup trades set-up seconds against extra iterations of the deeper hierarchy.  Round 3 priced those with three
// constants measured on config 5 (+11 / +22 / +38 at >= 8 / 6 / 5 points per wavelength on the direct level).  Now every solve through a
// layer-preserving hierarchy books its mean iteration count under (grid, coarsenings, points per wavelength of the direct level to the nearest
// 0.5, log10 rtol), and the decision uses the booked counts of both candidates where it has them; the constants remain only as the prior for a
// class that has never run in this process (the first frequency of the first job).
namespace {
struct ItKey { int nz, ny, nx, depth, ppw2, ltol; bool operator<(const ItKey &o) const { return std::tie(nz, ny, nx, depth, ppw2, ltol) < std::tie(o.nz, o.ny, o.nx, o.depth, o.ppw2, o.ltol); } };
std::mutex g_its_mu;
std::map<ItKey, std::pair<double, int>> &g_its = *new std::map<ItKey, std::pair<double, int>>();     // key -> (sum of mean iterations, solves)
ItKey it_key(const helm_op *op, int depth, double ppwd, double rtol) {
    return ItKey{op->nz, op->ny, op->nx, depth, (int)std::lround(2.0 * ppwd), (int)std::lround(-std::log10(std::max(rtol, 1e-16)))};
}
// mean iterations booked for the class, < 0 when it has never run
double its_lookup(const helm_op *op, int depth, double ppwd, double rtol) {
    std::lock_guard<std::mutex> lk(g_its_mu);
    auto it = g_its.find(it_key(op, depth, ppwd, rtol));
    return it == g_its.end() || it->second.second == 0 ? -1.0 : it->second.first / it->second.second;
}
}
// The tolerance class a set-up looks its iteration counts up under: the handle's stated tolerance (helm_set_tolerance_hint, or a solve on it), else --
// a C caller that prefactors a fresh handle without stating one -- the tolerance of the last solve booked on this grid in the process, so that
// what was recorded under the solves' real rtol is found again instead of the prior constants being used silently.
namespace {
std::map<std::tuple<int, int, int>, double> &g_last_rtol = *new std::map<std::tuple<int, int, int>, double>();       // (g_its_mu held)
double lookup_rtol(const helm_op *op) {
    if (op->rtol_hint_set) return op->rtol_hint;
    std::lock_guard<std::mutex> lk(g_its_mu);
    auto it = g_last_rtol.find(std::make_tuple(op->nz, op->ny, op->nx));
    return it == g_last_rtol.end() ? op->rtol_hint : it->second;
}
}
void mg3_record_iterations(helm_op *op, double mean_iterations, double rtol) {
    if (!op || !op->mg3 || !op->mg3->keep || !(mean_iterations > 0)) return;
    std::lock_guard<std::mutex> lk(g_its_mu);
    g_last_rtol[std::make_tuple(op->nz, op->ny, op->nx)] = rtol;
    std::pair<double, int> &e = g_its[it_key(op, op->mg3->kept_levels, op->mg3->ppw_direct, rtol)];
    e.first += mean_iterations; e.second += 1;
}

namespace {

// the four work vectors of a level ([batch][N] each) come from the size-keyed buffer pool: the next frequency's hierarchy has the same shapes
bool level_vectors(helm_op *op, Mg3Level &L, int batch) {
    L.vbytes = (size_t)batch * L.N * sizeof(cplx);
    cplx **v[4] = {&L.u, &L.f, &L.r, &L.t};
    for (int i = 0; i < 4; ++i) { *v[i] = (cplx *)helm_pool_alloc(op->device, L.vbytes); if (!*v[i]) return false; }
    return true;
}
void level_vectors_free(helm_op *op, Mg3Level &L) {
    cplx **v[4] = {&L.u, &L.f, &L.r, &L.t};
    for (int i = 0; i < 4; ++i) { if (*v[i]) helm_pool_free(op->device, *v[i], L.vbytes); *v[i] = nullptr; }
}

double envd(const char *n, double d) { const char *v = getenv(n); return v ? atof(v) : d; }
int envi(const char *n, int d) { const char *v = getenv(n); return v ? atoi(v) : d; }

template <class TU>
__global__ void k3_jac0(const cplx *__restrict__ f, const cplx *__restrict__ dinv, TU *__restrict__ u, long long N, double w) {
    const cplx *fb = f + (long long)blockIdx.y * N; TU *ub = u + (long long)blockIdx.y * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
        ub[i] = vfrom<TU>(cmul(cscale(dinv[i], w), fb[i]));
}
__device__ __forceinline__ cplx mg_cvt64(cplx a) { return a; }
__device__ __forceinline__ cplx mg_cvt64(cplxf a) { return to_f64(a); }

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
    HELM_LAUNCH(k3_jac0<cplx>, vgrid(L.N, nrhs), dim3(256), 0, st, (const cplx *)L.f, (const cplx *)L.op->d_dinv, L.u, L.N, P->omega_j);
    int rc;
    for (int s = 1; s < P->nu1; ++s) {
        rc = level_apply(op, L, L.u, L.t, L.f, nrhs, EPI_JACOBI, P->omega_j); if (rc) return rc;
        std::swap(L.u, L.t);
    }
    rc = level_apply(op, L, L.u, L.r, L.f, nrhs, EPI_RESID, 0.0); if (rc) return rc;
    HELM_LAUNCH(k3_restrict, vgrid(C.N, nrhs), dim3(256), 0, st, L.r, C.f, L.nz, L.ny, L.nx, C.nz, C.ny, C.nx);
    rc = cycle(op, P, l + 1, nrhs); if (rc) return rc;
    HELM_LAUNCH(k3_prolong_add, vgrid(L.N, nrhs), dim3(256), 0, st, C.u, L.u, L.nz, L.ny, L.nx, C.nz, C.ny, C.nx);
    for (int s = 0; s < P->nu2; ++s) {
        if (final_out && s == P->nu2 - 1) return level_apply(op, L, L.u, final_out, L.f, nrhs, EPI_JACOBI, P->omega_j);
        rc = level_apply(op, L, L.u, L.t, L.f, nrhs, EPI_JACOBI, P->omega_j); if (rc) return rc;
        std::swap(L.u, L.t);
    }
    if (final_out) HIP_TRY(op, hipMemcpyAsync(final_out, L.u, (size_t)nrhs * L.N * sizeof(cplx), hipMemcpyDeviceToDevice, st));
    return HELM_OK;
}

}  // namespace


// ================================================================================================================
// Layer-preserving hierarchy (oversampled grids: >= 20 points per wavelength).
//
// The cycle above needs a large shift and a weak layer because (a) point Jacobi DIVERGES where two stretched directions
// overlap (the directional parts of the diagonal have different complex phases and partly cancel: |lambda / d| reaches 4),
// and (b) inside a strongly stretched layer the coupling normal to the boundary is weak, so error that oscillates along the
// normal is neither smoothed nor representable on a grid coarsened in that direction.  Here instead:
//   * coarse grids keep EVERY node of the absorbing layers and halve only the interior: tensor-product grids with
//     non-uniform spacing, rediscretised with the same 27-point formula (the spacing enters the 1-D factors like a stretch),
//     per-axis interpolation / weighting tables;
//   * the smoother is l1-Jacobi (d_i = -sum_j |a_ij|, weight 1.6 = plain 0.8 in the interior);
//   * the preconditioner is the operator itself with its TRUE layer and a small shift (beta = 0.1);
//   * coarsening stops while the interior still has >= 10 points per wavelength and that level is solved directly:
//     block-tridiagonal elimination over the planes normal to the longest axis, dense plane inverses in HBM
//     (8.7 GB in single precision for the 79 x 79 x 47 level of config 5), applied as split-K batched GEMMs.
// numpy prototype (96 x 96 x 64, 40 / 100 points per wavelength): 9 / 7 BiCGSTAB iterations against 219 / 811 for the recipe above.
// ================================================================================================================
struct Ax3 {
    std::vector<double> x, gam;     // node coordinates, damping gamma at the nodes
    std::vector<char> lay;          // node belongs to an absorbing layer (never dropped)
    int n() const { return (int)x.size(); }
};
struct PTab { int c0, c1; double w0, w1; };         // fine node -> its two coarse nodes and weights (kept node: c0 = c1, w = 1, 0)
struct RTab { int f; double wl, wc, wr; };          // coarse node -> fine nodes f-1, f, f+1 with normalised weights

struct Bt3 {                        // direct solver of the coarsest level
    int axis = 0, np = 0, na = 0, nb = 0, m = 0, mpad = 0, ksplit = 1, kc = 0, batch = 0, nparts = 1;
    int mid = 0;                    // twisted elimination: planes 0 .. mid-1 from the left, np-1 .. mid+1 from the right, plane mid last
    bool own = true;                // k_bt_apply (memory-bound product) instead of the generic batched GEMM
    int device = 0; size_t tbytes = 0;   // Tinv comes from the size-keyed buffer pool (the next frequency takes it over without a hipMalloc)
    long long ss = 0, sa = 0, sb = 0, N = 0;      // node strides of the sweep axis / the two in-plane axes
    cplx *Tinv = nullptr;           // np x (mpad x m): inverse of the transposed Schur complement of plane k (rows >= m are zero)
    float2 *Tinv32 = nullptr;       // single-precision copy, np x (m x ld32), used INSTEAD of Tinv (f32 = true: Tinv is then not kept)
    bool f32 = false; int ld32 = 0; size_t tbytes32 = 0;
    cplx *Y[2] = {nullptr, nullptr};      // per chain: batch x mpad, packed right-hand side of one plane (columns >= m stay zero)
    cplx *Z = nullptr;                    // np x batch x m: forward-substituted planes, then the solution
    cplx *parts[2] = {nullptr, nullptr};  // per chain: ksplit x batch x m partial products of the split-K product
    helm_op *aux = nullptr;         // carries the stream (and the look-ahead stream) of the right-hand chain
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};

// Nested-dissection alternative to Bt3 (HELM_MG3_COARSE=nd): the multifrontal solver of the 2-D path (direct.hip) run over the (ny, nx) grid of
// z-columns of the level -- a "cell" is a column of nz unknowns (NdPlan::dof = nz), its 27-point coupling to the nine neighbour columns a
// block-tridiagonal nz x nz block.  The top separator is one plane of the level (the size Bt3 inverts np times); below it the fronts shrink.
struct Nd3 {
    std::shared_ptr<NdPlanDev> pd;
    NdFactor *f = nullptr;
    cplx *ws = nullptr; size_t ws_bytes = 0;        // solve scratch (nd_solve_ws_elems), from the pool
    int device = 0, batch = 0;
    bool on() const { return f != nullptr; }
};

struct Mg3Keep {
    std::vector<cplx *> dl1;                         // per level: l1-Jacobi inverse diagonal
    std::vector<size_t> dl1_bytes; int device = 0;   // (everything here comes from the size-keyed pool: hipMalloc / hipFree beside another handle's solve stall)
    std::vector<std::pair<void *, size_t>> tabs;     // the transfer tables' buffers
    std::vector<PTab *> pt[3]; std::vector<RTab *> rt[3];   // per transfer (level l -> l+1) and axis (z, y, x): device tables
    Bt3 bt;
    Nd3 nd;
    double omega_l1 = 1.6;
};

namespace {

void coarsen_axis(const Ax3 &a, bool keep_layer, Ax3 &c, std::vector<int> &kept, std::vector<PTab> &pt, std::vector<RTab> &rt) {
    const int n = a.n();
    std::vector<char> keep(n, 0);
    if (keep_layer) {
        for (int i = 0; i < n;) {
            if (a.lay[i]) { keep[i] = 1; ++i; continue; }
            int j = i;
            while (j < n && !a.lay[j]) ++j;
            for (int t = i; t < j; ++t) keep[t] = (char)((t - i) & 1);      // the first node of an interior run is dropped
            i = j;
        }
    } else {
        for (int i = 0; i < n; ++i) keep[i] = (char)!(i & 1);
    }
    keep[0] = keep[n - 1] = 1;
    for (int i = 1; i + 1 < n; ++i) if (!keep[i] && !(keep[i - 1] && keep[i + 1])) keep[i] = 1;   // a dropped node interpolates from kept neighbours
    std::vector<int> cmap(n, -1);
    c = Ax3(); kept.clear();
    for (int i = 0; i < n; ++i) if (keep[i]) {
        cmap[i] = (int)kept.size(); kept.push_back(i);
        c.x.push_back(a.x[i]); c.gam.push_back(a.gam[i]); c.lay.push_back(a.lay[i]);
    }
    pt.resize(n);
    for (int i = 0; i < n; ++i) {
        if (keep[i]) { pt[i].c0 = pt[i].c1 = cmap[i]; pt[i].w0 = 1.0; pt[i].w1 = 0.0; continue; }
        const double da = a.x[i] - a.x[i - 1], db = a.x[i + 1] - a.x[i];
        pt[i].c0 = cmap[i - 1]; pt[i].c1 = cmap[i + 1]; pt[i].w0 = db / (da + db); pt[i].w1 = da / (da + db);
    }
    rt.resize(kept.size());
    for (size_t I = 0; I < kept.size(); ++I) {
        const int f = kept[I];
        double wl = (f > 0 && !keep[f - 1]) ? pt[f - 1].w1 : 0.0, wr = (f + 1 < n && !keep[f + 1]) ? pt[f + 1].w0 : 0.0;
        const double s = 1.0 + wl + wr;
        rt[I].f = f; rt[I].wl = wl / s; rt[I].wc = 1.0 / s; rt[I].wr = wr / s;
    }
}

// 1-D factors L(-1), L(0), L(+1) of d/dx (1/xi) d/dx / xi on a non-uniform axis, 1/h^2 included (for uniform spacing h this is
// profile3() of helm3d.hip divided by h^2)
void lap_from_axis(const Ax3 &a, std::complex<double> om, std::vector<cplx> &Lt) {
    const int n = a.n();
    auto xi = [&](int i) { i = std::min(std::max(i, 0), n - 1); return 1.0 - std::complex<double>(0.0, a.gam[i]) / om; };
    Lt.resize((size_t)3 * n);
    for (int i = 0; i < n; ++i) {
        const double hm = i > 0 ? a.x[i] - a.x[i - 1] : a.x[1] - a.x[0], hp = i + 1 < n ? a.x[i + 1] - a.x[i] : a.x[n - 1] - a.x[n - 2];
        const double hbar = 0.5 * (hm + hp);
        const std::complex<double> c = xi(i);
        const std::complex<double> lm = 1.0 / (c * hbar * (c + xi(i - 1)) * 0.5 * hm), lp = 1.0 / (c * hbar * (c + xi(i + 1)) * 0.5 * hp);
        const std::complex<double> l0 = -(lm + lp);
        Lt[i] = cmake(lm.real(), lm.imag());
        Lt[(size_t)n + i] = cmake(l0.real(), l0.imag());
        Lt[(size_t)2 * n + i] = cmake(lp.real(), lp.imag());
    }
}

// l1-Jacobi: 1 / d with d = -sum_k |a_k| (the centre coefficient of this operator is negative real in the interior); identity rows
// (the box boundary) get 1 / w so that the weighted step is exact
__global__ void k3_l1_dinv(const cplx *__restrict__ planes, cplx *__restrict__ dl1, long long N, double w) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int k = 0; k < 27; ++k) { const cplx v = planes[(long long)k * N + i]; s += sqrt(v.x * v.x + v.y * v.y); }
        const cplx c = planes[13LL * N + i];
        const double ac = sqrt(c.x * c.x + c.y * c.y);
        if (s <= ac * (1.0 + 1e-14)) { const double r = 1.0 / (w * ac * ac); dl1[i] = cmake(c.x * r, -c.y * r); }
        else dl1[i] = cmake(-1.0 / s, 0.0);
    }
}

// (TF: element type of the fine vector -- complex64 for the finest level of a cycle that keeps its work vectors in single precision)
template <class TF>
__global__ void k3_restrict_t(const TF *__restrict__ fine, cplx *__restrict__ coarse, int ny, int nx, int nzc, int nyc, int nxc, long long Nf,
                              const RTab *__restrict__ tz, const RTab *__restrict__ ty, const RTab *__restrict__ tx) {
    const long long Nc = (long long)nzc * nyc * nxc;
    const TF *fb = fine + (long long)blockIdx.y * Nf; cplx *cb = coarse + (long long)blockIdx.y * Nc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < Nc; i += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(i % nxc), Y = (int)((i / nxc) % nyc), Z = (int)(i / ((long long)nxc * nyc));
        const RTab rz = tz[Z], ry = ty[Y], rx = tx[X];
        const double wz[3] = {rz.wl, rz.wc, rz.wr}, wy[3] = {ry.wl, ry.wc, ry.wr}, wx[3] = {rx.wl, rx.wc, rx.wr};
        cplx acc = cmake(0.0, 0.0);
        for (int a = 0; a < 3; ++a) { if (wz[a] == 0.0) continue;
            for (int b = 0; b < 3; ++b) { if (wy[b] == 0.0) continue;
                const double wab = wz[a] * wy[b];
                const TF *row = fb + ((long long)(rz.f + a - 1) * ny + (ry.f + b - 1)) * nx + rx.f;
                for (int c = 0; c < 3; ++c) { if (wx[c] == 0.0) continue;
                    const cplx v = mg_cvt64(row[c - 1]);
                    const double w = wab * wx[c];
                    acc.x += w * v.x; acc.y += w * v.y;
                } } }
        cb[i] = acc;
    }
}

template <class TF>
__global__ void k3_prolong_add_t(const cplx *__restrict__ coarse, TF *__restrict__ fine, int nz, int ny, int nx, int nyc, int nxc, long long Nc,
                                 const PTab *__restrict__ tz, const PTab *__restrict__ ty, const PTab *__restrict__ tx) {
    const long long Nf = (long long)nz * ny * nx;
    TF *fb = fine + (long long)blockIdx.y * Nf; const cplx *cb = coarse + (long long)blockIdx.y * Nc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < Nf; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((long long)nx * ny));
        const PTab pz = tz[z], py = ty[y], px = tx[x];
        const int cz[2] = {pz.c0, pz.c1}, cy[2] = {py.c0, py.c1}, cx[2] = {px.c0, px.c1};
        const double wz[2] = {pz.w0, pz.w1}, wy[2] = {py.w0, py.w1}, wx[2] = {px.w0, px.w1};
        cplx acc = cmake(0.0, 0.0);
        for (int a = 0; a < 2; ++a) { if (wz[a] == 0.0) continue;
            for (int b = 0; b < 2; ++b) { if (wy[b] == 0.0) continue;
                for (int c = 0; c < 2; ++c) { if (wx[c] == 0.0) continue;
                    const cplx v = cb[((long long)cz[a] * nyc + cy[b]) * nxc + cx[c]];
                    const double w = wz[a] * wy[b] * wx[c];
                    acc.x += w * v.x; acc.y += w * v.y;
                } } }
        fb[i] = vfrom<TF>(cadd(mg_cvt64(fb[i]), acc));
    }
}

// Galerkin coarse operator A_c = R A_f P of a 27-point fine operator with the tensor-product transfers of the tables: again 27-point.
// One thread per coarse node (a set-up kernel: the 27 accumulators are indexed dynamically and live in scratch).  Used for the directly
// solved level only: at 5-8 points per wavelength the rediscretised operator carries waves of a different length than the level above
// (numpy prototype, 5 points: 38 instead of 87 iterations).
__global__ __launch_bounds__(256) void k3_galerkin(const cplx *__restrict__ pf, int nzf, int nyf, int nxf, cplx *__restrict__ pc, int nzc, int nyc, int nxc,
                                                   const RTab *__restrict__ rz_, const RTab *__restrict__ ry_, const RTab *__restrict__ rx_,
                                                   const PTab *__restrict__ pz_, const PTab *__restrict__ py_, const PTab *__restrict__ px_) {
    const long long Nc = (long long)nzc * nyc * nxc, Nf = (long long)nzf * nyf * nxf;
    const long long I = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= Nc) return;
    const int X = (int)(I % nxc), Y = (int)((I / nxc) % nyc), Z = (int)(I / ((long long)nxc * nyc));
    cplx acc[27];
    for (int k = 0; k < 27; ++k) acc[k] = cmake(0.0, 0.0);
    const RTab rz = rz_[Z], ry = ry_[Y], rx = rx_[X];
    const double wz[3] = {rz.wl, rz.wc, rz.wr}, wy[3] = {ry.wl, ry.wc, ry.wr}, wx[3] = {rx.wl, rx.wc, rx.wr};
    for (int a = 0; a < 3; ++a) { if (wz[a] == 0.0) continue; const int iz = rz.f + a - 1;
        for (int b = 0; b < 3; ++b) { if (wy[b] == 0.0) continue; const int iy = ry.f + b - 1;
            for (int c = 0; c < 3; ++c) { if (wx[c] == 0.0) continue; const int ix = rx.f + c - 1;
                const double wr = wz[a] * wy[b] * wx[c];
                const long long i = ((long long)iz * nyf + iy) * nxf + ix;
                for (int k = 0; k < 27; ++k) {
                    const cplx cf = pf[(long long)k * Nf + i];
                    if (cf.x == 0.0 && cf.y == 0.0) continue;
                    const int jz = iz + k / 9 - 1, jy = iy + (k / 3) % 3 - 1, jx = ix + k % 3 - 1;
                    if (jz < 0 || jz >= nzf || jy < 0 || jy >= nyf || jx < 0 || jx >= nxf) continue;
                    const PTab qz = pz_[jz], qy = py_[jy], qx = px_[jx];
                    const int cz[2] = {qz.c0, qz.c1}, cy[2] = {qy.c0, qy.c1}, cx[2] = {qx.c0, qx.c1};
                    const double vz[2] = {qz.w0, qz.w1}, vy[2] = {qy.w0, qy.w1}, vx[2] = {qx.w0, qx.w1};
                    for (int ua = 0; ua < 2; ++ua) { if (vz[ua] == 0.0) continue; const int dz = cz[ua] - Z; if (dz < -1 || dz > 1) continue;
                        for (int ub = 0; ub < 2; ++ub) { if (vy[ub] == 0.0) continue; const int dy = cy[ub] - Y; if (dy < -1 || dy > 1) continue;
                            for (int uc = 0; uc < 2; ++uc) { if (vx[uc] == 0.0) continue; const int dx = cx[uc] - X; if (dx < -1 || dx > 1) continue;
                                const double w = wr * vz[ua] * vy[ub] * vx[uc];
                                cplx &t = acc[9 * (dz + 1) + 3 * (dy + 1) + (dx + 1)];
                                t.x += w * cf.x; t.y += w * cf.y;
                            } } }
                }
            } } }
    for (int k = 0; k < 27; ++k) pc[(long long)k * Nc + I] = acc[k];
}

// ---- block-tridiagonal direct solver of the coarsest level ------------------------------------------------------------
__device__ __forceinline__ int bt_slot(int axis, int os, int da, int db) {
    const int oz = axis == 0 ? os : da, oy = axis == 0 ? da : (axis == 1 ? os : db), ox = axis == 2 ? os : db;
    return 9 * (oz + 1) + 3 * (oy + 1) + (ox + 1);
}

struct BtGeom { int axis, np, na, nb, m; long long ss, sa, sb, N; };

// T_k = S_k^T with S_k = A_kk - sum over the eliminated neighbour planes k + d (d = -1 and / or +1) of A_{k,k+d} S_{k+d}^{-1} A_{k+d,k};
// Tm / Tp = T_{k-1}^{-1} / T_{k+1}^{-1} or null (element [b][a] of T^{-1} is S^{-1}[a][b]).
// One thread per entry, i (the row of S) fastest: coalesced writes of T[j][i] and reads of T^{-1}[b][a ~ i].
__global__ __launch_bounds__(256) void k_bt_schur_t(const cplx *__restrict__ planes, BtGeom g, int k, const cplx *__restrict__ Tm, const cplx *__restrict__ Tp,
                                                    cplx *__restrict__ T) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)g.m * g.m) return;
    const int i = (int)(e % g.m), j = (int)(e / g.m);
    const int ia = i / g.nb, ib = i % g.nb, ja = j / g.nb, jb = j % g.nb;
    const long long node_i = (long long)k * g.ss + ia * g.sa + ib * g.sb;
    cplx v = cmake(0.0, 0.0);
    if (abs(ja - ia) <= 1 && abs(jb - ib) <= 1) v = planes[(long long)bt_slot(g.axis, 0, ja - ia, jb - ib) * g.N + node_i];
    for (int side = 0; side < 2; ++side) {
        const cplx *Tn = side ? Tp : Tm;
        if (!Tn) continue;
        const int d = side ? 1 : -1;
        // the nine entries of row i of A_{k,k+d} (kept in registers) and of column j of A_{k+d,k}
        cplx am9[9]; int off9[9];
        #pragma unroll
        for (int d1 = 0; d1 < 9; ++d1) {
            const int aa = ia + (d1 / 3 - 1), ab = ib + (d1 % 3 - 1);
            const bool in = aa >= 0 && aa < g.na && ab >= 0 && ab < g.nb;
            off9[d1] = in ? aa * g.nb + ab : -1;
            am9[d1] = in ? planes[(long long)bt_slot(g.axis, d, d1 / 3 - 1, d1 % 3 - 1) * g.N + node_i] : cmake(0.0, 0.0);
        }
        for (int d2 = 0; d2 < 9; ++d2) {
            const int ba = ja - (d2 / 3 - 1), bb = jb - (d2 % 3 - 1);
            if (ba < 0 || ba >= g.na || bb < 0 || bb >= g.nb) continue;
            const long long node_b = (long long)(k + d) * g.ss + ba * g.sa + bb * g.sb;
            const cplx ap = planes[(long long)bt_slot(g.axis, -d, d2 / 3 - 1, d2 % 3 - 1) * g.N + node_b];
            if (ap.x == 0.0 && ap.y == 0.0) continue;
            const cplx *trow = Tn + (long long)(ba * g.nb + bb) * g.m;
            cplx acc = cmake(0.0, 0.0);
            #pragma unroll
            for (int d1 = 0; d1 < 9; ++d1) if (off9[d1] >= 0) cfma(acc, am9[d1], trow[off9[d1]]);
            v = csub(v, cmul(acc, ap));
        }
    }
    T[(long long)j * g.m + i] = v;
}

// packed right-hand side of plane k:  Y = [f_k] - A_{k,k-1} Zm - A_{k,k+1} Zp  (each neighbour optional; without f the sign is +:
// Y = A_{k,k-1} Zm + A_{k,k+1} Zp, the back-substitution term)
__global__ __launch_bounds__(256) void k_bt_rhs(const cplx *__restrict__ planes, BtGeom g, int k, const cplx *__restrict__ f, const cplx *__restrict__ Zm,
                                                const cplx *__restrict__ Zp, cplx *__restrict__ Y, int mpad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (i >= g.m) return;
    const int ia = i / g.nb, ib = i % g.nb;
    const long long node = (long long)k * g.ss + ia * g.sa + ib * g.sb;
    cplx v = cmake(0.0, 0.0);
    for (int side = 0; side < 2; ++side) {
        const cplx *Zn = side ? Zp : Zm;
        if (!Zn) continue;
        const cplx *zr = Zn + (long long)r * g.m;
        for (int d = 0; d < 9; ++d) {
            const int aa = ia + (d / 3 - 1), ab = ib + (d % 3 - 1);
            if (aa < 0 || aa >= g.na || ab < 0 || ab >= g.nb) continue;
            cfma(v, planes[(long long)bt_slot(g.axis, side ? 1 : -1, d / 3 - 1, d % 3 - 1) * g.N + node], zr[aa * g.nb + ab]);
        }
    }
    if (f) v = csub(f[(long long)r * g.N + node], v);
    Y[(long long)r * mpad + i] = v;
}

// parts[ks][r][c] = sum over the k rows of chunk ks of Y[r][k] T[k][c]   (r < 16 right-hand sides).
// The plane inverses are read once per solve and nothing else is: a memory-bound product (8 flop per byte at 16 right-hand sides).
// A workgroup takes 128 columns and one K chunk; every lane owns two columns (c, c + 64) so that one LDS broadcast of a right-hand-side
// value feeds two multiply-adds (with one column per lane the 16 broadcasts per row bound the kernel at 2.8 TB/s); wave w streams rows
// w, w + 4, ... of T (two coalesced 1-KB segments per row) and the four waves add their partial sums through LDS at the end.
#define BTA_KS 128
template <int NR>
__global__ __launch_bounds__(256, 2) void k_bt_apply(const cplx *__restrict__ Y, int ldy, const cplx *__restrict__ T, int m, int kc, int nrhs,
                                                     cplx *__restrict__ parts) {
    __shared__ cplx ys[BTA_KS][NR];                      // 32 KB; doubles as the reduction buffer (3 waves x 8 values x 64 lanes = 24 KB)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 128 + lane, c1 = c0 + 64;
    const int kbeg = blockIdx.y * kc, kend = min(m, kbeg + kc);
    cplx acc0[NR], acc1[NR];
    #pragma unroll
    for (int r = 0; r < NR; ++r) { acc0[r] = cmake(0.0, 0.0); acc1[r] = cmake(0.0, 0.0); }
    const bool live0 = c0 < m, live1 = c1 < m;
    for (int k0 = kbeg; k0 < kend; k0 += BTA_KS) {
        __syncthreads();
        for (int e = threadIdx.x; e < BTA_KS * NR; e += 256) {
            const int kk = e % BTA_KS, r = e / BTA_KS;
            ys[kk][r] = (r < nrhs && k0 + kk < kend) ? Y[(long long)r * ldy + k0 + kk] : cmake(0.0, 0.0);
        }
        __syncthreads();
        const int kn = min(BTA_KS, kend - k0);
        for (int kk = w; kk < kn; kk += 16) {            // four rows of this wave per step: eight 16-byte loads in flight per lane
            cplx t0[4], t1[4];
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kk + 4 * u;
                const cplx *row = T + (long long)(k0 + k) * m;
                t0[u] = (live0 && k < kn) ? row[c0] : cmake(0.0, 0.0);
                t1[u] = (live1 && k < kn) ? row[c1] : cmake(0.0, 0.0);
            }
            #pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = min(kk + 4 * u, BTA_KS - 1);
                #pragma unroll
                for (int r = 0; r < NR; ++r) { const cplx y = ys[k][r]; cfma(acc0[r], y, t0[u]); cfma(acc1[r], y, t1[u]); }
            }
        }
    }
    // waves 1-3 hand their sums to wave 0, eight values per lane and round
    cplx *red = &ys[0][0];
    #pragma unroll
    for (int half = 0; half < 2; ++half) {
        #pragma unroll
        for (int g = 0; g < NR; g += 8) {
            __syncthreads();
            if (w > 0) {
                #pragma unroll
                for (int r = 0; r < 8; ++r) red[((w - 1) * 8 + r) * 64 + lane] = half ? acc1[g + r] : acc0[g + r];
            }
            __syncthreads();
            if (w == 0) {
                #pragma unroll
                for (int r = 0; r < 8; ++r) {
                    cplx v = half ? acc1[g + r] : acc0[g + r];
                    #pragma unroll
                    for (int q = 0; q < 3; ++q) v = cadd(v, red[(q * 8 + r) * 64 + lane]);
                    if (half) acc1[g + r] = v; else acc0[g + r] = v;
                }
            }
        }
    }
    if (w != 0) return;
    cplx *out = parts + ((long long)blockIdx.y * nrhs) * m;
    #pragma unroll
    for (int r = 0; r < NR; ++r) if (r < nrhs) {
        if (live0) out[(long long)r * m + c0] = acc0[r];
        if (live1) out[(long long)r * m + c1] = acc1[r];
    }
}

// Single-precision variant: the plane inverses are stored as float2 (half the bytes, half the 17 GB) and the products run in fp32 --
// the cycle is a preconditioner, its coarse solve does not need more than ~1e-5.  A lane owns two ADJACENT columns (one 16-byte load).
template <int NR>
__global__ __launch_bounds__(256, 2) void k_bt_apply32(const cplx *__restrict__ Y, int ldy, const float2 *__restrict__ T, int m, int ld, int kc, int nrhs,
                                                       cplx *__restrict__ parts) {
    __shared__ float2 ys[BTA_KS][NR];                    // 16 KB; doubles as the reduction buffer (3 waves x 8 values x 64 lanes = 12 KB)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 128 + 2 * lane, c1 = c0 + 1;
    const int kbeg = blockIdx.y * kc, kend = min(m, kbeg + kc);
    float2 acc0[NR], acc1[NR];
    #pragma unroll
    for (int r = 0; r < NR; ++r) { acc0[r] = make_float2(0.f, 0.f); acc1[r] = make_float2(0.f, 0.f); }
    const bool live0 = c0 < m, live1 = c1 < m;
    for (int k0 = kbeg; k0 < kend; k0 += BTA_KS) {
        __syncthreads();
        for (int e = threadIdx.x; e < BTA_KS * NR; e += 256) {
            const int kk = e % BTA_KS, r = e / BTA_KS;
            const cplx v = (r < nrhs && k0 + kk < kend) ? Y[(long long)r * ldy + k0 + kk] : cmake(0.0, 0.0);
            ys[kk][r] = make_float2((float)v.x, (float)v.y);
        }
        __syncthreads();
        const int kn = min(BTA_KS, kend - k0);
        for (int kk = w; kk < kn; kk += 32) {            // eight rows of this wave per step
            float4 t[8];
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = kk + 4 * u;
                t[u] = (live0 && k < kn) ? *reinterpret_cast<const float4 *>(T + (long long)(k0 + k) * ld + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = min(kk + 4 * u, BTA_KS - 1);
                #pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const float2 y = ys[k][r];
                    acc0[r].x = fmaf(y.x, t[u].x, acc0[r].x); acc0[r].x = fmaf(-y.y, t[u].y, acc0[r].x);
                    acc0[r].y = fmaf(y.x, t[u].y, acc0[r].y); acc0[r].y = fmaf(y.y, t[u].x, acc0[r].y);
                    acc1[r].x = fmaf(y.x, t[u].z, acc1[r].x); acc1[r].x = fmaf(-y.y, t[u].w, acc1[r].x);
                    acc1[r].y = fmaf(y.x, t[u].w, acc1[r].y); acc1[r].y = fmaf(y.y, t[u].z, acc1[r].y);
                }
            }
        }
    }
    float2 *red = &ys[0][0];
    #pragma unroll
    for (int half = 0; half < 2; ++half) {
        #pragma unroll
        for (int g = 0; g < NR; g += 8) {
            __syncthreads();
            if (w > 0) {
                #pragma unroll
                for (int r = 0; r < 8; ++r) red[((w - 1) * 8 + r) * 64 + lane] = half ? acc1[g + r] : acc0[g + r];
            }
            __syncthreads();
            if (w == 0) {
                #pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float2 v = half ? acc1[g + r] : acc0[g + r];
                    #pragma unroll
                    for (int q = 0; q < 3; ++q) { const float2 o = red[(q * 8 + r) * 64 + lane]; v.x += o.x; v.y += o.y; }
                    if (half) acc1[g + r] = v; else acc0[g + r] = v;
                }
            }
        }
    }
    if (w != 0) return;
    cplx *out = parts + ((long long)blockIdx.y * nrhs) * m;
    #pragma unroll
    for (int r = 0; r < NR; ++r) if (r < nrhs) {
        if (live0) out[(long long)r * m + c0] = cmake((double)acc0[r].x, (double)acc0[r].y);
        if (live1) out[(long long)r * m + c1] = cmake((double)acc1[r].x, (double)acc1[r].y);
    }
}

__global__ void k_bt_to_f32(const cplx *__restrict__ T, float2 *__restrict__ T32, int m, int ld) {
    const long long n = (long long)m * ld;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / ld), c = (int)(e % ld);
        const cplx v = c < m ? T[(long long)r * m + c] : cmake(0.0, 0.0);
        T32[e] = make_float2((float)v.x, (float)v.y);
    }
}

// Z (+)= sum of the split-K partial products: sub = 0: Z = sum, 1: Z -= sum
__global__ void k_bt_reduce(const cplx *__restrict__ parts, int nparts, long long n, cplx *__restrict__ Z, int sub) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        cplx s = parts[e];
        for (int p = 1; p < nparts; ++p) s = cadd(s, parts[(long long)p * n + e]);
        Z[e] = sub ? csub(Z[e], s) : s;
    }
}

__global__ void k_bt_scatter(const cplx *__restrict__ Z, BtGeom g, int nrhs, cplx *__restrict__ u) {
    const long long tot = (long long)g.np * g.m;
    const int r = blockIdx.y;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(e / g.m), i = (int)(e % g.m);
        const long long node = (long long)k * g.ss + (i / g.nb) * g.sa + (i % g.nb) * g.sb;
        u[(long long)r * g.N + node] = Z[((long long)k * nrhs + r) * g.m + i];
    }
}

BtGeom bt_geom(const Bt3 &B) { BtGeom g; g.axis = B.axis; g.np = B.np; g.na = B.na; g.nb = B.nb; g.m = B.m; g.ss = B.ss; g.sa = B.sa; g.sb = B.sb; g.N = B.N; return g; }

void bt_free(Bt3 &B) {
    if (B.Tinv) helm_pool_free(B.device, B.Tinv, B.tbytes);
    if (B.Tinv32) helm_pool_free(B.device, B.Tinv32, B.tbytes32);
    for (int c = 0; c < 2; ++c) { hipFree(B.Y[c]); hipFree(B.parts[c]); }
    hipFree(B.Z);
    for (int e = 0; e < 3; ++e) if (B.ev[e]) hipEventDestroy(B.ev[e]);
    if (B.aux) helm_destroy(B.aux);
    B = Bt3();
}

// Twisted block elimination: the planes left of `mid` are eliminated left to right, those right of it right to left, plane mid last.
// The two chains are independent, so they run on two streams (set-up: two dense inversions in flight, whose latency-bound pivot and
// panel steps fill each other's gaps; solve: two half-length chains of small launches).
int bt_setup(helm_op *op, Bt3 &B, const Mg3Level &L, int batch) {
    const int dims[3] = {L.nz, L.ny, L.nx};
    const long long strides[3] = {(long long)L.ny * L.nx, L.nx, 1};
    int axis = 0;
    for (int a = 1; a < 3; ++a) if (dims[a] > dims[axis]) axis = a;       // planes normal to the longest axis are the smallest
    const int ia = axis == 0 ? 1 : 0, ib = axis == 2 ? 1 : 2;
    B.axis = axis; B.np = dims[axis]; B.na = dims[ia]; B.nb = dims[ib]; B.m = B.na * B.nb;
    B.ss = strides[axis]; B.sa = strides[ia]; B.sb = strides[ib]; B.N = L.N; B.batch = batch;
    B.mid = envi("HELM_MG3_BT_TWIST", 1) ? B.np / 2 : B.np - 1;
    // split-K: ~512 workgroups of 128 columns each (k_bt_apply); HELM_MG3_BT_GEMM=1 goes through the generic batched GEMM instead
    B.own = batch <= 16;
    B.ksplit = B.own ? std::max(1, std::min(16, 512 / ((B.m + 127) / 128))) : std::max(1, std::min(16, 255 / ((B.m + 63) / 64)));
    B.kc = (B.m + B.ksplit - 1) / B.ksplit;
    B.mpad = B.own ? B.m : B.kc * B.ksplit;          // (the generic GEMM wants equal K chunks: zero rows / columns up to mpad)
    B.nparts = B.ksplit;
    B.device = op->device;
    // single-precision plane inverses (default): only two double-precision planes per chain exist at a time during the set-up
    B.f32 = B.own && helm_tuning_now().mg3_bt_f32 != 0;
    B.ld32 = (B.m + 1) & ~1;
    const long long mm = (long long)B.m * B.m;
    const size_t wbytes = (size_t)mm * sizeof(cplx);
    const size_t tb = B.f32 ? 4 * wbytes : (size_t)B.np * B.mpad * B.m * sizeof(cplx);
    B.tbytes = tb;
    B.tbytes32 = B.f32 ? (size_t)B.np * B.m * B.ld32 * sizeof(float2) : 0;
    {   // leave room for the Krylov workspace: the plane inverses may take a third of the device memory (HELM_MG3_BT_MAXGB overrides)
        size_t freeb = 0, totb = 0;
        hipMemGetInfo(&freeb, &totb);
        freeb += helm_pool_idle_bytes(op->device);       // (r4: idle buffers of the library's own pool are available to it)
        const double cap = std::min(totb / 3.0, 0.95 * (double)freeb);     // (free memory: several 3-D handles may be alive)
        if ((double)(tb + B.tbytes32) > cap)
            HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "3-D multigrid: the plane inverses of the directly solved level (%.1f GB) exceed the budget of %.1f GB", (tb + B.tbytes32) / 1e9, cap / 1e9);
    }
    hipStreamSynchronize(op->stream);          // (buffers of the previous frequency go back to the pool only when their work is done)
    B.aux = helm_create3d(op->device, 3, 3, 3, 1.0, 1.0, 1.0, 2);
    if (!B.aux) HELM_FAIL(op, HELM_ERR_DEVICE, "%s", helm_last_error(nullptr));
    // set-up: two inversions in flight pay while they are latency-bound (m = 1617 at 2 Hz: 0.37 -> 0.30 s); two saturating ones only get in each
    // other's way (m = 3713: 1.49 -> 1.93 s), so from the size at which the look-ahead Gauss-Jordan takes over both chains share one stream
    const bool conc = B.m < 2048;
    hipStream_t sts[2] = {op->stream, conc ? B.aux->stream : op->stream};
    helm_op *ctx[2] = {op, conc ? B.aux : op};
    B.Tinv = (cplx *)helm_pool_alloc(op->device, tb);
    if (B.f32) B.Tinv32 = (float2 *)helm_pool_alloc(op->device, B.tbytes32);
    cplx *W[2] = {(cplx *)helm_pool_alloc(op->device, wbytes), (cplx *)helm_pool_alloc(op->device, wbytes)};
    bool ok = B.Tinv && W[0] && W[1] && (!B.f32 || B.Tinv32) && helm_malloc_retry(op->device, (void **)&B.Z, (size_t)B.np * batch * B.m * sizeof(cplx)) == hipSuccess;
    for (int c = 0; c < 2 && ok; ++c)
        ok = helm_malloc_retry(op->device, (void **)&B.Y[c], (size_t)batch * B.mpad * sizeof(cplx)) == hipSuccess &&
             helm_malloc_retry(op->device, (void **)&B.parts[c], (size_t)B.nparts * batch * B.m * sizeof(cplx)) == hipSuccess;
    for (int e = 0; e < 3 && ok; ++e) ok = hipEventCreateWithFlags(&B.ev[e], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        for (int c = 0; c < 2; ++c) if (W[c]) helm_pool_free(op->device, W[c], wbytes);
        bt_free(B);
        HELM_FAIL(op, HELM_ERR_DEVICE, "3-D multigrid: the plane inverses of the coarsest level (%.1f GB) do not fit", (tb + B.tbytes32) / 1e9);
    }
    if (!B.f32 && B.mpad != B.m) hipMemsetAsync(B.Tinv, 0, tb, op->stream);
    for (int c = 0; c < 2; ++c) hipMemsetAsync(B.Y[c], 0, (size_t)batch * B.mpad * sizeof(cplx), op->stream);
    hipEventRecord(B.ev[0], op->stream);
    hipStreamWaitEvent(sts[1], B.ev[0], 0);
    const BtGeom g = bt_geom(B);
    const cplx *planes = L.op->d_C;
    const unsigned sg = (unsigned)((mm + 255) / 256);
    // where the (transposed, double-precision) Schur complement of plane k lives: its own slot, or a ping-pong pair per chain (f32)
    auto slot = [&](int chain, int step, int k) { return B.f32 ? B.Tinv + ((long long)chain * 2 + (step & 1)) * mm : B.Tinv + (long long)k * B.mpad * B.m; };
    auto finish = [&](int chain, cplx *Tk, int k) -> int {
        const int rc = nd_dense_inverse(ctx[chain], Tk, B.m, W[chain]);
        if (rc) { helm_set_error(op, helm_last_error(ctx[chain])); return rc; }
        if (B.f32) HELM_LAUNCH(k_bt_to_f32, dim3(4096), dim3(256), 0, sts[chain], (const cplx *)Tk, B.Tinv32 + (long long)k * B.m * B.ld32, B.m, B.ld32);
        return HELM_OK;
    };
    int rc = HELM_OK;
    const int nl = B.mid, nr = B.np - 1 - B.mid;
    const cplx *lastL = nullptr, *lastR = nullptr;
    for (int step = 0; step < std::max(nl, nr) && !rc; ++step) {        // launches of the two chains interleaved
        if (step < nl) {
            const int k = step;
            cplx *Tk = slot(0, step, k);
            HELM_LAUNCH(k_bt_schur_t, dim3(sg), dim3(256), 0, sts[0], planes, g, k, lastL, (const cplx *)nullptr, Tk);
            rc = finish(0, Tk, k); lastL = Tk;
        }
        if (step < nr && !rc) {
            const int k = B.np - 1 - step;
            cplx *Tk = slot(1, step, k);
            HELM_LAUNCH(k_bt_schur_t, dim3(sg), dim3(256), 0, sts[1], planes, g, k, (const cplx *)nullptr, lastR, Tk);
            rc = finish(1, Tk, k); lastR = Tk;
        }
    }
    if (!rc) {                                                           // the plane where the chains meet
        hipEventRecord(B.ev[1], sts[1]);
        hipStreamWaitEvent(sts[0], B.ev[1], 0);
        cplx *Tk = slot(0, nl, B.mid);
        HELM_LAUNCH(k_bt_schur_t, dim3(sg), dim3(256), 0, sts[0], planes, g, B.mid, lastL, lastR, Tk);
        rc = finish(0, Tk, B.mid);
    }
    hipStreamSynchronize(sts[1]);
    hipStreamSynchronize(sts[0]);
    for (int c = 0; c < 2; ++c) helm_pool_free(op->device, W[c], wbytes);
    if (B.f32 && B.Tinv) { helm_pool_free(op->device, B.Tinv, B.tbytes); B.Tinv = nullptr; }
    if (rc) { bt_free(B); return rc; }
    return HELM_OK;
}

void nd3_free(Nd3 &D) {
    if (D.ws) helm_pool_free(D.device, D.ws, D.ws_bytes);
    if (D.f) nd_free(D.f);
    D = Nd3();
}

bool nd3_applicable(int nz, int ny, int nx) { return nz >= 3 && nz < 128 && ny >= 3 && nx >= 3 && ny < 4096 && nx < 4096; }

// what the column dissection of an (nz, ny, nx) level costs: flops of its factorisation, bytes of its factors and of the factorisation scratch
// (host side: the plan only; cached, the plan of a 79 x 79 grid has 2000 fronts)
struct Nd3Cost { double flops = 0, fac_bytes = 0, ws_bytes = 0; int top = 0; };
Nd3Cost nd3_cost(int nz, int ny, int nx) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, Nd3Cost> cache;
    const int leaf = envi("HELM_MG3_ND_LEAF", 2);
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_tuple(nz, ny, nx, leaf);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    NdPlan P;
    nd_build_plan(P, ny, nx, leaf, nz);
    Nd3Cost c;
    for (const NdGroup &g : P.groups) {
        c.flops += (double)g.cnt * 8.0 * (2.0 * g.smax * g.smax * g.smax + (double)g.smax * g.smax * g.mmax + (double)g.smax * g.mmax * g.mmax);
        c.top = std::max(c.top, g.smax);
    }
    c.fac_bytes = (double)P.fac_elems * sizeof(cplx);
    c.ws_bytes = (double)nd_factor_ws_elems(P) * sizeof(cplx);
    cache[key] = c;
    return c;
}

// Which direct solver the level gets.  HELM_MG3_COARSE = nd | bt forces one; otherwise the column dissection wherever it applies and needs fewer
// flops than the plane-by-plane elimination (np inversions of m^3): on config 5's 47 x 79 x 79 level 10.5 against 32.6 TFLOP.
bool coarse_is_nd(int nz, int ny, int nx) {
    const char *cs = getenv("HELM_MG3_COARSE");
    if (cs && !strcmp(cs, "bt")) return false;
    if (!nd3_applicable(nz, ny, nx)) return false;
    if (cs && !strcmp(cs, "nd")) return true;
    const int d[3] = {nz, ny, nx};
    int sI = 0;
    for (int a = 1; a < 3; ++a) if (d[a] > d[sI]) sI = a;
    const double m = (double)d[(sI + 1) % 3] * d[(sI + 2) % 3];
    return nd3_cost(nz, ny, nx).flops < (double)d[sI] * 8.0 * m * m * m;
}

int nd3_setup(helm_op *op, Nd3 &D, const Mg3Level &L, int batch) {
    if (!nd3_applicable(L.nz, L.ny, L.nx)) HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "3-D multigrid: the column dissection takes levels with fewer than 128 layers");
    D.device = op->device; D.batch = batch;
    int rc = nd_get_plan_dims(op, L.ny, L.nx, envi("HELM_MG3_ND_LEAF", 2), L.nz, &D.pd);
    if (rc) return rc;
    const NdPlan &P = D.pd->plan;
    const size_t fwb = (size_t)nd_factor_ws_elems(P) * sizeof(cplx);
    D.ws_bytes = (size_t)nd_solve_ws_elems(P, batch) * sizeof(cplx);
    {
        size_t freeb = 0, totb = 0;
        hipMemGetInfo(&freeb, &totb);
        freeb += helm_pool_idle_bytes(op->device);       // (r4: idle buffers of the library's own pool are available to it)
        const double need = (double)P.fac_elems * sizeof(cplx) + (double)fwb + (double)D.ws_bytes;
        if (need > 0.9 * (double)freeb) HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "3-D multigrid: the factors of the directly solved level (%.1f GB) do not fit", need / 1e9);
    }
    hipStreamSynchronize(op->stream);
    D.f = new NdFactor();
    D.f->pd = D.pd;
    cplx *fw = (cplx *)helm_pool_alloc(op->device, fwb);
    D.ws = (cplx *)helm_pool_alloc(op->device, D.ws_bytes);
    if (!fw || !D.ws) { if (fw) helm_pool_free(op->device, fw, fwb); nd3_free(D); HELM_FAIL(op, HELM_ERR_DEVICE, "3-D multigrid: scratch of the column dissection does not fit"); }
    rc = nd_factor(op, 0, D.f, fw, L.op->d_C);
    helm_pool_free(op->device, fw, fwb);
    if (rc) { nd3_free(D); return rc; }
    if (envi("HELM_MG3_TRACE", 0))
        fprintf(stderr, "[helm mg3] column dissection of the %d x %d x %d level: %zu fronts, factors %.2f GB, %.2f TFLOP\n", L.nz, L.ny, L.nx, P.nodes.size(),
                P.fac_elems * 16e-9, D.f->flops * 1e-12);
    return HELM_OK;
}

int nd3_solve(helm_op *op, Nd3 &D, const cplx *f, cplx *u, int nrhs) { return nd_solve(op, D.f, f, u, nrhs, D.ws); }

// u = A^-1 f on the coarsest level (f, u: [nrhs][N])
int bt_solve(helm_op *op, Bt3 &B, const Mg3Level &L, const cplx *f, cplx *u, int nrhs) {
    hipStream_t sts[2] = {op->stream, B.aux->stream};
    const BtGeom g = bt_geom(B);
    const cplx *planes = L.op->d_C;
    const dim3 rg((B.m + 255) / 256, nrhs);
    const long long pz = (long long)nrhs * B.m;              // one packed plane of Z
    const unsigned redg = (unsigned)std::min<long long>((pz + 255) / 256, 4096);
    // Z_k (-)= Y S_k^-T on the stream of `chain`
    auto apply_inverse = [&](int chain, int k, int sub) -> int {
        hipStream_t st = sts[chain];
        cplx *Zk = B.Z + k * pz;
        if (B.f32) {
            HELM_LAUNCH(k_bt_apply32<16>, dim3((B.m + 127) / 128, B.ksplit), dim3(256), 0, st, (const cplx *)B.Y[chain], B.mpad,
                               (const float2 *)(B.Tinv32 + (long long)k * B.m * B.ld32), B.m, B.ld32, B.kc, nrhs, B.parts[chain]);
            HELM_LAUNCH(k_bt_reduce, dim3(redg), dim3(256), 0, st, (const cplx *)B.parts[chain], B.nparts, pz, Zk, sub);
            return HELM_OK;
        }
        const cplx *Tk = B.Tinv + (long long)k * B.mpad * B.m;
        if (B.own) {
            HELM_LAUNCH(k_bt_apply<16>, dim3((B.m + 127) / 128, B.ksplit), dim3(256), 0, st, (const cplx *)B.Y[chain], B.mpad, Tk, B.m, B.kc, nrhs, B.parts[chain]);
            HELM_LAUNCH(k_bt_reduce, dim3(redg), dim3(256), 0, st, (const cplx *)B.parts[chain], B.nparts, pz, Zk, sub);
            return HELM_OK;
        }
        // generic batched GEMM (more than 16 right-hand sides): it launches on the handle's own stream, so this path keeps to one chain order
        const int rc = nd_dense_gemm_batched(chain ? B.aux : op, nrhs, B.m, B.kc, cmake(1, 0), B.Y[chain], B.mpad, B.kc, Tk, B.m, (long long)B.kc * B.m, cmake(0, 0),
                                             B.parts[chain], B.m, pz, B.ksplit);
        if (rc) return rc;
        HELM_LAUNCH(k_bt_reduce, dim3(redg), dim3(256), 0, st, (const cplx *)B.parts[chain], B.ksplit, pz, Zk, sub);
        return HELM_OK;
    };
    auto rhs = [&](int chain, int k, const cplx *fk, const cplx *Zm, const cplx *Zp) {
        HELM_LAUNCH(k_bt_rhs, rg, dim3(256), 0, sts[chain], planes, g, k, fk, Zm, Zp, B.Y[chain], B.mpad);
    };
    const int nl = B.mid, nr = B.np - 1 - B.mid;
    int rc = HELM_OK;
    hipEventRecord(B.ev[0], sts[0]);                       // f is ready
    hipStreamWaitEvent(sts[1], B.ev[0], 0);
    for (int step = 0; step < std::max(nl, nr); ++step) {  // forward: z_k = S_k^-1 (f_k - A_{k,k-+1} z_{k-+1}), both chains
        if (step < nl) { const int k = step; rhs(0, k, f, k ? B.Z + (k - 1) * pz : nullptr, nullptr); rc = apply_inverse(0, k, 0); if (rc) return rc; }
        if (step < nr) { const int k = B.np - 1 - step; rhs(1, k, f, nullptr, step ? B.Z + (k + 1) * pz : nullptr); rc = apply_inverse(1, k, 0); if (rc) return rc; }
    }
    hipEventRecord(B.ev[1], sts[1]);
    hipStreamWaitEvent(sts[0], B.ev[1], 0);
    rhs(0, B.mid, f, nl ? B.Z + (B.mid - 1) * pz : nullptr, nr ? B.Z + (B.mid + 1) * pz : nullptr);     // the plane where the chains meet: x_mid
    rc = apply_inverse(0, B.mid, 0); if (rc) return rc;
    hipEventRecord(B.ev[2], sts[0]);
    hipStreamWaitEvent(sts[1], B.ev[2], 0);
    for (int step = 0; step < std::max(nl, nr); ++step) {  // back substitution outwards: x_k = z_k - S_k^-1 A_{k,k+-1} x_{k+-1}
        if (step < nl) { const int k = B.mid - 1 - step; rhs(0, k, nullptr, nullptr, B.Z + (k + 1) * pz); rc = apply_inverse(0, k, 1); if (rc) return rc; }
        if (step < nr) { const int k = B.mid + 1 + step; rhs(1, k, nullptr, B.Z + (k - 1) * pz, nullptr); rc = apply_inverse(1, k, 1); if (rc) return rc; }
    }
    hipEventRecord(B.ev[1], sts[1]);
    hipStreamWaitEvent(sts[0], B.ev[1], 0);
    HELM_LAUNCH(k_bt_scatter, dim3((unsigned)std::min<long long>(((long long)B.np * B.m + 255) / 256, 4096), nrhs), dim3(256), 0, sts[0],
                       (const cplx *)B.Z, g, nrhs, u);
    return HELM_OK;
}

}  // namespace

namespace {

int cycle_keep(helm_op *op, Mg3Precond *P, size_t l, int nrhs, cplx *final_out = nullptr) {
    Mg3Keep *K = P->keep;
    Mg3Level &L = P->lv[l];
    hipStream_t st = op->stream;
    if (l + 1 == P->lv.size()) return K->nd.on() ? nd3_solve(op, K->nd, L.f, L.u, nrhs) : bt_solve(op, K->bt, L, L.f, L.u, nrhs);
    Mg3Level &C = P->lv[l + 1];
    const cplx *dl1 = K->dl1[l];
    const double w = K->omega_l1;
    // f32 (round 6, helm_tuning.mg3_f32): the work vectors u, t, r of the FINEST level hold complex64 -- the cycle is a preconditioner, its input and its result
    // stay complex128 (the outer BiCGSTAB recurrences, the operator applies that form its residuals and the convergence check are untouched), and single precision
    // in between costs the Krylov method nothing it can see (iteration counts in the tests).  What it buys: half the bytes of every vector the sweeps, the residual
    // and the transfers of that level move, and half the staging of the 27-point kernel's tiles, which is what bounds it.
    const bool f32 = l == 0 && P->fine32 && final_out != nullptr && P->nu2 >= 1;
    auto smooth = [&](const cplx *x, cplx *y) -> int {
        ApplyArgs a;
        a.planes = L.op->d_C; a.X = x; a.Y = y; a.W = L.f; a.ld = L.N; a.nrhs = nrhs; a.epi = EPI_JACOBI; a.scaled = 0; a.adjoint = 0;
        a.scal = nullptr; a.part = (double *)op->d_part; a.dinv = dl1; a.omega_j = w; a.profile = 0;
        if (f32) { a.x32 = 1; a.y32 = (y == final_out) ? 0 : 1; }
        return helm_launch_apply(L.op, a);
    };
    if (f32) HELM_LAUNCH(k3_jac0<cplxf>, vgrid(L.N, nrhs), dim3(256), 0, st, (const cplx *)L.f, dl1, (cplxf *)L.u, L.N, w);
    else HELM_LAUNCH(k3_jac0<cplx>, vgrid(L.N, nrhs), dim3(256), 0, st, (const cplx *)L.f, dl1, L.u, L.N, w);
    int rc;
    for (int s = 1; s < P->nu1; ++s) { rc = smooth(L.u, L.t); if (rc) return rc; std::swap(L.u, L.t); }
    if (f32) {
        ApplyArgs a;
        a.planes = L.op->d_C; a.X = L.u; a.Y = L.r; a.W = L.f; a.ld = L.N; a.nrhs = nrhs; a.epi = EPI_RESID; a.scaled = 0; a.adjoint = 0;
        a.scal = nullptr; a.part = (double *)op->d_part; a.dinv = L.op->d_dinv; a.omega_j = 0.0; a.profile = 0; a.x32 = 1; a.y32 = 1;
        rc = helm_launch_apply(L.op, a);
    } else rc = level_apply(op, L, L.u, L.r, L.f, nrhs, EPI_RESID, 0.0);
    if (rc) return rc;
    if (f32) HELM_LAUNCH(k3_restrict_t<cplxf>, vgrid(C.N, nrhs), dim3(256), 0, st, (const cplxf *)L.r, C.f, L.ny, L.nx, C.nz, C.ny, C.nx, L.N,
                         (const RTab *)K->rt[0][l], (const RTab *)K->rt[1][l], (const RTab *)K->rt[2][l]);
    else HELM_LAUNCH(k3_restrict_t<cplx>, vgrid(C.N, nrhs), dim3(256), 0, st, (const cplx *)L.r, C.f, L.ny, L.nx, C.nz, C.ny, C.nx, L.N,
                       (const RTab *)K->rt[0][l], (const RTab *)K->rt[1][l], (const RTab *)K->rt[2][l]);
    rc = cycle_keep(op, P, l + 1, nrhs); if (rc) return rc;
    if (f32) HELM_LAUNCH(k3_prolong_add_t<cplxf>, vgrid(L.N, nrhs), dim3(256), 0, st, (const cplx *)C.u, (cplxf *)L.u, L.nz, L.ny, L.nx, C.ny, C.nx, C.N,
                         (const PTab *)K->pt[0][l], (const PTab *)K->pt[1][l], (const PTab *)K->pt[2][l]);
    else HELM_LAUNCH(k3_prolong_add_t<cplx>, vgrid(L.N, nrhs), dim3(256), 0, st, (const cplx *)C.u, L.u, L.nz, L.ny, L.nx, C.ny, C.nx, C.N,
                       (const PTab *)K->pt[0][l], (const PTab *)K->pt[1][l], (const PTab *)K->pt[2][l]);
    for (int s = 0; s < P->nu2; ++s) {
        if (final_out && s == P->nu2 - 1) return smooth(L.u, final_out);
        rc = smooth(L.u, L.t); if (rc) return rc;
        std::swap(L.u, L.t);
    }
    if (final_out) HIP_TRY(op, hipMemcpyAsync(final_out, L.u, (size_t)nrhs * L.N * sizeof(cplx), hipMemcpyDeviceToDevice, st));
    return HELM_OK;
}

void keep_free(Mg3Precond *P) {
    Mg3Keep *K = P->keep;
    if (!K) return;
    for (size_t i = 0; i < K->dl1.size(); ++i) helm_pool_free(K->device, K->dl1[i], K->dl1_bytes[i]);
    for (auto &t : K->tabs) helm_pool_free(K->device, t.first, t.second);
    bt_free(K->bt);
    nd3_free(K->nd);
    delete K;
    P->keep = nullptr;
}

template <typename T> T *upload(Mg3Keep *K, const std::vector<T> &v) {
    const size_t b = v.size() * sizeof(T);
    T *d = (T *)helm_pool_alloc(K->device, b);
    if (!d) return nullptr;
    K->tabs.push_back(std::make_pair((void *)d, b));
    if (hipMemcpy(d, v.data(), b, hipMemcpyHostToDevice) != hipSuccess) return nullptr;      // (the buffer goes back with the others in keep_free)
    return d;
}

// geometry of the directly solved level if it is the one reached after `l` layer-preserving coarsenings: np planes of m x m;
// returns the bytes of its plane inverses as bt_setup will store them for `batch` right-hand sides (single precision for up to 16
// with the library's own plane product, double precision -- padded for the generic batched GEMM -- beyond that or on request)
// nodes per axis (z, y, x) of level l of the layer-preserving hierarchy
void keep_level_dims(const helm_op *op, int l, int out[3]) {
    const int dims[3] = {op->nz, op->ny, op->nx};
    for (int a = 0; a < 3; ++a) {
        Ax3 ax; const int n = dims[a], np = op->nPML;
        ax.x.resize(n); ax.gam.assign(n, 0.0); ax.lay.assign(n, 0);
        for (int i = 0; i < n; ++i) ax.x[i] = i;
        for (int k = 0; k < np && k < n; ++k) { ax.lay[k] = 1; ax.lay[n - np + k] = 1; }
        Ax3 c; std::vector<int> kept; std::vector<PTab> pt; std::vector<RTab> rt;
        for (int i = 0; i < l; ++i) { coarsen_axis(ax, true, c, kept, pt, rt); ax = c; }
        out[a] = ax.n();
    }
}

double keep_direct_bytes(const helm_op *op, int l, int batch, int *np_out = nullptr, int *m_out = nullptr) {
    const int dims[3] = {op->nz, op->ny, op->nx};
    int out[3];
    for (int a = 0; a < 3; ++a) {
        Ax3 ax; const int n = dims[a], np = op->nPML;
        ax.x.resize(n); ax.gam.assign(n, 0.0); ax.lay.assign(n, 0);
        for (int i = 0; i < n; ++i) ax.x[i] = i;
        for (int k = 0; k < np && k < n; ++k) { ax.lay[k] = 1; ax.lay[n - np + k] = 1; }
        Ax3 c; std::vector<int> kept; std::vector<PTab> pt; std::vector<RTab> rt;
        for (int i = 0; i < l; ++i) { coarsen_axis(ax, true, c, kept, pt, rt); ax = c; }
        out[a] = ax.n();
    }
    int s = 0;
    for (int a = 1; a < 3; ++a) if (out[a] > out[s]) s = a;
    const double m = (double)out[(s + 1) % 3] * out[(s + 2) % 3];
    if (np_out) *np_out = out[s];
    if (m_out) *m_out = (int)m;
    const bool own = batch <= 16;
    const bool f32 = own && helm_tuning_now().mg3_bt_f32 != 0;
    if (f32) return (double)out[s] * m * m * sizeof(float2) + 4.0 * m * m * sizeof(cplx);        // + the set-up's double-precision ping-pong planes
    double mpad = m;
    if (!own) { const int ks = std::max(1, std::min(16, 255 / (((int)m + 63) / 64))); mpad = (double)(((int)m + ks - 1) / ks) * ks; }
    return (double)out[s] * mpad * m * sizeof(cplx);
}

// ---- what the depth decision is made from: timed on this device at set-up, once per process and size class ---------------------------------
__global__ void k3_cal_fill(cplx *A, int n) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e / n), j = (int)(e % n);
        const unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(j * 40503u);
        A[e] = i == j ? cmake(4.0, 1.0) : cmake(((h & 1023) / 1024.0 - 0.5) / n, (((h >> 10) & 1023) / 1024.0 - 0.5) / n);
    }
}
std::mutex g_cal_mu;
std::map<std::pair<int, int>, double> g_cal_inverse;       // (device, size) -> seconds of one dense inversion
// seconds the dense blocked Gauss-Jordan takes for one m x m plane: timed on a synthetic matrix of min(m, 4096) rows, scaled with the cube of
// the size above that (the rate still rises a little there, so large planes are over- rather than under-estimated)
double inverse_seconds_class(helm_op *op, int mc);
// All size classes are timed the first time any of them is asked for -- the first set-up of the process, before anything else runs on the GPU:
// a dispatcher later builds the next frequency's preconditioner BESIDE the current frequency's iterations, and a timing taken there would
// measure the sharing (set-ups that look slow send the depth decision one level deeper than it should go).
double inverse_seconds(helm_op *op, int m) {
    static const int classes[5] = {256, 512, 1024, 2048, 4096};
    {
        bool have = false;
        { std::lock_guard<std::mutex> lk(g_cal_mu); have = g_cal_inverse.count(std::make_pair(op->device, 4096)) != 0; }
        if (!have) for (int c : classes) (void)inverse_seconds_class(op, c);
    }
    int mc = 4096;
    for (int c : classes) if (m <= c) { mc = c; break; }
    const double t = inverse_seconds_class(op, mc);
    const double r = (double)m / mc;
    return t * r * r * r;
}
double inverse_seconds_class(helm_op *op, int m) {
    const int mc = std::max(32, std::min(m, 4096));
    double t = -1.0;
    {
        std::lock_guard<std::mutex> lk(g_cal_mu);
        auto it = g_cal_inverse.find(std::make_pair(op->device, mc));
        if (it != g_cal_inverse.end()) t = it->second;
    }
    if (t < 0) {
        const size_t mb = (size_t)mc * mc * sizeof(cplx);
        cplx *A = (cplx *)helm_pool_alloc(op->device, mb), *W = (cplx *)helm_pool_alloc(op->device, mb);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (A && W && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            for (int rep = 0; rep < 2; ++rep) {                       // the second run is the timed one
                HELM_LAUNCH(k3_cal_fill, dim3(1024), dim3(256), 0, op->stream, A, mc);
                hipEventRecord(e0, op->stream);
                nd_dense_inverse(op, A, mc, W);
                hipEventRecord(e1, op->stream);
            }
            float ms = 0.f;
            if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0) t = ms * 1e-3;
        }
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
        hipStreamSynchronize(op->stream);
        helm_pool_free(op->device, A, mb); helm_pool_free(op->device, W, mb);
        if (t < 0) t = 8.0 * mc * (double)mc * mc / 20e12;             // (could not time it: a nominal rate)
        std::lock_guard<std::mutex> lk(g_cal_mu);
        g_cal_inverse[std::make_pair(op->device, mc)] = t;
    }
    const double r = (double)m / mc;
    return t * r * r * r;
}
// seconds one fine-grid 27-point apply takes per right-hand side at the batch width of this call (the unit an iteration is priced in)
double apply_seconds_per_rhs(helm_op *op, int batch) {
    const int nb = std::max(1, std::min(batch, 16));
    const size_t vb = (size_t)nb * op->N * sizeof(cplx);
    cplx *X = (cplx *)helm_pool_alloc(op->device, vb), *Y = (cplx *)helm_pool_alloc(op->device, vb);
    double t = -1.0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (X && Y && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        hipMemsetAsync(X, 0, vb, op->stream);
        ApplyArgs a = ApplyArgs();
        a.planes = op->d_C; a.X = X; a.Y = Y; a.ld = op->N; a.nrhs = nb; a.epi = EPI_NONE; a.profile = 0;
        int rc = helm_launch_apply(op, a);
        hipEventRecord(e0, op->stream);
        if (!rc) rc = helm_launch_apply(op, a);
        hipEventRecord(e1, op->stream);
        float ms = 0.f;
        if (!rc && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0) t = ms * 1e-3 / nb;
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipStreamSynchronize(op->stream);
    helm_pool_free(op->device, X, vb); helm_pool_free(op->device, Y, vb);
    if (t < 0) t = (double)op->N * (32.0 + 432.0 / nb) / 3.5e12;       // (could not time it: a nominal streaming rate)
    return t;
}

// memory and set-up time of the direct solver of level l, whichever kind it gets (coarse_is_nd): the plane-by-plane elimination is np timed
// inversions; the column dissection is priced at its flop count over the rate of a timed inversion of its top separator's size (its big
// fronts run the same blocked Gauss-Jordan and the same tile kernel: 10.5 TFLOP in 0.42 s on config 5 = the 25 TFLOP/s of the 3713^2 inversion)
struct CoarseEst { bool nd = false; double bytes = 0, seconds = 0; int np = 0, m = 0; };
CoarseEst coarse_estimate(helm_op *op, int l, int batch, bool timed) {
    CoarseEst e;
    int d[3];
    keep_level_dims(op, l, d);
    e.bytes = keep_direct_bytes(op, l, batch, &e.np, &e.m);
    e.nd = coarse_is_nd(d[0], d[1], d[2]);
    if (e.nd) {
        const Nd3Cost c = nd3_cost(d[0], d[1], d[2]);
        e.bytes = c.fac_bytes + c.ws_bytes;
        if (timed) { const int mt = std::max(64, c.top); e.seconds = c.flops / (8.0 * mt * (double)mt * mt / inverse_seconds(op, mt)); }
        e.np = 1; e.m = c.top;
    } else if (timed) e.seconds = e.np * inverse_seconds(op, e.m);
    return e;
}

// smallest Re(c) of the model, on the device (positive doubles order like their bit patterns)
__global__ void k3_min_re(const cplx *__restrict__ c, long long n, unsigned long long *out) {
    double m = 1e300;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) m = fmin(m, c[e].x);
    for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMin(out, (unsigned long long)__double_as_longlong(m));
}
// model of a coarser level: the values at the nodes it keeps (kz / ky / kx: kept node indices per axis)
__global__ void k3_inject_model(const cplx *__restrict__ c, const double *__restrict__ rho, int fny, int fnx, const int *__restrict__ kz, const int *__restrict__ ky,
                                const int *__restrict__ kx, int nzc, int nyc, int nxc, cplx *__restrict__ cc, double *__restrict__ rc) {
    const long long n = (long long)nzc * nyc * nxc;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(e % nxc), Y = (int)((e / nxc) % nyc), Z = (int)(e / ((long long)nxc * nyc));
        const long long src = ((long long)kz[Z] * fny + ky[Y]) * fnx + kx[X];
        cc[e] = c[src]; rc[e] = rho[src];
    }
}

// levels 0 .. ncoarsen of the layer-preserving hierarchy + the direct solver of the last one; on failure the caller falls back
int setup_keep(helm_op *op, Mg3Precond *P, int batch, int ncoarsen, double tauM) {
    const bool trace = envi("HELM_MG3_TRACE", 0) != 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto tprev = now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        hipStreamSynchronize(op->stream);
        const auto t = now();
        fprintf(stderr, "[helm mg3 set-up] %-34s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(t - tprev).count());
        tprev = t;
    };
    Mg3Keep *K = new Mg3Keep();
    P->keep = K;
    P->fine32 = helm_tuning_now().mg3_f32 != 0;
    K->omega_l1 = 1.6;
    K->device = op->device;
    std::complex<double> om(2.0 * M_PI * op->a_freq_re, 2.0 * M_PI * op->a_freq_im);
    om -= std::complex<double>(0.0, 1.0 / tauM);
    const double cpml = op->a_cpml;
    Ax3 ax[3];
    const int dims[3] = {op->nz, op->ny, op->nx};
    const double hs[3] = {op->dz, op->dy, op->dx};
    for (int a = 0; a < 3; ++a) {
        const int n = dims[a], np = op->nPML;
        ax[a].x.resize(n); ax[a].gam.assign(n, 0.0); ax[a].lay.assign(n, 0);
        for (int i = 0; i < n; ++i) ax[a].x[i] = i * hs[a];
        const double Lh = hs[a] * (np - 1);
        for (int k = 0; k < np && k < n; ++k) {        // the profile of helm3d.hip profile3()
            ax[a].gam[k] = cpml * cos((M_PI / 2) * (k * hs[a] / Lh)); ax[a].lay[k] = 1;
            ax[a].gam[n - np + k] = cpml * cos((M_PI / 2) * ((np - 1 - k) * hs[a] / Lh)); ax[a].lay[n - np + k] = 1;
        }
    }
    // the levels' models never visit the host: level 0 copies the caller's device arrays, a coarser level takes the values at the nodes it keeps
    std::vector<int> kept[3];
    lap("axes");
    for (int l = 0; l <= ncoarsen; ++l) {
        if (l) lap("level (operator, vectors, tables)");
        Mg3Level L;
        L.nz = ax[0].n(); L.ny = ax[1].n(); L.nx = ax[2].n(); L.N = (long long)L.nz * L.ny * L.nx;
        L.op = helm_create3d(op->device, L.nz, L.ny, L.nx, 1.0, 1.0, 1.0, 2);
        if (!L.op) HELM_FAIL(op, HELM_ERR_DEVICE, "%s", helm_last_error(nullptr));
        P->lv.push_back(L);
        Mg3Level &Lr = P->lv.back();
        if (helm_set_stream(Lr.op, op->stream)) HELM_FAIL(op, HELM_ERR_DEVICE, "3-D multigrid: cannot share the stream");
        std::vector<cplx> Lz, Ly, Lx;
        lap_from_axis(ax[0], om, Lz); lap_from_axis(ax[1], om, Ly); lap_from_axis(ax[2], om, Lx);
        Lr.op->lap_override = Lx;
        Lr.op->lap_override.insert(Lr.op->lap_override.end(), Ly.begin(), Ly.end());
        Lr.op->lap_override.insert(Lr.op->lap_override.end(), Lz.begin(), Lz.end());
        int rc = HELM_OK;
        if (l == 0) rc = helm_adopt_model_device(Lr.op, op->d_c, op->d_rho);
        else {
            const Mg3Level &Lf = P->lv[l - 1];
            const size_t kb = (kept[0].size() + kept[1].size() + kept[2].size()) * sizeof(int);
            int *dk = (int *)helm_pool_alloc(op->device, kb);
            if (!dk) HELM_FAIL(op, HELM_ERR_DEVICE, "3-D multigrid: transfer tables do not fit");
            size_t off = 0;
            const int *dka[3];
            for (int a = 0; a < 3; ++a) {
                hipMemcpyAsync(dk + off, kept[a].data(), kept[a].size() * sizeof(int), hipMemcpyHostToDevice, op->stream);
                dka[a] = dk + off; off += kept[a].size();
            }
            HELM_LAUNCH(k3_inject_model, dim3((unsigned)std::min<long long>((Lr.N + 255) / 256, 65535)), dim3(256), 0, op->stream, (const cplx *)Lf.op->d_c,
                               (const double *)Lf.op->d_rho, Lf.ny, Lf.nx, dka[0], dka[1], dka[2], Lr.nz, Lr.ny, Lr.nx, Lr.op->d_c, Lr.op->d_rho);
            hipStreamSynchronize(op->stream);                    // (kept[] is overwritten below; the table buffer goes back to the pool)
            helm_pool_free(op->device, dk, kb);
            rc = helm_adopt_model_device(Lr.op, nullptr, nullptr);
        }
        if (!rc) rc = helm_assemble(Lr.op, op->a_freq_re, op->a_freq_im, tauM, 0.0, cpml);
        if (rc) HELM_FAIL(op, rc, "%s", helm_last_error(Lr.op));
        if (!level_vectors(op, Lr, batch)) HELM_FAIL(op, HELM_ERR_DEVICE, "3-D multigrid: level vectors do not fit");
        if (l == ncoarsen) break;
        cplx *dl1 = (cplx *)helm_pool_alloc(op->device, (size_t)Lr.N * sizeof(cplx));
        if (!dl1) HELM_FAIL(op, HELM_ERR_DEVICE, "3-D multigrid: level vectors do not fit");
        K->dl1.push_back(dl1); K->dl1_bytes.push_back((size_t)Lr.N * sizeof(cplx));
        HELM_LAUNCH(k3_l1_dinv, dim3((unsigned)std::min<long long>((Lr.N + 255) / 256, 65535)), dim3(256), 0, op->stream, (const cplx *)Lr.op->d_C, dl1, Lr.N, K->omega_l1);
        // next level
        Ax3 cx[3];
        for (int a = 0; a < 3; ++a) {
            std::vector<PTab> pt; std::vector<RTab> rt;
            coarsen_axis(ax[a], true, cx[a], kept[a], pt, rt);
            PTab *dp = upload(K, pt); RTab *dr = upload(K, rt);
            K->pt[a].push_back(dp); K->rt[a].push_back(dr);
            if (!dp || !dr) HELM_FAIL(op, HELM_ERR_DEVICE, "3-D multigrid: transfer tables do not fit");
        }
        for (int a = 0; a < 3; ++a) ax[a] = cx[a];
    }
    lap("last level");
    if (ncoarsen > 0 && helm_tuning_now().mg3_galerkin) {        // the directly solved level carries the Galerkin product of the level above it
        const Mg3Level &Lf = P->lv[ncoarsen - 1]; Mg3Level &Lc = P->lv[ncoarsen];
        const int t = ncoarsen - 1;
        Lc.op->otf3 = false;            // (the coarse level's planes are the Galerkin product from here on, not what its c, rho and factor tables would rebuild)
        HELM_LAUNCH(k3_galerkin, dim3((unsigned)((Lc.N + 255) / 256)), dim3(256), 0, op->stream, (const cplx *)Lf.op->d_C, Lf.nz, Lf.ny, Lf.nx,
                           Lc.op->d_C, Lc.nz, Lc.ny, Lc.nx, (const RTab *)K->rt[0][t], (const RTab *)K->rt[1][t], (const RTab *)K->rt[2][t],
                           (const PTab *)K->pt[0][t], (const PTab *)K->pt[1][t], (const PTab *)K->pt[2][t]);
        HIP_TRY(op, hipGetLastError());
    }
    lap("Galerkin product");
    const Mg3Level &Ld = P->lv.back();
    int rcd = HELM_ERR_UNSUPPORTED;
    if (coarse_is_nd(Ld.nz, Ld.ny, Ld.nx)) rcd = nd3_setup(op, K->nd, Ld, batch);
    if (rcd == HELM_ERR_UNSUPPORTED) rcd = bt_setup(op, K->bt, Ld, batch);          // (also when the dissection's factors do not fit: the plane inverses are single precision)
    lap("direct solver of the last level");
    return rcd;
}

}  // namespace

// the hierarchy's level operators launch on the stream they were built on: point them at another one (helm_prefactor_n builds on a
// low-priority stream, the solves run on the handle's own)
void mg3_retarget_stream(helm_op *op, hipStream_t st) {
    if (!op->mg3) return;
    for (Mg3Level &L : op->mg3->lv) if (L.op) L.op->stream = st;
}

void mg3_destroy(helm_op *op) {
    Mg3Precond *P = op->mg3;
    if (!P) return;
    keep_free(P);
    for (Mg3Level &L : P->lv) {
        level_vectors_free(op, L);
        if (L.op) { L.op->own_stream = false; L.op->stream = nullptr; helm_destroy(L.op); }
    }
    hipFree(P->cinvT);
    delete P;
    op->mg3 = nullptr;
}

int mg3_setup(helm_op *op, int batch) {
    if (op->mg3 && op->mg3->batch >= batch) return HELM_OK;
    if (op->mg3) mg3_destroy(op);
    int rc = HELM_OK;
    Mg3Precond *P = new Mg3Precond();
    op->mg3 = P;
    P->batch = batch;
    // Jacobi damping: measured at 256 x 256 x 128, 4 sources (tools/sweep3d.sh): 0.8 / 0.9 / 1.0 / 1.1 -> 10.9 / 9.6 / 10.0 / 16.4 s at 3 Hz and
    // 7.4 / 7.0 / 6.6 / 8.9 s at 5 Hz
    P->omega_j = helm_tuning_now().mg3_omega;
    P->nu1 = 1; P->nu2 = 1; P->min_n = 8;
    const double omega = 2.0 * M_PI * std::abs(std::complex<double>(op->a_freq_re, op->a_freq_im));
    // shift: 0.6 at 10 grid points per wavelength, growing with the square of the oversampling up to 8 -- measured at
    // 256 x 256 x 128, 40-100 points per wavelength: beta 0.6 / 3 / 6 / 12 -> 26 / 16 / 14 / 14 s per 4 sources at 3 Hz
    double cmin = 1e300;
    {
        unsigned long long *dmin = (unsigned long long *)helm_pool_alloc(op->device, 64);     // (64 bytes: the pool's smallest size class)
        unsigned long long hmin = ~0ULL;
        if (!dmin) HELM_FAIL(op, HELM_ERR_DEVICE, "3-D multigrid: scratch allocation failed");
        hipMemcpyAsync(dmin, &hmin, sizeof(hmin), hipMemcpyHostToDevice, op->stream);
        HELM_LAUNCH(k3_min_re, dim3(2048), dim3(256), 0, op->stream, (const cplx *)op->d_c, op->N, dmin);
        hipMemcpyAsync(&hmin, dmin, sizeof(hmin), hipMemcpyDeviceToHost, op->stream);
        HIP_TRY(op, hipStreamSynchronize(op->stream));
        helm_pool_free(op->device, dmin, 64);
        if (hmin != ~0ULL) { long long b = (long long)hmin; cmin = __builtin_bit_cast(double, b); }
    }
    const double hmax = std::max(op->dx, std::max(op->dy, op->dz));
    const double ppw = omega > 0 ? cmin / (omega / (2.0 * M_PI) * hmax) : 10.0;
    const double over = std::max(1.0, ppw / 10.0);
    P->beta = envd("HELM_MG3_BETA", std::min(8.0, 0.6 * over * over));
    // Oversampled grids: the layer-preserving hierarchy with a direct solve where the interior still has >= 10 points per wavelength
    // (section above).  Falls back to the standard cycle when no level can be dropped or the plane inverses do not fit.
    {
        const double ppwc = 9.9;
        int ncoarsen = 0;
        while (ncoarsen < 5 && ppw / (double)(2 << ncoarsen) >= ppwc) ++ncoarsen;
        const int interior = std::min(op->nz, std::min(op->ny, op->nx)) - 2 * op->nPML;
        while (ncoarsen > 0 && (interior >> ncoarsen) < 3) --ncoarsen;
        // if the plane inverses of that level do not fit the budget, go one level deeper as long as it keeps HELM_MG3_PPWF (6) points per
        // wavelength: 20-35 iterations instead of 6-15, still an order of magnitude fewer than the standard cycle
        if (ncoarsen > 0) {
            size_t freeb = 0, totb = 0;
            hipMemGetInfo(&freeb, &totb);
            freeb += helm_pool_idle_bytes(op->device);       // (r4: idle buffers of the library's own pool are available to it)
            // budget of the plane inverses: a third of the device, and never more than what is free now less the Krylov vectors of this call
            const double krylov = 11.0 * batch * (double)op->N * sizeof(cplx);
            const double cap = std::min(totb / 3.0, std::max(0.0, (double)freeb - (op->d_ws ? 0.0 : krylov)));
            const double ppwf = 6.0;
            while (ncoarsen < 5 && coarse_estimate(op, ncoarsen, batch, false).bytes > cap && ppw / (double)(2 << ncoarsen) >= ppwf && (interior >> (ncoarsen + 1)) >= 3) ++ncoarsen;
            // ... and one level deeper (down to 5 points) when that SAVES time for the right-hand sides of the call that builds the preconditioner:
            // the set-up of the deeper level is cheaper (np plane inversions of m^3 work each) but every right-hand side pays more iterations.
            //   set-up saved   = np_d t_inv(m_d) - np_{d+1} t_inv(m_{d+1}),  t_inv timed on this device (inverse_seconds)
            //   iterations paid = nrhs * extra * t_iter,  t_iter = 18 fine-grid applies per right-hand side, the apply timed on this grid at this
            //                    batch width (apply_seconds_per_rhs).  18: an iteration of the right-preconditioned BiCGSTAB is 2 applies, 2 cycles of
            //                    ~3.7 fine-grid-apply equivalents each (two smoothing sweeps + the residual on the finest level, ~20 % more for the
            //                    levels below it), 224 B per point of vector updates (~4 applies at 16 right-hand sides) and two coarse solves;
            //                    on config 5 this reproduces the 2.6 ms per right-hand side and iteration measured there.
            //   extra          = +11 / +22 / +38 iterations with the Galerkin direct level at >= 8 / 6 / 5 points per wavelength -- a property of the
            //                    cycle, not of the machine: measured on config 5 (homogeneous) and on the heterogeneous probes of DESIGN.md 5.3.
            if (op->mg3_rhs_hint > 0 && helm_tuning_now().mg3_depth_model && ncoarsen < 5 && (interior >> (ncoarsen + 1)) >= 3) {
                const double ppwd = ppw / (double)(2 << ncoarsen);
                if (ppwd >= 5.0) {
                    const CoarseEst e0 = coarse_estimate(op, ncoarsen, batch, true), e1 = coarse_estimate(op, ncoarsen + 1, batch, true);
                    const int np0 = e0.np, m0 = e0.m, np1 = e1.np, m1 = e1.m;
                    const double saved = envd("HELM_MG3_DEPTH_SETUP_SCALE", 1.0) * (e0.seconds - e1.seconds);
                    // iterations the deeper hierarchy costs per right-hand side: booked counts of both classes where this process has run them,
                    // the prior (+11 / +22 / +38) on top of the booked count of the other, or alone, where it has not
                    const double prior = ppwd >= 8.0 ? 11.0 : (ppwd >= 6.0 ? 22.0 : 38.0);
                    const double rt_cls = lookup_rtol(op);
                    const double its0 = its_lookup(op, ncoarsen, 2.0 * ppwd, rt_cls), its1 = its_lookup(op, ncoarsen + 1, ppwd, rt_cls);      // (ppwd: the DEEPER candidate's direct level)
                    const double extra_its = (its0 > 0 && its1 > 0) ? std::max(0.0, its1 - its0) : prior;
                    const double t_iter = 18.0 * apply_seconds_per_rhs(op, batch);
                    const double paid = op->mg3_rhs_hint * extra_its * t_iter;
                    const bool deeper = saved > paid || envi("HELM_MG3_DEPTH_FORCE_DEEPER", 0) != 0;
                    if (envi("HELM_MG3_TRACE", 0))
                        fprintf(stderr, "[mg3 depth] %d coarsenings (%.1f points per wavelength on the direct level): set-up %d x %d^2 (%s); one deeper: %d x %d^2 (%s); saves %.3f s, "
                                        "costs %d rhs x %.0f iterations x %.2f ms = %.3f s -> %s\n", ncoarsen, ppwd, np0, m0, e0.nd ? "column dissection, top separator" : "planes",
                                np1, m1, e1.nd ? "column dissection, top separator" : "planes", saved, op->mg3_rhs_hint, extra_its,
                                t_iter * 1e3, paid, deeper ? "deeper" : "stay");
                    if (envi("HELM_MG3_TRACE", 0))
                        fprintf(stderr, "[mg3 depth]   iterations booked in this process: this depth %.1f, one deeper %.1f (< 0: never run; prior +%.0f)\n", its0, its1, prior);
                    if (deeper) ++ncoarsen;
                }
            }
        }
        { const int kl = helm_tuning_now().mg3_keep_levels; if (kl >= 0) ncoarsen = kl; }
        if (helm_tuning_now().mg3_keep && !op->mg3_no_keep && ncoarsen > 0 && op->a_cpml > 0 && omega > 0) {
            const double betak = envd("HELM_MG3_BETA", 0.1);
            double inv_tau_k = omega * betak / 2.0;
            if (std::isfinite(op->a_tau) && op->a_tau != 0.0) inv_tau_k += 1.0 / op->a_tau;
            const int rck = setup_keep(op, P, batch, ncoarsen, 1.0 / inv_tau_k);
            if (rck == HELM_OK) { P->beta = betak; P->kept_levels = ncoarsen; P->ppw_direct = ppw / (double)(1 << ncoarsen); hipStreamSynchronize(op->stream); return HELM_OK; }
            if (helm_tuning_now().mg3_keep == 2) { const std::string msg = op->err; mg3_destroy(op); helm_set_error(op, msg.c_str()); return rck; }
            // not this time: release what was built and go on with the standard hierarchy
            keep_free(P);
            for (Mg3Level &L : P->lv) {
                level_vectors_free(op, L);
                if (L.op) { L.op->own_stream = false; L.op->stream = nullptr; helm_destroy(L.op); }
            }
            P->lv.clear();
        }
    }
    double inv_tau = omega * P->beta / 2.0;
    if (std::isfinite(op->a_tau) && op->a_tau != 0.0) inv_tau += 1.0 / op->a_tau;
    const double tauM = 1.0 / inv_tau;
    // layer of the preconditioner: gamma / omega = 2 (measured at 256 x 256 x 128 with the shift above: 0.2 omega -> 26 / 14 / 8.8 s
    // per 4 sources at 2 / 3 / 5 Hz, 2 omega -> 22 / 10.5 / 7.4 s, 5 omega worse again, the true layer (300) does not converge;
    // with the small shift beta = 0.6 only gamma / omega <= 0.4 was stable)
    P->cpml_m = 2.0 * omega;
    const double cpml = std::min(P->cpml_m, op->a_cpml > 0 ? op->a_cpml : P->cpml_m);
    rc = helm_ensure_host_model(op);               // (the standard hierarchy injects its models on the host)
    if (rc) return rc;
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
        if (!level_vectors(op, Lr, batch)) return fail(HELM_ERR_DEVICE, "3-D multigrid: level vectors do not fit");
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
    if (helm_malloc_retry(op->device, (void **)&A, mb) != hipSuccess || helm_malloc_retry(op->device, (void **)&W, mb) != hipSuccess || helm_malloc_retry(op->device, (void **)&P->cinvT, mb) != hipSuccess) {
        hipFree(A); hipFree(W); return fail(HELM_ERR_DEVICE, "3-D multigrid: coarsest inverse does not fit");
    }
    hipMemsetAsync(A, 0, mb, op->stream);
    HELM_LAUNCH(k3_dense, dim3((unsigned)((Lc.N + 255) / 256)), dim3(256), 0, op->stream, (const cplx *)Lc.op->d_C, A, Lc.nz, Lc.ny, Lc.nx);
    rc = nd_dense_inverse(op, A, P->nc, W);
    if (!rc) HELM_LAUNCH(k3_transpose_sq, dim3(4096), dim3(256), 0, op->stream, (const cplx *)A, P->cinvT, P->nc);
    hipStreamSynchronize(op->stream);
    hipFree(A); hipFree(W);
    if (rc) return fail(rc, "3-D multigrid: coarsest inverse failed");
    return HELM_OK;
}

bool mg3_is_layer_preserving(const helm_op *op) { return op->mg3 && op->mg3->keep; }

int mg3_retreat(helm_op *op, int batch) {
    op->mg3_no_keep = true;
    mg3_destroy(op);
    return mg3_setup(op, batch);
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
    const int rc = P->keep ? cycle_keep(op, P, 0, nrhs, P->lv.size() > 1 ? out : nullptr) : cycle(op, P, 0, nrhs, P->lv.size() > 1 ? out : nullptr);
    L0.f = own_f;
    if (rc) return rc;
    if (P->lv.size() == 1) HIP_TRY(op, hipMemcpyAsync(out, L0.u, (size_t)nrhs * L0.N * sizeof(cplx), hipMemcpyDeviceToDevice, op->stream));
    return HELM_OK;
}

// ---- diagnostics exported through the C ABI (host side of the layer-preserving hierarchy, no GPU needed) ----------------
// One axis of n nodes (spacing h, npml layer nodes at each end, damping amplitude cpml), coarsened `level` times.  Returns the number of
// nodes nc of that level and writes, if the pointers are not null: x[nc] node coordinates, lay[nc] layer flags, lap[3 * nc] the factors
// L(-1), L(0), L(+1) (complex, interleaved re / im) for omega = (om_re, om_im); and for the transfer from this level to the next one:
// pc[2 * nc] / pw[2 * nc] the two coarse nodes and weights each node interpolates from, rf[ncn] / rw[3 * ncn] the centre node and the
// three weights of every node of the next level (ncn through *n_next).
extern "C" int helm_mg3_axis(int n, int npml, double h, double cpml, double om_re, double om_im, int level, double *x, int *lay, double *lap,
                             int *pc, double *pw, int *n_next, int *rf, double *rw) {
    if (n < 3 || npml < 2 || 2 * npml > n || level < 0) return -1;
    Ax3 a;
    a.x.resize(n); a.gam.assign(n, 0.0); a.lay.assign(n, 0);
    const double Lh = h * (npml - 1);
    for (int i = 0; i < n; ++i) a.x[i] = i * h;
    for (int k = 0; k < npml; ++k) {
        a.gam[k] = cpml * cos((M_PI / 2) * (k * h / Lh)); a.lay[k] = 1;
        a.gam[n - npml + k] = cpml * cos((M_PI / 2) * ((npml - 1 - k) * h / Lh)); a.lay[n - npml + k] = 1;
    }
    Ax3 c; std::vector<int> kept; std::vector<PTab> pt; std::vector<RTab> rt;
    for (int l = 0; l < level; ++l) { coarsen_axis(a, true, c, kept, pt, rt); a = c; }
    const int nc = a.n();
    if (x) for (int i = 0; i < nc; ++i) x[i] = a.x[i];
    if (lay) for (int i = 0; i < nc; ++i) lay[i] = a.lay[i];
    if (lap) {
        std::vector<cplx> Lt;
        lap_from_axis(a, std::complex<double>(om_re, om_im), Lt);
        for (size_t i = 0; i < Lt.size(); ++i) { lap[2 * i] = Lt[i].x; lap[2 * i + 1] = Lt[i].y; }
    }
    if (pc || pw || n_next || rf || rw) {
        coarsen_axis(a, true, c, kept, pt, rt);
        if (n_next) *n_next = c.n();
        for (int i = 0; i < nc; ++i) {
            if (pc) { pc[2 * i] = pt[i].c0; pc[2 * i + 1] = pt[i].c1; }
            if (pw) { pw[2 * i] = pt[i].w0; pw[2 * i + 1] = pt[i].w1; }
        }
        for (int i = 0; i < c.n(); ++i) {
            if (rf) rf[i] = rt[i].f;
            if (rw) { rw[3 * i] = rt[i].wl; rw[3 * i + 1] = rt[i].wc; rw[3 * i + 2] = rt[i].wr; }
        }
    }
    return nc;
}
