// Direct solver, dense inversion: in-place inverses of batches of pivot blocks F11 -- pivoted Gauss-Jordan sweeps for blocks of up to 64
// unknowns, blocked Gauss-Jordan (pivot sweep + rank-32 update per 32 columns, one launch per block step for the tree top) above that,
// one level of 2 x 2 block recursion first for the large plane matrices of the 3-D coarse solve.
#include "nd_gemm_body.hpp"

int g_recurse_min = -1;     // (helm_debug_inverse_bench overrides the block-recursion threshold)

namespace {

// ---- fast path for blocks of at most 32 x 32: 256 threads, no serial phase ------------------------------------------------
// The generic routine above spends ~1.7 us per elimination step (4000 cycles: a one-wave pivot search through shuffles, two
// fp64 divisions, integer divisions in the update loop); this one needs two barriers and ~500 cycles:
//   * thread (i = tid / 8, columns 4 (tid % 8) .. +3) owns four entries of row i for the whole elimination;
//   * the pivot of the next column is chosen by EVERY thread from 32 keys in LDS -- float(max(|re|, |im|)) with the row index in
//     the low five bits, so the search is one v_max_u32 reduction over eight 16-byte broadcast reads; the keys are written by
//     the threads that produce that column in the previous step;
//   * no row exchange (implicit pivoting) and no scaling of the pivot row (its 1 / d is applied once at the end); all old values are read
//     before the barrier and all new ones written after it;
//   * the reciprocal of the pivot is v_rcp_f64 + two Newton steps.
// (Measured and reverted: exchanging the pivot row and the multiplier column through small LDS buffers instead of the full matrix --
// fewer LDS bytes, but write -> barrier -> read makes three dependent LDS round trips per step instead of two: 65 -> 88 us per block step.)
struct Gj32 {
    cplx a[32][33];       // the matrix on entry, the result on return
    cplx b[32][33];       // second buffer: step k reads one and writes the other, so a step needs ONE barrier
    unsigned cand[2][32];
    cplx dinv[2][32];     // reciprocal of every row's entry in the column that is eliminated next
    int piv[32];          // sigma: pivot row of step k
    int sinv[32];         // step at which row r was the pivot
};
__device__ __forceinline__ double gj_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ unsigned gj_key(cplx v, int i) {
    const float m = (float)fmax(fabs(v.x), fabs(v.y));
    return (__float_as_uint(m) & ~31u) | (unsigned)i;
}
__device__ __forceinline__ cplx gj_recip(cplx d) {
    const double rr = gj_rcp(d.x * d.x + d.y * d.y);
    return cmake(d.x * rr, -d.y * rr);
}
// S.a must hold the matrix padded with the identity to 32 x 32; all 256 threads call.
// Implicit pivoting (rows stay where they are, S.piv[k] = sigma(k) = pivot row of step k, S.sinv its inverse) and deferred scaling of the
// pivot rows (see k_gj32w_inverse).  A step is a chain of latencies, so it is kept short:
//   * a thread keeps its four entries in registers for the whole elimination and publishes them to the buffer the NEXT step reads:
//     one barrier per step;
//   * the thread that produces row i's entry of column k + 1 also publishes the pivot key and the reciprocal of that entry, so step k + 1
//     starts with   read keys -> p;  read row p, 1 / d, own multiplier   and goes straight to the multiply-adds (the division is off the
//     critical path: it runs beside the other three entries' updates of the previous step).
// On return S.a holds the storage rows R with   inverse[i][sigma(k)] = R[sigma(i)][k].
// Measured with clock64 around the call (2.39 GHz, one workgroup): 48 000 cycles for 32 steps = 1500 per step for ~110 instructions per
// wave -- the step is bound by the number of instructions one wave has to issue one after the other, not by a particular latency.  A form
// in panels of four steps (the four columns of a panel in one half-wave: pivot search and pivot row by v_readlane, the other threads apply
// four steps at once after one barrier; bit-for-bit the same result) was built and measured: 3550 cycles for the four narrow steps + 1700
// for the rank-4 update per panel = the same 46-50 000 cycles; reverted.  The pivot search as an LDS atomic (ds_max_u32 by the 32 threads that hold the
// column, one word read by everybody instead of 32 keys and their maximum): 22.5 -> 29.2 us per block; reverted.
__device__ __forceinline__ void gj32(Gj32 &S, int n, int tid) {
    const int i = tid >> 3, jc = tid & 7, j0 = jc * 4;
    bool used = i >= n;
    cplx srow = cmake(1.0, 0.0);
    cplx out[4];
    #pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = S.a[i][j0 + q];
    if (jc == 0) { S.cand[0][i] = used ? (unsigned)i : gj_key(out[0], i); S.dinv[0][i] = gj_recip(out[0]); S.piv[i] = i; S.sinv[i] = i; }
    __syncthreads();
    cplx (*cur)[33] = S.a, (*nxt)[33] = S.b;
    // four steps per trip, so that the register that holds column k (out[k & 3]) is known at compile time: one select per step instead of four
    // compare-and-select groups, and no select chain for the next column's key
#define GJ32_STEP(Q_) do {                                                                                                         \
        const int k = k4 + (Q_);                                                                                                   \
        if (k >= n) break;                                                                                                         \
        const int pb = k & 1;                                                                                                      \
        const uint4 *c4 = reinterpret_cast<const uint4 *>(S.cand[pb]);                                                             \
        unsigned m = 0;                                                                                                            \
        _Pragma("unroll")                                                                                                          \
        for (int q = 0; q < 8; ++q) { const uint4 v = c4[q]; m = max(m, max(max(v.x, v.y), max(v.z, v.w))); }                       \
        const int p = (int)(m & 31u);                                                                                              \
        const cplx dinv = S.dinv[pb][p], f = cur[i][k];                                                                            \
        cplx pr[4];                                                                                                                \
        _Pragma("unroll")                                                                                                          \
        for (int q = 0; q < 4; ++q) pr[q] = cur[p][j0 + q];                                                                        \
        if (tid == 0) { S.piv[k] = p; S.sinv[p] = k; }                                                                             \
        const cplx fp = (i == p) ? cmake(0.0, 0.0) : cmul(f, dinv);                                                                \
        _Pragma("unroll")                                                                                                          \
        for (int q = 0; q < 4; ++q) {                                                                                              \
            out[q].x = fma(-fp.x, pr[q].x, out[q].x); out[q].x = fma(fp.y, pr[q].y, out[q].x);                                     \
            out[q].y = fma(-fp.x, pr[q].y, out[q].y); out[q].y = fma(-fp.y, pr[q].x, out[q].y);                                    \
        }                                                                                                                          \
        if (jc == (k4 >> 2)) out[Q_] = (i == p) ? cmake(1.0, 0.0) : cneg(fp);        /* column k of the running inverse */            \
        if (i == p) { srow = dinv; used = true; }                                                                                  \
        if (k + 1 < n) {                                                                                                           \
            if (jc == ((k + 1) >> 2)) {        /* key and reciprocal of the next column (unused rows only: their scale is still 1) */ \
                const cplx v = out[((Q_) + 1) & 3];                                                                                \
                S.cand[pb ^ 1][i] = used ? (unsigned)i : gj_key(v, i);                                                             \
                S.dinv[pb ^ 1][i] = gj_recip(v);                                                                                   \
            }                                                                                                                      \
            _Pragma("unroll")                                                                                                      \
            for (int q = 0; q < 4; ++q) nxt[i][j0 + q] = out[q];                                                                   \
        }                                                                                                                          \
        __syncthreads();                                                                                                           \
        cplx (*t_)[33] = cur; cur = nxt; nxt = t_;                                                                                 \
    } while (0)
    for (int k4 = 0; k4 < n; k4 += 4) { GJ32_STEP(0); GJ32_STEP(1); GJ32_STEP(2); GJ32_STEP(3); }
#undef GJ32_STEP
    #pragma unroll
    for (int q = 0; q < 4; ++q) S.a[i][j0 + q] = cmul(out[q], srow);
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_gj32_inverse(cplx *A0, int ld, long long stride, int n) {
    __shared__ Gj32 S;
    cplx *A = A0 + (long long)blockIdx.x * stride;
    const int tid = threadIdx.x, i = tid >> 3, j0 = (tid & 7) * 4;
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        S.a[i][j] = (i < n && j < n) ? A[(long long)i * ld + j] : cmake(i == j ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();
    gj32(S, n, tid);
    // inverse[row][sigma(j)] = R[sigma(row)][j]: this thread holds storage row i = sigma(row), i.e. row = sinv[i]
    const int row = S.sinv[i] & 31;
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        const int col = S.piv[j] & 31;
        if (i < n && j < n && row < n && col < n) A[(long long)row * ld + col] = S.a[i][j];
    }
}

// ---- throughput variant for thousands of small blocks: one WAVE per matrix, the matrix in registers -------------------------------
// Lane l holds half a row: row r = l & 31, columns 16 h .. 16 h + 15 with h = l >> 5 (16 complex = 64 VGPRs).  The 32 elimination steps
// are unrolled so that every register index is static.  Per step the wave exchanges three things through its own 1.5 KB of LDS (LDS
// operations of one wave execute in order, no barrier): the pivot keys of column k, the pivot row (written by its two owner lanes, read
// as broadcasts), and the multipliers a[r][k] for the half that does not hold column k.  Pivoting is implicit -- rows stay where they
// are, sigma(k) records the pivot row of step k -- and is undone when the result is stored: inv[i][sigma(k)] = R[sigma(i)][k].
// ~220 wave-instructions per step against ~840 for the four-wave kernel above: the leaf level's two launches of 16 384 blocks are
// issue-bound, so this is what they cost.
struct Gj32w {
    cplx prow[32];
    cplx fcol[32];
    unsigned cand[32];
    int sigma[32];       // pivot row of step k
    int sinv[32];        // step at which row r was the pivot
};
// the elimination itself: lane (r, h) holds a[c] = entry (r, 16 h + c) of the block padded with the identity; on return the storage rows of the inverse
// (see k_gj32w_inverse for the permutation that undoes the implicit pivoting)
__device__ __forceinline__ void gj32w_core(cplx (&a)[16], Gj32w &S, int n, int r, int h) {
    bool used = r >= n;                            // padding rows never pivot
    if (h == 0) { S.sigma[r] = r; S.sinv[r] = r; }  // (a singular block may leave entries unset: keep every index in range)
    // The pivot row is NOT scaled when it is chosen: the other rows are eliminated with the multiplier f / d against the unscaled row,
    // and the factor 1 / d of the pivot row rides along in `srow` until the end (every later operation on that row is linear in it).
    // That leaves 16 complex multiply-adds per lane and step -- scaling the row at once would double the fp64 work of a step.
    cplx srow = cmake(1.0, 0.0);
    #pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (k < n) {
            const int kc = k & 15, kh = k >> 4;
            // pivot keys and multipliers of column k, from the half that holds it (candidates are unused rows: their scale is still 1)
            if (h == kh) {
                S.cand[r] = used ? (unsigned)r : gj_key(a[kc], r);
                S.fcol[r] = a[kc];
            }
            __builtin_amdgcn_wave_barrier();
            const uint4 *c4 = reinterpret_cast<const uint4 *>(S.cand);
            unsigned m = 0;
            #pragma unroll
            for (int q = 0; q < 8; ++q) { const uint4 v = c4[q]; m = max(m, max(max(v.x, v.y), max(v.z, v.w))); }
            const int p = (int)(m & 31u);
            const cplx f = S.fcol[r];
            __builtin_amdgcn_wave_barrier();
            // the pivot row, written by its two owner lanes
            if (r == p) {
                #pragma unroll
                for (int c = 0; c < 16; ++c) S.prow[16 * h + c] = a[c];
                if (h == 0) { S.sigma[k] = p; S.sinv[p] = k; }
                used = true;
            }
            __builtin_amdgcn_wave_barrier();
            const cplx d = S.prow[k];
            const double rr = gj_rcp(d.x * d.x + d.y * d.y);
            const cplx dinv = cmake(d.x * rr, -d.y * rr);
            const cplx fp = (r == p) ? cmake(0.0, 0.0) : cmul(f, dinv);
            #pragma unroll
            for (int c = 0; c < 16; ++c) {
                const cplx pr = S.prow[16 * h + c];
                a[c].x = fma(-fp.x, pr.x, a[c].x); a[c].x = fma(fp.y, pr.y, a[c].x);
                a[c].y = fma(-fp.x, pr.y, a[c].y); a[c].y = fma(-fp.y, pr.x, a[c].y);
            }
            if (h == kh) a[kc] = (r == p) ? cmake(1.0, 0.0) : cneg(fp);        // column k of the running inverse
            if (r == p) srow = dinv;
            __builtin_amdgcn_wave_barrier();
        }
    }
    #pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = cmul(a[c], srow);
}

__global__ __launch_bounds__(256) void k_gj32w_inverse(cplx *A0, int ld, long long stride, int n, int nmat) {
    __shared__ Gj32w SW[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int mat = blockIdx.x * 4 + w;
    if (mat >= nmat) return;                       // whole waves leave together (no block-level barrier below)
    Gj32w &S = SW[w];
    cplx *A = A0 + (long long)mat * stride;
    const int r = lane & 31, h = lane >> 5;
    cplx a[16];
    #pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int j = 16 * h + c;
        a[c] = (r < n && j < n) ? A[(long long)r * ld + j] : cmake(r == j ? 1.0 : 0.0, 0.0);
    }
    gj32w_core(a, S, n, r, h);
    // inv[i][sigma(kcol)] = R[sigma(i)][kcol]: this lane holds storage row r = sigma(i), i.e. output row i = sinv[r]
    if (r < n) {
        const int i = S.sinv[r] & 31;
        #pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int kcol = 16 * h + c;
            const int col = S.sigma[kcol & 31] & 31;
            if (kcol < n && i < n && col < n) A[(long long)i * ld + col] = a[c];
        }
    }
}

// ---- in-place inverse of n x n blocks, n <= 64: one workgroup per matrix -------------------------------------------------
// (bottom of the block inversions: 32 by default; 64 (HELM_ND_GJ=64) gains a digit of accuracy, but its 64-step elimination
// is slower overall: 41.8 vs 35.9 ms per factorisation at 1024^2)
template <int NMAX, int NT = 256>
__global__ __launch_bounds__(NT) void k_gj_inverse(cplx *A0, int ld, long long stride, int n) {
    __shared__ cplx a[NMAX][NMAX + 1];
    __shared__ cplx fcol[NMAX];
    __shared__ int piv[NMAX];
    cplx *A = A0 + (long long)blockIdx.x * stride;
    const int tid = threadIdx.x;
    for (int e = tid; e < n * n; e += blockDim.x) a[e / n][e % n] = A[(long long)(e / n) * ld + e % n];
    __syncthreads();
    gj_lds<NMAX>(a, fcol, piv, n, tid, blockDim.x);
    for (int e = tid; e < n * n; e += blockDim.x) A[(long long)(e / n) * ld + e % n] = a[e / n][e % n];
}

// ---- blocked Gauss-Jordan inversion: panel kernel ----------------------------------------------------------------------
// In-place inverse of T (n x n) by block steps of nb <= 32 columns.  Step k with pivot block T_kk (rows / columns [k0, k0+nb)):
//     P = T_kk^-1 ;  R = P T[k, :] with R_k := P ;  C = T[:, k] with C_k := -I ;  T <- Z(T) - C R
// where Z zeroes block row k and block column k (the GEMM's masked beta).  This kernel makes R (nb x n) and C (n x nb) in
// scratch; every workgroup inverts the pivot block for itself (25 us, redundant but parallel) and then produces a 64-wide
// slice of R and a 64-tall slice of C.  Compared with the recursive 2 x 2 block inversion the chain of dependent launches is
// n / nb steps of two fat launches instead of ~6.8 n / 32 thin ones, which is what the upper tree levels were spending
// their time on.  Same pivots (the block-LU Schur complements), same accuracy class.
__global__ __launch_bounds__(256) void k_gj_panel(cplx *T0, int ld, long long stride, int n, int k0, int nb, cplx *Wc0, cplx *Wr0, long long wstride) {
    __shared__ Gj32 S;
    __shared__ cplx t[PNB][64 + 1];
    __shared__ int cperm[PNB];
    cplx *T = T0 + (long long)blockIdx.y * stride;
    cplx *Wc = Wc0 + (long long)blockIdx.y * wstride, *Wr = Wr0 + (long long)blockIdx.y * wstride;
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * 64;                     // this workgroup's slice [s0, s0 + 64) of the columns of R / rows of C
    {
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            S.a[i][j] = (i < nb && j < nb) ? T[(long long)(k0 + i) * ld + k0 + j] : cmake(i == j ? 1.0 : 0.0, 0.0);
        }
    }
    // row-panel slice T[k-rows, s0 .. s0+63] -> LDS ; column-panel slice copied out (C_k = -I)
    for (int e = tid; e < PNB * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        t[r][c] = (r < nb && s0 + c < n) ? T[(long long)(k0 + r) * ld + s0 + c] : cmake(0.0, 0.0);
    }
    for (int e = tid; e < 64 * PNB; e += 256) {
        const int r = e >> 5, c = e & 31;
        if (s0 + r >= n || c >= nb) continue;
        const int gr = s0 + r;
        cplx v = T[(long long)gr * ld + k0 + c];
        if (gr >= k0 && gr < k0 + nb) v = (gr - k0 == c) ? cmake(-1.0, 0.0) : cmake(0.0, 0.0);
        Wc[(long long)gr * PNB + c] = v;
    }
    __syncthreads();
    gj32(S, nb, tid);
    if (tid < PNB) cperm[tid] = S.piv[tid] & 31;            // P[r][sigma(j)] = S.a[sigma(r)][j]
    __syncthreads();
    // R slice = P * t  (nb x 64) with P[r][cperm[j]] = S.a[r][j]; columns inside the pivot block get P itself.
    // Thread (r = tid / 8, eight consecutive columns): ten LDS reads per eight complex multiply-adds.
    {
        const int r = tid >> 3, c0 = (tid & 7) * 8;
        cplx acc[8];
        #pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = cmake(0.0, 0.0);
        const int sr = cperm[r];                             // storage row of output row r
        for (int j = 0; j < nb; ++j) {
            const cplx a = S.a[sr][j];
            const int tj = cperm[j];
            #pragma unroll
            for (int i = 0; i < 8; ++i) cfma(acc[i], a, t[tj][c0 + i]);
        }
        if (r < nb) {
            #pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gc = s0 + c0 + i;
                if (gc < n && !(gc >= k0 && gc < k0 + nb)) Wr[(long long)r * n + gc] = acc[i];
            }
        }
    }
    if (s0 < k0 + nb && s0 + 64 > k0)                    // the slice that holds the pivot block's columns
        for (int e = tid; e < nb * nb; e += 256) {
            const int r = e / nb, j = e % nb;
            const int gc = k0 + cperm[j];
            if (gc >= s0 && gc < s0 + 64) Wr[(long long)r * n + gc] = S.a[cperm[r]][j];
        }
}

// Look-ahead form of the panel step.  The Gauss-Jordan sweep of the next pivot block only needs that 32 x 32 block, so
// it runs on a second stream beside the rank-32 update of the whole front: k_gj_pivot applies the pending update to its
// block privately (the GEMM skips it, GemmRows::sk0/sk1), inverts it and leaves P in Pb; k_gj_slices then forms the
// panels R_k = P T[k-rows, :], C_k = T[:, k-cols] from the updated front.
// LDS of the sweep: wc | wr (2 x 32 x 33 complex) while the pending update is applied to the block, then the Gj32 state in the same place
constexpr int GJ_PIVOT_LDS = (int)sizeof(Gj32) + PNB * (int)sizeof(int);
static_assert(sizeof(Gj32) >= 2 * PNB * (PNB + 1) * sizeof(cplx), "the two panels lie over Gj32's buffers");

__device__ __forceinline__ void gj_pivot_body(const cplx *T0, int ld, long long stride, int n, int k0, int nb, const cplx *Wc0, const cplx *Wr0, long long wstride,
                                              cplx *Pb0, long long pstride, int mat, char *lds) {
    cplx (&wc)[PNB][PNB + 1] = *reinterpret_cast<cplx (*)[PNB][PNB + 1]>(lds);
    cplx (&wr)[PNB][PNB + 1] = *reinterpret_cast<cplx (*)[PNB][PNB + 1]>(lds + PNB * (PNB + 1) * sizeof(cplx));
    Gj32 &S = *reinterpret_cast<Gj32 *>(lds);
    int *cperm = reinterpret_cast<int *>(lds + sizeof(Gj32));
    const cplx *T = T0 + (long long)mat * stride;
    const cplx *Wc = Wc0 + (long long)mat * wstride, *Wr = Wr0 + (long long)mat * wstride;
    cplx *Pb = Pb0 + (long long)mat * pstride;
    const int tid = threadIdx.x;
    const int i = tid >> 3, j0 = (tid & 7) * 4;
    const bool pending = k0 > 0;                          // the update of step k-1 (full width PNB) has not touched this block
    if (pending) {
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            wc[i][j] = i < nb ? Wc[(long long)(k0 + i) * PNB + j] : cmake(0.0, 0.0);
            wr[i][j] = j < nb ? Wr[(long long)i * n + k0 + j] : cmake(0.0, 0.0);
        }
        __syncthreads();
    }
    {
        cplx v[4];
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            v[q] = (i < nb && j < nb) ? T[(long long)(k0 + i) * ld + k0 + j] : cmake(i == j ? 1.0 : 0.0, 0.0);
        }
        if (pending) {
            cplx acc[4];
            #pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = cmake(0.0, 0.0);
            for (int p = 0; p < PNB; ++p) {
                const cplx a = wc[i][p];
                #pragma unroll
                for (int q = 0; q < 4; ++q) cfma(acc[q], a, wr[p][j0 + q]);
            }
            #pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = csub(v[q], acc[q]);
            __syncthreads();                              // the panels are read: S may take their place
        }
        #pragma unroll
        for (int q = 0; q < 4; ++q) S.a[i][j0 + q] = v[q];
    }
    __syncthreads();
    gj32(S, nb, tid);
    if (tid < PNB) cperm[tid] = S.piv[tid] & 31;            // P[r][sigma(j)] = S.a[sigma(r)][j]
    __syncthreads();
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        if (i < nb && j < nb) Pb[i * PNB + cperm[j]] = S.a[cperm[i]][j];
    }
}

__global__ __launch_bounds__(256) void k_gj_pivot(const cplx *T0, int ld, long long stride, int n, int k0, int nb, const cplx *Wc0, const cplx *Wr0, long long wstride,
                                                  cplx *Pb0, long long pstride) {
    __shared__ __attribute__((aligned(16))) char lds[GJ_PIVOT_LDS];
    gj_pivot_body(T0, ld, stride, n, k0, nb, Wc0, Wr0, wstride, Pb0, pstride, blockIdx.x, lds);
}

// Rank-32 update of step k and the Gauss-Jordan sweep of pivot block k+1 in ONE launch: the workgroups of one extra z-slice of the grid
// do the sweeps (one per matrix, the rest of that slice leaves at once), all others are tiles of the masked update, which skips the pivot
// block.  The sweep (31 us, one workgroup) hides behind the update without a second stream: cross-stream event hops cost 15-20 us apiece.
// the same fused launch with the matrix-core tile body (generation 7)
template <int WM, int WN, int MT, int NT, int KS>
__global__ __launch_bounds__(256, 2) void k_zgemm3_la(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                                      const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, GemmRows R, GjPivotArgs pv) {
    constexpr int TBYTES = 2 * (KS / 4) * (MT * WM + NT * WN) * 64 * (int)sizeof(cplx);
    constexpr int LDS = TBYTES > GJ_PIVOT_LDS ? TBYTES : GJ_PIVOT_LDS;
    __shared__ __attribute__((aligned(16))) char lds[LDS];
    if (blockIdx.z == 0) {
        const int mat = blockIdx.y * gridDim.x + blockIdx.x;
        if (mat < pv.batch) gj_pivot_body(pv.T0, pv.ld, pv.stride, pv.n, pv.k0, pv.nb, pv.Wc0, pv.Wr0, pv.wstride, pv.Pb0, pv.pstride, mat, lds);
        return;
    }
    zgemm3_body<WM, WN, MT, NT, 0, KS>(M, Nn, K, alpha, A0 - sa, lda, sa, B0 - sb, ldb, sb, beta, C0 - sc, ldc, sc, R, reinterpret_cast<cplx *>(lds));
}

// ---- one launch per block step of the blocked Gauss-Jordan inversion (round 4) ---------------------------------------------------------------
// The step   T <- Z(T) - C R   with   R = P T[k rows, :] (R_k := P),  C = T[:, k cols] (C_k := -I)   needed two launches because the update overwrites
// the pivot rows and columns that the other tiles still read: k_gj_slices copied the panels out first (18 us of the 54 us a step of a 1024-wide
// front takes).  With TWO copies of the matrix -- a step reads one and writes the other -- nothing a tile reads is written in the same launch,
// so every 64 x 32 tile forms its own slab of R (P times the raw pivot rows of its 32 columns: 32^3 multiply-adds, redundant across the row
// tiles, hidden behind the sweep) and takes its slab of C straight from the source.  The sweep of the NEXT pivot block rides in the first
// z-slice as before; it applies the step to its 32 x 32 block privately, now from the source matrix and P (two 32^3 products) instead of
// the panels.  The workspace W holds the second copy (n^2 per matrix), the two alternating P buffers come from the handle.
struct GjStepArgs {
    const cplx *Ta; int lda; long long sa;     // source: read-only in this launch
    cplx *Tb; int ldb; long long sb;           // destination
    int n, k0, nb;                             // this step's pivot block: rows / columns [k0, k0 + nb)
    const cplx *P; cplx *Pn; long long sp;     // its inverse (PNB x PNB per matrix) ; where the sweep leaves the next block's
    int k1, nb1;                               // the next pivot block (nb1 == 0: none)
    int batch, nsw;                            // matrices; z-slices of the grid that hold the sweeps (one workgroup per matrix)
};
// LDS: the sweep state (35 KB) -- the two 32 x 32 blocks of the private update and the tiles' P and R slab lie over it.  (A first version kept four
// blocks, 68 KB: alone on the GPU the same speed, but beside the solve kernels of the previous work item a workgroup of that size waits for a
// compute unit with that much LDS free -- the products' launches took 171 instead of 156 us on average inside the pipeline; with 35 KB 156.)
constexpr int GJS_LDS = (int)sizeof(Gj32) + PNB * (int)sizeof(int);
static_assert(GJS_LDS >= 2 * PNB * (PNB + 1) * (int)sizeof(cplx), "two 32 x 32 blocks lie over the sweep state");

__global__ __launch_bounds__(256, 2) void k_gj_step(GjStepArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[GJS_LDS];
    const int tid = threadIdx.x;
    const int k0 = a.k0, nb = a.nb, n = a.n;
    cplx (&X0)[PNB][PNB + 1] = *reinterpret_cast<cplx (*)[PNB][PNB + 1]>(lds);
    cplx (&X1)[PNB][PNB + 1] = *reinterpret_cast<cplx (*)[PNB][PNB + 1]>(lds + PNB * (PNB + 1) * sizeof(cplx));
    if ((int)blockIdx.z < a.nsw) {                           // ---- sweep of the next pivot block (one workgroup per matrix, the first nsw z-slices)
        const int mat = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (mat >= a.batch || a.nb1 == 0) return;
        const cplx *Ta = a.Ta + (long long)mat * a.sa, *P = a.P + (long long)mat * a.sp;
        cplx *Pn = a.Pn + (long long)mat * a.sp;
        const int k1 = a.k1, nb1 = a.nb1, lda = a.lda;
        Gj32 &S = *reinterpret_cast<Gj32 *>(lds);
        int *cperm = reinterpret_cast<int *>(lds + sizeof(Gj32));
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        // The step applied to the next pivot block privately:  T11 - C_k (P * rows_k)  with two 32^3 products.  Round 5: both on the matrix cores (wave w owns
        // the 16 x 16 block (w >> 1, w & 1) of each product; fragments as in the tiles below) -- with one multiply-add per thread and entry they took 5 of a
        // step's 36 us, and the sweep of the next block is what a step waits for.
        const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
        const int br = wave >> 1, bc = wave & 1;
        cplx lreg[4], vm[4];
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            X0[i][j] = (i < nb && j < nb1) ? Ta[(long long)(k0 + i) * lda + k1 + j] : cmake(0.0, 0.0);       // pivot rows, columns of the next block
            X1[i][j] = (i < nb && j < nb) ? P[i * PNB + j] : cmake(0.0, 0.0);
            lreg[q] = (i < nb1 && j < nb) ? Ta[(long long)(k1 + i) * lda + k0 + j] : cmake(0.0, 0.0);        // C_k, rows of the next block
            const int mr = 16 * br + lq + 4 * q, mc = 16 * bc + lr;                                          // the next block itself, in the layout the products come out in
            vm[q] = (mr < nb1 && mc < nb1) ? Ta[(long long)(k1 + mr) * lda + k1 + mc] : cmake(mr == mc ? 1.0 : 0.0, 0.0);
        }
        __syncthreads();
        v4f64 er = {0.0, 0.0, 0.0, 0.0}, ei = {0.0, 0.0, 0.0, 0.0};
        #pragma unroll
        for (int ks = 0; ks < PNB / 4; ++ks) {                // R_k (columns of the next block) = P * rows_k
            const cplx x = X1[16 * br + lr][4 * ks + lq], y = X0[4 * ks + lq][16 * bc + lr];
            er = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.x, er, 0, 0, 0);
            ei = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.y, ei, 0, 0, 0);
            er = __builtin_amdgcn_mfma_f64_16x16x4f64(-x.y, y.y, er, 0, 0, 0);
            ei = __builtin_amdgcn_mfma_f64_16x16x4f64(x.y, y.x, ei, 0, 0, 0);
        }
        __syncthreads();
        #pragma unroll
        for (int q = 0; q < 4; ++q) { X0[16 * br + lq + 4 * q][16 * bc + lr] = cmake(er[q], ei[q]); X1[i][j0 + q] = lreg[q]; }
        __syncthreads();
        er = v4f64{0.0, 0.0, 0.0, 0.0}; ei = v4f64{0.0, 0.0, 0.0, 0.0};
        #pragma unroll
        for (int ks = 0; ks < PNB / 4; ++ks) {                // C_k * R_k
            const cplx x = X1[16 * br + lr][4 * ks + lq], y = X0[4 * ks + lq][16 * bc + lr];
            er = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.x, er, 0, 0, 0);
            ei = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.y, ei, 0, 0, 0);
            er = __builtin_amdgcn_mfma_f64_16x16x4f64(-x.y, y.y, er, 0, 0, 0);
            ei = __builtin_amdgcn_mfma_f64_16x16x4f64(x.y, y.x, ei, 0, 0, 0);
        }
        __syncthreads();                                     // the two blocks are read: S takes their place
        #pragma unroll
        for (int q = 0; q < 4; ++q) S.a[16 * br + lq + 4 * q][16 * bc + lr] = csub(vm[q], cmake(er[q], ei[q]));
        __syncthreads();
        // (round 5, measured and not kept: the sweep by one wave with the block in registers (gj32w_core) while the other three waves leave -- 36 -> 43 us
        // per step of the 1024-wide front: a wave that issues all 64 multiply-adds of a step itself takes longer than four waves and a barrier)
        gj32(S, nb1, tid);
        if (tid < PNB) cperm[tid] = S.piv[tid] & 31;         // P[r][sigma(j)] = S.a[sigma(r)][j]
        __syncthreads();
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            if (i < nb1 && j < nb1) Pn[i * PNB + cperm[j]] = S.a[cperm[i]][j];
        }
        return;
    }
    // ---- a 64 x 32 tile of the update, both products on the matrix cores (v_mfma_f64_16x16x4_f64: A lane l = A[l % 16][l / 16], B lane l = B[l / 16][l % 16],
    // D register q of lane l = D[l / 16 + 4 q][l % 16]; four real instructions per complex block and k step of 4, as in zgemm3_body)
    const int mat = blockIdx.z - a.nsw;
    const cplx *Ta = a.Ta + (long long)mat * a.sa, *P = a.P + (long long)mat * a.sp;
    cplx *Tb = a.Tb + (long long)mat * a.sb;
    const int lda = a.lda, ldb = a.ldb;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 32;
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    cplx (&Ps)[PNB][PNB + 1] = X0;
    cplx (&Bs)[PNB][PNB + 1] = X1;
    // this lane's fragments of the C slab T[tile rows, k cols] (-I on the pivot rows): row 16 wave + lr, columns 4 ks + lq -- straight from the source
    cplx afr[PNB / 4];
    {
        const int gr = r0 + 16 * wave + lr;
        const bool prow = gr >= k0 && gr < k0 + nb;
        #pragma unroll
        for (int ks = 0; ks < PNB / 4; ++ks) {
            const int j = 4 * ks + lq;
            cplx x = cmake(0.0, 0.0);
            if (gr < n && j < nb) x = prow ? cmake(gr - k0 == j ? -1.0 : 0.0, 0.0) : Ta[(long long)gr * lda + k0 + j];
            afr[ks] = x;
        }
    }
    #pragma unroll
    for (int l = 0; l < 4; ++l) {                            // P, and the raw pivot rows of the tile's columns (the identity where they are pivot columns: R_k = P)
        const int e = tid + 256 * l, r = e >> 5, c = e & 31, gc = c0 + c;
        Ps[r][c] = (r < nb && c < nb) ? P[r * PNB + c] : cmake(0.0, 0.0);
        cplx x = cmake(0.0, 0.0);
        if (r < nb && gc < n) {
            if (gc >= k0 && gc < k0 + nb) x = cmake(gc - k0 == r ? 1.0 : 0.0, 0.0);
            else x = Ta[(long long)(k0 + r) * lda + gc];
        }
        Bs[r][c] = x;
    }
    __syncthreads();
    {
        const int br = wave >> 1, bc = wave & 1;             // wave -> one 16 x 16 block of the 32 x 32 slab R = P * (raw pivot rows)
        v4f64 er = {0.0, 0.0, 0.0, 0.0}, ei = {0.0, 0.0, 0.0, 0.0};
        #pragma unroll
        for (int ks = 0; ks < PNB / 4; ++ks) {
            const cplx x = Ps[16 * br + lr][4 * ks + lq], y = Bs[4 * ks + lq][16 * bc + lr];
            er = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.x, er, 0, 0, 0);
            ei = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.y, ei, 0, 0, 0);
            er = __builtin_amdgcn_mfma_f64_16x16x4f64(-x.y, y.y, er, 0, 0, 0);
            ei = __builtin_amdgcn_mfma_f64_16x16x4f64(x.y, y.x, ei, 0, 0, 0);
        }
        __syncthreads();                                     // the raw rows are read: the slab takes their place
        #pragma unroll
        for (int q = 0; q < 4; ++q) Bs[16 * br + lq + 4 * q][16 * bc + lr] = cmake(er[q], ei[q]);
    }
    __syncthreads();
    v4f64 cr[2], ci[2];                                      // wave -> rows 16 wave .. + 15 of the tile, both 16-column blocks
    #pragma unroll
    for (int j = 0; j < 2; ++j) { cr[j] = v4f64{0.0, 0.0, 0.0, 0.0}; ci[j] = v4f64{0.0, 0.0, 0.0, 0.0}; }
    #pragma unroll
    for (int ks = 0; ks < PNB / 4; ++ks) {
        const cplx x = afr[ks];
        #pragma unroll
        for (int j = 0; j < 2; ++j) {
            const cplx y = Bs[4 * ks + lq][16 * j + lr];
            cr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.x, cr[j], 0, 0, 0);
            ci[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, y.y, ci[j], 0, 0, 0);
            cr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x.y, y.y, cr[j], 0, 0, 0);
            ci[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x.y, y.x, ci[j], 0, 0, 0);
        }
    }
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int gr = r0 + 16 * wave + lq + 4 * q;
        if (gr >= n) continue;
        const bool prow = gr >= k0 && gr < k0 + nb;
        #pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gc = c0 + 16 * j + lr;
            if (gc >= n) continue;
            const bool zero = prow || (gc >= k0 && gc < k0 + nb);
            const cplx cin = zero ? cmake(0.0, 0.0) : Ta[(long long)gr * lda + gc];
            Tb[(long long)gr * ldb + gc] = csub(cin, cmake(cr[j][q], ci[j][q]));
        }
    }
}

// dst[mat][r][c] = src[mat][r][c] for n x n blocks with different leading dimensions (the odd step count of k_gj_step leaves the result in W)
__global__ __launch_bounds__(256) void k_copy_blocks(const cplx *src, int ld_src, long long ss, cplx *dst, int ldd, long long sd, int n) {
    const cplx *s = src + (long long)blockIdx.y * ss;
    cplx *d = dst + (long long)blockIdx.y * sd;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * 256) {
        const int r = (int)(e / n), c = (int)(e % n);
        d[(long long)r * ldd + c] = s[(long long)r * ld_src + c];
    }
}

__global__ __launch_bounds__(256) void k_gj_slices(const cplx *T0, int ld, long long stride, int n, int k0, int nb, cplx *Wc0, cplx *Wr0, long long wstride,
                                                   const cplx *Pb0, long long pstride) {
    __shared__ cplx P[PNB][PNB + 1];
    __shared__ cplx t[PNB][64 + 1];
    const cplx *T = T0 + (long long)blockIdx.y * stride;
    cplx *Wc = Wc0 + (long long)blockIdx.y * wstride, *Wr = Wr0 + (long long)blockIdx.y * wstride;
    const cplx *Pb = Pb0 + (long long)blockIdx.y * pstride;
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * 64;
    for (int e = tid; e < PNB * PNB; e += 256) {
        const int r = e >> 5, c = e & 31;
        P[r][c] = (r < nb && c < nb) ? Pb[e] : cmake(0.0, 0.0);
    }
    for (int e = tid; e < PNB * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        t[r][c] = (r < nb && s0 + c < n) ? T[(long long)(k0 + r) * ld + s0 + c] : cmake(0.0, 0.0);
    }
    for (int e = tid; e < 64 * PNB; e += 256) {
        const int r = e >> 5, c = e & 31;
        if (s0 + r >= n || c >= nb) continue;
        const int gr = s0 + r;
        cplx v = T[(long long)gr * ld + k0 + c];
        if (gr >= k0 && gr < k0 + nb) v = (gr - k0 == c) ? cmake(-1.0, 0.0) : cmake(0.0, 0.0);
        Wc[(long long)gr * PNB + c] = v;
    }
    __syncthreads();
    const int r = tid >> 3, c0 = (tid & 7) * 8;
    cplx acc[8];
    #pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = cmake(0.0, 0.0);
    for (int j = 0; j < nb; ++j) {
        const cplx a = P[r][j];
        #pragma unroll
        for (int i = 0; i < 8; ++i) cfma(acc[i], a, t[j][c0 + i]);
    }
    if (r < nb) {
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gc = s0 + c0 + i;
            if (gc >= n) continue;
            Wr[(long long)r * n + gc] = (gc >= k0 && gc < k0 + nb) ? P[r][gc - k0] : acc[i];
        }
    }
}

// scratch for the alternating pivot-block inverses of k_gj_step (2 PNB^2 per matrix): the handle's own, or one process-wide buffer for the
// diagnostic entry points (single-threaded)
static cplx *gj_pbuf(helm_op *op, int batch) {
    const size_t need = (size_t)batch * 2 * PNB * PNB * sizeof(cplx);
    if (op) {
        if (op->gjp_bytes < need) {
            if (op->gjp_buf) { hipStreamSynchronize(op->stream); helm_pool_free(op->device, op->gjp_buf, op->gjp_bytes); op->gjp_buf = nullptr; op->gjp_bytes = 0; }
            op->gjp_buf = (cplx *)helm_pool_alloc(op->device, need);
            op->gjp_bytes = op->gjp_buf ? need : 0;
        }
        return op->gjp_buf;
    }
    static cplx *buf = nullptr; static size_t bytes = 0;
    if (bytes < need) {
        if (buf) { hipDeviceSynchronize(); hipFree(buf); buf = nullptr; bytes = 0; }
        if (hipMalloc((void **)&buf, need) == hipSuccess) bytes = need; else buf = nullptr;
    }
    return buf;
}

}  // namespace

void launch_zgemm3_la(hipStream_t st, bool latency_tile, int nb, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                      cplx beta, cplx *C, int ldc, long long sc, const GemmRows &R, const GjPivotArgs &pv) {
    if (latency_tile) {
        dim3 grid((Nn + 31) / 32, (M + 31) / 32, nb + 1);
        ZG_LAUNCH((k_zgemm3_la<2, 2, 1, 1, 16>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R, pv);
    } else {
        dim3 grid((Nn + 31) / 32, (M + 63) / 64, nb + 1);
        ZG_LAUNCH((k_zgemm3_la<2, 2, 2, 1, 8>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R, pv);
    }
}

// in-place inverse of `batch` n x n blocks (row-major, leading dimension ld, batch stride `stride`); W: workspace with batch stride ws, at least
// n * n elements per matrix.  align 2: the coupled two-field system (halves split between cells, 64-wide pivot windows: it is far worse conditioned);
// base: pivot-window size forced by the caller (0: 32, or 64 for align 2).
//   n <= base            one pivoted Gauss-Jordan sweep per matrix in LDS: a wave per matrix in registers from 2048 matrices up (k_gj32w_inverse: the chip
//                        is issue-bound there), the latency-built 256-thread kernel below that (k_gj32_inverse), k_gj_inverse<64> for 33..64
//   blocked Gauss-Jordan everything above the base with 32-wide windows: per 32 columns a pivot sweep + panels + ONE masked n x n x 32 update;
//                        fronts of nd_gjstep_min..1536 unknowns (the tree top) in one launch per block step (k_gj_step, two copies of the matrix);
//                        larger single matrices with the sweep of step k + 1 riding in the update of step k (k_zgemm3_la)
//   2 x 2 block recursion first for n >= 3000 (the 3713-wide plane inverses of the 3-D coarse solve: three quarters of the multiply-adds in products with
//                        an inner dimension of n / 2 instead of HBM-bound rank-32 updates, 15.9 -> 14.5 ms; a second level and recursion at 1857 lose),
//                        and all the way down for the 64-wide windows of the coupled system
void invert(helm_op *op, cplx *M, int ld, long long stride, int n, int batch, cplx *W, long long ws, int align, int base) {
    hipStream_t st = op ? op->stream : nullptr;
    const int gj_base = base ? base : (align == 2 ? 64 : 32);
    const int recurse_min = g_recurse_min > 0 ? g_recurse_min : 3000;
    const bool recurse = align == 1 && n >= recurse_min && (long long)n * n <= ws;
    const int dbatch = std::max(1, batch / tl_nf_div);            // the batch the choices are made for (one frequency's share of a multi-frequency launch)
    if (!recurse && dbatch <= 1024 && n > gj_base && gj_base == 32 && (long long)2 * PNB * n <= ws) {
        cplx *Wc = W, *Wr = W + (long long)PNB * n;
        const helm_tuning tune = helm_tuning_now();
        if (tune.nd_gjstep && n >= tune.nd_gjstep_min && n <= 1536 && (long long)n * n <= ws && dbatch <= 32768 && batch <= 32768) {
            // one launch per block step: the second copy of the matrix lives in W, the two P buffers in the handle's scratch
            cplx *Pb = gj_pbuf(op, batch);
            if (Pb) {
                const long long sp = 2LL * PNB * PNB;
                HELM_LAUNCH(k_gj_pivot, dim3(batch), dim3(256), 0, st, M, ld, stride, n, 0, std::min(PNB, n), (const cplx *)nullptr, (const cplx *)nullptr, ws, Pb, sp);
                const bool ext = op && op->profiling && tune.prof_ext != 0;
                int step = 0;
                for (int k0 = 0; k0 < n; k0 += PNB, ++step) {
                    const int nb = std::min(PNB, n - k0);
                    GjStepArgs a;
                    const bool fromM = (step & 1) == 0;
                    a.Ta = fromM ? M : W; a.lda = fromM ? ld : n; a.sa = fromM ? stride : ws;
                    a.Tb = fromM ? W : M; a.ldb = fromM ? n : ld; a.sb = fromM ? ws : stride;
                    a.n = n; a.k0 = k0; a.nb = nb;
                    a.P = Pb + (step & 1) * PNB * PNB; a.Pn = Pb + ((step + 1) & 1) * PNB * PNB; a.sp = sp;
                    a.k1 = k0 + PNB; a.nb1 = k0 + PNB < n ? std::min(PNB, n - k0 - PNB) : 0;
                    a.batch = batch;
                    const int per_slice = ((n + 31) / 32) * ((n + 63) / 64);
                    a.nsw = (batch + per_slice - 1) / per_slice;
                    // booked with the products (mode 5: update + pivot sweep): 8 n^2 nb flop, the matrix read and written once
                    ExtArm arm(op, ext, 8.0 * n * (double)n * nb * batch, 16.0 * (2.0 * n * (double)n + 2.0 * n * nb) * batch, n, n, nb, batch, 5);
                    const dim3 grid((n + 31) / 32, (n + 63) / 64, batch + a.nsw);
                    ZG_LAUNCH(k_gj_step, grid, a);
                }
                if (step & 1) HELM_LAUNCH(k_copy_blocks, dim3((unsigned)std::min<long long>(((long long)n * n + 255) / 256, 1024), batch), dim3(256), 0, st, (const cplx *)W, n, ws, M, ld, stride, n);
                return;
            }
        }
        // look-ahead: the pivot block of step k + 1 is inverted beside the update of step k, in the same launch (one sweep per matrix rides in the first
        // z-slice of the update's grid: that slice must have a workgroup for each).  The dense plane inverses of the 3-D coarse solve, n = 3713: -7 %.
        if (n >= 512 && (long long)((n + 63) / 64) * ((n + 31) / 32) >= batch && (long long)2 * PNB * n + PNB * PNB <= ws) {
            cplx *Pb = W + (long long)2 * PNB * n;
            HELM_LAUNCH(k_gj_pivot, dim3(batch), dim3(256), 0, st, M, ld, stride, n, 0, std::min(PNB, n), Wc, Wr, ws, Pb, ws);
            for (int k0 = 0; k0 < n; k0 += PNB) {
                const int nb = std::min(PNB, n - k0);
                HELM_LAUNCH(k_gj_slices, dim3((n + 63) / 64, batch), dim3(256), 0, st, M, ld, stride, n, k0, nb, Wc, Wr, ws, Pb, ws);
                GemmRows R; R.dense = 1; R.zr0 = k0; R.zr1 = k0 + nb; R.zc0 = k0; R.zc1 = k0 + nb;
                GjPivotArgs pv;
                if (k0 + PNB < n) {                                      // the sweep of the next pivot block rides along; the update leaves that block alone
                    const int k1 = k0 + PNB, nb1 = std::min(PNB, n - k1);
                    R.sk0 = k1; R.sk1 = k1 + nb1;
                    pv.T0 = M; pv.ld = ld; pv.stride = stride; pv.n = n; pv.k0 = k1; pv.nb = nb1; pv.Wc0 = Wc; pv.Wr0 = Wr; pv.wstride = ws;
                    pv.Pb0 = Pb; pv.pstride = ws; pv.batch = batch;
                    R.la = &pv;
                }
                gemm(op, n, n, nb, cmake(-1, 0), Wc, PNB, ws, Wr, n, ws, cmake(1, 0), M, ld, stride, batch, &R);
            }
            return;
        }
        for (int k0 = 0; k0 < n; k0 += PNB) {
            const int nb = std::min(PNB, n - k0);
            for (int b0 = 0; b0 < batch; b0 += 65535) {
                const int nbt = std::min(65535, batch - b0);
                HELM_LAUNCH(k_gj_panel, dim3((n + 63) / 64, nbt), dim3(256), 0, st, M + b0 * stride, ld, stride, n, k0, nb, Wc + b0 * ws, Wr + b0 * ws, ws);
            }
            GemmRows R; R.dense = 1; R.zr0 = k0; R.zr1 = k0 + nb; R.zc0 = k0; R.zc1 = k0 + nb;
            gemm(op, n, n, nb, cmake(-1, 0), Wc, PNB, ws, Wr, n, ws, cmake(1, 0), M, ld, stride, batch, &R);
        }
        return;
    }
    if (n <= gj_base) {
        for (int b0 = 0; b0 < batch; b0 += 1 << 20) {
            const int nb = std::min(1 << 20, batch - b0);
            if (n <= 32 && dbatch >= 2048) HELM_LAUNCH(k_gj32w_inverse, dim3((nb + 3) / 4), dim3(256), 0, st, M + b0 * stride, ld, stride, n, nb);
            else if (n <= 32) HELM_LAUNCH(k_gj32_inverse, dim3(nb), dim3(256), 0, st, M + b0 * stride, ld, stride, n);
            else HELM_LAUNCH(k_gj_inverse<64>, dim3(nb), dim3(256), 0, st, M + b0 * stride, ld, stride, n);
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

int nd_dense_inverse(helm_op *op, cplx *M, int n, cplx *W) {
    invert(op, M, n, (long long)n * n, n, 1, W, (long long)n * n);
    return check_kernels(op, "dense inverse");
}

extern "C" int helm_debug_inverse(int device, int n, double *A, int batch) {
    helm_tuning_refresh();
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

// times `reps` in-place inversions of one n x n matrix (the caller's A, uploaded once; an inverse of an inverse is as good a test
// matrix as the original) with the 2 x 2 block recursion applied from `recurse_n` unknowns up (0: the default policy)
extern "C" int helm_debug_inverse_bench(int device, int n, const double *A, int reps, int recurse_n, double *ms_out) {
    helm_tuning_refresh();
    if (hipSetDevice(device) != hipSuccess) return HELM_ERR_DEVICE;
    cplx *dA, *dW;
    const size_t na = (size_t)n * n;
    if (hipMalloc((void **)&dA, na * 16) != hipSuccess || hipMalloc((void **)&dW, na * 16) != hipSuccess) return HELM_ERR_DEVICE;
    hipMemcpy(dA, A, na * 16, hipMemcpyHostToDevice);
    g_recurse_min = recurse_n > 0 ? recurse_n : -1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    invert((helm_op *)nullptr, dA, n, (long long)n * n, n, 1, dW, (long long)n * n);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) invert((helm_op *)nullptr, dA, n, (long long)n * n, n, 1, dW, (long long)n * n);
    hipEventRecord(e1, nullptr);
    const hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / std::max(1, reps);
    g_recurse_min = -1;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(dA); hipFree(dW);
    return e == hipSuccess ? HELM_OK : HELM_ERR_DEVICE;
}

