// Direct solver, dense products: k_zgemm3 (the matrix-core tile kernel of nd_gemm_body.hpp), its tile choice per shape, split-K for
// thin products over a few big matrices, and the per-launch bookkeeping the roofline of bench.py is computed from.
#include "nd_gemm_body.hpp"

thread_local hipEvent_t tl_ev0 = nullptr, tl_ev1 = nullptr;
thread_local int tl_nf_div = 1;
int g_gemm_tile = -1;           // >= 0: forces the tile configuration (helm_debug_zgemm_bench)

namespace {

template <int WM, int WN, int MT, int NT, int IDX, int KS, int OCC, int XR = 0>
__global__ __launch_bounds__(256, OCC) void k_zgemm3(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                                     const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, GemmRows R) {
    __shared__ cplx tiles[2 * (KS / 4) * (MT * WM + XR + NT * WN) * 64];
    zgemm3_body<WM, WN, MT, NT, IDX, KS, XR>(M, Nn, K, alpha, A0, lda, sa, B0, ldb, sb, beta, C0, ldc, sc, R, tiles);
}

// (Round 4, measured and removed: the same kernel with LDS-DMA staging -- global_load_lds_dwordx4 writing each 1-KB fragment straight into a
// three- or four-stage LDS ring, counted vmcnt, one raw barrier per slab, every LDS read of the loop in inline asm because hipcc makes any
// ds_read it can see wait vmcnt(0) while a DMA is in flight.  Correct on the product shapes, and 3-8 % SLOWER than the register-staged
// kernel above on every shape of tools/zgemm_lab.py (leaf back substitution 2989 against 2839 us, 1024 x 1024 x 256 x 16: 619 against 597 us,
// under-filled 1025 x 256 x 512 x 4: 164 against 163 us): at 3-4 workgroups per CU the other workgroups already cover a slab's load latency, and
// the DMA's per-fragment address arithmetic costs what the staging registers did.  profiles/r04_zgemm_lab_mfma.txt keeps the table.)
// the 16 MT + 1-row tile (XR) for dense and row-table operands
template <int MT, int NT, int KS, int OCC = 2>
void launch_mfma_xr(hipStream_t st, int idx, int nb, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                    cplx beta, cplx *C, int ldc, long long sc, const GemmRows &R) {
    constexpr int TM = 16 * MT + 1, TN = 16 * NT * 4;
    dim3 grid((Nn + TN - 1) / TN, (M + TM - 1) / TM, nb);
    if (idx) ZG_LAUNCH((k_zgemm3<1, 4, MT, NT, 1, KS, OCC, 1>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else ZG_LAUNCH((k_zgemm3<1, 4, MT, NT, 0, KS, OCC, 1>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
}

template <int WM, int WN, int MT, int NT, int KS, int OCC = 2>
void launch_mfma(hipStream_t st, int idx, int nb, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                 cplx beta, cplx *C, int ldc, long long sc, const GemmRows &R) {
    constexpr int TM = 16 * MT * WM, TN = 16 * NT * WN;
    dim3 grid((Nn + TN - 1) / TN, (M + TM - 1) / TM, nb);
    if (idx == 4) ZG_LAUNCH((k_zgemm3<WM, WN, MT, NT, 4, KS, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else if (idx == 2) ZG_LAUNCH((k_zgemm3<WM, WN, MT, NT, 2, KS, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else if (idx) ZG_LAUNCH((k_zgemm3<WM, WN, MT, NT, 1, KS, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
    else ZG_LAUNCH((k_zgemm3<WM, WN, MT, NT, 0, KS, OCC>), grid, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, R);
}

// Leaf back substitution where a leaf has no right-hand side in a block of 64 columns (round 5) -- nearly every (leaf, block) pair of a survey's
// point sources.  There x_S = G x_B: 32 ring rows against the last 33 columns of [F11^-1 | G].  Through the slab pipeline of the tile kernel that is
// five slabs, i.e. five exposed load latencies in a row (row-table look-up, then one per slab; ~28 us per workgroup for 106 KB, the launch at 45 % of its
// traffic).  Here a workgroup issues EVERYTHING it will read back to back: the nine row-table entries a lane needs, then its nine B fragments straight into
// the registers the matrix instructions take them from (lane l of wave w: row 4 kg + l / 16, column 16 w + l % 16 -- a fragment IS a row-major 4 x 16
// piece of B, no LDS), and the 49 x 36 piece of A through registers into a fragment-ordered LDS image (one barrier).  Same k groups of four (aligned at
// 48 = 8 (k2 / 8), the y_S row 48 taken as the zero it is), same order of the matrix instructions per accumulator, same order of the vector-ALU sums of
// row 49 as zgemm3_body<1, 4, 3, 1, 1, 8, 1>: the results are bit for bit those of the tile kernel started at row 48, which are bit for bit those of
// the full product (tests/test_gpu_direct.py, sparse right-hand sides).  The tile kernel takes the blocks whose flag is up (GemmRows::idle_done).
constexpr int LBI_KG = 9;                                            // k groups of four from kbeg on: covers K - kbeg <= 36
__global__ __launch_bounds__(256, 3) void k_leaf_bwd_idle(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa, cplx beta, GemmRows R) {
    constexpr int FA = 4, TMM = 48, TN = 64;
    __shared__ cplx As[LBI_KG * FA * 64];
    int bxi = blockIdx.x, bzi = blockIdx.z;
    if (gridDim.x > 1 && gridDim.z >= 16 && R.xcd_map) {              // the four column blocks of a leaf on one XCD (see zgemm3_body)
        const int nt = gridDim.x, nbz = gridDim.z;
        const int L = blockIdx.x + nt * blockIdx.z;
        const int full = (nbz / 8) * 8 * nt;
        if (L < full) { const int grp = L / (8 * nt), w = L % (8 * nt); bzi = grp * 8 + (w & 7); bxi = w >> 3; }
        else { bzi = (nbz / 8) * 8 + (L - full) / nt; bxi = (L - full) % nt; }
    }
    const int n0 = bxi * TN;
    if (R.act_ro[(long long)(R.first + R.z0 + bzi) * R.nct + (n0 >> 6)]) return;      // a right-hand side in this block: the tile kernel's
    const cplx *A = A0 + (long long)bzi * sa;
    const int tid = threadIdx.x, wn = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const long long trow = (long long)(R.z0 + bzi) * R.tab_stride;
    const int kbeg = (R.k2 / 8) * 8;
    // row-table entries: the B rows of this lane's k, the C rows of its outputs
    int rk[LBI_KG], ro[3][4], ro48 = -1;
    #pragma unroll
    for (int kg = 0; kg < LBI_KG; ++kg) {
        const int k = kbeg + 4 * kg + lq;
        rk[kg] = (k >= R.k2 && k < K) ? R.tabB[trow + R.offB + k].x : -1;          // (k < k2: a y_S row of a block without right-hand side -- zero)
    }
    #pragma unroll
    for (int i = 0; i < 3; ++i)
        #pragma unroll
        for (int q = 0; q < 4; ++q) { const int r = 16 * i + lq + 4 * q; ro[i][q] = r < M ? R.tabCo[trow + R.offCo + r].x : -1; }
    if (TMM < M) ro48 = R.tabCo[trow + R.offCo + TMM].x;
    // A: rows 0 .. 48, columns kbeg .. kbeg + 35
    constexpr int NA = (49 * 4 * LBI_KG + 255) / 256;
    cplx ra[NA];
    #pragma unroll
    for (int e = 0; e < NA; ++e) {
        const int idx = tid + e * 256;
        const int ar = idx / (4 * LBI_KG), ak = idx % (4 * LBI_KG);
        cplx v = cmake(0.0, 0.0);
        if (ar < 49 && ar < M && kbeg + ak < K) v = A[(long long)ar * lda + kbeg + ak];
        ra[e] = v;
    }
    const int cc = n0 + 16 * wn + lr;
    cplx breg[LBI_KG];
    #pragma unroll
    for (int kg = 0; kg < LBI_KG; ++kg) breg[kg] = (rk[kg] >= 0 && cc < Nn) ? R.Bx[(long long)rk[kg] * R.ldx + cc] : cmake(0.0, 0.0);
    #pragma unroll
    for (int e = 0; e < NA; ++e) {
        const int idx = tid + e * 256;
        const int ar = idx / (4 * LBI_KG), ak = idx % (4 * LBI_KG);
        if (ar < 49) As[((ak >> 2) * FA + (ar >> 4)) * 64 + (ak & 3) * 16 + ((ar & 15) ^ (ak & 3) ^ (((ak >> 2) & 1) << 2))] = ra[e];
    }
    __syncthreads();
    v4f64 cr[3], ci[3];
    #pragma unroll
    for (int i = 0; i < 3; ++i) { cr[i] = (v4f64){0, 0, 0, 0}; ci[i] = (v4f64){0, 0, 0, 0}; }
    cplx xacc = cmake(0.0, 0.0);
    #pragma unroll
    for (int kg = 0; kg < LBI_KG; ++kg) {
        if (kbeg + 4 * kg >= K) break;                                 // (a k group that is all padding)
        cplx a[3];
        const cplx b = breg[kg];
        #pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = As[(kg * FA + i) * 64 + lq * 16 + (lr ^ lq ^ ((kg & 1) << 2))];
        #pragma unroll
        for (int i = 0; i < 3; ++i) {
            cr[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b.x, cr[i], 0, 0, 0);
            ci[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b.y, ci[i], 0, 0, 0);
        }
        #pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double nai = -a[i].y;
            cr[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, b.y, cr[i], 0, 0, 0);
            ci[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b.x, ci[i], 0, 0, 0);
        }
        // row 49 on the vector ALUs: this lane's share of the k range is k = 4 kg + lq (the tile kernel's thread (column, share lq) takes the same k in the same order)
        const cplx av = As[(kg * FA + FA - 1) * 64 + lq * 16 + (lq ^ ((kg & 1) << 2))];
        cfma(xacc, av, b);
    }
    const cplx zero = cmake(0.0, 0.0);
    #pragma unroll
    for (int i = 0; i < 3; ++i)
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (ro[i][q] < 0 || cc >= Nn) continue;
            cplx v = cmul(alpha, cmake(cr[i][q], ci[i][q]));
            v = cadd(v, cmul(beta, zero));
            if (R.cj_out) v = conj_scaled(R.oscale, v);
            cplx *dst = R.Cox + (long long)ro[i][q] * R.ldx + cc;
            if (R.ntc) __builtin_nontemporal_store((v2f64){v.x, v.y}, reinterpret_cast<v2f64 *>(dst)); else *dst = v;
        }
    // row 49: the four shares of a column sit in lanes lr, lr + 16, lr + 32, lr + 48 of this wave; added up in the tile kernel's order
    {
        cplx s = cmake(__shfl(xacc.x, lr), __shfl(xacc.y, lr));
        #pragma unroll
        for (int h = 1; h < 4; ++h) s = cadd(s, cmake(__shfl(xacc.x, lr + 16 * h), __shfl(xacc.y, lr + 16 * h)));
        if (lq == 0 && TMM < M && ro48 >= 0 && cc < Nn) {
            cplx v = cmul(alpha, s);
            if (R.cj_out) v = conj_scaled(R.oscale, v);
            R.Cox[(long long)ro48 * R.ldx + cc] = v;
        }
    }
}

// The same idea for the back substitution of the small separator fronts (at most 16 unknowns, rings of up to 96):  x_S = F11^-1 (y_S - F12 x_B)  in ONE
// launch instead of two, with every load of a workgroup in flight at once.  A lane's KGN fragments of x_B go straight into registers through the row
// table; F12 (16 x m) and F11^-1 (16 x 16) go through one LDS image each.  The first product leaves t = y_S - F12 x_B in the layout the matrix cores write
// -- rows l / 16 + 4 q of column l % 16 -- and that IS the fragment layout of a B operand with k group q, so the second product takes t from the registers
// it is in.  Same k groups, same order of instructions and of the alpha / beta arithmetic as the two launches of the tile kernel it replaces
// (t = -1 acc + 1 y_S; x = 1 acc' + 0): bit for bit their result.  Reads all its rows before it stores any (y_S -> x_S in place in Xt).
// (A first version kept G = -F11^-1 F12 where F12 was, the leaves' one-product form: one more product per level at factor time, 0.1-0.15 ms each on the
// critical path of the pipelined job, for 0.06-0.1 ms per pass.)
template <int KGN>
__global__ __launch_bounds__(256, 3) void k_sep_bwd_small(int M, int Nn, int Km, const cplx *Finv0, const cplx *F120, int lda, long long sa, GemmRows R) {
    constexpr int TN = 64;
    __shared__ cplx As[KGN * 64];
    __shared__ cplx Fs[4 * 64];
    int bxi = blockIdx.x, bzi = blockIdx.z;
    if (gridDim.x > 1 && gridDim.z >= 16 && R.xcd_map) {              // the column blocks of a front on one XCD (see zgemm3_body)
        const int nt = gridDim.x, nbz = gridDim.z;
        const int L = blockIdx.x + nt * blockIdx.z;
        const int full = (nbz / 8) * 8 * nt;
        if (L < full) { const int grp = L / (8 * nt), w = L % (8 * nt); bzi = grp * 8 + (w & 7); bxi = w >> 3; }
        else { bzi = (nbz / 8) * 8 + (L - full) / nt; bxi = (L - full) % nt; }
    }
    const int n0 = bxi * TN;
    const cplx *Finv = Finv0 + (long long)bzi * sa, *F12 = F120 + (long long)bzi * sa;
    const int tid = threadIdx.x, wn = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const long long trow = (long long)(R.z0 + bzi) * R.tab_stride;
    int rk[KGN], ro[4];
    #pragma unroll
    for (int kg = 0; kg < KGN; ++kg) { const int k = 4 * kg + lq; rk[kg] = k < Km ? R.tabB[trow + R.offB + k].x : -1; }      // ring rows (offB = the front's separator size)
    #pragma unroll
    for (int q = 0; q < 4; ++q) { const int r = lq + 4 * q; ro[q] = r < M ? R.tabCo[trow + R.offCo + r].x : -1; }
    constexpr int NA = (16 * 4 * KGN + 255) / 256;
    cplx ra[NA];
    #pragma unroll
    for (int e = 0; e < NA; ++e) {
        const int idx = tid + e * 256;
        const int ar = idx / (4 * KGN), ak = idx % (4 * KGN);
        cplx v = cmake(0.0, 0.0);
        if (ar < 16 && ar < M && ak < Km) v = F12[(long long)ar * lda + ak];
        ra[e] = v;
    }
    cplx rf = cmake(0.0, 0.0);
    { const int ar = tid >> 4, ak = tid & 15; if (ar < M && ak < M) rf = Finv[(long long)ar * lda + ak]; }
    const int cc = n0 + 16 * wn + lr;
    cplx breg[KGN], ys[4];
    #pragma unroll
    for (int kg = 0; kg < KGN; ++kg) breg[kg] = (rk[kg] >= 0 && cc < Nn) ? R.Bx[(long long)rk[kg] * R.ldx + cc] : cmake(0.0, 0.0);
    #pragma unroll
    for (int q = 0; q < 4; ++q) ys[q] = (ro[q] >= 0 && cc < Nn) ? R.Cix[(long long)ro[q] * R.ldx + cc] : cmake(0.0, 0.0);
    #pragma unroll
    for (int e = 0; e < NA; ++e) {
        const int idx = tid + e * 256;
        const int ar = idx / (4 * KGN), ak = idx % (4 * KGN);
        if (ar < 16) As[(ak >> 2) * 64 + (ak & 3) * 16 + ar] = ra[e];
    }
    { const int ar = tid >> 4, ak = tid & 15; Fs[(ak >> 2) * 64 + (ak & 3) * 16 + ar] = rf; }
    __syncthreads();
    v4f64 cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
    #pragma unroll
    for (int kg = 0; kg < KGN; ++kg) {
        const cplx a = As[kg * 64 + lq * 16 + lr], b = breg[kg];          // (k groups past the ring hold zeros)
        cr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.y, ci, 0, 0, 0);
        cr = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, b.y, cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.x, ci, 0, 0, 0);
    }
    // t = y_S - F12 x_B as the tile kernel forms it (alpha = -1, beta = 1)
    const cplx one = cmake(1.0, 0.0), mone = cmake(-1.0, 0.0), zero = cmake(0.0, 0.0);
    cplx t[4];
    #pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = cadd(cmul(mone, cmake(cr[q], ci[q])), cmul(one, ys[q]));
    // x_S = F11^-1 t: row l / 16 + 4 q of t is row (k group q, k = l / 16) of a B operand
    v4f64 xr = {0, 0, 0, 0}, xi = {0, 0, 0, 0};
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (4 * q >= M) break;
        const cplx a = Fs[q * 64 + lq * 16 + lr], b = t[q];
        xr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, xr, 0, 0, 0);
        xi = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.y, xi, 0, 0, 0);
        xr = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, b.y, xr, 0, 0, 0);
        xi = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.x, xi, 0, 0, 0);
    }
    #pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (ro[q] < 0 || cc >= Nn) continue;
        cplx v = cmul(one, cmake(xr[q], xi[q]));
        v = cadd(v, cmul(zero, zero));
        cplx *dst = R.Cox + (long long)ro[q] * R.ldx + cc;
        if (R.ntc) __builtin_nontemporal_store((v2f64){v.x, v.y}, reinterpret_cast<v2f64 *>(dst)); else *dst = v;
        if (R.Cox2) { const cplx u = conj_scaled(R.oscale, v); __builtin_nontemporal_store((v2f64){u.x, u.y}, reinterpret_cast<v2f64 *>(dst + (R.Cox2 - R.Cox))); }
    }
}

// C = beta C + alpha (sum of the ksplit partial products of a split launch); parts: [chunk][matrix][M x Nn]
__global__ __launch_bounds__(256) void k_splitk_reduce(const cplx *__restrict__ parts, int ksplit, long long pstride, int M, int Nn, cplx alpha, cplx beta,
                                                       cplx *__restrict__ C, int ldc, long long sc, long long total) {
    const bool rd = !(beta.x == 0.0 && beta.y == 0.0);
    const long long per = (long long)M * Nn;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        cplx sum = parts[e];
        for (int k = 1; k < ksplit; ++k) sum = cadd(sum, parts[(long long)k * pstride + e]);
        const long long b = e / per, rem = e - b * per;
        cplx *dst = C + b * sc + (rem / Nn) * ldc + rem % Nn;
        cplx o = cmul(alpha, sum);
        if (rd) o = cadd(o, cmul(beta, *dst));
        *dst = o;
    }
}

// Tile choice.  Nine shapes, WM x WN waves of MT x NT blocks of 16 x 16 (rows x columns of C per workgroup):
//   0: 64 x 64   1: 32 x 128   2: 16 x 256   3: 64 x 32   4: 32 x 64   5: 16 x 128   6: 32 x 32   7: 16 x 64   8: 128 x 16
// chosen by padded area, the narrow tiles weighed down by how much less they re-use an operand fragment (x 1 / 0.7, x 1 / 0.45), then
// adjusted by what was measured on the shapes the 1024^2 plan issues (tools/tile_lab.py, tools/zgemm_lab.py; HISTORY.md has the numbers):
int choose_tile(int M, int Nn, int K, int batch, const GemmRows *rows, bool *latency_mode) {
    static const int vc_tm[9] = {64, 32, 16, 64, 32, 16, 32, 16, 128}, vc_rn[9] = {4, 4, 4, 2, 2, 2, 1, 1, 2};
    static const double vc_eff[9] = {1.0, 1.0, 1.0, 0.7, 0.7, 0.7, 0.45, 0.45, 0.7};
    int vsel = 0; double vcost = -1;
    for (int c = 0; c < 9; ++c) {
        if (c == 8 && Nn > 16) continue;                  // (128 x 16: products with at most 16 columns, the passes of the 3-D coarse level)
        const int tm = vc_tm[c], tn = 1024 / tm * vc_rn[c];
        double cost = (double)((M + tm - 1) / tm) * tm * ((Nn + tn - 1) / tn) * tn / vc_eff[c];
        // a 16-row tile is one block row per wave column -- right for fronts of 8 and 16 rows, 40-50 % slower than the 64- and 32-row tiles on
        // tall operands, where only the padding of an odd row count (1025, 1281) made it look cheap
        if (tm == 16 && M > 32) cost *= 1.6;
        if (vcost < 0 || cost < vcost * 0.999) { vsel = c; vcost = cost; }
    }
    const long long tiles64 = (long long)batch * ((M + 63) / 64) * ((Nn + 63) / 64);
    // under-filled launches (the few big fronts at the top of the tree: fewer 64 x 64 tiles than compute units): 32 x 32 or 16 x 64 tiles
    // quadruple the number of workgroups and take a K slab of 16 -- with one wave per SIMD nothing else hides the LDS and HBM latencies
    *latency_mode = false;
    if (tiles64 < 256) {
        const long long a6 = (long long)((M + 31) / 32) * 32 * ((Nn + 31) / 32) * 32, a7 = (long long)((M + 15) / 16) * 16 * ((Nn + 63) / 64) * 64;
        vsel = a7 < a6 ? 7 : 6;
        *latency_mode = true;
    }
    // fronts of 8 and 16 rows: the 16 x 64 tile (four workgroups per front) is 4-15 % ahead of 16 x 256
    if (!*latency_mode && M <= 16 && vsel == 2) vsel = 7;
    // a few hundred 64 x 64 tiles (one or two per compute unit, gone in a single round): 32 x 32 tiles give every unit four to eight
    // workgroups to overlap (1025 x 256 x 512 x 4: 152 -> 106 us)
    if (!*latency_mode && M > 64 && tiles64 < 600) vsel = 6;
    // forward-gather launches are HBM-bound and every row-tile repeats the three-source gather of the B rows: one row-tile per front
    // wherever the front has at most 64 rows, whatever the padding costs in flops
    if (rows && rows->fwd3 && M <= 64 && !*latency_mode) vsel = 0;
    // rank-32 updates of one large matrix (blocked Gauss-Jordan of the 3-D plane inverses): HBM-bound, the 64 x 32 tile is the fastest
    if (rows && rows->dense && K <= 32 && !*latency_mode && batch == 1) vsel = 3;
    // (round 5 measured a 128 x 64 tile -- WM x WN = 2 x 2 waves of 4 x 2 blocks, 216-224 VGPRs -- on the large Schur / G21 products: 56.8 against 57.5 TFLOP/s on
    // 1024 x 1024 x 256 x 16, 46 against 52 on the 1281-row fronts, headline -1.6 %: the 64 x 64 tile is not bound by its operand traffic; HISTORY.md / profiles/r05_zgemm_lab.txt)
    if (g_gemm_tile >= 0) { vsel = g_gemm_tile & 15; *latency_mode = false; }
    // the fused update + sweep launch exists for two tiles: 64 x 32 (large matrices) and the 32 x 32 latency tile (under-filled launches)
    if (rows && rows->la) vsel = *latency_mode ? 6 : 3;
    // (one row tile per matrix, so that C may overwrite B; fronts of at most 16 rows -- the small separator fronts of the one-product back substitution -- take
    // the 16 x 64 tile: the 64-row tile cost those levels 0.15 ms each at factor time for a product of a few MFLOP)
    if (rows && rows->tm64 && M <= 64) { vsel = M <= 16 ? 7 : (Nn <= 32 ? 3 : 0); *latency_mode = false; }
    return vsel;
}

}  // namespace

int gemm(helm_op *op, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
         cplx beta, cplx *C, int ldc, long long sc, int batch, const GemmRows *rows) {
    if (M <= 0 || Nn <= 0 || batch <= 0) return 0;
    hipStream_t st = op ? op->stream : nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // profiling: every launch carries its own start / stop events (ext launch, see ZG_LAUNCH); helm_tuning.prof_ext = 0: hipEventRecord markers
    // around launches / runs of launches
    const bool ext = op && op->profiling && helm_tuning_now().prof_ext != 0;
    const bool in_run = !ext && op && op->gemm_run_depth > 0;
    if (!ext && op && op->profiling && !(in_run && op->gemm_run_pair >= 0)) {
        if (op->ev_used + 2 > op->ev_pool.size() && op->ev_pool.size() < 16384)
            (void)helm_events_grow(op, 64);
        if (op->ev_used + 2 <= op->ev_pool.size()) { e0 = op->ev_pool[op->ev_used]; e1 = op->ev_pool[op->ev_used + 1]; }
    }
    if (e0) hipEventRecord(e0, st);
    if (in_run && e0) { op->gemm_run_pair = (int)op->ev_used; op->ev_used += 2; }
    // Products with at most 16 columns over a few big matrices (the top of the 3-D coarse level's passes: 16 right-hand sides through fronts of
    // thousands of unknowns) fill a few dozen 128 x 16 tiles: the inner dimension is split over up to 16 workgroups per tile, each writes its
    // partial product to the handle's scratch and a small launch adds them up (with alpha / beta applied there).
    // (measured on the 47 x 79 x 79 level: fewer than 96 tiles / 192 workgroups -> 5.8 ms per coarse solve, 300 / 768 -> 5.0, more changes nothing)
    bool split_done = false;
    const int dbatch = std::max(1, batch / tl_nf_div);        // the batch the choices below are made for (one frequency's share of a multi-frequency launch)
    if (op && !rows && !ext && Nn <= 16 && K >= 1024 && dbatch <= 64) {
        const long long tiles = (long long)dbatch * ((M + 127) / 128);
        if (tiles < 300) {
            const int ks = (int)std::min<long long>(16, std::max<long long>(2, 768 / tiles));
            const int kc = (((K + ks - 1) / ks) + 7) & ~7;
            const long long per = (long long)batch * M * Nn;
            const size_t need = (size_t)ks * per * sizeof(cplx);
            if (op->sk_bytes < need) {
                if (op->sk_buf) { hipStreamSynchronize(st); helm_pool_free(op->device, op->sk_buf, op->sk_bytes); op->sk_buf = nullptr; op->sk_bytes = 0; }
                op->sk_buf = (cplx *)helm_pool_alloc(op->device, need);
                op->sk_bytes = op->sk_buf ? need : 0;
            }
            if (op->sk_buf) {
                GemmRows R; R.dense = 1; R.ksplit = ks; R.kc = kc; R.pstride = per;
                launch_mfma<4, 1, 2, 1, 8>(st, 0, batch * ks, M, Nn, K, cmake(1, 0), A, lda, sa, B, ldb, sb, cmake(0, 0), op->sk_buf, Nn, (long long)M * Nn, R);
                HELM_LAUNCH(k_splitk_reduce, dim3((unsigned)std::min<long long>((per + 255) / 256, 4096)), dim3(256), 0, st, (const cplx *)op->sk_buf, ks, per, M, Nn, alpha, beta,
                                   C, ldc, sc, per);
                split_done = true;
            }
        }
    }
    bool latency_mode = false;
    const int vsel = choose_tile(M, Nn, K, dbatch, rows, &latency_mode);
    const int idxmode = rows && rows->schur4 ? 4 : (rows && !rows->dense ? (rows->fwd3 ? 2 : 1) : 0);
    const int xcd_map = helm_tuning_now().nd_xcd_map;
    for (int b0 = 0; b0 < batch && !split_done; b0 += 65535) {
        const int nb = std::min(65535, batch - b0);
        GemmRows R; if (rows) R = *rows;
        R.z0 = b0;
        R.xcd_map = xcd_map;
        // gathered products over many fronts (the row-table and forward-gather levels of both passes) store C with nontemporal stores: the rows are not read
        // again before a whole level has gone by (headline +1.5 %; the Schur complements, read back one level later: no difference either way)
        if (rows && !rows->dense && !rows->schur4 && nb >= 64) R.ntc = 1;
        const cplx *Ab = A + b0 * sa, *Bb = B ? B + b0 * sb : B;
        cplx *Cb = C ? C + b0 * sc : C;
        // leaf back substitution on sparse right-hand sides: the (leaf, block of 64 columns) pairs without a right-hand side go to k_leaf_bwd_idle
        if (idxmode == 1 && M == 49 && rows->act_ro && rows->k2 > 0 && K - (rows->k2 / 8) * 8 <= 4 * LBI_KG && Nn % 64 == 0 && rows->tabB && rows->tabCo && !rows->tabCi &&
            !rows->Cox2 && beta.x == 0.0 && beta.y == 0.0 && g_gemm_tile < 0 && R.zr1 == 0 && R.zc1 == 0 && R.sk1 == 0 && helm_tuning_now().nd_leaf_idle != 0) {
            R.idle_done = 1;
            ExtArm arm2(op, ext, 0.0, 0.0, M, Nn, K, nb, 6);          // (booked with its time and no flops: the tile launch below carries the product's)
            ZG_LAUNCH(k_leaf_bwd_idle, dim3(Nn / 64, 1, nb), M, Nn, K, alpha, Ab, lda, sa, beta, R);
        }
        ExtArm arm(op, ext, 8.0 * M * (double)Nn * K * nb, gemm_operand_bytes(M, Nn, K, beta, rows) * nb, M, Nn, K, nb, idxmode ? idxmode : (rows && rows->la ? 5 : 0));
#define ZG_ARGS st, idxmode, nb, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R
#define ZG_MFMA(WM_, WN_, MT_, NT_, KS_) launch_mfma<WM_, WN_, MT_, NT_, KS_>(ZG_ARGS)
        if (rows && rows->la) {           // update + pivot sweep of the next block in one launch
            launch_zgemm3_la(st, latency_mode, nb, M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R, *rows->la);
            continue;
        }
        // fronts of 16 a + 1 rows (the 49-unknown leaves): a rows of blocks on the matrix cores, the last row on the vector ALUs.
        // (49 x 64 tile, four workgroups per compute unit, against 49 x 128 with two: leaf back substitution 3.23 -> 2.90 ms with every front
        // computed, 1.89 -> 1.70 on point sources; a K slab of 16 halves the occupancy again: 3.7 / 4.7 ms)
        const bool plain = !rows || (!rows->schur4 && !rows->fwd3 && !rows->ksplit && rows->zr1 == 0 && rows->zc1 == 0 && rows->sk1 == 0);
        if (M == 49 && Nn >= 64 && plain && idxmode <= 1 && g_gemm_tile < 0) { launch_mfma_xr<3, 1, 8>(ZG_ARGS); continue; }
        if (idxmode == 2 && rows->list) {             // forward gather dealt from the level's list of active (front, block) pairs: 64 x 64 tiles, one x per pair
            ZG_LAUNCH((k_zgemm3<2, 2, 2, 2, 2, 8, 2>), dim3((unsigned)nb * (unsigned)R.nct, (M + 63) / 64, 1), M, Nn, K, alpha, Ab, lda, sa, Bb, ldb, sb, beta, Cb, ldc, sc, R);
            continue;
        }
        if (latency_mode) { if (vsel == 6) ZG_MFMA(2, 2, 1, 1, 16); else ZG_MFMA(1, 4, 1, 1, 16); continue; }
        switch (vsel) {
            case 0: ZG_MFMA(2, 2, 2, 2, 8); break;
            case 1: ZG_MFMA(1, 4, 2, 2, 8); break;
            case 2: ZG_MFMA(1, 4, 1, 4, 8); break;
            case 3: ZG_MFMA(2, 2, 2, 1, 8); break;
            case 4: ZG_MFMA(2, 2, 1, 2, 8); break;
            case 5: ZG_MFMA(1, 4, 1, 2, 8); break;
            case 6: ZG_MFMA(2, 2, 1, 1, 8); break;
            case 8: ZG_MFMA(4, 1, 2, 1, 8); break;
            case 9: ZG_MFMA(2, 2, 4, 2, 8); break;        // 128 x 64 (tools/zgemm_lab.py only)
            default: ZG_MFMA(1, 4, 1, 1, 8); break;
        }
#undef ZG_MFMA
#undef ZG_ARGS
    }
    const double flops = 8.0 * M * (double)Nn * K * batch;
    const double obytes = gemm_operand_bytes(M, Nn, K, beta, rows) * batch;
    if (ext) return 0;
    if (in_run) {                       // the run's end event is recorded by GemmRun's destructor
        if (op->gemm_run_pair >= 0) { op->gemm_run_flops += flops; op->gemm_run_bytes += obytes; op->gemm_run_sol += gemm_sol_ms(flops, obytes); op->gemm_run_launches += 1; }
    } else if (e0) {
        hipEventRecord(e1, st);
        op->ev_pending_gemm.push_back(std::make_pair((int)op->ev_used, flops));
        op->ev_pending_gemm_n.push_back(1);
        op->ev_pending_gemm_bytes.push_back(obytes);
        op->ev_pending_gemm_sol.push_back(gemm_sol_ms(flops, obytes));
        op->ev_used += 2;
    }
    return 0;
}

// Back substitution of a group of small separator fronts in one launch (k_sep_bwd_small); false: not a case for it, the caller issues the two products.
// rows: tabB / offB = the ring rows, tabCo / offCo = the separator rows (Cix: where y_S is read, Cox: where x_S goes, Cox2: its transformed copy).
bool gemm_sep_bwd_small(helm_op *op, int smax, int mmax, int nrhs, const cplx *Finv, const cplx *F12, int lda, long long sa, int batch, const GemmRows &rows) {
    if (!op || smax > 16 || mmax > 96 || batch < 256 || batch > 65535 || nrhs % 64 != 0 || !rows.tabB || !rows.tabCo || !rows.Cix || !rows.Cox || g_gemm_tile >= 0 ||
        helm_tuning_now().nd_leaf_idle == 0) return false;
    hipStream_t st = op->stream;
    const bool ext = op->profiling && helm_tuning_now().prof_ext != 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!ext && op->profiling) {
        if (op->ev_used + 2 > op->ev_pool.size() && op->ev_pool.size() < 16384) (void)helm_events_grow(op, 64);
        if (op->ev_used + 2 <= op->ev_pool.size()) { e0 = op->ev_pool[op->ev_used]; e1 = op->ev_pool[op->ev_used + 1]; }
    }
    if (e0) hipEventRecord(e0, st);
    GemmRows R = rows;
    R.z0 = 0; R.xcd_map = helm_tuning_now().nd_xcd_map; R.ntc = 1;
    const double flops = 8.0 * smax * (double)nrhs * (mmax + smax) * batch;
    const double obytes = (gemm_operand_bytes(smax, nrhs, mmax, cmake(1.0, 0.0), &rows) + 16.0 * smax * smax) * batch;
    {
        ExtArm arm(op, ext, flops, obytes, smax, nrhs, mmax + smax, batch, 1);
        const dim3 grid(nrhs / 64, 1, batch);
        if (mmax <= 48) ZG_LAUNCH(k_sep_bwd_small<12>, grid, smax, nrhs, mmax, Finv, F12, lda, sa, R);
        else if (mmax <= 64) ZG_LAUNCH(k_sep_bwd_small<16>, grid, smax, nrhs, mmax, Finv, F12, lda, sa, R);
        else ZG_LAUNCH(k_sep_bwd_small<24>, grid, smax, nrhs, mmax, Finv, F12, lda, sa, R);
    }
    if (e0) {
        hipEventRecord(e1, st);
        op->ev_pending_gemm.push_back(std::make_pair((int)op->ev_used, flops));
        op->ev_pending_gemm_n.push_back(1);
        op->ev_pending_gemm_bytes.push_back(obytes);
        op->ev_pending_gemm_sol.push_back(gemm_sol_ms(flops, obytes));
        op->ev_used += 2;
    }
    return true;
}

// dense helpers for other translation units (3-D multigrid: coarsest-level inverse and its application)
int nd_dense_gemm(helm_op *op, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, const cplx *B, int ldb, cplx beta, cplx *C, int ldc) {
    gemm(op, M, Nn, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1);
    return check_kernels(op, "dense GEMM");
}
int nd_dense_gemm_batched(helm_op *op, int M, int Nn, int K, cplx alpha, const cplx *A, int lda, long long sa, const cplx *B, int ldb, long long sb,
                          cplx beta, cplx *C, int ldc, long long sc, int batch) {
    gemm(op, M, Nn, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, batch);
    return check_kernels(op, "dense GEMM");
}
extern "C" int helm_debug_zgemm(int device, int M, int Nn, int K, const double *alpha, const double *A, const double *B, const double *beta, double *C, int batch) {
    helm_tuning_refresh();
    if (hipSetDevice(device) != hipSuccess) return HELM_ERR_DEVICE;
    cplx *dA, *dB, *dC;
    const size_t na = (size_t)batch * M * K, nb = (size_t)batch * K * Nn, nc = (size_t)batch * M * Nn;
    if (hipMalloc((void **)&dA, na * 16) != hipSuccess || hipMalloc((void **)&dB, nb * 16) != hipSuccess || hipMalloc((void **)&dC, nc * 16) != hipSuccess) return HELM_ERR_DEVICE;
    hipMemcpy(dA, A, na * 16, hipMemcpyHostToDevice); hipMemcpy(dB, B, nb * 16, hipMemcpyHostToDevice); hipMemcpy(dC, C, nc * 16, hipMemcpyHostToDevice);
    // (with a handle, like the solver's own calls: the paths that keep scratch on it -- the split over the inner dimension -- are taken too)
    const int fs[4] = {0, 0, 0, 0};
    helm_op *tmp = helm_create(device, 0, 8, 8, 1.0, 1.0, 2, fs);
    gemm(tmp, M, Nn, K, cmake(alpha[0], alpha[1]), dA, K, (long long)M * K, dB, Nn, (long long)K * Nn, cmake(beta[0], beta[1]), dC, Nn, (long long)M * Nn, batch);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(C, dC, nc * 16, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dC);
    if (tmp) helm_destroy(tmp);
    return e == hipSuccess ? HELM_OK : HELM_ERR_DEVICE;
}

// times `reps` launches of one strided-batched GEMM shape on random operands; variant < 16: the tile gemm() would choose, 16 (t + 1) + anything: tile
// configuration t forced (choose_tile); returns the average milliseconds per launch in *ms
extern "C" int helm_debug_zgemm_bench(int device, int M, int Nn, int K, int batch, int variant, int reps, double *ms_out) {
    helm_tuning_refresh();
    if (hipSetDevice(device) != hipSuccess) return HELM_ERR_DEVICE;
    cplx *dA, *dB, *dC;
    const size_t na = (size_t)batch * M * K, nb = (size_t)batch * K * Nn, nc = (size_t)batch * M * Nn;
    if (hipMalloc((void **)&dA, na * 16) != hipSuccess || hipMalloc((void **)&dB, nb * 16) != hipSuccess || hipMalloc((void **)&dC, nc * 16) != hipSuccess) return HELM_ERR_DEVICE;
    std::vector<cplx> h(std::max(na, nb));
    unsigned long long st = 88172645463325252ULL;
    for (size_t i = 0; i < h.size(); ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = cmake((double)(st & 0xffff) / 65536.0 - 0.5, (double)((st >> 16) & 0xffff) / 65536.0 - 0.5); }
    hipMemcpy(dA, h.data(), na * 16, hipMemcpyHostToDevice); hipMemcpy(dB, h.data(), nb * 16, hipMemcpyHostToDevice);
    hipMemset(dC, 0, nc * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    g_gemm_tile = variant >= 16 ? (variant >> 4) - 1 : -1;
    for (int w = 0; w < 2; ++w) gemm((helm_op *)nullptr, M, Nn, K, cmake(1, 0), dA, K, (long long)M * K, dB, Nn, (long long)K * Nn, cmake(0, 0), dC, Nn, (long long)M * Nn, batch);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) gemm((helm_op *)nullptr, M, Nn, K, cmake(1, 0), dA, K, (long long)M * K, dB, Nn, (long long)K * Nn, cmake(0, 0), dC, Nn, (long long)M * Nn, batch);
    hipEventRecord(e1, nullptr);
    hipError_t e = hipEventSynchronize(e1);
    g_gemm_tile = -1;
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = ms / std::max(1, reps);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(dA); hipFree(dB); hipFree(dC);
    return e == hipSuccess ? HELM_OK : HELM_ERR_DEVICE;
}

