// The tile kernel of the direct solver's dense products: strided-batched complex C = beta C + alpha A B on the matrix cores
// (v_mfma_f64_16x16x4_f64).  A device-side template: nd_gemm.hip instantiates it as k_zgemm3, nd_gj.hip inside the fused
// update + pivot-sweep launch (k_zgemm3_la).
#pragma once
#include "nd_internal.hpp"

// ---- third-generation tile kernel: the same products on the matrix cores (v_mfma_f64_16x16x4_f64) -----------------------
// Round 4.  tools/fp64_clock.hip with the occupancy pinned (profiles/r04_fp64_clock_probe.txt) settled what the fp64 units give: the
// vector FMAs of the 4 x 4 complex register block saturate at 55-56 TFLOP/s at 2, 4 and 8 waves per SIMD (the chip holds 2.04-2.09 GHz
// under that load), whereas v_mfma_f64_16x16x4_f64 runs at 77.8 TFLOP/s = 99 % of nominal from two waves per SIMD up at 2.38 GHz, 73.4
// in the complex product's own mix of 16 MFMAs on 8 rotating operand registers.  (Rounds 2 and 3 measured 35-47 for the MFMA and
// concluded it could not win: that was one wave per SIMD under __launch_bounds__(256), where hipcc parks the accumulators in AGPRs
// and copies them around every instruction.)  One MFMA replaces 16 vector FMAs per lane-pair of operands, so the inner loop issues
// (MT + NT) 16-byte LDS reads for 4 MT NT matrix instructions of 64 cycles each: LDS and VALU issue are out of the picture.
//   tile      256 threads = WM x WN waves; a wave owns MT x NT blocks of 16 x 16 outputs, real and imaginary accumulators apart
//             (C = (Ar Br - Ai Bi) + i (Ar Bi + Ai Br): four real MFMAs per block and k step of 4, -Ai formed once per fragment)
//   LDS       fragment-ordered: every (k group of 4, 16-row block of A | 16-column block of B) is one 1-KB run that a wave reads with
//             lane l at offset 16 l -- conflict-free by construction -- A[row l & 15][k l >> 4], B[k l >> 4][col l & 15].  The B
//             fragment IS a row-major 4 x 16 piece of B.  A arrives k-contiguous (8 lanes = 8 k of one row): its slot inside the
//             fragment is XOR-swizzled with (k & 3) ^ 4 (k / 4 & 1) so that those 8 lanes hit 8 different 16-byte bank groups on the
//             store and the 16-lane groups of ds_read_b128 still cover all 64 banks on the load.
//   C / D     lane l holds rows (l >> 4) + 4 q, q = 0..3, of column l & 15: a store is four 256-byte row segments.
// Addressing modes (IDX), masks, split-K and the fused gathers are those of zgemm2_body, operand for operand.
static __device__ cplx g_zero_page[4];       // 64 bytes of zeros: what masked lanes load instead of branching around a load
// XR = 1 (WM == 1 only): the tile has ONE more row than its 16 MT rows of matrix-core blocks -- row 16 MT goes through the vector ALUs, which the
// MFMA loop leaves idle (thread = column x share of the k range, partial sums added up through LDS at the end).  98 % of the leaves of a 2^k grid
// have 49 = 3 x 16 + 1 unknowns: a fourth block of 16 rows for the 49th would spend a quarter of the leaf level's matrix instructions on padding.
template <int WM, int WN, int MT, int NT, int IDX, int KS, int XR = 0>
__device__ __forceinline__ void zgemm3_body(int M, int Nn, int K, cplx alpha, const cplx *A0, int lda, long long sa,
                                            const cplx *B0, int ldb, long long sb, cplx beta, cplx *C0, int ldc, long long sc, const GemmRows &R,
                                            cplx *lds) {
    static_assert(WM * WN == 4 && KS % 4 == 0, "four waves per workgroup, k groups of 4");
    static_assert(!XR || (WM == 1 && (IDX == 0 || IDX == 1)), "the extra row needs all four waves side by side and plain / row-table addressing");
    constexpr int TMM = 16 * MT * WM;                                   // rows on the matrix cores
    constexpr int TM = TMM + (XR ? 1 : 0), TN = 16 * NT * WN;           // rows / columns of C per workgroup
    constexpr int FA = TMM / 16 + (XR ? 1 : 0), FB = TN / 16, KG = KS / 4;
    constexpr int NA = (FA * 16 * KS + 255) / 256, NB = (TN * KS + 255) / 256;
    constexpr int XP = 256 / TN;                                        // (XR) shares of the k range
    constexpr int ABUF = KG * FA * 64, BBUF = KG * FB * 64;           // elements per buffer
    cplx *As = lds, *Bs = lds + 2 * ABUF;
    __shared__ int kidx[IDX == 1 ? GB_KIDX : 1];
    __shared__ int4 kidx4[IDX == 2 ? GB_KIDX : 1];
    // Which (front, column tile) this workgroup takes.  Workgroups are dealt round-robin over the eight XCDs in launch order (x fastest), so the
    // column tiles of one front -- which all read the same A operand, the front's factors -- would land on different XCDs with different L2s and the
    // factors would cross the fabric once per tile.  With one row tile per front the ids are regrouped in blocks of eight fronts: ids L and L + 8 are
    // the same front's neighbouring column tiles, i.e. the same XCD (speed only: nothing depends on where a workgroup runs).
    int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
    if (gridDim.x * gridDim.y > 1 && gridDim.z >= 16 && R.la == nullptr && !(IDX == 0 && R.ksplit > 1) && R.xcd_map && (gridDim.y == 1 || R.xcd_map > 1)) {
        // (xcd_map > 1: fronts with several row tiles too -- every row tile repeats the gather of the B rows, which then comes out of that XCD's L2)
        const int nxt = gridDim.x, nt = gridDim.x * gridDim.y, nbz = gridDim.z;
        const int L = blockIdx.x + nxt * blockIdx.y + nt * blockIdx.z;
        const int full = (nbz / 8) * 8 * nt;                          // ids covered by whole blocks of eight fronts
        int tile;
        if (L < full) { const int grp = L / (8 * nt), w = L % (8 * nt); bzi = grp * 8 + (w & 7); tile = w >> 3; }
        else { bzi = (nbz / 8) * 8 + (L - full) / nt; tile = (L - full) % nt; }
        bxi = tile % nxt; byi = tile / nxt;
    }
    if (IDX == 2 && R.list) {                            // the launch is dealt from a list of the pairs that have work (see GemmRows::list)
        if ((int)blockIdx.x >= *R.lcount) return;
        const int pair = R.list[blockIdx.x];
        bzi = pair / R.nct; bxi = pair % R.nct; byi = blockIdx.y;
    }
    int zb = bzi;
    if (IDX == 0 && R.ksplit > 1) {                      // this workgroup's share of the inner dimension
        const int kch = zb % R.ksplit;
        zb /= R.ksplit;
        const int kbeg = kch * R.kc;
        A0 += kbeg; B0 += (long long)kbeg * ldb; C0 += (long long)kch * R.pstride;
        K = K - kbeg < R.kc ? (K - kbeg > 0 ? K - kbeg : 0) : R.kc;
    }
    const cplx *A = A0 + (long long)zb * sa;
    const cplx *B = B0 + (long long)zb * sb;
    cplx *C = C0 + (long long)zb * sc;
    const int m0 = byi * TM, n0 = bxi * TN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN, lr = lane & 15, lq = lane >> 4;
    const long long trow = IDX ? (long long)(IDX == 4 ? (R.z0 + bzi) / R.nf : R.z0 + bzi) * R.tab_stride : 0;
    const bool idxB = IDX == 1 && R.tabB != nullptr;
    if (idxB) {
        for (int k = tid; k < K; k += 256) kidx[k] = R.tabB[trow + R.offB + k].x;
        __syncthreads();
    }
    if (IDX == 2) {
        for (int k = tid; k < K; k += 256) kidx4[k] = R.tabB[trow + R.offB + k];
        __syncthreads();
    }
    // (IDX 2, sparse right-hand sides) bit j of am0 / am1: child 0 / 1 has outgoing rows for the j-th block of 64 columns of this tile
    unsigned am0 = ~0u, am1 = ~0u;
    if (IDX == 2 && R.act) {
        const int node = R.first + R.z0 + bzi;
        const NdDev nd = R.nodes[node];
        const int ct0 = n0 >> 6, ncl = ((Nn - n0 < TN ? Nn - n0 : TN) + 63) >> 6;
        am0 = 0; am1 = 0;
        for (int j = 0; j < ncl; ++j) {
            if (nd.kid[0] >= 0 && R.act[nd.kid[0] * R.nct + ct0 + j]) am0 |= 1u << j;
            if (nd.kid[1] >= 0 && R.act[nd.kid[1] * R.nct + ct0 + j]) am1 |= 1u << j;
        }
        int nzq = 0;                                                     // any nonzero among the front's own right-hand-side rows in these columns?
        // (over whole blocks of 64 columns: the workgroups of a tile narrower than that share a flag and must all come to the same verdict)
        constexpr int TS = TN < 64 ? 64 : TN;
        const int ns0 = TN < 64 ? (n0 & ~63) : n0;
        if (!(am0 | am1) && !R.list)
            for (int e = tid; e < K * TS; e += 256) {
                const int k = e / TS, bc = e % TS;
                const int4 t4 = kidx4[k];
                if (t4.w && ns0 + bc < Nn) { const cplx v = R.Bx[(long long)t4.x * R.ldx + ns0 + bc]; nzq |= (v.x != 0.0 || v.y != 0.0); }
            }
        if (!R.list && !(am0 | am1) && !__syncthreads_or(nzq)) {
            if (byi == 0 && R.Cox)                                // y_S = 0 where the back substitution will look for it
                for (int e = tid; e < K * TN; e += 256) {
                    const int k = e / TN, bc = e % TN;
                    const int4 t4 = kidx4[k];
                    if (t4.w && n0 + bc < Nn) R.Cox[(long long)t4.x * R.ldx + n0 + bc] = cmake(0.0, 0.0);
                }
            return;
        }
        if (!R.list && byi == 0 && tid < ncl) R.act[node * R.nct + ct0 + tid] = 1;      // (list: k_fwd_flags has raised it)
    }
    if (IDX == 1 && R.act && R.hint) {                                   // declared support: a leaf none of whose blocks of 64 columns carries a right-hand side is not read
        const int *fl = R.act + (long long)(R.first + R.z0 + bzi) * R.nct + (n0 >> 6);
        const int ncl = ((Nn - n0 < TN ? Nn - n0 : TN) + 63) >> 6;
        int any = 0;
        for (int j = 0; j < ncl; ++j) any |= fl[j];
        if (!any) return;
    }
    if (IDX == 1 && R.act && !R.hint && idxB && TN >= 64 && R.k2 == 0) {
        // No declared support: the leaf has to look at its right-hand-side rows.  Through the slab pipeline below that is one exposed load latency per
        // slab of 8 rows for what is nearly always a block of zeros; here every thread issues its share of the K x TN entries back to back and the
        // workgroup leaves if none of them is nonzero (the flags stay 0, nothing is stored: the state the exit after the pipeline leaves behind).
        int nzq = 0;
        const int tot = K * TN;
        #pragma unroll 7
        for (int e = tid; e < tot; e += 256) {
            const int k = e / TN, bc = e - k * TN;
            const int r = kidx[k];
            if (r >= 0 && n0 + bc < Nn) { const cplx v = R.Bx[(long long)r * R.ldx + n0 + bc]; nzq |= (v.x != 0.0 || v.y != 0.0) ? 1 : 0; }
        }
        if (!__syncthreads_or(nzq)) return;
    }
    int nzb = 0;                                                         // (IDX 1 with act: leaf level) bit j: a nonzero right-hand-side entry in the j-th block of 64 columns
    __shared__ int2 sgr[IDX == 4 ? TM : 1], sgc[IDX == 4 ? TN : 1];
    const cplx *S0 = nullptr, *S1 = nullptr;
    int ld0 = 0, ld1 = 0;
    if (IDX == 4) {
        const int bg = R.z0 + bzi, kf = bg % R.nf;                 // (R.nf > 1: batch index = front * nf + frequency)
        const NdDev nd = R.nodes[R.first + bg / R.nf];
        int base0 = 0, base1 = 0;
        if (nd.kid[0] >= 0) { const NdDev c0 = R.nodes[nd.kid[0]]; ld0 = c0.smax + c0.mmax; S0 = R.arenaS + (long long)R.nf * c0.foff + (long long)kf * c0.mmax * ld0 + c0.smax; base0 = (int)(c0.voff + c0.smax); }
        if (nd.kid[1] >= 0) { const NdDev c1 = R.nodes[nd.kid[1]]; ld1 = c1.smax + c1.mmax; S1 = R.arenaS + (long long)R.nf * c1.foff + (long long)kf * c1.mmax * ld1 + c1.smax; base1 = (int)(c1.voff + c1.smax); }
        for (int t = tid; t < TM + TN; t += 256) {
            const int q = t < TM ? m0 + t : n0 + (t - TM);
            int2 e = make_int2(-1, -1);
            if (q < (t < TM ? M : Nn)) {
                const int4 t4 = R.tabCi[trow + nd.smax + q];
                if (t4.y >= 0 && S0) e.x = t4.y - base0;
                if (t4.z >= 0 && S1) e.y = t4.z - base1;
            }
            if (t < TM) sgr[t] = e; else sgc[t - TM] = e;
        }
        __syncthreads();
    }
    v4f64 cr[MT][NT], ci[MT][NT];
    #pragma unroll
    for (int i = 0; i < MT; ++i)
        #pragma unroll
        for (int j = 0; j < NT; ++j) { cr[i][j] = (v4f64){0, 0, 0, 0}; ci[i][j] = (v4f64){0, 0, 0, 0}; }
    cplx xacc = cmake(0.0, 0.0);                                       // (XR) this thread's share of row TMM, column tid % TN
    const int xc = tid % TN, xh = tid / TN;
    // (one register stage.  A second one -- the loads of slab k + 2 issued before slab k is multiplied -- was measured: 16-18 more registers take the 49 x 64 tile
    // from four workgroups per compute unit to three, leaf back substitution 1.70 -> 1.96 ms, headline 13 900 -> 13 700: reverted)
    cplx ra[NA], rb[NB];
    auto fetch = [&](int k0) {
        #pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int idx = tid + e * 256;
            const int ar = idx / KS, ak = idx % KS;
            cplx v = cmake(0.0, 0.0);
            if (idx < FA * 16 * KS && ar < TM && m0 + ar < M && k0 + ak < K) v = A[(long long)(m0 + ar) * lda + k0 + ak];
            ra[e] = v;
        }
        #pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int idx = tid + e * 256;
            const int bk = idx / TN, bc = idx % TN;
            cplx v = cmake(0.0, 0.0);
            if (idx < TN * KS && k0 + bk < K && n0 + bc < Nn) {
                if (IDX == 2) {
                    const int4 t4 = kidx4[k0 + bk];
                    if (t4.w) v = R.Bx[(long long)t4.x * R.ldx + n0 + bc];
                    if (t4.y >= 0 && ((am0 >> (bc >> 6)) & 1)) v = cadd(v, R.Cix[(long long)t4.y * R.ldx + n0 + bc]);
                    if (t4.z >= 0 && ((am1 >> (bc >> 6)) & 1)) v = cadd(v, R.Cix[(long long)t4.z * R.ldx + n0 + bc]);
                    if (byi == 0 && t4.w && R.Cox) R.Cox[(long long)t4.x * R.ldx + n0 + bc] = v;      // y_S
                }
                else if (idxB) {
                    const int r = kidx[k0 + bk];
                    if (r >= 0) v = (k0 + bk < R.k2 ? R.Bx2 : R.Bx)[(long long)r * R.ldx + n0 + bc];
                }
                else v = B[(long long)(k0 + bk) * ldb + n0 + bc];
            }
            rb[e] = v;
        }
    };
    auto stash = [&](int buf) {
        #pragma unroll
        for (int e = 0; e < NA; ++e) {
            const int idx = tid + e * 256;
            const int ar = idx / KS, ak = idx % KS;
            if (idx < FA * 16 * KS) As[buf * ABUF + ((ak >> 2) * FA + (ar >> 4)) * 64 + (ak & 3) * 16 + ((ar & 15) ^ (ak & 3) ^ (((ak >> 2) & 1) << 2))] = ra[e];
        }
        #pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int idx = tid + e * 256;
            const int bk = idx / TN, bc = idx % TN;
            if (idx < TN * KS) Bs[buf * BBUF + ((bk >> 2) * FB + (bc >> 4)) * 64 + (bk & 3) * 16 + (bc & 15)] = rb[e];
            // (leaf level of the sparse-right-hand-side pass: looked at HERE, where the slab is in registers anyway -- testing the value in fetch()
            // made every load wait for its data and cost the kernel half of its bandwidth)
            if (IDX == 1 && R.act) { const long long bx = __double_as_longlong(rb[e].x) | __double_as_longlong(rb[e].y); nzb |= (bx << 1) ? 1 << (bc >> 6) : 0; }
        }
    };
    int kbeg = 0;
    if (IDX == 1 && R.act_ro && R.k2 > 0) {                             // (leaf back substitution on sparse right-hand sides, see GemmRows::act_ro)
        const int *fl = R.act_ro + (long long)(R.first + R.z0 + bzi) * R.nct + (n0 >> 6);
        const int ncl = ((Nn - n0 < TN ? Nn - n0 : TN) + 63) >> 6;
        int any = 0;
        for (int j = 0; j < ncl; ++j) any |= fl[j];
        if (!any && R.idle_done) return;                              // (uniform across the workgroup)
        if (!any) kbeg = (R.k2 / KS) * KS;
    }
    fetch(kbeg);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < K; k0 += KS) {
        const bool more = k0 + KS < K;
        if (more) fetch(k0 + KS);
        #pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            if (KG > 1 && k0 + 4 * kg >= K) break;                  // (a k group that is all padding)
            cplx a[MT], b[NT];
            #pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = As[cur * ABUF + (kg * FA + wm * MT + i) * 64 + lq * 16 + (lr ^ lq ^ ((kg & 1) << 2))];
            #pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = Bs[cur * BBUF + (kg * FB + wn * NT + j) * 64 + lane];
            #pragma unroll
            for (int i = 0; i < MT; ++i)
                #pragma unroll
                for (int j = 0; j < NT; ++j) {
                    cr[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, cr[i][j], 0, 0, 0);
                    ci[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].y, ci[i][j], 0, 0, 0);
                }
            #pragma unroll
            for (int i = 0; i < MT; ++i) {
                const double nai = -a[i].y;
                #pragma unroll
                for (int j = 0; j < NT; ++j) {
                    cr[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, b[j].y, cr[i][j], 0, 0, 0);
                    ci[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].x, ci[i][j], 0, 0, 0);
                }
            }
        }
        if (XR) {
            #pragma unroll
            for (int kk = 0; kk < KS / XP; ++kk) {
                const int k = kk * XP + xh;                            // k of the slab
                const int kg = k >> 2, kq = k & 3;
                const cplx av = As[cur * ABUF + (kg * FA + FA - 1) * 64 + kq * 16 + (kq ^ ((kg & 1) << 2))];
                const cplx bv = Bs[cur * BBUF + (kg * FB + (xc >> 4)) * 64 + kq * 16 + (xc & 15)];
                cfma(xacc, av, bv);
            }
        }
        if (more) { stash(cur ^ 1); __syncthreads(); cur ^= 1; }
    }
    unsigned colmask = ~0u;                                              // blocks of 64 columns of this tile whose results are stored
    if (IDX == 1 && R.act) {                                             // leaf level: which blocks of 64 columns carry a right-hand side at all
        const int node = R.first + R.z0 + bzi;
        colmask = 0;
        #pragma unroll
        for (int j = 0; j < (TN + 63) / 64; ++j)
            if (__syncthreads_or((nzb >> j) & 1)) {
                colmask |= 1u << j;
                if (byi == 0 && tid == 0 && (n0 >> 6) + j < R.nct) R.act[node * R.nct + (n0 >> 6) + j] = 1;
            }
        // a declared support may be a superset of the nonzeros (explicit zeros among the triplets, a conservative caller): k_nd_support_act has raised
        // the flag already and the parent will read this front's ring rows on its word -- so the zeros are stored wherever the flag is up
        if (R.hint) {
            const int *fl = R.act + (long long)node * R.nct + (n0 >> 6);
            #pragma unroll
            for (int j = 0; j < (TN + 63) / 64; ++j)
                if ((n0 >> 6) + j < R.nct && fl[j]) colmask |= 1u << j;
        }
        // tiles narrower than a block of 64 columns share its flag with their neighbours: another workgroup may raise it, so this one writes its
        // zeros; from 64 columns up nothing but zeros coming in means the rows stay unwritten and the flag stays 0
        if (TN < 64) colmask = ~0u;
        if (!colmask) return;
    }
    const bool b0 = (beta.x == 0.0 && beta.y == 0.0);
    // Epilogue, one block row of 16 at a time: first the addresses of its four rows (row-table look-ups), then EVERY value of C that has to be
    // read (beta != 0, the forward gather's child rows, the Schur gather's child entries) with the loads issued back to back -- masked elements
    // read a zero page instead of being branched around (a per-element `if (...) load` made hipcc wait vmcnt(0) after each one: up to 16
    // dependent round trips per thread) -- then the arithmetic and the stores.
    #pragma unroll
    for (int i = 0; i < MT; ++i) {
        cplx *dstq[4];
        const cplx *cinq[4], *cin2q[4];
        int2 erq[4];
        bool rowok[4];
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = m0 + (wm * MT + i) * 16 + lq + 4 * q;
            rowok[q] = r < M;
            const int rr = rowok[q] ? r : m0;                              // (a valid row for the look-ups of a masked one)
            cplx *dst = C + (long long)rr * ldc;
            const cplx *cin = dst, *cin2 = nullptr;
            if (IDX == 2) {
                const int4 t4 = R.tabCi[trow + R.offCi + rr];
                cin = t4.y >= 0 ? R.Cix + (long long)t4.y * R.ldx : nullptr;
                cin2 = t4.z >= 0 ? R.Cix + (long long)t4.z * R.ldx : nullptr;
            }
            if (IDX == 1 && R.tabCo) {
                const int ix = R.tabCo[trow + R.offCo + rr].x;
                if (ix < 0) rowok[q] = false;
                dst = R.Cox + (long long)(ix >= 0 ? ix : 0) * R.ldx;
            }
            if (IDX == 1 && R.tabCi && !b0) {
                const int ix = R.tabCi[trow + R.offCi + rr].x;
                cin = ix >= 0 ? R.Cix + (long long)ix * R.ldx : nullptr;
            }
            if (IDX == 4) erq[q] = sgr[rr - m0];
            if ((IDX == 0 || IDX == 1) && (b0 || (rr >= R.zr0 && rr < R.zr1))) cin = nullptr;
            dstq[q] = dst; cinq[q] = cin; cin2q[q] = cin2;
        }
        cplx cv[4][NT];
        #pragma unroll
        for (int q = 0; q < 4; ++q)
            #pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cc = n0 + (wn * NT + j) * 16 + lr;
                const bool ok = rowok[q] && cc < Nn && ((colmask >> (((wn * NT + j) * 16 + lr) >> 6)) & 1);
                if (IDX == 4) {
                    const int2 ec = sgc[cc < Nn ? cc - n0 : 0];
                    const cplx *p0 = (ok && erq[q].x >= 0 && ec.x >= 0) ? S0 + (long long)erq[q].x * ld0 + ec.x : g_zero_page;
                    const cplx *p1 = (ok && erq[q].y >= 0 && ec.y >= 0) ? S1 + (long long)erq[q].y * ld1 + ec.y : g_zero_page;
                    cv[q][j] = cadd(*p0, *p1);
                } else if (IDX == 2) {
                    const int cl = ((wn * NT + j) * 16 + lr) >> 6;          // block of 64 columns inside the tile
                    const cplx *p0 = (ok && cinq[q] && ((am0 >> cl) & 1)) ? cinq[q] + cc : g_zero_page;
                    const cplx *p1 = (ok && cin2q[q] && ((am1 >> cl) & 1)) ? cin2q[q] + cc : g_zero_page;
                    cv[q][j] = cadd(*p0, *p1);
                } else {
                    const cplx *p0 = (ok && cinq[q] && !(cc >= R.zc0 && cc < R.zc1)) ? cinq[q] + cc : g_zero_page;
                    cv[q][j] = *p0;
                }
            }
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = m0 + (wm * MT + i) * 16 + lq + 4 * q;
            const bool srow = r >= R.sk0 && r < R.sk1;
            #pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cc = n0 + (wn * NT + j) * 16 + lr;
                if (!rowok[q] || cc >= Nn || !((colmask >> (((wn * NT + j) * 16 + lr) >> 6)) & 1)) continue;
                if (srow && cc >= R.sk0 && cc < R.sk1) continue;
                cplx v = cmul(alpha, cmake(cr[i][j][q], ci[i][j][q]));
                if (IDX == 4) v = cadd(cv[q][j], v);
                else v = cadd(v, cmul(beta, cv[q][j]));
                if (IDX == 1 && R.cj_out) v = conj_scaled(R.oscale, v);
                if (R.ntc) __builtin_nontemporal_store((v2f64){v.x, v.y}, reinterpret_cast<v2f64 *>(dstq[q] + cc)); else dstq[q][cc] = v;
                if (IDX == 1 && R.Cox2) {                         // (the caller's wavefield array: written once, read by the residual check only)
                    const cplx u = conj_scaled(R.oscale, v);
                    __builtin_nontemporal_store((v2f64){u.x, u.y}, reinterpret_cast<v2f64 *>(dstq[q] + (R.Cox2 - R.Cox) + cc));
                }
            }
        }
    }
    if (XR) {
        __syncthreads();                                               // every wave is done with the operand tiles: their LDS holds the partial sums now
        cplx *part = lds;
        if (xh > 0) part[(xh - 1) * TN + xc] = xacc;
        __syncthreads();
        const int r = m0 + TMM, cc = n0 + xc;
        if (xh == 0 && r < M && cc < Nn) {
            #pragma unroll
            for (int h = 1; h < XP; ++h) xacc = cadd(xacc, part[(h - 1) * TN + xc]);
            cplx *dst = C + (long long)r * ldc;
            const cplx *cin = dst;
            bool keep = true;
            if (IDX == 1 && R.tabCo) {
                const int ix = R.tabCo[trow + R.offCo + r].x;
                keep = ix >= 0;
                dst = R.Cox + (long long)(ix >= 0 ? ix : 0) * R.ldx;
            }
            if (IDX == 1 && R.tabCi && !b0) {
                const int ix = R.tabCi[trow + R.offCi + r].x;
                cin = ix >= 0 ? R.Cix + (long long)ix * R.ldx : nullptr;
            }
            if (keep) {
                cplx v = cmul(alpha, xacc);
                if (!b0 && cin) v = cadd(v, cmul(beta, cin[cc]));
                if (IDX == 1 && R.cj_out) v = conj_scaled(R.oscale, v);
                dst[cc] = v;
                if (IDX == 1 && R.Cox2 && R.tabCo) { const cplx u = conj_scaled(R.oscale, v); dst[(R.Cox2 - R.Cox) + cc] = u; }
            }
        }
    }
}

