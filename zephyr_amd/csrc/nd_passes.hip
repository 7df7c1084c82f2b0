// Direct solver, the passes: forward elimination (front vectors travel up the tree like the Schur complements) and back substitution
// x_S = F11^-1 (y_S - F12 x_B) top-down, level by level as batched products on node-major right-hand sides X[cell][rhs]; fronts that see
// nothing but zeros are skipped (sparse sources); ill-conditioned fronts go through their own pivoted LU.
#include "nd_internal.hpp"

namespace {

// mask[cell]: bit b = the right-hand sides may be nonzero at this cell in block b of 64 columns (leaf cells: the leaf's flag; separator cells: always)
__global__ __launch_bounds__(256) void k_nd_qmask(const int *cellnode, const int *act, int nct, long long N, unsigned char *mask) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (long long)gridDim.x * blockDim.x) {
        const int nd = cellnode[c];
        unsigned m = 0xFF;
        if (nd >= 0) { m = 0; for (int b = 0; b < nct && b < 8; ++b) if (act[(long long)nd * nct + b]) m |= 1u << b; }
        mask[c] = (unsigned char)m;
    }
}

// forward pass, one group: V[row] = [separator row: Xt[cell]] + outgoing rows of the children; separator rows of
// non-leaf fronts are final (y_S) and written back to Xt.  blockDim = (LX, 256 / LX), LX lanes over the right-hand sides.
// act (may be null): flags of the sparse-right-hand-side forward pass (GemmRows::act) -- a child's rows count only where its flag is set;
// nodes / first / nmax: the fronts these rows belong to (row / nmax-th front from `first`)
__global__ __launch_bounds__(256) void k_nd_fwd_rows(const int4 *tab, cplx *V, const cplx *arenaV, const cplx *Qt, cplx *Xt, long long rows, int nrhs, int write_back,
                                                     const int *act = nullptr, int nct = 0, const NdDev *nodes = nullptr, int first = 0, int nmax = 1) {
    for (long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y; row < rows; row += (long long)gridDim.x * blockDim.y) {
        const int4 e = tab[row];
        cplx *dst = V + row * nrhs;
        const cplx *s0 = e.w ? Qt + (long long)e.x * nrhs : nullptr;
        const cplx *s1 = e.y >= 0 ? arenaV + (long long)e.y * nrhs : nullptr;
        const cplx *s2 = e.z >= 0 ? arenaV + (long long)e.z * nrhs : nullptr;
        const int *a1 = nullptr, *a2 = nullptr;
        if (act) { const NdDev nd = nodes[first + (int)(row / nmax)]; a1 = nd.kid[0] >= 0 ? act + (long long)nd.kid[0] * nct : nullptr; a2 = nd.kid[1] >= 0 ? act + (long long)nd.kid[1] * nct : nullptr; }
        for (int r = threadIdx.x; r < nrhs; r += blockDim.x) {
            cplx acc = s0 ? s0[r] : cmake(0.0, 0.0);
            if (s1 && (!act || (a1 && a1[r >> 6]))) acc = cadd(acc, s1[r]);
            if (s2 && (!act || (a2 && a2[r >> 6]))) acc = cadd(acc, s2[r]);
            dst[r] = acc;
            if (write_back && e.w) Xt[(long long)e.x * nrhs + r] = acc;
        }
    }
}

// backward pass: V[row] = Xt[cell of the row] (separator and ring rows), 0 for padding
__global__ __launch_bounds__(256) void k_nd_bwd_gather(const int4 *tab, cplx *V, const cplx *XS, const cplx *Xt, long long rows, int nrhs) {
    // XS: where the separator rows' y_S lives (the right-hand sides themselves for leaves, Xt otherwise)
    for (long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y; row < rows; row += (long long)gridDim.x * blockDim.y) {
        const int4 e = tab[row];
        cplx *dst = V + row * nrhs;
        const cplx *src = e.x >= 0 ? (e.w ? XS : Xt) + (long long)e.x * nrhs : nullptr;
        for (int r = threadIdx.x; r < nrhs; r += blockDim.x) dst[r] = src ? src[r] : cmake(0.0, 0.0);
    }
}

// backward pass: Xt[separator cells] = XS (cnt x smax rows)
// (U2 != null: the caller's wavefield array takes conj(oscale x) of the same rows, see GemmRows::Cox2)
__global__ __launch_bounds__(256) void k_nd_bwd_store(const int4 *tab, const cplx *XS, cplx *Xt, long long rows, int smax, int nmax, int nrhs,
                                                      cplx *U2 = nullptr, cplx oscale = {1.0, 0.0}) {
    for (long long row = (long long)blockIdx.x * blockDim.y + threadIdx.y; row < rows; row += (long long)gridDim.x * blockDim.y) {
        const long long j = row / smax;
        const int a = (int)(row - j * smax);
        const int4 e = tab[j * nmax + a];
        if (!e.w) continue;
        const cplx *src = XS + row * nrhs;
        cplx *dst = Xt + (long long)e.x * nrhs;
        for (int r = threadIdx.x; r < nrhs; r += blockDim.x) dst[r] = src[r];
        if (U2) {
            cplx *d2 = U2 + (long long)e.x * nrhs;
            for (int r = threadIdx.x; r < nrhs; r += blockDim.x) d2[r] = conj_scaled(oscale, src[r]);
        }
    }
}

// Forward pass on sparse right-hand sides, separator levels (round 5): which (front, block of 64 columns) pairs of a level have anything to do -- a child with
// outgoing rows there, or a nonzero among the front's own right-hand-side rows -- decided by ONE WAVE per pair before the level's product is launched.
// The product's workgroups hold 144 registers per lane (three per compute unit): with the decision inside them, a level of 8192 fronts x 4 blocks cost
// 200 us whether 8192 or 8 of its fronts had work (43 rounds of workgroups that fetch a node record, two flags and eight rows and leave).  Here 32 waves
// per compute unit do the looking, the pairs with work go on a list (in no particular order: every pair is computed by itself), the others get the zeros
// the back substitution expects in their y_S rows, and the product is dealt from the list.
// NR rows per wave, WPP waves per pair (a workgroup of four waves takes 4 / WPP pairs).  All row-table entries of a wave are requested first, then all the
// right-hand-side values they point to (masked rows read a zero instead of being branched around): two memory round trips per wave, beside the two of the
// children's flags -- a first version with `if (valid) load` inside the row loop paid two per ROW (117 us for the 8192-front level).
static __device__ cplx g_fwd_zero[4];
template <int NR, int WPP>
__global__ __launch_bounds__(256) void k_fwd_flags(const int4 *__restrict__ tab, int nmax, int smax, const NdDev *__restrict__ nodes, int first, int cnt, int nct,
                                                   int *__restrict__ act, const cplx *__restrict__ Q, cplx *__restrict__ Xt, int ldx, int nrhs,
                                                   int *__restrict__ count, int *__restrict__ list, int leaf) {
    __shared__ int wnz[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * (4 / WPP) + wave / WPP, part = wave % WPP;
    const bool live = pair < cnt * nct;
    const int j = live ? pair / nct : 0, b = live ? pair - j * nct : 0;
    const int k0 = nodes[first + j].kid[0], k1 = nodes[first + j].kid[1];
    const int a0 = k0 >= 0 ? act[(long long)k0 * nct + b] : 0, a1 = k1 >= 0 ? act[(long long)k1 * nct + b] : 0;
    const int col = 64 * b + lane;
    const int4 *t = tab + (long long)j * nmax;
    int rows[NR];
    #pragma unroll
    for (int i = 0; i < NR; ++i) { const int r = part + WPP * i; const int4 e = (live && r < smax) ? t[r] : make_int4(-1, -1, -1, 0); rows[i] = (e.w && col < nrhs) ? e.x : -1; }
    cplx v[NR];
    #pragma unroll
    for (int i = 0; i < NR; ++i) v[i] = *(rows[i] >= 0 ? Q + (long long)rows[i] * ldx + col : g_fwd_zero);
    int nz = 0;
    #pragma unroll
    for (int i = 0; i < NR; ++i) nz |= (v[i].x != 0.0 || v[i].y != 0.0) ? 1 : 0;
    int anyw = __any(nz) ? 1 : 0;
    if (WPP > 1) {
        if (lane == 0) wnz[wave] = anyw;
        __syncthreads();
        anyw = 0;
        #pragma unroll
        for (int q = 0; q < WPP; ++q) anyw |= wnz[(wave / WPP) * WPP + q];
    }
    if (!live) return;
    if (a0 || a1 || anyw) {
        if (part == 0 && lane == 0) { act[(long long)(first + j) * nct + b] = 1; if (!leaf) list[atomicAdd(count, 1)] = pair; }
        return;
    }
    if (leaf) return;                                      // (a leaf's y_S is the right-hand side itself: nothing to zero, and its product leaves on the flag)
    #pragma unroll
    for (int i = 0; i < NR; ++i) if (rows[i] >= 0) Xt[(long long)rows[i] * ldx + col] = cmake(0.0, 0.0);      // y_S = 0 where the back substitution will look for it
}

// The leaves' flags (no declared support): one workgroup per leaf, wave w looks at the column blocks w, w + 4, ... -- the four waves of a workgroup read the 4 KB of
// a right-hand-side row side by side (k_fwd_flags<32, 2> gave a pair of waves 1 KB of it each: 736 us for the 3.2 GB of leaf rows of the 1024^2 x 256 job, 0.54 of
// the HBM rate).  Rows in passes of NR, every load of a pass in flight at once.  Same decisions, same flags.
template <int NR>
__global__ __launch_bounds__(256) void k_leaf_flags(const int4 *__restrict__ tab, int nmax, int smax, int first, int nct, int *__restrict__ act,
                                                    const cplx *__restrict__ Q, int ldx, int nrhs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x;
    const int4 *t = tab + (long long)j * nmax;
    for (int b = wave; b < nct; b += 4) {
        const int col = 64 * b + lane;
        int nz = 0;
        for (int r0 = 0; r0 < smax; r0 += NR) {
            int rows[NR];
            #pragma unroll
            for (int i = 0; i < NR; ++i) { const int r = r0 + i; const int4 e = r < smax ? t[r] : make_int4(-1, -1, -1, 0); rows[i] = (e.w && col < nrhs) ? e.x : -1; }
            cplx v[NR];
            #pragma unroll
            for (int i = 0; i < NR; ++i) v[i] = *(rows[i] >= 0 ? Q + (long long)rows[i] * ldx + col : g_fwd_zero);
            #pragma unroll
            for (int i = 0; i < NR; ++i) nz |= (v[i].x != 0.0 || v[i].y != 0.0) ? 1 : 0;
        }
        if (__any(nz) && lane == 0) act[(long long)(first + j) * nct + b] = 1;
    }
}

struct SolveCtx {
    const int4 *tab; cplx *Xt, *arenaV; int nrhs; dim3 rb; int use_idx;
    const cplx *Qt;       // node-major right-hand sides (read only); == Xt for an in-place solve
    int *act = nullptr; int nct = 0;     // sparse-right-hand-side flags of the forward pass (null: every front is computed)
    int act_hint = 0;                    // the leaves' flags come from the support the caller declared (helm_set_rhs_support): no scan of q
    int *fcount = nullptr, *flist = nullptr;      // (with act) per group: number of active (front, block) pairs of a separator level; the list of the level at hand
    cplx *Uout = nullptr; cplx oscale = {1.0, 0.0};      // direct output (NdDirectOut): the back substitution leaves u = conj(oscale x) in the caller's array [cell][nrhs]
    dim3 rgrid(long long rows) const { return dim3((unsigned)std::min<long long>((rows + rb.y - 1) / rb.y, 1 << 20)); }
};

SolveCtx solve_ctx(const NdFactor *f, cplx *ws, int nrhs) {
    const int use_idx = 1;      // lower tree levels address the Xt rows through the row table inside the products (no gather / scatter passes)
    const NdPlan &P = f->pd->plan;
    SolveCtx c;
    c.tab = f->pd->d_tab; c.Xt = ws; c.arenaV = ws + (long long)P.dof * P.nz * P.nx * nrhs; c.nrhs = nrhs; c.use_idx = use_idx;
    c.Qt = c.Xt;
    int lx = 1;
    while (lx < nrhs && lx < 256) lx <<= 1;
    c.rb = dim3(lx, 256 / lx);
    return c;
}

// ill-conditioned fronts of group gi (NdStable): their outgoing rows once more, through the front's own LU
void forward_stable(helm_op *op, NdFactor *f, size_t gi, const SolveCtx &c) {
    const NdPlan &P = f->pd->plan;
    const NdGroup &g = P.groups[gi];
    const cplx one = cmake(1, 0), mone = cmake(-1, 0);
    for (NdStable &S : f->stable) {
        if (S.group != gi) continue;
        const NdDev &n = P.nodes[S.node];
        const int nmax = S.smax + S.mmax, nrhs = c.nrhs;
        cplx *V = c.arenaV + n.voff * nrhs;
        // the front vector gathered again: separator rows q_S + the children's rows (written to Xt as y_S for the back substitution), ring rows
        // the children's rows; then z = F11^-1 y_S through the LU and V_B -= F21 z
        HELM_LAUNCH(k_nd_fwd_rows, c.rgrid(nmax), c.rb, 0, op->stream, c.tab + n.roff, V, c.arenaV, c.Qt, c.Xt, (long long)nmax, nrhs, g.leaf ? 0 : 1,
                           (const int *)c.act, c.nct, (const NdDev *)f->pd->d_nodes, S.node, nmax);
        if (c.act) hipMemsetAsync(c.act + (long long)S.node * c.nct, 1, (size_t)c.nct * sizeof(int), op->stream);      // (every row of this front has been written)
        launch_lu_solve(op->stream, S.lu, nmax, S.smax, S.piv, V, nrhs, nrhs);
        gemm(op, S.mmax, nrhs, S.smax, mone, S.f21, S.smax, 0, V, nrhs, 0, one, V + (long long)S.smax * nrhs, nrhs, 0, 1);
    }
}

void forward_group_batched(helm_op *op, NdFactor *f, size_t gi, const SolveCtx &c);
// forward elimination of one group on op->stream
void forward_group(helm_op *op, NdFactor *f, size_t gi, const SolveCtx &c) {
    forward_group_batched(op, f, gi, c);
    if (!f->stable.empty()) forward_stable(op, f, gi, c);
}

void forward_group_batched(helm_op *op, NdFactor *f, size_t gi, const SolveCtx &c) {
    const NdPlan &P = f->pd->plan;
    const cplx one = cmake(1, 0), mone = cmake(-1, 0), zero = cmake(0, 0);
    const NdGroup &g = P.groups[gi];
    const int nmax = g.smax + g.mmax, nrhs = c.nrhs;
    const long long rows = (long long)g.cnt * nmax;
    cplx *V = c.arenaV + g.voff * nrhs;
    if (c.use_idx && g.leaf && g.mmax > 0 && g.smax <= GB_KIDX) {
        // leaves have no children: the outgoing ring part is -G21 x_S with x_S read straight from Xt
        GemmRows R; R.tabB = c.tab + g.roff; R.offB = 0; R.tab_stride = nmax; R.Bx = c.Qt; R.ldx = nrhs;
        R.act = c.act; R.nct = c.nct; R.first = g.first; R.hint = R.act ? c.act_hint : 0;
        // r5: without a declared support the leaves' flags are found by a scan of their own (every load of a pass in flight at
        // once) and the product then behaves as with a declared one: a workgroup whose flags are down leaves on one load.  (It used to find out itself:
        // 753 us for the 3.2 GB of leaf rows, 54 % of the HBM rate, from workgroups of 134 registers per lane.)
        if (R.act && !R.hint && c.flist && g.smax <= 64 && c.Qt != c.Xt && helm_tuning_now().nd_leaf_idle != 0) {
            // (r6: one workgroup per leaf -- k_leaf_flags: 734 -> 561 us on the 1024^2 x 256 job, 0.54 -> 0.71 of the HBM rate)
            HELM_LAUNCH((k_leaf_flags<25>), dim3(g.cnt), dim3(256), 0, op->stream, c.tab + g.roff, nmax, g.smax, g.first, c.nct, c.act, c.Qt, nrhs, nrhs);
            R.hint = 1;
        }
        gemm(op, g.mmax, nrhs, g.smax, mone, nd_fac_at(f, g.g21, (long long)g.mmax * g.smax), g.smax, nd_fac_stride(f, (long long)g.mmax * g.smax), nullptr, 0, 0, zero,
             V + (long long)g.smax * nrhs, nrhs, (long long)nmax * nrhs, g.cnt, &R);
        if (c.act && !R.act) hipMemsetAsync(c.act + (long long)g.first * c.nct, 1, (size_t)g.cnt * c.nct * sizeof(int), op->stream);
        return;
    }
    if (c.use_idx && c.Qt != c.Xt && !g.leaf && g.mmax > 0 && g.mmax <= 256 && g.smax <= GB_KIDX) {
        // (lower levels only: with many row-tiles per front every one of them repeats the three-source gather -- measured slower from m = 1025 up)
        // (out-of-place solves only: in place, the y_S store of the first row-tile would race with the other row-tiles' reads of q_S)
        // the gather of k_nd_fwd_rows happens inside the GEMM's operand loads: V_B = (children's rows) - G21 (q_S + children's rows),
        // y_S stored to Xt on the way
        GemmRows R; R.fwd3 = 1; R.tabB = c.tab + g.roff; R.offB = 0; R.tabCi = c.tab + g.roff; R.offCi = g.smax; R.tab_stride = nmax;
        R.Bx = c.Qt; R.Cix = c.arenaV; R.Cox = c.Xt; R.ldx = nrhs;
        R.act = c.act; R.nct = c.nct; R.first = g.first; R.nodes = f->pd->d_nodes;
        if (c.act && c.flist && g.cnt <= 65535 && g.smax <= 128 && helm_tuning_now().nd_leaf_idle != 0) {      // who has work: decided before the launch, which is dealt from the list
            const int pairs = g.cnt * c.nct;
#define FWD_FLAGS(NR_, WPP_) HELM_LAUNCH((k_fwd_flags<NR_, WPP_>), dim3((pairs + 4 / WPP_ - 1) / (4 / WPP_)), dim3(256), 0, op->stream, c.tab + g.roff, nmax, g.smax, \
                                                (const NdDev *)f->pd->d_nodes, g.first, g.cnt, c.nct, c.act, c.Qt, c.Xt, nrhs, nrhs, c.fcount + gi, c.flist, 0)
            if (g.smax <= 8) FWD_FLAGS(8, 1); else if (g.smax <= 16) FWD_FLAGS(16, 1); else if (g.smax <= 32) FWD_FLAGS(16, 2); else if (g.smax <= 64) FWD_FLAGS(16, 4);
            else FWD_FLAGS(32, 4);
#undef FWD_FLAGS
            R.list = c.flist; R.lcount = c.fcount + gi;
        }
        { const NdDev &n0 = P.nodes[g.first]; R.child_rows = (n0.kid[0] >= 0 ? P.nodes[n0.kid[0]].mmax : 0) + (n0.kid[1] >= 0 ? P.nodes[n0.kid[1]].mmax : 0); }
        gemm(op, g.mmax, nrhs, g.smax, mone, nd_fac_at(f, g.g21, (long long)g.mmax * g.smax), g.smax, nd_fac_stride(f, (long long)g.mmax * g.smax), nullptr, 0, 0, one,
             V + (long long)g.smax * nrhs, nrhs, (long long)nmax * nrhs, g.cnt, &R);
        if (c.act && !R.act) hipMemsetAsync(c.act + (long long)g.first * c.nct, 1, (size_t)g.cnt * c.nct * sizeof(int), op->stream);
        return;
    }
    HELM_LAUNCH(k_nd_fwd_rows, c.rgrid(rows), c.rb, 0, op->stream, c.tab + g.roff, V, c.arenaV, c.Qt, c.Xt, rows, nrhs, g.leaf ? 0 : 1,
                       (const int *)c.act, c.nct, (const NdDev *)f->pd->d_nodes, g.first, nmax);
    if (c.act) hipMemsetAsync(c.act + (long long)g.first * c.nct, 1, (size_t)g.cnt * c.nct * sizeof(int), op->stream);      // (these fronts write every row)
    if (g.mmax > 0)     // outgoing ring part: V_B -= G21 V_S
        gemm(op, g.mmax, nrhs, g.smax, mone, nd_fac_at(f, g.g21, (long long)g.mmax * g.smax), g.smax, nd_fac_stride(f, (long long)g.mmax * g.smax), V, nrhs, (long long)nmax * nrhs, one,
             V + (long long)g.smax * nrhs, nrhs, (long long)nmax * nrhs, g.cnt);
}

// ill-conditioned fronts of group gk: [y_S; x_B] is set aside before the batched launches overwrite y_S in Xt (pre), and x_S = F11^-1 (y_S - F12 x_B)
// through the front's LU replaces what they wrote (post)
int backward_stable(helm_op *op, NdFactor *f, size_t gk, const SolveCtx &c, bool post) {
    const NdPlan &P = f->pd->plan;
    const NdGroup &g = P.groups[gk];
    const cplx one = cmake(1, 0), mone = cmake(-1, 0);
    for (NdStable &S : f->stable) {
        if (S.group != gk) continue;
        const NdDev &n = P.nodes[S.node];
        const int nmax = S.smax + S.mmax, nrhs = c.nrhs;
        if (!post) {
            const size_t need = (size_t)nmax * nrhs;
            if (S.vs_elems < need) {
                if (S.vs) { hipStreamSynchronize(op->stream); helm_pool_free(op->device, S.vs, S.vs_elems * sizeof(cplx)); S.vs = nullptr; S.vs_elems = 0; }
                S.vs = (cplx *)helm_pool_alloc(op->device, need * sizeof(cplx));
                if (!S.vs) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: scratch for an ill-conditioned front failed");
                S.vs_elems = need;
            }
            HELM_LAUNCH(k_nd_bwd_gather, c.rgrid(nmax), c.rb, 0, op->stream, c.tab + n.roff, S.vs, g.leaf ? c.Qt : (const cplx *)c.Xt, (const cplx *)c.Xt, (long long)nmax, nrhs);
        } else {
            if (S.mmax > 0) gemm(op, S.smax, nrhs, S.mmax, mone, S.lu + S.smax, nmax, 0, S.vs + (long long)S.smax * nrhs, nrhs, 0, one, S.vs, nrhs, 0, 1);
            launch_lu_solve(op->stream, S.lu, nmax, S.smax, S.piv, S.vs, nrhs, nrhs);
            HELM_LAUNCH(k_nd_bwd_store, c.rgrid(S.smax), c.rb, 0, op->stream, c.tab + n.roff, (const cplx *)S.vs, c.Xt, (long long)S.smax, S.smax, nmax, nrhs, c.Uout, c.oscale);
        }
    }
    return HELM_OK;
}

void backward_group_batched(helm_op *op, NdFactor *f, size_t gk, const SolveCtx &c);
// back substitution of one group on op->stream
void backward_group(helm_op *op, NdFactor *f, size_t gk, const SolveCtx &c) {
    if (!f->stable.empty()) (void)backward_stable(op, f, gk, c, false);
    backward_group_batched(op, f, gk, c);
    if (!f->stable.empty()) (void)backward_stable(op, f, gk, c, true);
}

void backward_group_batched(helm_op *op, NdFactor *f, size_t gk, const SolveCtx &c) {
    const NdPlan &P = f->pd->plan;
    const cplx one = cmake(1, 0), mone = cmake(-1, 0), zero = cmake(0, 0);
    const NdGroup &g = P.groups[gk];
    const int nmax = g.smax + g.mmax, nrhs = c.nrhs;
    const long long rows = (long long)g.cnt * nmax;
    const long long s1 = nd_fac_stride(f, (long long)g.smax * nmax);      // stride of a front's [F11^-1 | F12] rows (nf times the slot for a factor of a set, direct.hpp)
    const cplx *Finv = nd_fac_at(f, g.finv, (long long)g.smax * nmax), *F12 = Finv + g.smax;
    // leaves hold G = -F11^-1 F12 in place of F12: x_S = F11^-1 y_S + G x_B
    const bool gform = merged_group(P, g);
    cplx *V = c.arenaV + g.voff * nrhs;
    // the other region is free in this pass: separator results go there
    const long long xs_off = g.voff + ((g.level & 1) ? -P.vregion : P.vregion);
    cplx *XS = c.arenaV + xs_off * nrhs;
    if (c.use_idx && g.mmax > 0 && g.mmax <= GB_KIDX) {
        // lower tree levels (almost all rows): every Xt row addressed through the row table -- no gather / store pass
        if (gform && c.Qt != c.Xt && nmax <= GB_KIDX) {
            // leaves: x_S = [F11^-1 | G] [y_S; x_B] in ONE product -- y_S rows from the right-hand sides, x_B rows from Xt; the result goes
            // straight to the Xt rows (no intermediate: 2 x 3.2 GB less per pass at 1024^2 x 256)
            GemmRows R; R.tabB = c.tab + g.roff; R.offB = 0; R.tabCo = c.tab + g.roff; R.offCo = 0; R.tab_stride = nmax;
            R.Bx = c.Xt; R.Bx2 = c.Qt; R.k2 = g.smax; R.Cox = c.Xt; R.ldx = nrhs;
            // direct output: nothing on the GPU reads a leaf cell's x again but the residual check -- the rows go to the caller's array as u = conj(oscale x);
            // separator rows stay in Xt for the levels below and go to the caller's array beside it
            if (c.Uout) { R.Cox = c.Uout; R.cj_out = 1; R.oscale = c.oscale; }
            R.act_ro = g.leaf ? c.act : nullptr; R.nct = c.nct; R.first = g.first;
            gemm(op, g.smax, nrhs, nmax, one, Finv, nmax, s1, nullptr, 0, 0, zero, nullptr, 0, 0, g.cnt, &R);
            return;
        }
        if (gform) {        // right-hand sides and wavefields share their rows (Qt == Xt): V = F11^-1 y_S first, then x_S = V + G x_B
            GemmRows R1; R1.tabB = c.tab + g.roff; R1.offB = 0; R1.tab_stride = nmax; R1.Bx = c.Qt; R1.ldx = nrhs;
            gemm(op, g.smax, nrhs, g.smax, one, Finv, nmax, s1, nullptr, 0, 0, zero, V, nrhs, (long long)g.smax * nrhs, g.cnt, &R1);
            GemmRows R2; R2.tabB = c.tab + g.roff; R2.offB = g.smax; R2.tabCo = c.tab + g.roff; R2.offCo = 0; R2.tab_stride = nmax;
            R2.Bx = c.Xt; R2.Cox = c.Xt; R2.ldx = nrhs;
            gemm(op, g.smax, nrhs, g.mmax, one, F12, nmax, s1, nullptr, 0, 0, one, V, nrhs, (long long)g.smax * nrhs, g.cnt, &R2);
            return;
        }
        // small separator fronts (r5): both products in one launch, y_S -> x_S in place in Xt (k_sep_bwd_small)
        if (!g.leaf && P.dof == 1 && g.smax <= 16) {
            GemmRows R; R.tabB = c.tab + g.roff; R.offB = g.smax; R.tabCo = c.tab + g.roff; R.offCo = 0; R.tab_stride = nmax;
            R.Bx = c.Xt; R.Cix = c.Xt; R.Cox = c.Xt; R.ldx = nrhs;
            if (c.Uout) { R.Cox2 = c.Uout; R.oscale = c.oscale; }
            if (gemm_sep_bwd_small(op, g.smax, g.mmax, nrhs, Finv, F12, nmax, s1, g.cnt, R)) return;
        }
        // T = y_S - F12 x_B and x_S = F11^-1 T
        GemmRows R1; R1.tabB = c.tab + g.roff; R1.offB = g.smax; R1.tabCi = c.tab + g.roff; R1.offCi = 0; R1.tab_stride = nmax;
        R1.Bx = c.Xt; R1.Cix = g.leaf ? c.Qt : c.Xt; R1.ldx = nrhs;      // a leaf's y_S is still the right-hand side itself
        gemm(op, g.smax, nrhs, g.mmax, mone, F12, nmax, s1, nullptr, 0, 0, one, V, nrhs, (long long)g.smax * nrhs, g.cnt, &R1);
        GemmRows R2; R2.tabCo = c.tab + g.roff; R2.offCo = 0; R2.tab_stride = nmax; R2.Cox = c.Xt; R2.ldx = nrhs;
        if (c.Uout) { R2.Cox2 = c.Uout; R2.oscale = c.oscale; }        // (separator rows: x stays in Xt for the levels below, u goes to the caller's array beside it)
        gemm(op, g.smax, nrhs, g.smax, one, Finv, nmax, s1, V, nrhs, (long long)g.smax * nrhs, zero, nullptr, 0, 0, g.cnt, &R2);
        return;
    }
    HELM_LAUNCH(k_nd_bwd_gather, c.rgrid(rows), c.rb, 0, op->stream, c.tab + g.roff, V, g.leaf ? c.Qt : c.Xt, c.Xt, rows, nrhs);
    if (gform) {            // the gathered front vector is [y_S; x_B]: one dense product with [F11^-1 | G]
        gemm(op, g.smax, nrhs, nmax, one, Finv, nmax, s1, V, nrhs, (long long)nmax * nrhs, zero, XS, nrhs, (long long)g.smax * nrhs, g.cnt);
    } else {
        if (g.mmax > 0)
            gemm(op, g.smax, nrhs, g.mmax, mone, F12, nmax, s1, V + (long long)g.smax * nrhs, nrhs, (long long)nmax * nrhs,
                 one, V, nrhs, (long long)nmax * nrhs, g.cnt);
        gemm(op, g.smax, nrhs, g.smax, one, Finv, nmax, s1, V, nrhs, (long long)nmax * nrhs, zero, XS, nrhs, (long long)g.smax * nrhs, g.cnt);
    }
    const long long srows = (long long)g.cnt * g.smax;
    HELM_LAUNCH(k_nd_bwd_store, c.rgrid(srows), c.rb, 0, op->stream, c.tab + g.roff, XS, c.Xt, srows, g.smax, nmax, nrhs, c.Uout, c.oscale);
}

}  // namespace

// ---- solve: Xin (nrhs x N, each right-hand side contiguous) -> Xout (may alias Xin) --------------------------------------
// ws: workspace of nd_solve_ws_elems(plan, nrhs) elements
long long nd_solve_ws_elems(const NdPlan &P, int nrhs) { return ((long long)P.dof * P.nz * P.nx + 2 * P.vregion) * nrhs; }

int nd_solve(helm_op *op, NdFactor *f, const cplx *Xin, cplx *Xout, int nrhs, cplx *ws, int conj_out) {
    const NdPlan &P = f->pd->plan;
    hipStream_t st = op->stream;
    const long long N = (long long)P.dof * P.nz * P.nx;          // unknowns per right-hand side
    const SolveCtx c = solve_ctx(f, ws, nrhs);
    GroupTrace t0(st, "transpose");
    launch_transpose(st, Xin, (long long)nrhs, N, c.Xt, 0, 0);
    t0.mark(); t0.report(P, false);
    GroupTrace tf(st, "forward");
    for (size_t gi = 0; gi < P.groups.size(); ++gi) { forward_group(op, f, gi, c); tf.mark(); }       // leaves to root
    tf.report(P, false);
    GroupTrace tb(st, "backward");
    for (size_t gk = P.groups.size(); gk-- > 0;) { backward_group(op, f, gk, c); tb.mark(); }          // root to leaves
    tb.report(P, true);
    GroupTrace t1(st, "transpose");
    launch_transpose(st, c.Xt, N, (long long)nrhs, Xout, 1, conj_out);
    t1.mark(); t1.report(P, false);
    return check_kernels(op, "solve kernels");
}

// act[node][b] = 1 for every leaf that holds a cell whose declared support has bit b (block b of 64 columns) set
__global__ __launch_bounds__(256) void k_nd_support_act(const unsigned char *bits, const int *cellnode, int *act, int nct, long long N) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (long long)gridDim.x * blockDim.x) {
        const unsigned b = bits[c];
        if (!b) continue;
        const int nd = cellnode[c];
        if (nd < 0) continue;                                  // separator cells: their fronts look at q themselves
        for (int j = 0; j < nct; ++j) if ((b >> j) & 1) act[(long long)nd * nct + j] = 1;
    }
}
// (HELM_ND_SUPPORT_CHECK=1) bad[0] = 1 when a right-hand side is nonzero outside the declared support
__global__ __launch_bounds__(256) void k_nd_support_check(const unsigned char *bits, const cplx *Q, int ldq, int nrhs, long long N, int *bad) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < N * nrhs; e += (long long)gridDim.x * blockDim.x) {
        const long long c = e / nrhs; const int j = (int)(e - c * nrhs);
        const cplx v = Q[c * ldq + j];
        if ((v.x != 0.0 || v.y != 0.0) && !((bits[c] >> (j >> 6)) & 1)) bad[0] = 1;
    }
}

// node-major solve: Qt (cells x nrhs, read only) -> Xt (may alias Qt); arenaV: 2 * vregion * nrhs elements
// Flags of the forward pass on sparse right-hand sides, zeroed on `st` (HELM_ND_SPARSE_RHS=0: every front is computed, as before round 4).
// A survey's sources touch a handful of grid cells (81 per Kaiser-windowed source at the surface): below the few fronts that contain them the
// forward elimination multiplies zeros -- the reference hands such right-hand sides over as scipy-sparse matrices for the same reason
// (survey.py:86-89, discretization.py:101-103).  Dense right-hand sides set every flag and cost one flag read per workgroup.
static void arm_sparse_rhs(helm_op *op, NdFactor *f, SolveCtx &c, hipStream_t st) {
    const int on = helm_tuning_now().nd_sparse_rhs;
    c.act = nullptr; c.nct = 0;
    f->act_nct = 0;
    if (!on || c.Qt == c.Xt || f->pd->plan.dof != 1) return;
    const int nct = (c.nrhs + 63) / 64;
    // layout: [flags: fronts x nct][active-pair counts: one per group][the list of the level at hand: largest cnt x nct]
    const size_t nflags = f->pd->plan.nodes.size() * (size_t)nct, ngroups = f->pd->plan.groups.size();
    size_t maxpairs = 1;
    for (const NdGroup &g : f->pd->plan.groups) if (!g.leaf) maxpairs = std::max(maxpairs, (size_t)g.cnt * (size_t)std::max(4, nct));
    const size_t need = nflags + ngroups;
    if (f->act_elems < need + maxpairs) {
        if (f->d_act) { hipStreamSynchronize(st); helm_pool_free(op->device, f->d_act, f->act_elems * sizeof(int)); f->d_act = nullptr; f->act_elems = 0; }
        const size_t want = f->pd->plan.nodes.size() * (size_t)std::max(4, nct) + ngroups + maxpairs;
        f->d_act = (int *)helm_pool_alloc(op->device, want * sizeof(int));
        if (!f->d_act) return;
        f->act_elems = want;
    }
    if (hipMemsetAsync(f->d_act, 0, need * sizeof(int), st) != hipSuccess) { (void)hipGetLastError(); return; }
    c.act = f->d_act; c.nct = nct;
    c.fcount = f->d_act + nflags; c.flist = f->d_act + nflags + ngroups;
    f->act_nct = nct;
    // Declared support (helm_set_rhs_support: one byte per cell, bit b = block b of 64 columns may be nonzero there; what helm_rhs_support_from_coo makes of
    // the triplets of a scipy-sparse source matrix): the leaves' flags are set from it and the leaf level of the forward pass no longer reads q to find out --
    // 3.2 of the 4.3 GB of a 1024^2 x 256 batch.  Only for the pass whose right-hand sides are the caller's own array (refinement passes solve for residuals).
    if (op->rhs_bits && c.Qt == op->rhs_bits_q && c.nrhs == op->rhs_bits_nrhs && nct <= 8 && f->pd->d_cellnode && op->rhs_bits_rows == (long long)f->pd->plan.nz * f->pd->plan.nx) {
        const long long N = op->rhs_bits_rows;
        HELM_LAUNCH(k_nd_support_act, dim3((unsigned)std::min<long long>((N + 255) / 256, 4096)), dim3(256), 0, st, op->rhs_bits, (const int *)f->pd->d_cellnode, f->d_act, nct, N);
        c.act_hint = 1;
        if (getenv("HELM_ND_SUPPORT_CHECK") && atoi(getenv("HELM_ND_SUPPORT_CHECK"))) {
            int *d_bad = (int *)helm_pool_alloc(op->device, sizeof(int));
            if (d_bad) {
                hipMemsetAsync(d_bad, 0, sizeof(int), st);
                HELM_LAUNCH(k_nd_support_check, dim3(4096), dim3(256), 0, st, op->rhs_bits, c.Qt, c.nrhs, c.nrhs, N, d_bad);
                int bad = 0;
                hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st);
                hipStreamSynchronize(st);
                helm_pool_free(op->device, d_bad, sizeof(int));
                if (bad) { helm_set_error(op, "a right-hand side is nonzero outside the support declared with helm_set_rhs_support"); op->rhs_bits_violated = 1; }
            }
        }
    }
    // (tests) HELM_ND_POISON=1: the front-vector arena is filled with NaNs first, so that a read of rows no front has written shows up in the wavefield
    if (getenv("HELM_ND_POISON") && atoi(getenv("HELM_ND_POISON"))) (void)hipMemsetAsync(c.arenaV, 0xFF, (size_t)2 * f->pd->plan.vregion * c.nrhs * sizeof(cplx), st);
}

const unsigned char *nd_rhs_mask(helm_op *op, NdFactor *f) {
    if (!f || !f->act_nct || f->act_nct > 8 || !f->d_act || !f->pd->d_cellnode) return nullptr;
    const long long N = (long long)f->pd->plan.nz * f->pd->plan.nx;
    if (f->qmask_elems < (size_t)N) {
        if (f->d_qmask) { hipStreamSynchronize(op->stream); helm_pool_free(op->device, f->d_qmask, f->qmask_elems); f->d_qmask = nullptr; f->qmask_elems = 0; }
        f->d_qmask = (unsigned char *)helm_pool_alloc(op->device, (size_t)N);
        if (!f->d_qmask) return nullptr;
        f->qmask_elems = (size_t)N;
    }
    HELM_LAUNCH(k_nd_qmask, dim3((unsigned)std::min<long long>((N + 255) / 256, 4096)), dim3(256), 0, op->stream, (const int *)f->pd->d_cellnode, (const int *)f->d_act, f->act_nct, N, f->d_qmask);
    return f->d_qmask;
}

int nd_solve_nm(helm_op *op, NdFactor *f, const cplx *Qt, cplx *Xt, int nrhs, cplx *arenaV, const NdDirectOut *dout) {
    const NdPlan &P = f->pd->plan;
    SolveCtx c = solve_ctx(f, Xt, nrhs);
    c.Qt = Qt; c.Xt = Xt; c.arenaV = arenaV;
    if (dout && dout->U && P.dof == 1 && Qt != Xt) { c.Uout = dout->U; c.oscale = dout->oscale; }
    arm_sparse_rhs(op, f, c, op->stream);
    GroupTrace tf(op->stream, "forward");
    for (size_t gi = 0; gi < P.groups.size(); ++gi) { forward_group(op, f, gi, c); tf.mark(); }
    tf.report(P, false);
    GroupTrace tb(op->stream, "backward");
    for (size_t gk = P.groups.size(); gk-- > 0;) { backward_group(op, f, gk, c); tb.mark(); }
    tb.report(P, true);
    return check_kernels(op, "solve kernels");
}

// events of a factor + solve sweep, destroyed on every exit (an early error return must not leak the ones already created)
namespace {
struct EventSet {
    std::vector<hipEvent_t> plain, timed;
    bool create(size_t nplain, size_t ntimed) {
        for (size_t i = 0; i < nplain; ++i) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false; plain.push_back(e); }
        for (size_t i = 0; i < ntimed; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return false; timed.push_back(e); }
        return true;
    }
    ~EventSet() { for (hipEvent_t e : plain) hipEventDestroy(e); for (hipEvent_t e : timed) hipEventDestroy(e); }
};
}  // namespace

// Factorisation with the forward elimination of one batch running beside it: the forward pass of a tree level only needs that
// level's factors, so it follows the factorisation level by level on a second, LOW-priority stream.  The top of the tree is a chain of
// small dependent launches (80 block steps of a 16-workgroup panel kernel + one update each) that leaves most of the chip idle; the
// forward pass of the lower levels (big HBM-bound launches) fills it, and the priorities keep it from delaying the chain.
int nd_factor_solve_nm(helm_op *op, int block, NdFactor *f, cplx *ws_factor, const cplx *planes_in, const cplx *Qt, cplx *Xt, int nrhs, cplx *arenaV,
                       hipStream_t side, float *factor_ms, const NdDirectOut *dout) {
    const NdPlan &P = f->pd->plan;
    hipStream_t main = op->stream;
    const cplx *planes = nullptr;
    int rc = factor_prologue(op, block, f, planes_in, &planes);
    if (rc) return rc;
    SolveCtx c = solve_ctx(f, Xt, nrhs);
    c.Qt = Qt; c.Xt = Xt; c.arenaV = arenaV;
    if (dout && dout->U && P.dof == 1 && Qt != Xt) { c.Uout = dout->U; c.oscale = dout->oscale; }
    const size_t ng = P.groups.size();
    EventSet evs;
    if (!evs.create(ng + 2, 2)) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: hipEventCreate failed");
    std::vector<hipEvent_t> &ev = evs.plain;
    hipEvent_t t0 = evs.timed[0], t1 = evs.timed[1];
    hipEventRecord(ev[ng], main);                 // the right-hand sides were prepared on the main stream
    hipStreamWaitEvent(side, ev[ng], 0);
    arm_sparse_rhs(op, f, c, side);
    hipEventRecord(t0, main);
    for (size_t gi = 0; gi < ng && !rc; ++gi) {
        rc = factor_group(op, f, gi, ws_factor, ws_factor + 2 * P.fregion, planes);
        hipEventRecord(ev[gi], main);
        hipStreamWaitEvent(side, ev[gi], 0);
        op->stream = side;
        forward_group(op, f, gi, c);
        op->stream = main;
    }
    hipEventRecord(t1, main);
    hipEventRecord(ev[ng + 1], side);
    hipStreamWaitEvent(main, ev[ng + 1], 0);
    if (!rc) for (size_t gk = ng; gk-- > 0;) backward_group(op, f, gk, c);
    hipError_t e = hipStreamSynchronize(main);
    if (factor_ms) { float ms = 0.f; if (hipEventElapsedTime(&ms, t0, t1) == hipSuccess) *factor_ms = ms; }
    if (rc) return rc;
    if (e != hipSuccess) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: factor + solve failed: %s", hipGetErrorString(e));
    return check_kernels(op, "factor + solve kernels");
}
