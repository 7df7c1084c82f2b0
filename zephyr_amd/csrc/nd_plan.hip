// Direct solver, plan: the elimination tree of the (nz, nx) grid in closed form -- recursive bisection by one-cell separator lines, fronts of one
// tree level and kind padded to one shape -- its device tables, and the per-device plan cache.  (What SuperLU's symbolic phase does for the
// reference, zephyr/backend/discretization.py:78-103, is known in advance on a regular grid.)
#include "nd_internal.hpp"
#include <mutex>
#include <map>

namespace {

// ---- plan ------------------------------------------------------------------------------------------------------
struct Build {
    int nz, nx, leaf, dof;
    std::vector<NdDev> nodes;      // creation order
    std::vector<int> level;
    std::vector<int> parent;
};

void fill_geometry(NdDev &n, int nz, int nx, int dof) {
    const int h = n.z1 - n.z0, w = n.x1 - n.x0;
    n.dof = dof;
    n.s = dof * (n.cut < 0 ? h * w : (n.cut == 0 ? w : h));
    n.xlo = std::max(n.x0 - 1, 0);
    const int xhi = std::min(n.x1, nx - 1);
    const int wrow = xhi - n.xlo + 1;
    n.ntop = n.z0 > 0 ? wrow : 0;
    n.nbot = n.z1 < nz ? wrow : 0;
    n.nleft = n.x0 > 0 ? h : 0;
    n.nright = n.x1 < nx ? h : 0;
    n.m = dof * (n.ntop + n.nbot + n.nleft + n.nright);
}

int build_rec(Build &B, int z0, int z1, int x0, int x1, int lev, int parent) {
    NdDev n = NdDev();
    n.z0 = z0; n.z1 = z1; n.x0 = x0; n.x1 = x1; n.kid[0] = n.kid[1] = -1;
    const int h = z1 - z0, w = x1 - x0;
    const int me = (int)B.nodes.size();
    if (h <= B.leaf && w <= B.leaf) { n.cut = -1; n.pos = -1; }
    else if (h >= w) { n.cut = 0; n.pos = z0 + h / 2; }
    else { n.cut = 1; n.pos = x0 + w / 2; }
    fill_geometry(n, B.nz, B.nx, B.dof);
    B.nodes.push_back(n); B.level.push_back(lev); B.parent.push_back(parent);
    if (n.cut == 0) {
        int k = 0;
        if (n.pos > z0) { int c = build_rec(B, z0, n.pos, x0, x1, lev + 1, me); B.nodes[me].kid[k++] = c; }
        if (z1 > n.pos + 1) { int c = build_rec(B, n.pos + 1, z1, x0, x1, lev + 1, me); B.nodes[me].kid[k++] = c; }
    } else if (n.cut == 1) {
        int k = 0;
        if (n.pos > x0) { int c = build_rec(B, z0, z1, x0, n.pos, lev + 1, me); B.nodes[me].kid[k++] = c; }
        if (x1 > n.pos + 1) { int c = build_rec(B, z0, z1, n.pos + 1, x1, lev + 1, me); B.nodes[me].kid[k++] = c; }
    }
    return me;
}

}  // namespace

int nd_build_plan(NdPlan &P, int nz, int nx, int leaf, int dof) {
    Build B; B.nz = nz; B.nx = nx; B.leaf = std::max(2, leaf); B.dof = dof;
    build_rec(B, 0, nz, 0, nx, 0, -1);
    const int nn = (int)B.nodes.size();
    int maxlev = 0;
    for (int l : B.level) maxlev = std::max(maxlev, l);
    // processing order: deepest level first; within a level the leaves, then the separators.  Leaves come in two size classes per level:
    // on a 2^k grid with one-cell separators all but one leaf interval per axis are 7 cells long (1024 = 127 x 7 + 8 + 127 separators),
    // so 98 % of the leaves have 49 unknowns and a single padded shape of 64 would waste a quarter of the leaf level's inner dimension
    std::vector<int> order; order.reserve(nn);
    P.groups.clear();
    for (int lev = maxlev; lev >= 0; --lev) {
        // most frequent leaf size of this level
        int smode = 0;
        {
            std::vector<int> hist;
            for (int i = 0; i < nn; ++i)
                if (B.level[i] == lev && B.nodes[i].cut < 0) { if ((int)hist.size() <= B.nodes[i].s) hist.resize(B.nodes[i].s + 1, 0); hist[B.nodes[i].s] += 1; }
            for (int v = 0; v < (int)hist.size(); ++v) if (hist[v] > (smode < (int)hist.size() ? hist[smode] : 0)) smode = v;
        }
        for (int kind = 0; kind < 3; ++kind) {      // 0: leaves up to the usual size, 1: larger leaves, 2: separators
            NdGroup g = NdGroup(); g.first = (int)order.size(); g.level = lev; g.leaf = kind < 2;
            for (int i = 0; i < nn; ++i) {
                if (B.level[i] != lev) continue;
                const bool isleaf = B.nodes[i].cut < 0;
                const int k = !isleaf ? 2 : (B.nodes[i].s <= smode ? 0 : 1);
                if (k != kind) continue;
                order.push_back(i);
                g.smax = std::max(g.smax, B.nodes[i].s); g.mmax = std::max(g.mmax, B.nodes[i].m);
            }
            g.cnt = (int)order.size() - g.first;
            if (g.cnt > 0) P.groups.push_back(g);
        }
    }
    std::vector<int> newidx(nn);
    for (int k = 0; k < nn; ++k) newidx[order[k]] = k;
    P.nodes.resize(nn);
    for (int k = 0; k < nn; ++k) {
        NdDev n = B.nodes[order[k]];
        for (int c = 0; c < 2; ++c) if (n.kid[c] >= 0) n.kid[c] = newidx[n.kid[c]];
        P.nodes[k] = n;
    }
    P.nz = nz; P.nx = nx; P.leaf = B.leaf; P.nlevels = maxlev + 1; P.total_rows = 0; P.dof = dof;
    // arenas: fronts (factor) and front vectors (solve) of level L live in region L % 2
    std::vector<long long> lev_f(maxlev + 1, 0), lev_v(maxlev + 1, 0);
    long long fac = 0;
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        NdGroup &g = P.groups[gi];
        const long long nmax = g.smax + g.mmax;
        g.foff = lev_f[g.level]; g.voff = lev_v[g.level];
        g.roff = P.total_rows; P.total_rows += (long long)g.cnt * nmax;
        lev_f[g.level] += (long long)g.cnt * g.mmax * nmax;
        lev_v[g.level] += (long long)g.cnt * nmax;
        g.finv = fac; g.f12 = fac + g.smax; fac += (long long)g.cnt * g.smax * nmax;       // [F11^-1 | F12] rows of nmax per front
        g.g21 = fac; fac += (long long)g.cnt * g.mmax * g.smax;
    }
    P.fac_elems = fac;
    P.fregion = 0; P.vregion = 0; P.work_elems = 0;
    for (int l = 0; l <= maxlev; ++l) { P.fregion = std::max(P.fregion, lev_f[l]); P.vregion = std::max(P.vregion, lev_v[l]); }
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        NdGroup &g = P.groups[gi];
        g.foff += (long long)(g.level & 1) * P.fregion;
        g.voff += (long long)(g.level & 1) * P.vregion;
        // (the inversion's scratch; a leaf level with more than 64 unknowns per front also forms -F11^-1 F12 there: smax x mmax per front)
        P.work_elems = std::max(P.work_elems, (long long)g.cnt * g.smax * (g.leaf ? std::max(g.smax, g.mmax) : g.smax));
        for (int j = 0; j < g.cnt; ++j) {
            NdDev &n = P.nodes[g.first + j];
            const long long nmax = g.smax + g.mmax;
            n.smax = g.smax; n.mmax = g.mmax;
            n.foff = g.foff + (long long)j * g.mmax * nmax;
            n.finv_off = g.finv + (long long)j * g.smax * nmax;
            n.f12_off = n.finv_off + g.smax;
            n.voff = g.voff + (long long)j * nmax;
            n.roff = g.roff + (long long)j * nmax;
        }
    }
    return HELM_OK;
}
namespace {

// cellnode[cell] = front of the leaf that eliminates the cell (rows of a leaf group's table with the separator flag)
__global__ __launch_bounds__(256) void k_nd_cellnode(const int4 *tab, long long rows, int nmax, int first, int *cellnode) {
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
        const int4 e = tab[r];
        if (e.w && e.x >= 0) cellnode[e.x] = first + (int)(r / nmax);
    }
}

// row table (see NdPlanDev): one thread per padded row of the group's fronts
__global__ __launch_bounds__(256) void k_nd_build_tab(const NdDev *nodes, int first, int4 *tab, int nz, int nx) {
    const NdDev n = nodes[first + blockIdx.y];
    const int nmax = n.smax + n.mmax;
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < nmax; row += gridDim.x * blockDim.x) {
        int a = -1;
        if (row < n.s) a = row;
        else if (row >= n.smax && row - n.smax < n.m) a = n.s + row - n.smax;
        int4 e = make_int4(-1, -1, -1, row < n.s ? 1 : 0);
        if (a >= 0) {
            int z, x, comp;
            nd_cell(n, a, z, x, comp);
            e.x = comp * nz * nx + z * nx + x;            // row of Xt: the fields are stacked [u; v] like the right-hand sides
            for (int k = 0; k < 2; ++k) {
                if (n.kid[k] < 0) continue;
                const NdDev c = nodes[n.kid[k]];
                const int la = nd_local(c, nz, nx, z, x, comp);
                if (la >= c.s) { const int src = (int)(c.voff + c.smax + (la - c.s)); if (k == 0) e.y = src; else e.z = src; }
            }
        }
        tab[n.roff + row] = e;
    }
}

}  // namespace

// ---- plan cache ---------------------------------------------------------------------------------------------------
// Per device, most recently used last, HELM_ND_PLANS (default 6) kept alive per device by the cache (a factor keeps its own plan alive whatever the
// cache does).  r4: round 3 had ONE list of four for the whole process, searched and FILLED under one mutex: with the in-process dispatcher dealing
// operators over eight GPUs (or a 2-D plan beside the 3-D column-dissection plans) every new operator missed, and each miss rebuilt the host
// plan -- tens of milliseconds of recursion -- and ran hipMalloc / hipFree with the lock held, i.e. with every other GPU's prepare thread waiting.
// Now: look-up under the lock, build outside it (two threads that miss on the same key both build; the second finds the first's entry and
// drops its own), tables from the size-keyed device pool.
NdPlanDev::~NdPlanDev() {
    if (d_nodes) helm_pool_free(device, d_nodes, plan.nodes.size() * sizeof(NdDev));
    if (d_tab) helm_pool_free(device, d_tab, (size_t)plan.total_rows * sizeof(int4));
    if (d_cellnode) helm_pool_free(device, d_cellnode, (size_t)plan.nz * plan.nx * sizeof(int));
}

namespace {
std::mutex g_plan_mu;
// (never destroyed: a plan's destructor hands its tables to the device pool of capi.hip, which may be gone first when the process exits)
std::map<int, std::vector<std::shared_ptr<NdPlanDev>>> &g_plans = *new std::map<int, std::vector<std::shared_ptr<NdPlanDev>>>();

std::shared_ptr<NdPlanDev> plan_lookup(int device, int pnz, int pnx, int leaf, int dof) {      // (g_plan_mu held)
    std::vector<std::shared_ptr<NdPlanDev>> &L = g_plans[device];
    for (size_t i = 0; i < L.size(); ++i) {
        const NdPlanDev &c = *L[i];
        if (c.plan.nz == pnz && c.plan.nx == pnx && c.plan.leaf == std::max(2, leaf) && c.plan.dof == dof) {
            std::shared_ptr<NdPlanDev> hit = L[i];
            L.erase(L.begin() + i); L.push_back(hit);
            return hit;
        }
    }
    return nullptr;
}
}

int nd_get_plan(helm_op *op, int leaf, int dof, std::shared_ptr<NdPlanDev> *out) { return nd_get_plan_dims(op, op->nz, op->nx, leaf, dof, out); }

// the same for a grid that is not the handle's own: the 3-D coarse solve runs the 2-D dissection over (ny, nx) columns of nz unknowns (dof = nz)
int nd_get_plan_dims(helm_op *op, int pnz, int pnx, int leaf, int dof, std::shared_ptr<NdPlanDev> *out) {
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        if (std::shared_ptr<NdPlanDev> hit = plan_lookup(op->device, pnz, pnx, leaf, dof)) { *out = hit; return HELM_OK; }
    }
    std::shared_ptr<NdPlanDev> pd(new NdPlanDev());
    pd->device = op->device;
    nd_build_plan(pd->plan, pnz, pnx, leaf, dof);
    const NdPlan &P = pd->plan;
    if (2 * P.vregion >= (1LL << 31) || P.total_rows >= (1LL << 31)) HELM_FAIL(op, HELM_ERR_UNSUPPORTED, "direct solver: grid too large for 32-bit row indices");
    pd->d_nodes = (NdDev *)helm_pool_alloc(op->device, P.nodes.size() * sizeof(NdDev));
    pd->d_tab = (int4 *)helm_pool_alloc(op->device, (size_t)P.total_rows * sizeof(int4));
    if (!pd->d_nodes || !pd->d_tab) HELM_FAIL(op, HELM_ERR_DEVICE, "direct solver: allocation of the plan tables failed");
    HIP_TRY(op, hipMemcpyAsync(pd->d_nodes, P.nodes.data(), P.nodes.size() * sizeof(NdDev), hipMemcpyHostToDevice, op->stream));
    for (size_t gi = 0; gi < P.groups.size(); ++gi) {
        const NdGroup &g = P.groups[gi];
        const int nmax = g.smax + g.mmax;
        for (int j0 = 0; j0 < g.cnt; j0 += 65535) {
            const int nb = std::min(65535, g.cnt - j0);
            HELM_LAUNCH(k_nd_build_tab, dim3((nmax + 255) / 256, nb), dim3(256), 0, op->stream, pd->d_nodes, g.first + j0, pd->d_tab, P.nz, P.nx);
        }
    }
    if (dof == 1) {                                      // which leaf eliminates a cell (the residual's q mask on sparse right-hand sides)
        pd->d_cellnode = (int *)helm_pool_alloc(op->device, (size_t)pnz * pnx * sizeof(int));
        if (pd->d_cellnode) {
            HIP_TRY(op, hipMemsetAsync(pd->d_cellnode, 0xFF, (size_t)pnz * pnx * sizeof(int), op->stream));
            for (size_t gi = 0; gi < P.groups.size(); ++gi) {
                const NdGroup &g = P.groups[gi];
                if (!g.leaf) continue;
                const long long rows = (long long)g.cnt * (g.smax + g.mmax);
                HELM_LAUNCH(k_nd_cellnode, dim3((unsigned)std::min<long long>((rows + 255) / 256, 65535)), dim3(256), 0, op->stream,
                                   (const int4 *)(pd->d_tab + g.roff), rows, g.smax + g.mmax, g.first, pd->d_cellnode);
            }
        }
    }
    HIP_TRY(op, hipStreamSynchronize(op->stream));      // the tables are complete before another handle (another stream) can find them
    const int keep = helm_tuning_now().nd_plans;
    std::shared_ptr<NdPlanDev> evicted;                  // (destroyed after the lock is released)
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        if (std::shared_ptr<NdPlanDev> hit = plan_lookup(op->device, pnz, pnx, leaf, dof)) { *out = hit; return HELM_OK; }      // another thread was faster: ours goes back to the pool
        std::vector<std::shared_ptr<NdPlanDev>> &L = g_plans[op->device];
        L.push_back(pd);
        if ((int)L.size() > keep) { evicted = L.front(); L.erase(L.begin()); }
    }
    *out = pd;
    return HELM_OK;
}

// (tests) number of plans the cache holds for `device`
extern "C" int helm_debug_plan_cache(int device) {
    helm_tuning_refresh();
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plans.find(device);
    return it == g_plans.end() ? 0 : (int)it->second.size();
}

// ---- diagnostics exported through the C ABI (host side of the plan; dense kernels on small inputs) ---------------------
extern "C" int helm_direct_plan(int nz, int nx, int leaf, int *out, int cap) {
    helm_tuning_refresh();
    NdPlan P;
    nd_build_plan(P, nz, nx, leaf);
    const int nn = (int)P.nodes.size();
    if (!out) return nn;
    for (int i = 0; i < nn && i < cap; ++i) {
        const NdDev &n = P.nodes[i];
        int *o = out + 12 * i;
        o[0] = n.z0; o[1] = n.z1; o[2] = n.x0; o[3] = n.x1; o[4] = n.cut; o[5] = n.pos; o[6] = n.s; o[7] = n.m;
        o[8] = n.kid[0]; o[9] = n.kid[1]; o[10] = n.smax; o[11] = n.mmax;
    }
    return nn;
}

// cells (z * nx + x) of the front of node `node` in local order; returns s + m, or a negative value when the
// inverse map nd_local disagrees with nd_cell (self-check of the closed-form index maps)
extern "C" int helm_direct_plan_front(int nz, int nx, int leaf, int node, long long *cells, int cap) {
    helm_tuning_refresh();
    NdPlan P;
    nd_build_plan(P, nz, nx, leaf);
    if (node < 0 || node >= (int)P.nodes.size()) return HELM_ERR_ARG;
    const NdDev &n = P.nodes[node];
    for (int a = 0; a < n.s + n.m; ++a) {
        int z, x;
        nd_cell(n, a, z, x);
        if (z < 0 || z >= nz || x < 0 || x >= nx) return -100;
        if (nd_local(n, nz, nx, z, x) != a) return -101;
        if (cells && a < cap) cells[a] = (long long)z * nx + x;
    }
    return n.s + n.m;
}

