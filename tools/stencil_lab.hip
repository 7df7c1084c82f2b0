// Kernel lab (development tool, not part of the library): times variants of the batched 9-point
// complex128 stencil apply on the GPU box and checks them against a naive reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/stencil_lab tools/stencil_lab.hip && tools/stencil_lab [n] [B]
#include "../zephyr_amd/csrc/kernels.hip"
#include <vector>
#include <random>
#include <cstdio>

void helm_set_error(helm_op *, const char *) {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_ref(const cplx *planes, const cplx *X, cplx *Y, int nz, int nx, long long N, int nrhs) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    if (i >= N) return;
    int iz = i / nx, ix = i % nx;
    cplx acc = cmake(0, 0);
    for (int k = 0; k < 9; ++k) {
        int jz = iz + k / 3 - 1, jx = ix + k % 3 - 1;
        if (jz < 0 || jz >= nz || jx < 0 || jx >= nx) continue;
        cfma(acc, planes[(long long)k * N + i], X[(long long)b * N + (long long)jz * nx + jx]);
    }
    Y[(long long)b * N + i] = acc;
}

__global__ void k_copy(const cplx *__restrict__ a, cplx *__restrict__ b, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) b[i] = a[i];
}

// ---- register-marching variant ---------------------------------------------------------------------
// One wave owns a 64-column strip and marches down ZC rows for BT right-hand sides, keeping a 3-row
// window (centre value + halo value per lane) in registers; left/right neighbours come from lane
// shuffles, the two halo columns from lanes 0 / 63.  The 4 waves of a workgroup share (strip, chunk)
// and take different right-hand-side groups, so the coefficient rows are fetched from HBM once.
__device__ inline cplx shfl_up_c(cplx v) { cplx r; r.x = __shfl_up(v.x, 1, 64); r.y = __shfl_up(v.y, 1, 64); return r; }
__device__ inline cplx shfl_dn_c(cplx v) { cplx r; r.x = __shfl_down(v.x, 1, 64); r.y = __shfl_down(v.y, 1, 64); return r; }

template <int BT, int GROUPS>   // GROUPS rhs-groups per block (1, 2 or 4); 4/GROUPS z-chunks per block
__global__ __launch_bounds__(256) void k_march(const cplx *__restrict__ planes, const cplx *__restrict__ X, cplx *__restrict__ Y,
                                               int nz, int nx, long long N, int nrhs, int ZC, int nstrips, int nblk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int CHUNKS = 4 / GROUPS;
    const int t = xcd_swizzle(blockIdx.x, nblk);
    const int strip = t % nstrips, cb = t / nstrips;
    const int chunk = cb * CHUNKS + wave / GROUPS;
    const int grp = blockIdx.y * GROUPS + wave % GROUPS;
    const int b0 = grp * BT;
    if (b0 >= nrhs) return;
    const int z0 = chunk * ZC;
    if (z0 >= nz) return;
    const int z1 = min(z0 + ZC, nz);
    const int col = strip * 64 + lane;
    const bool colok = col < nx;
    const int hcol = lane == 0 ? col - 1 : (lane == 63 ? col + 1 : -1);
    const bool hok = hcol >= 0 && hcol < nx;
    const cplx zero = cmake(0.0, 0.0);

    cplx wc[3][BT], wh[3][BT];
    auto load_row = [&](int r, cplx (&c)[BT], cplx (&h)[BT]) {
        const bool rok = r >= 0 && r < nz;
#pragma unroll
        for (int j = 0; j < BT; ++j) {
            const bool bok = b0 + j < nrhs;
            const cplx *xb = X + (long long)(b0 + j) * N + (long long)r * nx;
            c[j] = (rok && colok && bok) ? xb[col] : zero;
            h[j] = (rok && hok && bok) ? xb[hcol] : zero;
        }
    };
    load_row(z0 - 1, wc[0], wh[0]);
    load_row(z0, wc[1], wh[1]);
    load_row(z0 + 1, wc[2], wh[2]);
    cplx cf[9], cfn[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cf[k] = colok ? planes[(long long)k * N + (long long)z0 * nx + col] : zero;

    for (int z = z0; z < z1; ++z) {
        cplx nc[BT], nh[BT];
        load_row(z + 2, nc, nh);                       // prefetch
        const bool nok = (z + 1 < z1) && colok;
#pragma unroll
        for (int k = 0; k < 9; ++k) cfn[k] = nok ? planes[(long long)k * N + (long long)(z + 1) * nx + col] : zero;
#pragma unroll
        for (int j = 0; j < BT; ++j) {
            cplx acc = zero;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const cplx c = wc[r][j];
                cplx l = shfl_up_c(c), rr = shfl_dn_c(c);
                if (lane == 0) l = wh[r][j];
                if (lane == 63) rr = wh[r][j];
                cfma(acc, cf[r * 3 + 0], l);
                cfma(acc, cf[r * 3 + 1], c);
                cfma(acc, cf[r * 3 + 2], rr);
            }
            if (colok && b0 + j < nrhs) Y[(long long)(b0 + j) * N + (long long)z * nx + col] = acc;
        }
#pragma unroll
        for (int j = 0; j < BT; ++j) { wc[0][j] = wc[1][j]; wc[1][j] = wc[2][j]; wc[2][j] = nc[j]; wh[0][j] = wh[1][j]; wh[1][j] = wh[2][j]; wh[2][j] = nh[j]; }
#pragma unroll
        for (int k = 0; k < 9; ++k) cf[k] = cfn[k];
    }
}

template <typename F>
double time_it(F f, int reps = 20) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

double max_err(const std::vector<cplx> &a, const std::vector<cplx> &b) {
    double e = 0, s = 0;
    for (size_t i = 0; i < a.size(); ++i) { e = fmax(e, hypot(a[i].x - b[i].x, a[i].y - b[i].y)); s = fmax(s, hypot(b[i].x, b[i].y)); }
    return e / s;
}

template <int P>
void run_tile(const char *name, const cplx *dC, const cplx *dX, cplx *dY, int n, long long N, int B, const std::vector<cplx> &ref, double bytes, int tiled = 0) {
    StencilParams q;
    q.planes_tiled = tiled; q.U = nullptr; q.E = nullptr; q.nzc = q.nxc = 0; q.acc = 0; q.part_stride = 1; q.part_off = 0;
    q.planes = dC; q.X = dX; q.Y = dY; q.W = nullptr; q.ld = N; q.N = N; q.nz = n; q.nx = n; q.nrhs = B;
    q.ntx = (n + 63) / 64; q.ntz = (n + 4 * P - 1) / (4 * P); q.nblk = q.ntx * q.ntz; q.scal = nullptr; q.part = nullptr; q.dinv = nullptr; q.omega_j = 0; q.tiles = nullptr;
    int split = 1;
    if (q.nblk < 1024) { split = (1024 + q.nblk - 1) / q.nblk; if (split > B) split = B; }
    dim3 grid(q.nblk, split);
    CK(hipMemset(dY, 0, (size_t)B * N * sizeof(cplx)));
    double ms = time_it([&] { hipLaunchKernelGGL((k_stencil_t<cplx, P, false, false, EPI_NONE>), grid, dim3(256), 0, 0, q); });
    std::vector<cplx> out((size_t)B * N);
    CK(hipMemcpy(out.data(), dY, out.size() * sizeof(cplx), hipMemcpyDeviceToHost));
    printf("%-28s %8.1f us  %7.1f GB/s alg  err %.1e\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e9, max_err(out, ref));
}

template <int BT, int GROUPS>
void run_march(const char *name, int ZC, const cplx *dC, const cplx *dX, cplx *dY, int n, long long N, int B, const std::vector<cplx> &ref, double bytes) {
    const int nstrips = (n + 63) / 64, nchunks = (n + ZC - 1) / ZC;
    constexpr int CHUNKS = 4 / GROUPS;
    const int nblk = nstrips * ((nchunks + CHUNKS - 1) / CHUNKS);
    const int ngrp = (B + BT - 1) / BT;
    dim3 grid(nblk, (ngrp + GROUPS - 1) / GROUPS);
    CK(hipMemset(dY, 0, (size_t)B * N * sizeof(cplx)));
    double ms = time_it([&] { hipLaunchKernelGGL((k_march<BT, GROUPS>), grid, dim3(256), 0, 0, dC, dX, dY, n, n, N, B, ZC, nstrips, nblk); });
    std::vector<cplx> out((size_t)B * N);
    CK(hipMemcpy(out.data(), dY, out.size() * sizeof(cplx), hipMemcpyDeviceToHost));
    char nm[96]; snprintf(nm, sizeof(nm), "%s ZC=%d", name, ZC);
    printf("%-28s %8.1f us  %7.1f GB/s alg  err %.1e\n", nm, ms * 1e3, bytes / (ms * 1e-3) / 1e9, max_err(out, ref));
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    const int B = argc > 2 ? atoi(argv[2]) : 8;
    const long long N = (long long)n * n;
    std::mt19937_64 rng(1234);
    std::normal_distribution<double> nd;
    std::vector<cplx> hC((size_t)9 * N), hX((size_t)B * N);
    for (auto &v : hC) v = cmake(nd(rng), nd(rng));
    for (auto &v : hX) v = cmake(nd(rng), nd(rng));
    cplx *dC, *dX, *dY;
    CK(hipMalloc(&dC, hC.size() * sizeof(cplx))); CK(hipMalloc(&dX, hX.size() * sizeof(cplx))); CK(hipMalloc(&dY, hX.size() * sizeof(cplx)));
    CK(hipMemcpy(dC, hC.data(), hC.size() * sizeof(cplx), hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, hX.data(), hX.size() * sizeof(cplx), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ref, dim3((N + 255) / 256, B), dim3(256), 0, 0, dC, dX, dY, n, n, N, B);
    std::vector<cplx> ref((size_t)B * N);
    CK(hipMemcpy(ref.data(), dY, ref.size() * sizeof(cplx), hipMemcpyDeviceToHost));
    const double bytes = (double)N * (32.0 * B + 144.0);
    printf("n=%d B=%d  algorithmic bytes per launch %.1f MB\n", n, B, bytes / 1e6);
    {
        const long long cn = (long long)B * N;
        double ms = time_it([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const cplx *)dX, dY, cn); });
        printf("%-28s %8.1f us  %7.1f GB/s (read+write)\n", "copy 16B/lane", ms * 1e3, 2.0 * cn * 16 / (ms * 1e-3) / 1e9);
    }
    run_tile<1>("lds-tile P=1", dC, dX, dY, n, N, B, ref, bytes);
    run_tile<2>("lds-tile P=2 (current)", dC, dX, dY, n, N, B, ref, bytes);
    run_tile<4>("lds-tile P=4", dC, dX, dY, n, N, B, ref, bytes);
    {   // tile-blocked coefficient layout, P=2 (64 x 8 tiles)
        const int ntx = (n + 63) / 64, ntz = (n + 7) / 8;
        std::vector<cplx> hT((size_t)ntx * ntz * 9 * 512, cmake(0, 0));
        for (int tz = 0; tz < ntz; ++tz) for (int tx = 0; tx < ntx; ++tx) for (int k = 0; k < 9; ++k) for (int r = 0; r < 8; ++r) for (int l = 0; l < 64; ++l) {
            const int row = tz * 8 + r, col = tx * 64 + l;
            if (row < n && col < n) hT[(((size_t)(tz * ntx + tx) * 9 + k) * 8 + r) * 64 + l] = hC[(size_t)k * N + (size_t)row * n + col];
        }
        cplx *dT; CK(hipMalloc(&dT, hT.size() * sizeof(cplx)));
        CK(hipMemcpy(dT, hT.data(), hT.size() * sizeof(cplx), hipMemcpyHostToDevice));
        run_tile<2>("lds-tile P=2 tiled planes", dT, dX, dY, n, N, B, ref, bytes, 1);
        CK(hipFree(dT));
    }
    for (int ZC : {16}) {
        run_march<2, 4>("march BT=2 G=4", ZC, dC, dX, dY, n, N, B, ref, bytes);
        run_march<4, 2>("march BT=4 G=2", ZC, dC, dX, dY, n, N, B, ref, bytes);
        run_march<4, 1>("march BT=4 G=1", ZC, dC, dX, dY, n, N, B, ref, bytes);
        run_march<1, 4>("march BT=1 G=4", ZC, dC, dX, dY, n, N, B, ref, bytes);
    }
    return 0;
}
