// What do the fp64 units of one MI355X sustain, and at how many waves per SIMD?  (round 4: the round-3 version ran 1 and 2 waves per
// SIMD only and was latency-limited by its own occupancy -- its fp32 row reached 70 % of nominal.)
// Every probe kernel reads the shader-cycle counter (clock64: s_memtime) and the constant-rate wall clock (wall_clock64:
// s_memrealtime, hipDeviceAttributeWallClockRate kHz) at the start and end of a long run, so the effective shader clock DURING the
// kernel and the instructions issued per cycle per SIMD come out of the same launch.
// Occupancy is pinned, not hoped for: every workgroup is 256 threads (one wave on each SIMD of its CU) and declares 160 KiB / W of
// dynamic LDS, so exactly W workgroups fit on a CU; 256 * W workgroups are launched, i.e. every CU holds W waves per SIMD for the whole
// run (W = 1, 2, 4, 8; the register budgets of the kernels allow it: __launch_bounds__(256, W)).
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_clock.hip -o tools/fp64_clock && tools/fp64_clock [scale] [waves-list, e.g. 1,2,4,8]
//   counters: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -- tools/fp64_clock 0.1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

struct Stamp { long long c0, c1, w0, w1; };

static double g_wall_khz = 100000.0;
static int g_ncu = 256;

static size_t lds_for(int waves) { return (size_t)(160 * 1024 / waves) - (waves > 1 ? 512 : 0); }

// flops: per thread and iteration; wave_instr_iter: arithmetic instructions per wave and iteration; nominal_cyc: cycles one of them holds its pipe at the nominal rate
template <class K, class... A>
void report(const char *name, K kern, int waves, int iters, double flops_thread_iter, double wave_instr_iter, double nominal_cyc, Stamp *d_st, A... args) {
    const int nblocks = g_ncu * waves;
    const size_t lds = lds_for(waves);
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, 0, args..., d_st, iters / 50 + 1, 1.000001, 1e-9);   // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, 0, args..., d_st, iters, 1.000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) { printf("%-40s %d wave/SIMD  launch failed\n", name, waves); return; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(nblocks);
    hipMemcpy(h.data(), d_st, nblocks * sizeof(Stamp), hipMemcpyDeviceToHost);
    double mhz = 0, mhz_min = 1e30, mhz_max = 0, cyc = 0;
    long long first = h[0].w0, last_start = h[0].w0;
    for (int b = 0; b < nblocks; ++b) {
        const double dc = (double)(h[b].c1 - h[b].c0), dw = (double)(h[b].w1 - h[b].w0);
        const double f = dc / (dw / (g_wall_khz * 1e3)) / 1e6;
        mhz += f; if (f < mhz_min) mhz_min = f; if (f > mhz_max) mhz_max = f; cyc += dc;
        if (h[b].w0 < first) first = h[b].w0;
        if (h[b].w0 > last_start) last_start = h[b].w0;
    }
    mhz /= nblocks; cyc /= nblocks;
    const double tf = (double)nblocks * 256 * iters * flops_thread_iter / ms / 1e9;
    // cycles the SIMD spends per issued instruction (its W resident waves together); all workgroups resident at once when the last one
    // starts within a sliver of the run (start skew printed)
    const double cyc_per_instr = cyc / ((double)iters * wave_instr_iter * waves);
    printf("%-40s %d wave/SIMD (occupancy query %d)  %7.1f ms  %6.1f TFLOP/s  sclk %4.0f MHz (min %4.0f max %4.0f)  %5.2f cycles per instruction (nominal %g)  start skew %.2f %% of the run\n",
           name, waves, occ, ms, tf, mhz, mhz_min, mhz_max, cyc_per_instr, nominal_cyc,
           100.0 * (double)(last_start - first) / (g_wall_khz * 1e3) / (ms * 1e-3));
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

// the probe kernels; arguments: (out, [in,] stamps, iters, a0, b0)
// w_cblock: the tile kernel's mix -- RM x RN complex register block, 4 RM RN FMAs on RM + RN operand pairs and RM RN accumulator pairs per step
//   (4 x 4: 64 FMAs, ~100 VGPRs -> up to 4 waves per SIMD; 2 x 4: 32 FMAs on 16 accumulator chains, <= 64 VGPRs -> 8 waves)
// w_fma / w_fma32: NACC independent chains acc = a * acc + b with two loop-invariant operands
// w_mfma: v_mfma_f64_16x16x4_f64 on NACC independent accumulators, loop-invariant operands
// w_mfma_cblock: the complex product as an MFMA tile kernel would issue it: a 32 x 32 complex output block per wave = 2 x 2 MFMA tiles
//   x (re, im); per k step of 4, two A fragments (re, im) x two B fragments (re, im) -> 16 MFMAs on 8 operand registers (rotated)
template <int RM, int RN, int W> __global__ __launch_bounds__(256, W) void w_cblock(double *out, const double *in, Stamp *st, int it, double, double) {
    double ax[RM], ay[RM], bx[RN], by[RN], cx[RM * RN], cy[RM * RN];
    for (int i = 0; i < RM; ++i) { ax[i] = in[threadIdx.x + 256 * i]; ay[i] = in[threadIdx.x + 256 * (4 + i)]; }
    for (int i = 0; i < RN; ++i) { bx[i] = in[threadIdx.x + 256 * (8 + i)]; by[i] = in[threadIdx.x + 256 * (12 + i)]; }
    for (int i = 0; i < RM * RN; ++i) { cx[i] = 0; cy[i] = 0; }
    Stamp s0; s0.c0 = clock64(); s0.w0 = wall_clock64();
    for (int k = 0; k < it; ++k) {
        #pragma unroll
        for (int i = 0; i < RM; ++i)
            #pragma unroll
            for (int j = 0; j < RN; ++j) {
                cx[RN * i + j] = fma(ax[i], bx[j], cx[RN * i + j]); cx[RN * i + j] = fma(-ay[i], by[j], cx[RN * i + j]);
                cy[RN * i + j] = fma(ax[i], by[j], cy[RN * i + j]); cy[RN * i + j] = fma(ay[i], bx[j], cy[RN * i + j]);
            }
        double t = ax[0];
        #pragma unroll
        for (int i = 0; i + 1 < RM; ++i) ax[i] = ax[i + 1];
        ax[RM - 1] = t;
        t = by[0];
        #pragma unroll
        for (int i = 0; i + 1 < RN; ++i) by[i] = by[i + 1];
        by[RN - 1] = t;
    }
    s0.c1 = clock64(); s0.w1 = wall_clock64();
    if (threadIdx.x == 0) st[blockIdx.x] = s0;
    double s = 0;
    for (int i = 0; i < RM * RN; ++i) s += cx[i] + cy[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int W> __global__ __launch_bounds__(256, W) void w_fma(double *out, Stamp *st, int it, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    Stamp s0; s0.c0 = clock64(); s0.w0 = wall_clock64();
    for (int k = 0; k < it; ++k) {
        #pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(a, acc[i], b);
    }
    s0.c1 = clock64(); s0.w1 = wall_clock64();
    if (threadIdx.x == 0) st[blockIdx.x] = s0;
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int W> __global__ __launch_bounds__(256, W) void w_fma32(double *out, Stamp *st, int it, double a0d, double b0d) {
    float acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    float a = (float)a0d + threadIdx.x * 1e-7f, b = (float)b0d;
    Stamp s0; s0.c0 = clock64(); s0.w0 = wall_clock64();
    for (int k = 0; k < it; ++k) {
        #pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fmaf(a, acc[i], b);
    }
    s0.c1 = clock64(); s0.w1 = wall_clock64();
    if (threadIdx.x == 0) st[blockIdx.x] = s0;
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int W> __global__ __launch_bounds__(256, W) void w_mfma(double *out, Stamp *st, int it, double a0, double b0) {
    v4f64 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    Stamp s0; s0.c0 = clock64(); s0.w0 = wall_clock64();
    for (int k = 0; k < it; ++k) {
        #pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    s0.c1 = clock64(); s0.w1 = wall_clock64();
    if (threadIdx.x == 0) st[blockIdx.x] = s0;
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int W> __global__ __launch_bounds__(256, W) void w_mfma_cblock(double *out, const double *in, Stamp *st, int it, double, double) {
    double ar[2], ai[2], br[2], bi[2];
    v4f64 cr[4], ci[4];
    for (int i = 0; i < 2; ++i) { ar[i] = in[threadIdx.x + 256 * i]; ai[i] = in[threadIdx.x + 256 * (4 + i)]; br[i] = in[threadIdx.x + 256 * (8 + i)]; bi[i] = in[threadIdx.x + 256 * (12 + i)]; }
    for (int i = 0; i < 4; ++i) { cr[i] = (v4f64){0, 0, 0, 0}; ci[i] = (v4f64){0, 0, 0, 0}; }
    Stamp s0; s0.c0 = clock64(); s0.w0 = wall_clock64();
    for (int k = 0; k < it; ++k) {
        #pragma unroll
        for (int i = 0; i < 2; ++i)
            #pragma unroll
            for (int j = 0; j < 2; ++j) {
                cr[2 * i + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[i], br[j], cr[2 * i + j], 0, 0, 0);
                ci[2 * i + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[i], bi[j], ci[2 * i + j], 0, 0, 0);
            }
        #pragma unroll
        for (int i = 0; i < 2; ++i)
            #pragma unroll
            for (int j = 0; j < 2; ++j) {
                cr[2 * i + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai[i], bi[j], cr[2 * i + j], 0, 0, 0);
                ci[2 * i + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[i], br[j], ci[2 * i + j], 0, 0, 0);
            }
        double t = ar[0]; ar[0] = ar[1]; ar[1] = t;
        t = bi[0]; bi[0] = bi[1]; bi[1] = t;
    }
    s0.c1 = clock64(); s0.w1 = wall_clock64();
    if (threadIdx.x == 0) st[blockIdx.x] = s0;
    double s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) s += cr[i][e] + ci[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int W>
void run_all(double scale, double *d, const double *din, Stamp *d_st) {
    const int it = (int)(6000000 * scale / W);
    // fp64 vector FMA: 4 cycles per wave instruction at the nominal 16 lanes per clock and SIMD; fp32: 2
    report("vector fma f64, 16 chains", w_fma<16, W>, W, it, 16 * 2.0, 16.0, 4.0, d_st, d);
    if (W <= 4) report("complex 4x4 block (tile-kernel mix)", w_cblock<4, 4, (W <= 4 ? W : 4)>, W, it / 4, 128.0, 64.0, 4.0, d_st, d, din);
    if (W <= 4) report("complex 2x4 block (16 chains)", w_cblock<2, 4, (W <= 4 ? W : 4)>, W, it / 2, 64.0, 32.0, 4.0, d_st, d, din);
    report("complex 2x2 block (8 chains)", w_cblock<2, 2, W>, W, it, 32.0, 16.0, 4.0, d_st, d, din);
    // one MFMA 16x16x4 = 2048 flops per wave = 32 per thread; nominal 64 cycles (32 flop per clock and SIMD)
    if (W <= 4) report("mfma f64 16x16x4, 8 accumulators", w_mfma<8, (W <= 4 ? W : 4)>, W, it / 8, 8 * 32.0, 8.0, 64.0, d_st, d);
    if (W <= 4) report("mfma f64 16x16x4, 4 accumulators", w_mfma<4, (W <= 4 ? W : 4)>, W, it / 4, 4 * 32.0, 4.0, 64.0, d_st, d);
    report("mfma f64 16x16x4, 2 accumulators", w_mfma<2, W>, W, it / 2, 2 * 32.0, 2.0, 64.0, d_st, d);
    if (W <= 4) report("mfma f64 complex 32x32 block (16 mfma)", w_mfma_cblock<(W <= 4 ? W : 4)>, W, it / 16, 16 * 32.0, 16.0, 64.0, d_st, d, din);
    report("vector fma f32, 16 chains", w_fma32<16, W>, W, it, 16 * 2.0, 16.0, 2.0, d_st, d);
}

int main(int argc, char **argv) {
    const double scale = argc > 1 ? atof(argv[1]) : 1.0;
    const char *wl = argc > 2 ? argv[2] : "1,2,4,8";
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0) == hipSuccess && khz > 0) g_wall_khz = khz;
    int sclk_khz = 0; hipDeviceGetAttribute(&sclk_khz, hipDeviceAttributeClockRate, 0);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); g_ncu = prop.multiProcessorCount;
    printf("wall clock rate %.0f kHz; hipDeviceAttributeClockRate %d kHz; %d CUs; nominal fp64 at 2.4 GHz: %.1f TFLOP/s\n", g_wall_khz, sclk_khz, g_ncu, g_ncu * 4 * 32 * 2.4e9 / 1e12);
    double *d; hipMalloc(&d, 256 * 4096 * 8);
    Stamp *d_st; hipMalloc(&d_st, 4096 * sizeof(Stamp));
    double *din; hipMalloc(&din, 256 * 16 * 8);
    { double h[256 * 16]; for (int i = 0; i < 256 * 16; ++i) h[i] = 1e-3 * ((i * 7919) % 1000) - 0.5; hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice); }
    if (strstr(wl, "1")) run_all<1>(scale, d, din, d_st);
    if (strstr(wl, "2")) run_all<2>(scale, d, din, d_st);
    if (strstr(wl, "4")) run_all<4>(scale, d, din, d_st);
    if (strstr(wl, "8")) run_all<8>(scale, d, din, d_st);
    return 0;
}
