// Is the ~48 TFLOP/s the complex tile kernel's instruction mix sustains on MI355X a CLOCK limit (power management under fp64 load)
// or an ISSUE limit (operand delivery)?  Every probe kernel reads the shader-cycle counter (clock64: s_memtime) and the constant-rate
// wall clock (wall_clock64: s_memrealtime, hipDeviceAttributeWallClockRate kHz) at the start and end of a long run (>= 0.3 s), so the
// effective shader clock DURING the kernel and the fp64 FMAs issued per cycle per SIMD come out of the same launch.
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_clock.hip -o /tmp/fp64_clock && /tmp/fp64_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

struct Stamp { long long c0, c1, w0, w1; };

#define STAMP_BEGIN Stamp st; st.c0 = clock64(); st.w0 = wall_clock64();
#define STAMP_END(out) st.c1 = clock64(); st.w1 = wall_clock64(); if (threadIdx.x == 0) (out)[blockIdx.x] = st;

// (a) dependent-free chain with two loop-invariant operands: acc = a * acc + b
template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double *out, Stamp *stamps, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(a, acc[i], b);
    }
    STAMP_END(stamps)
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// (b) the tile kernel's mix: 4 x 4 complex register block, 64 FMAs on 8 operand pairs and 32 accumulators per step
__global__ __launch_bounds__(256) void k_cblock(double *out, Stamp *stamps, int iters, const double *in) {
    double ax[4], ay[4], bx[4], by[4], cx[16], cy[16];
    for (int i = 0; i < 4; ++i) { ax[i] = in[threadIdx.x + 256 * i]; ay[i] = in[threadIdx.x + 256 * (4 + i)]; bx[i] = in[threadIdx.x + 256 * (8 + i)]; by[i] = in[threadIdx.x + 256 * (12 + i)]; }
    for (int i = 0; i < 16; ++i) { cx[i] = 0; cy[i] = 0; }
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 4; ++i)
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                cx[4 * i + j] = fma(ax[i], bx[j], cx[4 * i + j]); cx[4 * i + j] = fma(-ay[i], by[j], cx[4 * i + j]);
                cy[4 * i + j] = fma(ax[i], by[j], cy[4 * i + j]); cy[4 * i + j] = fma(ay[i], bx[j], cy[4 * i + j]);
            }
        double t = ax[0]; ax[0] = ax[1]; ax[1] = ax[2]; ax[2] = ax[3]; ax[3] = t;
        t = by[0]; by[0] = by[1]; by[1] = by[2]; by[2] = by[3]; by[3] = t;
    }
    STAMP_END(stamps)
    double s = 0;
    for (int i = 0; i < 16; ++i) s += cx[i] + cy[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// (c) real 8 x 4 outer product: 32 FMAs on 12 operands, 32 accumulators (every FMA has three distinct, changing sources like (b),
// but no negated operand and twice the accumulators per operand)
__global__ __launch_bounds__(256) void k_rblock(double *out, Stamp *stamps, int iters, const double *in) {
    double a[8], b[4], c[32];
    for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x + 256 * i];
    for (int i = 0; i < 4; ++i) b[i] = in[threadIdx.x + 256 * (8 + i)];
    for (int i = 0; i < 32; ++i) c[i] = 0;
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 8; ++i)
            #pragma unroll
            for (int j = 0; j < 4; ++j) c[4 * i + j] = fma(a[i], b[j], c[4 * i + j]);
        double t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = a[3]; a[3] = t;
        t = b[0]; b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = t;
    }
    STAMP_END(stamps)
    double s = 0;
    for (int i = 0; i < 32; ++i) s += c[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// (d) fp64 MFMA 16x16x4, 8 independent accumulators
__global__ __launch_bounds__(256) void k_mfma(double *out, Stamp *stamps, int iters, double a0, double b0) {
    v4f64 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4f64){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    STAMP_END(stamps)
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// (e) fp32 FMA chain for comparison (same structure as (a)): does the clock drop only under fp64?
__global__ __launch_bounds__(256) void k_fma32(double *out, Stamp *stamps, int iters, float a0, float b0) {
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    float a = a0 + threadIdx.x * 1e-7f, b = b0;
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(a, acc[i], b);
    }
    STAMP_END(stamps)
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double g_wall_khz = 100000.0;

// flops: per thread and iteration; fmas: fp64 FMA (or MFMA-equivalent scalar FMA) instructions per wave and iteration
template <class L>
void report(const char *name, int nblocks, int iters, double flops_thread_iter, double wave_instr_iter, Stamp *d_st, L launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(iters / 50 + 1);                        // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(nblocks);
    hipMemcpy(h.data(), d_st, nblocks * sizeof(Stamp), hipMemcpyDeviceToHost);
    double mhz = 0, mhz_min = 1e30, mhz_max = 0, cyc = 0;
    for (int b = 0; b < nblocks; ++b) {
        const double dc = (double)(h[b].c1 - h[b].c0), dw = (double)(h[b].w1 - h[b].w0);
        const double f = dc / (dw / (g_wall_khz * 1e3)) / 1e6;
        mhz += f; if (f < mhz_min) mhz_min = f; if (f > mhz_max) mhz_max = f; cyc += dc;
    }
    mhz /= nblocks; cyc /= nblocks;
    const double tf = (double)nblocks * 256 * iters * flops_thread_iter / ms / 1e9;
    const int waves_per_simd = (nblocks + 255) / 256;                // one 256-thread workgroup = one wave on each SIMD of its CU
    // cycles the SIMD spends per issued instruction (all its resident waves together)
    const double cyc_per_instr = cyc / ((double)iters * wave_instr_iter * waves_per_simd);
    printf("%-44s %2d wave/SIMD  %7.1f ms  %6.1f TFLOP/s  sclk %6.0f MHz (min %4.0f max %4.0f)  %.2f cycles per wave-instruction  -> at this clock the nominal rate would be %.1f TFLOP/s\n",
           name, waves_per_simd, ms, tf, mhz, mhz_min, mhz_max, cyc_per_instr, 256.0 * 4 * 16 * 2 * mhz * 1e6 / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0) == hipSuccess && khz > 0) g_wall_khz = khz;
    int sclk_khz = 0; hipDeviceGetAttribute(&sclk_khz, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %.0f kHz; hipDeviceAttributeClockRate %d kHz\n", g_wall_khz, sclk_khz);
    double *d; hipMalloc(&d, 256 * 4096 * 8);
    Stamp *d_st; hipMalloc(&d_st, 4096 * sizeof(Stamp));
    double *din; hipMalloc(&din, 256 * 16 * 8);
    { double h[256 * 16]; for (int i = 0; i < 256 * 16; ++i) h[i] = 1e-3 * ((i * 7919) % 1000) - 0.5; hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice); }
    for (int waves = 1; waves <= 2; ++waves) {
        const int nb = 256 * waves;
        const int it = 12000000 / waves;
        report("vector fma f64 chain (2 invariant operands)", nb, it, 16 * 2.0, 16.0, d_st, [&](int n) { hipLaunchKernelGGL(k_fma<16>, dim3(nb), dim3(256), 0, 0, d, d_st, n, 1.000001, 1e-9); });
        report("complex 4x4 block (tile-kernel mix)", nb, it / 4, 128.0, 64.0, d_st, [&](int n) { hipLaunchKernelGGL(k_cblock, dim3(nb), dim3(256), 0, 0, d, d_st, n, din); });
        report("real 8x4 outer product", nb, it / 2, 64.0, 32.0, d_st, [&](int n) { hipLaunchKernelGGL(k_rblock, dim3(nb), dim3(256), 0, 0, d, d_st, n, din); });
        // one MFMA 16x16x4 = 2048 flops per wave = 32 per thread; at the nominal rate it occupies the pipe like 16 vector FMAs
        report("mfma f64 16x16x4 (8 accumulators)", nb, it / 8, 8 * 32.0, 8.0 * 16.0, d_st, [&](int n) { hipLaunchKernelGGL(k_mfma, dim3(nb), dim3(256), 0, 0, d, d_st, n, 1.0, 1.0); });
        report("vector fma f32 chain", nb, it, 16 * 2.0, 16.0, d_st, [&](int n) { hipLaunchKernelGGL(k_fma32, dim3(nb), dim3(256), 0, 0, d, d_st, n, 1.000001f, 1e-9f); });
    }
    return 0;
}
