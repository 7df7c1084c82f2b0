// fp64 issue-rate probe for MI355X: register-only loops of v_mfma_f64_16x16x4_f64 and of v_fma_f64.
// Measured (hipcc -O3 --offload-arch=gfx950, one MI355X): MFMA 44-48 TFLOP/s, vector FMA 65-67 TFLOP/s -- the reason the
// dense kernels of zephyr_amd/csrc/direct.hip run on the vector ALUs.   Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_rate.hip -o fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double a0, double b0) {
    v4f64 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double *out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(a, acc[i], b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// complex 4 x 4 register block exactly as the tile kernel's inner loop issues it (64 fp64 FMAs on 8 operand pairs and 32 accumulator
// pairs), operands re-used from registers: what the vector ALUs give that instruction mix with no LDS or memory traffic at all
__global__ __launch_bounds__(256) void k_cblock(double *out, int iters, const double *in) {
    double ax[4], ay[4], bx[4], by[4], cx[16], cy[16];
    for (int i = 0; i < 4; ++i) { ax[i] = in[threadIdx.x + 256 * i]; ay[i] = in[threadIdx.x + 256 * (4 + i)]; bx[i] = in[threadIdx.x + 256 * (8 + i)]; by[i] = in[threadIdx.x + 256 * (12 + i)]; }
    for (int i = 0; i < 16; ++i) { cx[i] = 0; cy[i] = 0; }
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 4; ++i)
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                cx[4 * i + j] = fma(ax[i], bx[j], cx[4 * i + j]); cx[4 * i + j] = fma(-ay[i], by[j], cx[4 * i + j]);
                cy[4 * i + j] = fma(ax[i], by[j], cy[4 * i + j]); cy[4 * i + j] = fma(ay[i], bx[j], cy[4 * i + j]);
            }
        // keep the operands live and changing without adding arithmetic: rotate them through the block
        double t = ax[0]; ax[0] = ax[1]; ax[1] = ax[2]; ax[2] = ax[3]; ax[3] = t;
        t = by[0]; by[0] = by[1]; by[1] = by[2]; by[2] = by[3]; by[3] = t;
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += cx[i] + cy[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same register block with NM fp64 MFMAs (independent accumulators) in every iteration: do the matrix pipe and the vector ALUs
// of one wave overlap for fp64?  (flops counted for both)
template <int NM>
__global__ __launch_bounds__(256) void k_cblock_mfma(double *out, int iters, const double *in) {
    double ax[4], ay[4], bx[4], by[4], cx[16], cy[16];
    v4f64 m[NM > 0 ? NM : 1];
    for (int i = 0; i < 4; ++i) { ax[i] = in[threadIdx.x + 256 * i]; ay[i] = in[threadIdx.x + 256 * (4 + i)]; bx[i] = in[threadIdx.x + 256 * (8 + i)]; by[i] = in[threadIdx.x + 256 * (12 + i)]; }
    for (int i = 0; i < 16; ++i) { cx[i] = 0; cy[i] = 0; }
    for (int i = 0; i < NM; ++i) m[i] = (v4f64){0, 0, 0, 0};
    const double ma = in[threadIdx.x + 1], mb = in[threadIdx.x + 2];
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < NM) m[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, m[i], 0, 0, 0);
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                cx[4 * i + j] = fma(ax[i], bx[j], cx[4 * i + j]); cx[4 * i + j] = fma(-ay[i], by[j], cx[4 * i + j]);
                cy[4 * i + j] = fma(ax[i], by[j], cy[4 * i + j]); cy[4 * i + j] = fma(ay[i], bx[j], cy[4 * i + j]);
            }
        }
        double t = ax[0]; ax[0] = ax[1]; ax[1] = ax[2]; ax[2] = ax[3]; ax[3] = t;
        t = by[0]; by[0] = by[1]; by[1] = by[2]; by[2] = by[3]; by[3] = t;
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += cx[i] + cy[i];
    for (int i = 0; i < NM; ++i) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NM>
void run_mix(double *d, const double *din) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 2; ++waves) {
        const int nb = 256 * waves, it2 = 4000;
        hipLaunchKernelGGL(k_cblock_mfma<NM>, dim3(nb), dim3(256), 0, 0, d, it2, din);
        hipEventRecord(e0); hipLaunchKernelGGL(k_cblock_mfma<NM>, dim3(nb), dim3(256), 0, 0, d, it2, din); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fv = (double)nb * 256 * it2 * 128.0, fm = (double)nb * 4 * it2 * NM * 2048.0;
        printf("4x4 complex block + %d MFMA per iteration, %d wave(s)/SIMD: %.1f TFLOP/s total (vector %.1f + matrix %.1f), %.3f ms\n", NM, waves, (fv + fm) / ms / 1e9, fv / ms / 1e9, fm / ms / 1e9, ms);
    }
}
int main() {
    double *d; hipMalloc(&d, 256 * 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 8;
    double *din; hipMalloc(&din, 256 * 16 * 8);
    { double h[256 * 16]; for (int i = 0; i < 256 * 16; ++i) h[i] = 1e-3 * ((i * 7919) % 1000) - 0.5; hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice); }
    for (int waves = 1; waves <= 3; ++waves) {         // workgroups per CU = waves per SIMD
        const int nb = 256 * waves, it2 = 4000;
        hipLaunchKernelGGL(k_cblock, dim3(nb), dim3(256), 0, 0, d, it2, din);
        hipEventRecord(e0); hipLaunchKernelGGL(k_cblock, dim3(nb), dim3(256), 0, 0, d, it2, din); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("complex 4x4 register block, %d wave(s) per SIMD: %.1f TFLOP/s\n", waves, (double)nb * 256 * it2 * 128.0 / ms / 1e9);
    }
    run_mix<0>(d, din); run_mix<1>(d, din); run_mix<2>(d, din); run_mix<3>(d, din); run_mix<4>(d, din);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 4 /*waves*/ * iters * 8 * 2048.0;
        printf("mfma f64 16x16x4, 8 acc: %.1f TFLOP/s\n", fl / ms / 1e9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(256), 0, 0, d, iters * 8, 1.000001, 1e-9); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        fl = (double)blocks * 256 * iters * 8 * 16 * 2.0;
        printf("vector fma f64, 16 acc: %.1f TFLOP/s\n", fl / ms / 1e9);
    }
    return 0;
}
