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
int main() {
    double *d; hipMalloc(&d, 256 * 4096 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 8;
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
