// Does data written by one kernel and read by the next come back from the 256 MB Infinity Cache instead of HBM when the working set is
// small enough?  Pattern A (level by level): write all of a 4 GB array, then read all of it.  Pattern B (chunked): write a chunk, read it,
// next chunk.  Same bytes either way.    hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o tools/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_write(double2 *p, long long n, double v) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = make_double2(v + i, v);
}
__global__ __launch_bounds__(256) void k_read(const double2 *p, long long n, double *out) {
    double s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { const double2 a = p[i]; s += a.x + a.y; }
    if (s == 1.234567) out[0] = s;
}
int main() {
    const long long total = 4LL << 30;      // bytes
    double2 *p; double *o; hipMalloc(&p, total); hipMalloc(&o, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long nel = total / 16;
    for (long long chunkMB : {4096LL, 1024LL, 256LL, 128LL, 64LL, 32LL, 16LL}) {
        const long long cel = chunkMB * (1 << 20) / 16;
        const int grid = (int)((cel / 256 / 8) < 8192 ? (cel / 256 / 8) : 8192);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (long long off = 0; off < nel; off += cel) {
                hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, p + off, cel, 1.0);
                hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, p + off, cel, o);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("chunk %5lld MB: write + read of 4 GB in %.3f ms  -> %.2f TB/s of traffic (%lld launches)\n", chunkMB, ms, 2.0 * total / ms / 1e9, 2 * nel / cel);
        }
    }
    return 0;
}
