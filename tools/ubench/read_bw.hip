// read-only HBM bandwidth of a 512 MiB buffer on MI355X: 16 B/lane streaming loads, several
// launch shapes.  hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ p, size_t n4, float* out, int per) {
    float acc = 0.f;
    const size_t base = (size_t)blockIdx.x * 256 * per + threadIdx.x;
    for (int u = 0; u < per; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n4) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    const size_t bytes = 512ull << 20, n4 = bytes / 16;
    float4* d; float* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int per : {1, 4, 12, 48}) {
        const unsigned grid = (unsigned)((n4 + 256ull * per - 1) / (256ull * per));
        for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, d, n4, o, per);
        hipEventRecord(e0);
        for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, d, n4, o, per);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read-only 512 MiB, %2d x 16 B per lane, grid %u: %.1f us/launch = %.2f TB/s\n", per, grid, ms / 200 * 1e3, bytes / (ms / 200 * 1e-3) / 1e12);
    }
    return 0;
}
