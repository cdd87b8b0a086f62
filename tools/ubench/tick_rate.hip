// round 6: what does __builtin_readcyclecounter() (s_memtime) count on this part?  One wave spins on a dependent chain for a while; the counter's
// advance over the kernel against the kernel's duration by HIP events, with the shader clock read from rocm-smi beside it by the caller.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out, unsigned long long* cyc, int reps, float a) {
    float x = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = wall_clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) x = __builtin_fmaf(x, a, 1.0f);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4 * 64); hipMalloc(&cyc, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 3; ++pass) {
        const int reps = 1 << 18;
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, reps, 0.999f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
        int cr = 0; hipDeviceGetAttribute(&cr, hipDeviceAttributeClockRate, 0);
        printf("kernel %.3f ms: readcyclecounter advanced %llu = %.1f MHz; wall_clock64 advanced %llu = %.1f MHz (attribute %d kHz); %d x 64 dependent v_fma_f32: %.2f counter ticks each = %.2f ns each; clockRate attribute %d kHz\n",
               ms, h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3), wc, reps, (double)h[0] / ((double)reps * 64), ms * 1e6 / ((double)reps * 64), cr);
    }
    return 0;
}
