// round 6: v_fma_f64 -- cycles per instruction and wave by the number of independent dependent-chains, one and two waves per SIMD
// (the IIR block passes are a float64 recurrence: two dependent operations per sample, eleven independent ones beside them)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CH>
__global__ void __launch_bounds__(512) k(double* out, unsigned long long* cyc, double a0, double b0) {
    double acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = (double)(threadIdx.x + c);
    const double a = a0, b = b0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_fma(acc[c], a, b);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
static void run(int threads) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8 * 512 * 256); hipMalloc(&cyc, 8 * 256);
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, cyc, 0.999, 0.001);
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, cyc, 0.999, 0.001);
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; ++i) m += (double)h[i];
    m /= 256;
    printf("chains %d, %d waves per SIMD: %.1f cycles per instruction and wave (%.1f per SIMD)\n", CH, threads / 256, m / (64.0 * 16 * CH), m / (64.0 * 16 * CH) / (threads / 256));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>(256); run<2>(256); run<4>(256); run<8>(256);
    run<1>(512); run<2>(512); run<4>(512); run<8>(512);
    return 0;
}
