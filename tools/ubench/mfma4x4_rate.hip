// round 6: issue rate of v_mfma_f32_4x4x1_16b_f32 (cbsz = 4) by the number of independent accumulator chains, one and two waves per SIMD.
//   mfma4x4_rate -> cycles per instruction and wave (s_memtime around 256 x CH instructions x 64 repeats)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, float a0, float b0) {
    v4f acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 4, 3, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CH>
static void run(int threads) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4 * 512 * 256); hipMalloc(&cyc, 8 * 256);
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f, 2.0f);
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f, 2.0f);
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
