// micro-benchmark: aggregate VALU issue rate per SIMD (inline asm, 16 independent chains),
// and the s_memtime tick vs wall clock
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; i += (MODE == 1 ? 2 : 1)) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (MODE == 1) {
                    typedef float v2f __attribute__((ext_vector_type(2)));
                    v2f p = {x[i], x[i + 1]}, pa = {a, a}, pb = {b, b};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(pa), "v"(pb));
                    x[i] = p.x; x[i + 1] = p.y;
                }
                if (MODE == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                if (MODE == 3) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x[i]));
                if (MODE == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long*)out)[1 << 20] = t1 - t0;
}
int main() {
    float* d; hipMalloc(&d, (1 << 23) + 64);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[5] = {"v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "v_cvt_f16_f32", "v_cndmask_b32"};
    for (int w = 1; w <= 4; w *= 2) {
        for (int mode = 0; mode < 5; ++mode) {
            dim3 g(256), b(256 * w);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, d, iters, 1.0001f, 0.5f);
                if (mode == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, d, iters, 1.0001f, 0.5f);
                if (mode == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, d, iters, 1.0001f, 0.5f);
                if (mode == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, d, iters, 1.0001f, 0.5f);
                if (mode == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, d, iters, 1.0001f, 0.5f);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long cyc; hipMemcpy(&cyc, ((unsigned long long*)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
            const double n = (double)iters * 4 * (mode == 1 ? 8 : 16);
            printf("waves/SIMD=%d %-14s: %.2f ticks/wave-instr, %.2f ticks/instr/SIMD, %.2f ns/instr/SIMD (tick=%.3f ns)\n", w, names[mode],
                   cyc / n, cyc / n / w, ms * 1e6 / n / w, ms * 1e6 / cyc);
        }
    }
    return 0;
}
