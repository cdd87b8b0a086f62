// micro-benchmark: VALU issue rate of 1..3 vector waves per SIMD beside one wave per SIMD that issues
// v_mfma_f32_32x32x16_f16 back to back (the shape of the headline kernel), and the MFMA rate beside them.
// hipcc --offload-arch=gfx950 -O3 -o valu_beside_mfma valu_beside_mfma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
// mode: 0 v_fma_f32, 1 v_cvt_f16_f32, 2 v_mov_b32 dpp, 3 v_fma_mixlo_f16, 4 v_rcp_f32, 5 v_cndmask
template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* cyc, int iters, int mfma_waves, int valu_waves, float a, float b) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (wave >= mfma_waves) return;
        v16f c0, c1;
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
        v8h fa, fb;
        for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(threadIdx.x * 0.001f + i); fb[i] = (_Float16)(0.5f + i * 0.01f); }
        __builtin_amdgcn_s_setprio(3);
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, c1, 0, 0, 0);
            }
        }
        unsigned long long t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
        out[blockIdx.x * 1024 + threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
        return;
    }
    if (wave >= 4 + 4 * valu_waves) return;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x[i]));
            if (MODE == 2) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
            if (MODE == 3) asm volatile("v_fma_mixlo_f16 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (MODE == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            if (MODE == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}
int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 256 * 1024 * 4); hipMalloc(&c, 256 * 16 * 8);
    const int iters = 2000;
    const char* names[6] = {"v_fma_f32", "v_cvt_f16_f32", "v_mov_b32_dpp", "v_fma_mixlo_f16", "v_rcp_f32", "v_cndmask_b32"};
    unsigned long long h[256 * 16];
    for (int mode = 0; mode < 6; ++mode) {
        for (int mf = 0; mf <= 4; mf += 4) {
            for (int vw = (mf ? 0 : 1); vw <= 3; ++vw) {
                hipMemset(c, 0, sizeof(h));
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, d, c, iters, mf, vw, 1.0001f, 0.5f);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, d, c, iters, mf, vw, 1.0001f, 0.5f);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, d, c, iters, mf, vw, 1.0001f, 0.5f);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, d, c, iters, mf, vw, 1.0001f, 0.5f);
                    if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(1024), 0, 0, d, c, iters, mf, vw, 1.0001f, 0.5f);
                    if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(1024), 0, 0, d, c, iters, mf, vw, 1.0001f, 0.5f);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
                double m = 0, v = 0; int nm = 0, nv = 0;
                for (int b = 0; b < 256; ++b) for (int w = 0; w < 16; ++w) {
                    if (!h[b * 16 + w]) continue;
                    if (w < 4) { m += h[b * 16 + w]; ++nm; } else { v += h[b * 16 + w]; ++nv; }
                }
                printf("%-16s mfma waves/SIMD=%d valu waves/SIMD=%d :", names[mode], mf ? 1 : 0, vw);
                if (nm) printf("  %.1f cycles/MFMA", m / nm / (iters * 16.0));
                if (nv) printf("  %.2f cycles/VALU/wave = %.2f cycles/VALU/SIMD", v / nv / (iters * 16.0), v / nv / (iters * 16.0) / vw);
                printf("\n");
            }
        }
    }
    return 0;
}
