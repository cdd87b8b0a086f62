// micro-benchmark: issue rate and dependent latency of the packed-f32 VALU instructions the overlap-save FFT kernel is
// made of, for 1..8 independent chains per wave and 1..3 waves per SIMD; plus ds_write_b64 / ds_read_b64 and
// global_store_dword issue costs.   hipcc --offload-arch=gfx950 -O3 -o pk_chain pk_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
// MODE 0 v_pk_fma_f32, 1 v_pk_add_f32, 2 v_fma_f32, 3 v_pk_mul_f32 ; CH independent chains (of 16 registers)
template <int MODE, int CH>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters, v2f a, v2f b) {
    v2f x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (v2f){(float)threadIdx.x + i, 1.0f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8 * 16 / CH; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(a.x), "v"(b.x));
                if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(x[i]) : "v"(a));
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// LDS / store issue: MODE 0 16 x ds_write_b64, 1 16 x ds_read_b64 (+wait), 2 15 x global_store_dword, 3 4 x global_store_dwordx4 (same bytes as 16 dword stores)
template <int MODE>
__global__ void __launch_bounds__(256) kmem(float* out, unsigned long long* cyc, int iters) {
    __shared__ v2f buf[16 * 272];
    v2f x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (v2f){(float)threadIdx.x + i, 1.0f};
    const int t = threadIdx.x;
    float* o = out + (size_t)blockIdx.x * 4096 * 64;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) buf[t + 272 * k] = x[k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] += buf[(t >> 4) * 272 + (t & 15) + 16 * k];
            asm volatile("" ::: "memory");
        }
        if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < 15; ++k) o[(it & 63) * 4096 + t + 256 * k] = x[k].x;
        }
        if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < 4; ++k) reinterpret_cast<float4*>(o + (it & 63) * 4096)[t + 256 * k] = make_float4(x[k].x, x[k].y, x[k + 4].x, x[k + 4].y);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y;
    if (s == 12345.f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
static double avg(unsigned long long* c, int n) { double s = 0; for (int i = 0; i < n; ++i) s += (double)c[i]; return s / n; }
int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, (size_t)1024 * 4096 * 64 * 4); hipMalloc(&c, 1024 * 4 * 8);
    unsigned long long h[4096];
    const int iters = 500;
    const v2f a = {1.0001f, 0.9999f}, b = {0.001f, -0.001f};
    const char* nm[4] = {"v_pk_fma_f32", "v_pk_add_f32", "v_fma_f32", "v_pk_mul_f32"};
#define RUN(MODE, CH, WG)                                                                                       \
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, CH>), dim3(256 * WG), dim3(256), 0, 0, d, c, iters, a, b); \
    hipDeviceSynchronize(); hipMemcpy(h, c, 256 * WG * 4 * 8, hipMemcpyDeviceToHost);                          \
    printf("%-14s chains %d  waves/SIMD %d : %.2f cycles per instruction per wave\n", nm[MODE], CH, WG, avg(h, 256 * WG * 4) / (iters * 128.0));
    RUN(0, 1, 1) RUN(0, 2, 1) RUN(0, 4, 1) RUN(0, 8, 1) RUN(0, 16, 1) RUN(0, 1, 2) RUN(0, 4, 2) RUN(0, 16, 2) RUN(0, 16, 3)
    RUN(1, 1, 1) RUN(1, 2, 1) RUN(1, 4, 1) RUN(1, 16, 1) RUN(1, 16, 2)
    RUN(3, 1, 1) RUN(3, 4, 1) RUN(3, 16, 1)
    RUN(2, 1, 1) RUN(2, 2, 1) RUN(2, 4, 1) RUN(2, 16, 1) RUN(2, 16, 2) RUN(2, 16, 3)
    const char* mn[4] = {"16 x ds_write_b64 + wait", "16 x ds_read_b64 + use", "15 x global_store_dword", "4 x global_store_dwordx4"};
#define RUNM(MODE, WG)                                                                                          \
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kmem<MODE>), dim3(256 * WG), dim3(256), 0, 0, d, c, 500); \
    hipDeviceSynchronize(); hipMemcpy(h, c, 256 * WG * 4 * 8, hipMemcpyDeviceToHost);                          \
    printf("%-26s waves/SIMD %d : %.0f cycles per group per wave\n", mn[MODE], WG, avg(h, 256 * WG * 4) / 500.0);
    RUNM(0, 1) RUNM(0, 2) RUNM(1, 1) RUNM(1, 2) RUNM(2, 1) RUNM(2, 2) RUNM(3, 1) RUNM(3, 2)
    return 0;
}
