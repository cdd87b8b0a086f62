// micro-benchmark: what a global load / store INSTRUCTION costs the issuing wave when the data is cache resident
// (each workgroup re-reads / re-writes its own 32 KB), by access width, 1 and 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o vmem_issue vmem_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int W, bool ST>     // W dwords per lane: 1, 2, 4
__global__ void __launch_bounds__(256) k(float* buf, unsigned long long* cyc, int iters) {
    float* p = buf + (size_t)blockIdx.x * 8192 + threadIdx.x * W;
    float acc = 0.f;
    v4f v4 = {(float)threadIdx.x, 1.f, 2.f, 3.f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* q = p + (r * 256 * W) % 8192;
            if (!ST) {
                if (W == 1) { float v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(q)); asm volatile("" :: "v"(v)); }
                if (W == 2) { v2f v; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(q)); asm volatile("" :: "v"(v)); }
                if (W == 4) { v4f v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(q)); asm volatile("" :: "v"(v)); }
            } else {
                if (W == 1) asm volatile("global_store_dword %0, %1, off" :: "v"(q), "v"(v4.x) : "memory");
                if (W == 2) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(q), "v"(v4.xy) : "memory");
                if (W == 4) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(q), "v"(v4) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 123.f) buf[0] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
static double avg(unsigned long long* c, int n) { double s = 0; for (int i = 0; i < n; ++i) s += (double)c[i]; return s / n; }
int main() {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, (size_t)1024 * 8192 * 4); (void)hipMalloc(&c, 1024 * 4 * 8);
    (void)hipMemset(d, 0, (size_t)1024 * 8192 * 4);
    unsigned long long h[4096];
    const int iters = 500;
#define RUN(W, ST, WG)                                                                                          \
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<W, ST>), dim3(256 * WG), dim3(256), 0, 0, d, c, iters); \
    (void)hipDeviceSynchronize(); (void)hipMemcpy(h, c, 256 * WG * 4 * 8, hipMemcpyDeviceToHost);              \
    printf("%s dwordx%d  waves/SIMD %d : %.1f cycles per instruction per wave (16 per drain)\n", ST ? "store" : "load ", W, WG, avg(h, 256 * WG * 4) / (iters * 16.0));
    RUN(1, false, 1) RUN(2, false, 1) RUN(4, false, 1) RUN(1, false, 2) RUN(2, false, 2) RUN(4, false, 2)
    RUN(1, true, 1) RUN(2, true, 1) RUN(4, true, 1) RUN(1, true, 2) RUN(2, true, 2) RUN(4, true, 2)
    return 0;
}
