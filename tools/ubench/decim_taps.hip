// The tap loop of k_chain_decim_w alone (dd_decimw.hip): each wave owns an LDS image and runs the loop over it again and again.
//   variant 1: one output per lane   -- window start lane * 34, 160 taps: 80 ds_read_b128 + 160 v_pk_fma_f32 per output
//   variant 2: two outputs per lane  -- window start lane * 70 (padded image), the union of two windows 34 samples apart, 192 sample
//              positions: 96 reads + 2 x 192 multiply-adds for TWO outputs
// Question: the kernel's tap loop is bound by the LDS read rate -- do shared reads buy time, at the occupancy the bigger rows leave
// (4 waves per CU instead of 8)?   usage: decim_taps <variant> <waves per CU> [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v2f* const_f2p;

__device__ __forceinline__ void mac_lo(v2f& acc, v2f c, v2f x) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(c), "v"(x)); }
__device__ __forceinline__ void mac_hi(v2f& acc, v2f c, v2f x) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(c), "v"(x)); }

template <int TWO>
__global__ void __launch_bounds__(64) k(const float* taps, float2* out, int iters, int img, int ntaps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* buf = reinterpret_cast<float2*>(smem);
    const int lane = threadIdx.x;
    for (int i = lane; i < img; i += 64) buf[i] = make_float2(0.001f * i, 1.0f - 0.002f * i);
    const v4f* w4 = reinterpret_cast<const v4f*>(buf + lane * (TWO ? 70 : 34));
    const const_f2p G = (const_f2p)taps;
    const const_f2p G2 = (const_f2p)(taps + 256);                  // (variant 2: the second output's taps, shifted by M)
    v2f acc = (v2f){0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        v2f a0 = (v2f){0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0, b0 = a0, b1 = a0, b2 = a0, b3 = a0;
        v4f xa[8], xb[8];
        v2f ca[8], cb[8], da[8], db[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { xa[u] = w4[u]; ca[u] = G[u]; if (TWO) da[u] = G2[u]; }
        auto mac = [&](const v4f (&x)[8], const v2f (&c)[8], const v2f (&d)[8]) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                mac_lo(a0, c[u], (v2f){x[u].x, x[u].y}); mac_hi(a1, c[u], (v2f){x[u].z, x[u].w});
                mac_lo(a2, c[u + 1], (v2f){x[u + 1].x, x[u + 1].y}); mac_hi(a3, c[u + 1], (v2f){x[u + 1].z, x[u + 1].w});
                if (TWO) {
                    mac_lo(b0, d[u], (v2f){x[u].x, x[u].y}); mac_hi(b1, d[u], (v2f){x[u].z, x[u].w});
                    mac_lo(b2, d[u + 1], (v2f){x[u + 1].x, x[u + 1].y}); mac_hi(b3, d[u + 1], (v2f){x[u + 1].z, x[u + 1].w});
                }
            }
        };
        for (int j = 0; j < ntaps; j += 32) {
            if (j + 16 < ntaps) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { xb[u] = w4[(j + 16) / 2 + u]; cb[u] = G[(j + 16) / 2 + u]; if (TWO) db[u] = G2[(j + 16) / 2 + u]; }
            }
            mac(xa, ca, da);
            if (j + 16 < ntaps) {
                if (j + 32 < ntaps) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) { xa[u] = w4[(j + 32) / 2 + u]; ca[u] = G[(j + 32) / 2 + u]; if (TWO) da[u] = G2[(j + 32) / 2 + u]; }
                }
                mac(xb, cb, db);
            }
        }
        acc += (a0 + a1) + (a2 + a3);
        if (TWO) acc += (b0 + b1) + (b2 + b3);
        // (keep the iterations apart: the image changes a little)
        if (lane == 0) buf[it & 1023] = make_float2(acc.x * 1e-30f, acc.y * 1e-30f);
    }
    out[blockIdx.x * 64 + lane] = make_float2(acc.x, acc.y);
}

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 1, wpc = argc > 2 ? atoi(argv[2]) : 8, iters = argc > 3 ? atoi(argv[3]) : 2000;
    const int ntaps = variant == 2 ? 192 : 160;
    const int img = variant == 2 ? 64 * 70 + 256 : 64 * 34 + 256;
    // the LDS allocation decides the occupancy
    const size_t lds = (size_t)((160 * 1024) / wpc) & ~(size_t)255;
    if (lds < sizeof(float2) * img) { printf("image does not fit\n"); return 1; }
    float* taps; float2* out;
    hipMalloc(&taps, 4096); hipMalloc(&out, sizeof(float2) * 64 * 256 * 16);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 1.0f / (1 + i % 37);
    hipMemcpy(taps, h, 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int grid = 256 * wpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (variant == 2) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), lds, 0, taps, out, iters, img, ntaps);
        else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), lds, 0, taps, out, iters, img, ntaps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double outputs = (double)grid * iters * 64 * (variant == 2 ? 2 : 1);
        printf("variant %d, %d waves per CU: %.3f ms for %d iterations, %.2f ns per wave pass, %.1f G outputs/s (%.3f ns per 64 outputs and CU)\n", variant, wpc, ms, iters,
               ms * 1e6 / iters, outputs / ms * 1e-6, ms * 1e6 / (outputs / 64 / 256));
    }
    return 0;
}
