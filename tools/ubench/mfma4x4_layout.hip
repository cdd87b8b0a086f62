// round 6: v_mfma_f32_4x4x1_16b_f32 -- the operand layout, the A broadcast (cbsz = 4, abid) and whether its arithmetic is one fmaf per element.
// D_t[m][n] = C_t[m][n] + A_t[m] B_t[n] for 16 blocks t: lane 4 t + m holds A_t[m], lane 4 t + n holds B_t[n], register m of lane 4 t + n holds
// D_t[m][n]; with cbsz = 4 every block uses the A of block `abid`.  Prints mismatches against fmaf.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int ABID>
__global__ void k(const float* a, const float* b, const float* c, float* d_plain, float* d_bc, float* d_chain) {
    const int l = threadIdx.x;
    v4f acc = {c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
    v4f r0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    v4f r1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 4, ABID, 0);
    // a chain of four on one accumulator (the compiler places the waits)
    v4f r2 = acc;
    r2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], r2, 4, 0, 0);
    r2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[64 + l], r2, 4, 1, 0);
    r2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[128 + l], r2, 4, 2, 0);
    r2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[192 + l], r2, 4, 3, 0);
    for (int m = 0; m < 4; ++m) { d_plain[l * 4 + m] = r0[m]; d_bc[l * 4 + m] = r1[m]; d_chain[l * 4 + m] = r2[m]; }
}

int main() {
    float ha[64], hb[256], hc[256], h0[256], h1[256], h2[256];
    srand(7);
    auto rnd = []() { return (float)((rand() % 20001) - 10000) / 977.0f * ((rand() & 1) ? 1e-3f : 37.f); };
    for (int i = 0; i < 64; ++i) ha[i] = rnd();
    for (int i = 0; i < 256; ++i) { hb[i] = rnd(); hc[i] = rnd(); }
    float *a, *b, *c, *d0, *d1, *d2;
    hipMalloc(&a, 256); hipMalloc(&b, 1024); hipMalloc(&c, 1024); hipMalloc(&d0, 1024); hipMalloc(&d1, 1024); hipMalloc(&d2, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 1024, hipMemcpyHostToDevice); hipMemcpy(c, hc, 1024, hipMemcpyHostToDevice);
    const int ABID = 5;
    hipLaunchKernelGGL(k<ABID>, dim3(1), dim3(64), 0, 0, a, b, c, d0, d1, d2);
    hipMemcpy(h0, d0, 1024, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, 1024, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, 1024, hipMemcpyDeviceToHost);
    int bad0 = 0, bad1 = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l)
        for (int m = 0; m < 4; ++m) {
            const int t = l / 4;
            const float e0 = fmaf(ha[4 * t + m], hb[l], hc[l * 4 + m]);
            const float e1 = fmaf(ha[4 * ABID + m], hb[l], hc[l * 4 + m]);
            float e2 = hc[l * 4 + m];
            for (int s = 0; s < 4; ++s) e2 = fmaf(ha[4 * s + m], hb[64 * s + l], e2);
            if (memcmp(&e0, &h0[l * 4 + m], 4)) { if (bad0 < 4) printf("plain  lane %d m %d: got %.9g want %.9g\n", l, m, h0[l * 4 + m], e0); ++bad0; }
            if (memcmp(&e1, &h1[l * 4 + m], 4)) { if (bad1 < 4) printf("bcast  lane %d m %d: got %.9g want %.9g\n", l, m, h1[l * 4 + m], e1); ++bad1; }
            if (memcmp(&e2, &h2[l * 4 + m], 4)) { if (bad2 < 4) printf("chain  lane %d m %d: got %.9g want %.9g\n", l, m, h2[l * 4 + m], e2); ++bad2; }
        }
    printf("mfma_f32_4x4x1 layout: plain %d, broadcast(abid=%d) %d, chain %d mismatches of 256 (0 = the layout above, one fmaf per element)\n", bad0, ABID, bad1, bad2);
    return 0;
}
