// micro-benchmark: cycles per v_mfma_f32_32x32x16_f16 when every MFMA's operands stream from LDS (one
// ds_read_b128 per MFMA, as in the headline kernel), for a few ways of placing reads and waits.
// One wave per SIMD (256-thread workgroup), nothing else on the CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define NKS 18
// MODE 0: operands in registers (no LDS)            MODE 1: 6 reads grouped per k-step, 2 k-steps ahead, one wait per k-step
// MODE 2: one read per MFMA gap, pinned (the kernel) MODE 3: as 2, reads 3 k-steps ahead
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40000; i += 256) reinterpret_cast<_Float16*>(smem)[i] = (_Float16)(0.001f * (i % 977));
    __syncthreads();
    const char* abase = smem + wave * 2560 + 80 * (lane & 31) + 16 * (lane >> 5);
    const v8h* tb = reinterpret_cast<const v8h*>(smem + 45000 - (45000 % 16)) + lane;
    v16f cre, cim;
    for (int r = 0; r < 16; ++r) { cre[r] = 0.f; cim[r] = 0.f; }
    __builtin_amdgcn_s_setprio(3);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles; ++t) {
        constexpr int D = MODE == 3 ? 4 : 3;
        v8h f[D][6];
#define LOADF(buf, ks) { const int o = 32 * (ks) + 16 * ((ks) >> 1); \
        f[buf][0] = *reinterpret_cast<const v8h*>(abase + o); f[buf][1] = *reinterpret_cast<const v8h*>(abase + 10880 + o); \
        f[buf][2] = *reinterpret_cast<const v8h*>(abase + 21760 + o); f[buf][3] = *reinterpret_cast<const v8h*>(abase + 32640 + o); \
        f[buf][4] = tb[(ks) * 64]; f[buf][5] = tb[(NKS + (ks)) * 64]; }
        if (MODE == 0) { LOADF(0, 0) }
        else { for (int d = 0; d < D - 1; ++d) LOADF(d, d) }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int b = MODE == 0 ? 0 : ks % D, nb = (ks + D - 1) % D, ksn = ks + D - 1;
            const bool pre = MODE != 0 && ksn < NKS;
            const int o = 32 * ksn + 16 * (ksn >> 1);
            if (MODE == 1 && pre) LOADF(nb, ksn)
            cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[b][0], f[b][4], cre, 0, 0, 0);
            if (MODE >= 2 && pre) f[nb][0] = *reinterpret_cast<const v8h*>(abase + o);
            if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
            cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[b][2], f[b][4], cim, 0, 0, 0);
            if (MODE >= 2 && pre) f[nb][4] = tb[ksn * 64];
            if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
            cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[b][1], f[b][4], cre, 0, 0, 0);
            if (MODE >= 2 && pre) f[nb][2] = *reinterpret_cast<const v8h*>(abase + 21760 + o);
            if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
            cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[b][3], f[b][4], cim, 0, 0, 0);
            if (MODE >= 2 && pre) f[nb][1] = *reinterpret_cast<const v8h*>(abase + 10880 + o);
            if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
            cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[b][0], f[b][5], cre, 0, 0, 0);
            if (MODE >= 2 && pre) f[nb][3] = *reinterpret_cast<const v8h*>(abase + 32640 + o);
            if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
            cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[b][2], f[b][5], cim, 0, 0, 0);
            if (MODE >= 2 && pre) f[nb][5] = tb[(NKS + ksn) * 64];
            if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += cre[r] + cim[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 256 * 256 * 4); hipMalloc(&c, 256 * 4 * 8);
    unsigned long long h[1024];
    const int tiles = 200;
    const char* names[4] = {"operands in registers", "reads grouped per k-step (compiler-placed waits)", "one read per MFMA gap, pinned (kernel form)", "one read per gap, 3 k-steps ahead"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 100000, 0, d, c, tiles);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 100000, 0, d, c, tiles);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 100000, 0, d, c, tiles);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 100000, 0, d, c, tiles);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 1024; ++i) m += h[i];
        printf("%-52s %.2f cycles per MFMA\n", names[mode], m / 1024 / (tiles * 108.0));
    }
    return 0;
}
