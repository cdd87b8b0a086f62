// micro-benchmark: the matrix stream of the Toeplitz FIR in its two limb formats, operands streaming from LDS on random
// data, one wave per SIMD, every CU busy, long enough for the clock to settle:
//   F16: per 16 taps 6 x v_mfma_f32_32x32x16_f16 (three limb products x re/im), 6 ds_read_b128        (today's kernel)
//   I8 : per 32 taps 12 x v_mfma_i32_32x32x32_i8 (exact 8-bit data x three tap limbs x 4 real products), 9 ds_read_b128
// Reports cycles per MFMA, the clock the chip holds (s_memtime / s_memrealtime) and the time per 1024-output strip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
template <bool I8>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int strips) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned s = 12345u + threadIdx.x * 7919u + blockIdx.x * 104729u;
    for (int i = threadIdx.x; i < 24000; i += 256) {
        s = s * 1664525u + 1013904223u;
        if (I8) reinterpret_cast<unsigned*>(smem)[i] = s;                                          // random bytes
        else { const _Float16 a = (_Float16)(int)((s >> 8) % 2001 - 1000), b = (_Float16)(int)((s >> 20) % 2001 - 1000);
               reinterpret_cast<_Float16*>(smem)[2 * i] = a; reinterpret_cast<_Float16*>(smem)[2 * i + 1] = b; }
    }
    __syncthreads();
    const char* abase = smem + wave * 2560 + 80 * (lane & 31) + 16 * (lane >> 5);
    const char* tb = smem + 48000 + lane * 16;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    if (!I8) {
        v16f cre, cim;
        for (int r = 0; r < 16; ++r) { cre[r] = 0.f; cim[r] = 0.f; }
        for (int t = 0; t < strips; ++t) {
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) {
                const int o = 32 * ks + 16 * (ks >> 1);
                const v8h a0 = *reinterpret_cast<const v8h*>(abase + o), a1 = *reinterpret_cast<const v8h*>(abase + 10880 + o);
                const v8h a2 = *reinterpret_cast<const v8h*>(abase + 21760 + o), a3 = *reinterpret_cast<const v8h*>(abase + 32640 + o);
                const v8h b0 = *reinterpret_cast<const v8h*>(tb + ks * 1024), b1 = *reinterpret_cast<const v8h*>(tb + (18 + ks) * 1024);
                cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, cre, 0, 0, 0);
                cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b0, cim, 0, 0, 0);
                cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, cre, 0, 0, 0);
                cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3, b0, cim, 0, 0, 0);
                cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, cre, 0, 0, 0);
                cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, cim, 0, 0, 0);
            }
            for (int r = 0; r < 16; ++r) { cre[r] *= 1e-6f; cim[r] *= 1e-6f; }
        }
        for (int r = 0; r < 16; ++r) acc += cre[r] + cim[r];
    } else {
        v16i c0r, c0i, c1r, c1i, c2r, c2i;
        for (int r = 0; r < 16; ++r) { c0r[r] = c0i[r] = c1r[r] = c1i[r] = c2r[r] = c2i[r] = 0; }
        for (int t = 0; t < strips; ++t) {
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                const int o = 32 * ks + 16 * (ks >> 1);
                const v4i dr = *reinterpret_cast<const v4i*>(abase + o), di = *reinterpret_cast<const v4i*>(abase + 10880 + o), dn = *reinterpret_cast<const v4i*>(abase + 21760 + o);
                const v4i g0r = *reinterpret_cast<const v4i*>(tb + (6 * ks) * 1024), g0i = *reinterpret_cast<const v4i*>(tb + (6 * ks + 1) * 1024);
                const v4i g1r = *reinterpret_cast<const v4i*>(tb + (6 * ks + 2) * 1024), g1i = *reinterpret_cast<const v4i*>(tb + (6 * ks + 3) * 1024);
                const v4i g2r = *reinterpret_cast<const v4i*>(tb + (6 * ks + 4) * 1024), g2i = *reinterpret_cast<const v4i*>(tb + (6 * ks + 5) * 1024);
                c0r = __builtin_amdgcn_mfma_i32_32x32x32_i8(dr, g0r, c0r, 0, 0, 0);
                c0i = __builtin_amdgcn_mfma_i32_32x32x32_i8(di, g0r, c0i, 0, 0, 0);
                c0r = __builtin_amdgcn_mfma_i32_32x32x32_i8(dn, g0i, c0r, 0, 0, 0);
                c0i = __builtin_amdgcn_mfma_i32_32x32x32_i8(dr, g0i, c0i, 0, 0, 0);
                c1r = __builtin_amdgcn_mfma_i32_32x32x32_i8(dr, g1r, c1r, 0, 0, 0);
                c1i = __builtin_amdgcn_mfma_i32_32x32x32_i8(di, g1r, c1i, 0, 0, 0);
                c1r = __builtin_amdgcn_mfma_i32_32x32x32_i8(dn, g1i, c1r, 0, 0, 0);
                c1i = __builtin_amdgcn_mfma_i32_32x32x32_i8(dr, g1i, c1i, 0, 0, 0);
                c2r = __builtin_amdgcn_mfma_i32_32x32x32_i8(dr, g2r, c2r, 0, 0, 0);
                c2i = __builtin_amdgcn_mfma_i32_32x32x32_i8(di, g2r, c2i, 0, 0, 0);
                c2r = __builtin_amdgcn_mfma_i32_32x32x32_i8(dn, g2i, c2r, 0, 0, 0);
                c2i = __builtin_amdgcn_mfma_i32_32x32x32_i8(dr, g2i, c2i, 0, 0, 0);
            }
            for (int r = 0; r < 16; ++r) { c0r[r] >>= 8; c0i[r] >>= 8; c1r[r] >>= 8; c1i[r] >>= 8; c2r[r] >>= 8; c2i[r] >>= 8; }
        }
        for (int r = 0; r < 16; ++r) acc += (float)(c0r[r] + c0i[r] + c1r[r] + c1i[r] + c2r[r] + c2i[r]);
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0) { cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}
int main() {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 256 * 256 * 4); (void)hipMalloc(&c, 256 * 4 * 2 * 8);
    unsigned long long h[2048];
    (void)hipFuncSetAttribute((const void*)k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 120000);
    (void)hipFuncSetAttribute((const void*)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 120000);
    const int strips = 20000;       // ~100 ms
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            if (mode == 0) hipLaunchKernelGGL((k<false>), dim3(256), dim3(256), 120000, 0, d, c, strips);
            else hipLaunchKernelGGL((k<true>), dim3(256), dim3(256), 120000, 0, d, c, strips);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
            double cy = 0, rt = 0; for (int i = 0; i < 1024; ++i) { cy += h[2 * i]; rt += h[2 * i + 1]; }
            cy /= 1024; rt /= 1024;
            printf("%s run %d: %.2f cycles per MFMA, clock %.3f GHz, %.3f us per 1024-output strip -> %.4f ms per 2^26 samples (MFMA stream only)\n", mode ? "i8  32x32x32" : "f16 32x32x16", rep,
                   cy / (strips * 108.0), cy / rt * 0.1, rt * 0.01 / strips, rt * 0.01 / strips * 65536 / 1024 * 1e-3);
        }
    return 0;
}
