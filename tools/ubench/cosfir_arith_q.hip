// round 6: cosfir_arith.hip with NQ cosine terms -- NQ = 1 is the Hamming form the product kernel k_chain_cos1k was built from,
// NQ = 3 the four-term blackmanHarris window (filters.py:139): seven running sums instead of three.  Gate for a product kernel: arithmetic-only
// <= 0.11 ms and <= 0.12 J per 2^26 samples.  Usage: cosfir_arith_q [seconds] [waves per SIMD] [NQ = 1 | 3]
// micro-benchmark: the ARITHMETIC of an f32 cosine-series ("modulated running sum") form of the
// M = 1 chain -- NCO, Hamming-255 FIR as three running sums, FM discriminator -- with no loads, stores or LDS traffic, so that its
// time and its joules per 2^26 samples can be set beside the overlap-save FFT kernel's arithmetic-only build
// (profiles/r04_clock_power.txt: 0.1227 ms, 0.129 J).  Gate: go on to a product kernel only if <= 0.08 ms and <= 0.07 J.
//
// The form (filters.py:199 hamming = 0.54 - 0.46 cos(2 pi k / 254), comm.py:63-78 NCO, demod_fm.py:40-49):
//   xt[n] = x[n] e^{-j w n}
//   R[n] = sum_{k<255} xt[n-k],  C[n] = sum_{k<255} cos(phi k) xt[n-k],  S[n] = sum_{k<255} sin(phi k) xt[n-k],   phi = 2 pi / 254
//   (C, S)[n] = Rot_phi((C, S)[n-1]) + (xt[n] - cos(phi) xt[n-255], -sin(phi) xt[n-255]),   R[n] = R[n-1] + xt[n] - xt[n-255]
//   y[n] = 0.54 R[n] - 0.46 C[n],   out[n] = angle(y[n] conj y[n-1])
// Layout priced here (the cheapest one found): a wave walks rows of 1024 samples, a lane owns 16 consecutive samples of
// the row (xt[n-255] is then another lane's register: one LDS exchange of 8 B per sample in a product kernel, a register of the
// previous row here), so the recurrence is a two-pass scan: pass A (lane totals from a zero state), a weighted Kogge-Stone scan of
// the 64 totals through DPP moves (weights = rotations by 16 phi 2^k), pass B (the recurrence again from each lane's true state).
//   hipcc --offload-arch=gfx950 -O3 -o cosfir_arith cosfir_arith.hip ;  ./cosfir_arith [seconds] [waves per SIMD 1..4]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef float v2f __attribute__((ext_vector_type(2)));

#define MAXQ 3
struct CosfirConsts {
    float c[MAXQ], s[MAXQ];        // cos(q phi), sin(q phi)
    v2f nco[16];                   // e^{-j w i}, i = 0..15 (wave-uniform: scalar registers)
    v2f rowstep;                   // e^{-j w 1024}
    float wc[MAXQ][4], ws[MAXQ][4];            // rotation by 16 q phi 2^k, k = 0..3 (row_shr 1, 2, 4, 8)
    float a0, a[MAXQ], theta, eps;
};

#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define DPP_WAVE_SHR1 0x138

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dppf(float v) {       // lanes without a source (or masked rows) read 0
    if (ROW_MASK == 0xF) return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));   // bound_ctrl: 0 fill
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ v2f dpp2(v2f v) { return (v2f){dppf<CTRL, ROW_MASK>(v.x), dppf<CTRL, ROW_MASK>(v.y)}; }

__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f fma2(float a, v2f b, v2f c) { return __builtin_elementwise_fma((v2f){a, a}, b, c); }
// (a.x b.x - a.y b.y, a.x b.y + a.y b.x) as one packed multiply and one packed multiply-add
__device__ __forceinline__ v2f cmul(v2f a, v2f b) { return fma2(a.x, b, (v2f){-a.y, a.y} * (v2f){b.y, b.x}); }
// one step of the rotation recurrence:  C' = c C - s S + bC,  S' = s C + c S + bS   (four packed multiply-adds)
#define ROT_STEPQ(Cq, Sq, cq, sq, bc, bs) do { v2f Cn_ = fma2(cq, Cq, fma2(-(sq), Sq, (bc))); Sq = fma2(sq, Cq, fma2(cq, Sq, (bs))); Cq = Cn_; } while (0)

// atan(y/x) for x > 0, |y| <= tan(pi/8) x: odd minimax polynomial (the small-angle path of the FFT kernel's discriminator)
__device__ __forceinline__ float atan_small(float y, float x) {
    float t = y * __builtin_amdgcn_rcpf(x);
    float u = t * t;
    float p = fmaf(u, -0.0752896400f, 0.1065626393f);
    p = fmaf(p, u, -0.1420889944f);
    p = fmaf(p, u, 0.1999355085f);
    p = fmaf(p, u, -0.3333314528f);
    return fmaf(p * u, t, t);
}

__device__ __forceinline__ float rl63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
template <int NQ> struct St { v2f C[NQ], S[NQ], R; };

template <int NQ, int STEP, int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_step(St<NQ>& t, const float (&wc)[NQ], const float (&ws)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        v2f Cs = dpp2<CTRL, ROW_MASK>(t.C[q]), Ss = dpp2<CTRL, ROW_MASK>(t.S[q]);
        t.C[q] = fma2(wc[q], Cs, fma2(-ws[q], Ss, t.C[q]));
        t.S[q] = fma2(ws[q], Cs, fma2(wc[q], Ss, t.S[q]));
    }
    t.R += dpp2<CTRL, ROW_MASK>(t.R);
}

template <int NQ>
__global__ void __launch_bounds__(256) k_cosfir_arith(float* out, int rows, CosfirConsts k, const v2f* seed) {
    const int lane = threadIdx.x & 63;
    v2f base[16], xa[16], xb[16];
    float fb[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        base[i] = seed[(threadIdx.x * 16 + i) & 4095];           // the "loaded" samples: values on the u8 grid, |x| <= 181
        xb[i] = base[15 - i];
        fb[i] = 0.0f;
    }
    // per-lane constants: the lane's NCO phasor e^{-j w 16 lane}, the scan weights of the two broadcast steps
    float sn, cs;
    __sincosf(-k.theta * 16.0f * lane, &sn, &cs);
    v2f qlane = (v2f){cs, sn};
    const float phi16 = 16.0f * 6.283185307f / 254.0f;
    float b15c[NQ], b15s[NQ], b31c[NQ], b31s[NQ], cq[NQ], sq[NQ], w0c[NQ], w0s[NQ], w1c[NQ], w1s[NQ], w2c[NQ], w2s[NQ], w3c[NQ], w3s[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        __sincosf((q + 1) * phi16 * ((lane & 15) + 1), &b15s[q], &b15c[q]);
        __sincosf((q + 1) * phi16 * ((lane & 31) + 1), &b31s[q], &b31c[q]);
        cq[q] = k.c[q]; sq[q] = k.s[q];
        w0c[q] = k.wc[q][0]; w0s[q] = k.ws[q][0]; w1c[q] = k.wc[q][1]; w1s[q] = k.ws[q][1];
        w2c[q] = k.wc[q][2]; w2s[q] = k.ws[q][2]; w3c[q] = k.wc[q][3]; w3s[q] = k.ws[q][3];
    }
    St<NQ> carry;
#pragma unroll
    for (int q = 0; q < NQ; ++q) { carry.C[q] = (v2f){0, 0}; carry.S[q] = (v2f){0, 0}; }
    carry.R = (v2f){0, 0};
    v2f zcarry = (v2f){1.0f, 0.0f};
    auto row = [&](v2f (&xt)[16], const v2f (&xprev)[16], int r) __attribute__((always_inline)) {
        qlane = cmul(qlane, k.rowstep);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v2f x = fma2(k.eps, (v2f){fb[i], fb[i]}, base[i]);
            xt[i] = cmul(x, cmul(qlane, k.nco[i]));
        }
        // ---- pass A: lane totals from a zero state (the comb inputs xt[i] - c_q d, -s_q d formed on the fly: NQ sets of them do not fit registers)
        St<NQ> t;
        {
            const v2f d = xprev[1];
#pragma unroll
            for (int q = 0; q < NQ; ++q) { t.C[q] = fma2(-cq[q], d, xt[0]); t.S[q] = -sq[q] * d; }
            t.R = xt[0] - d;
        }
#pragma unroll
        for (int i = 1; i < 16; ++i) {
            const v2f d = xprev[(i + 1) & 15];
#pragma unroll
            for (int q = 0; q < NQ; ++q) ROT_STEPQ(t.C[q], t.S[q], cq[q], sq[q], fma2(-cq[q], d, xt[i]), -sq[q] * d);
            t.R += xt[i] - d;
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                t.C[q] = fma2(w0c[q], carry.C[q], fma2(-w0s[q], carry.S[q], t.C[q]));
                t.S[q] = fma2(w0s[q], carry.C[q], fma2(w0c[q], carry.S[q], t.S[q]));
            }
            t.R += carry.R;
        }
        scan_step<NQ, 1, DPP_ROW_SHR(1), 0xF>(t, w0c, w0s);
        scan_step<NQ, 2, DPP_ROW_SHR(2), 0xF>(t, w1c, w1s);
        scan_step<NQ, 4, DPP_ROW_SHR(4), 0xF>(t, w2c, w2s);
        scan_step<NQ, 8, DPP_ROW_SHR(8), 0xF>(t, w3c, w3s);
        scan_step<NQ, 16, DPP_ROW_BCAST15, 0xA>(t, b15c, b15s);
        scan_step<NQ, 32, DPP_ROW_BCAST31, 0xC>(t, b31c, b31s);
        St<NQ> b0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { b0.C[q] = dpp2<DPP_WAVE_SHR1, 0xF>(t.C[q]); b0.S[q] = dpp2<DPP_WAVE_SHR1, 0xF>(t.S[q]); }
        b0.R = dpp2<DPP_WAVE_SHR1, 0xF>(t.R);
        if (lane == 0) b0 = carry;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { carry.C[q] = (v2f){rl63(t.C[q].x), rl63(t.C[q].y)}; carry.S[q] = (v2f){rl63(t.S[q].x), rl63(t.S[q].y)}; }
        carry.R = (v2f){rl63(t.R.x), rl63(t.R.y)};
        // ---- pass B: the recurrence from the true state; y = a0 R + sum a_q C_q; FM
        v2f z[16];
        St<NQ> u = b0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const v2f d = xprev[(i + 1) & 15];
            v2f y = (v2f){0, 0};
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                ROT_STEPQ(u.C[q], u.S[q], cq[q], sq[q], fma2(-cq[q], d, xt[i]), -sq[q] * d);
                y = fma2(k.a[q], u.C[q], y);
            }
            u.R += xt[i] - d;
            z[i] = fma2(k.a0, u.R, y);
        }
        v2f zl = dpp2<DPP_WAVE_SHR1, 0xF>(z[15]);
        if (lane == 0) zl = zcarry;
        zcarry = (v2f){rl63(z[15].x), rl63(z[15].y)};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v2f zp = i ? z[i - 1] : zl;
            float re = z[i].x * zp.x + z[i].y * zp.y, im = z[i].y * zp.x - z[i].x * zp.y;
            fb[i] = atan_small(im, fmaxf(re, fmaf(fabsf(im), 2.4142136f, 1e-30f))) - k.theta;
        }
        (void)r;
    };
    for (int r = 0; r < rows; r += 2) {           // (rows is even: the two sample arrays swap roles without a copy)
        row(xa, xb, r);
        row(xb, xa, r + 1);
    }
    float acc = carry.C[0].x + carry.S[NQ - 1].y + carry.R.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += fb[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    int wps = argc > 2 ? atoi(argv[2]) : 3;
    int nq = argc > 3 ? atoi(argv[3]) : 1;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * wps;                       // 256 threads = 4 waves, one per SIMD: wps workgroups per CU
    const long total_rows = (1L << 26) / 1024;
    const int rows = 2 * (int)((total_rows + blocks * 8 - 1) / (blocks * 8));
    CosfirConsts k;
    const double phi = 2.0 * M_PI / 254.0, w = 2.0 * M_PI * 25000.0 / 2400000.0;
    for (int q = 0; q < MAXQ; ++q) { k.c[q] = (float)cos((q + 1) * phi); k.s[q] = (float)sin((q + 1) * phi); }
    for (int i = 0; i < 16; ++i) k.nco[i] = (v2f){(float)cos(w * i), (float)-sin(w * i)};
    k.rowstep = (v2f){(float)cos(w * 1024), (float)-sin(w * 1024)};
    for (int q = 0; q < MAXQ; ++q)
        for (int j = 0; j < 4; ++j) { k.wc[q][j] = (float)cos(16 * (q + 1) * phi * (1 << j)); k.ws[q][j] = (float)sin(16 * (q + 1) * phi * (1 << j)); }
    if (nq == 1) { k.a0 = 0.54f; k.a[0] = -0.46f; k.a[1] = k.a[2] = 0.f; }
    else { k.a0 = 0.35875f; k.a[0] = -0.48829f; k.a[1] = 0.14128f; k.a[2] = -0.01168f; }
    k.theta = (float)w; k.eps = 1e-3f;
    v2f* hseed = (v2f*)malloc(4096 * sizeof(v2f));
    srand(1234);
    for (int i = 0; i < 4096; ++i) hseed[i] = (v2f){(rand() & 255) - 127.5f, (rand() & 255) - 127.5f};
    v2f* seed; float* out;
    CK(hipMalloc(&seed, 4096 * sizeof(v2f)));
    CK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
    CK(hipMemcpy(seed, hseed, 4096 * sizeof(v2f), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() {
        if (nq == 1) hipLaunchKernelGGL(k_cosfir_arith<1>, dim3(blocks), dim3(256), 0, 0, out, rows, k, seed);
        else hipLaunchKernelGGL(k_cosfir_arith<3>, dim3(blocks), dim3(256), 0, 0, out, rows, k, seed);
    };
    for (int i = 0; i < 50; ++i) launch();
    CK(hipDeviceSynchronize());
    long launches = 0;
    double ev_ms = 0;
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 200; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ev_ms += ms; launches += 200;
    }
    float h0; CK(hipMemcpy(&h0, out, 4, hipMemcpyDeviceToHost));
    printf("cosfir_arith: %d CUs, %d workgroups of 4 waves (%d waves per SIMD asked), %d rows of 1024 samples per wave (%.3f x 2^26 samples per launch), %d cosine terms (%d running sums)\n",
           cus, blocks, wps, rows, (double)rows * blocks * 4 * 1024 / (1L << 26), nq, 1 + 2 * nq);
    printf("cosfir_arith: %ld launches, %.4f ms per launch (HIP events), %.4f ms per 2^26 samples, out[0] = %g\n", launches, ev_ms / launches,
           ev_ms / launches / ((double)rows * blocks * 4 * 1024 / (1L << 26)), h0);
    return 0;
}
