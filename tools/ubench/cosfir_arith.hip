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

struct CosfirConsts {
    float c, s;                    // cos(phi), sin(phi)
    v2f nco[16];                   // e^{-j w i}, i = 0..15 (wave-uniform: scalar registers)
    v2f rowstep;                   // e^{-j w 1024}
    float wc[4], ws[4];            // rotation by 16 phi 2^k, k = 0..3 (row_shr 1, 2, 4, 8)
    float a0, a1, theta, eps;
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
#define ROT_STEP(st, bc, bs) do { v2f Cn_ = fma2(c, (st).C, fma2(-s, (st).S, (bc))); (st).S = fma2(s, (st).C, fma2(c, (st).S, (bs))); (st).C = Cn_; } while (0)

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
struct St { v2f C, S, R; };

template <int STEP, int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_step(St& t, float wc, float ws) {
    v2f Cs = dpp2<CTRL, ROW_MASK>(t.C), Ss = dpp2<CTRL, ROW_MASK>(t.S), Rs = dpp2<CTRL, ROW_MASK>(t.R);
    t.C = fma2(wc, Cs, fma2(-ws, Ss, t.C));
    t.S = fma2(ws, Cs, fma2(wc, Ss, t.S));
    t.R += Rs;
}

template <bool REFRESH>
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
    float b15c, b15s, b31c, b31s;
    __sincosf(phi16 * ((lane & 15) + 1), &b15s, &b15c);
    __sincosf(phi16 * ((lane & 31) + 1), &b31s, &b31c);
    St carry = {(v2f){0, 0}, (v2f){0, 0}, (v2f){0, 0}};
    v2f zcarry = (v2f){1.0f, 0.0f};
    const float c = k.c, s = k.s;
    // one row; xt = this row's samples after the NCO (kept: they are the next row's xt[n - 255]), xprev = the previous row's
    auto row = [&](v2f (&xt)[16], const v2f (&xprev)[16], int r) __attribute__((always_inline)) {
        // ---- NCO (comm.py:77) and the comb inputs
        qlane = cmul(qlane, k.rowstep);
        v2f bC[16], bS[16], bR[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v2f x = fma2(k.eps, (v2f){fb[i], fb[i]}, base[i]);                      // (stands for the load: keeps every row's arithmetic live)
            xt[i] = cmul(x, cmul(qlane, k.nco[i]));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v2f d = xprev[(i + 1) & 15];                          // xt[n - 255]: lane - 16's sample i + 1 (an LDS exchange in a product kernel)
            bC[i] = fma2(-c, d, xt[i]);
            bS[i] = -s * d;
            bR[i] = xt[i] - d;
        }
        // ---- pass A: lane totals from a zero state
        St t = {bC[0], bS[0], bR[0]};
#pragma unroll
        for (int i = 1; i < 16; ++i) {
            ROT_STEP(t, bC[i], bS[i]);
            t.R += bR[i];
        }
        // ---- lane 0 takes the state the previous row ended in, then a weighted inclusive scan over the 64 lanes
        if (lane == 0) {
            t.C = fma2(k.wc[0], carry.C, fma2(-k.ws[0], carry.S, t.C));
            t.S = fma2(k.ws[0], carry.C, fma2(k.wc[0], carry.S, t.S));
            t.R += carry.R;
        }
        scan_step<1, DPP_ROW_SHR(1), 0xF>(t, k.wc[0], k.ws[0]);
        scan_step<2, DPP_ROW_SHR(2), 0xF>(t, k.wc[1], k.ws[1]);
        scan_step<4, DPP_ROW_SHR(4), 0xF>(t, k.wc[2], k.ws[2]);
        scan_step<8, DPP_ROW_SHR(8), 0xF>(t, k.wc[3], k.ws[3]);
        scan_step<16, DPP_ROW_BCAST15, 0xA>(t, b15c, b15s);
        scan_step<32, DPP_ROW_BCAST31, 0xC>(t, b31c, b31s);
        // state at the start of this lane's run = inclusive value of the lane before (lane 0: the carry)
        St b0;
        b0.C = dpp2<DPP_WAVE_SHR1, 0xF>(t.C);
        b0.S = dpp2<DPP_WAVE_SHR1, 0xF>(t.S);
        b0.R = dpp2<DPP_WAVE_SHR1, 0xF>(t.R);
        if (lane == 0) b0 = carry;
        carry.C = (v2f){rl63(t.C.x), rl63(t.C.y)};
        carry.S = (v2f){rl63(t.S.x), rl63(t.S.y)};
        carry.R = (v2f){rl63(t.R.x), rl63(t.R.y)};
        // ---- pass B: the recurrence from the true state; y = 0.54 R - 0.46 C (filters.py:199); FM (demod_fm.py:40-49)
        v2f z[16];
        St u = b0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ROT_STEP(u, bC[i], bS[i]);
            u.R += bR[i];
            z[i] = fma2(k.a0, u.R, k.a1 * u.C);
        }
        v2f zl = dpp2<DPP_WAVE_SHR1, 0xF>(z[15]);
        if (lane == 0) zl = zcarry;
        zcarry = (v2f){rl63(z[15].x), rl63(z[15].y)};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v2f zp = i ? z[i - 1] : zl;
            float re = z[i].x * zp.x + z[i].y * zp.y, im = z[i].y * zp.x - z[i].x * zp.y;
            // (the product kernel takes this path when a wave-uniform test says |im| <= tan(pi/8) re for the whole group; the stand-in
            //  data here is not a filtered signal, so the quotient is bounded by hand: one multiply-add more than the real path)
            fb[i] = atan_small(im, fmaxf(re, fmaf(fabsf(im), 2.4142136f, 1e-30f))) - k.theta;
        }
        if (REFRESH && (r & 3) == 3) {
            // every fourth row the carried state is rebuilt from the row's last 255 samples alone (lanes 48..63, no comb): the
            // comb form never forgets a rounding error, this bounds its age to 4096 samples
            St f = {xt[0], (v2f){0, 0}, xt[0]};
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                ROT_STEP(f, xt[i], ((v2f){0, 0}));
                f.R += xt[i];
            }
            scan_step<1, DPP_ROW_SHR(1), 0x8>(f, k.wc[0], k.ws[0]);
            scan_step<2, DPP_ROW_SHR(2), 0x8>(f, k.wc[1], k.ws[1]);
            scan_step<4, DPP_ROW_SHR(4), 0x8>(f, k.wc[2], k.ws[2]);
            scan_step<8, DPP_ROW_SHR(8), 0x8>(f, k.wc[3], k.ws[3]);
            carry.C = (v2f){rl63(f.C.x), rl63(f.C.y)} - xt[0];        // (the 256th sample back leaves the window)
            carry.S = (v2f){rl63(f.S.x), rl63(f.S.y)};
            carry.R = (v2f){rl63(f.R.x), rl63(f.R.y)} - xt[0];
        }
    };
    for (int r = 0; r < rows; r += 2) {           // (rows is even: the two sample arrays swap roles without a copy)
        row(xa, xb, r);
        row(xb, xa, r + 1);
    }
    float acc = carry.C.x + carry.S.y + carry.R.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += fb[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    int wps = argc > 2 ? atoi(argv[2]) : 3;
    int refresh = argc > 3 ? atoi(argv[3]) : 1;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * wps;                       // 256 threads = 4 waves, one per SIMD: wps workgroups per CU
    const long total_rows = (1L << 26) / 1024;
    const int rows = 2 * (int)((total_rows + blocks * 8 - 1) / (blocks * 8));
    CosfirConsts k;
    const double phi = 2.0 * M_PI / 254.0, w = 2.0 * M_PI * 25000.0 / 2400000.0;
    k.c = (float)cos(phi); k.s = (float)sin(phi);
    for (int i = 0; i < 16; ++i) k.nco[i] = (v2f){(float)cos(w * i), (float)-sin(w * i)};
    k.rowstep = (v2f){(float)cos(w * 1024), (float)-sin(w * 1024)};
    for (int j = 0; j < 4; ++j) { k.wc[j] = (float)cos(16 * phi * (1 << j)); k.ws[j] = (float)sin(16 * phi * (1 << j)); }
    k.a0 = 0.54f; k.a1 = -0.46f; k.theta = (float)w; k.eps = 1e-3f;
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
        if (refresh) hipLaunchKernelGGL(k_cosfir_arith<true>, dim3(blocks), dim3(256), 0, 0, out, rows, k, seed);
        else hipLaunchKernelGGL(k_cosfir_arith<false>, dim3(blocks), dim3(256), 0, 0, out, rows, k, seed);
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
    printf("cosfir_arith: %d CUs, %d workgroups of 4 waves (%d waves per SIMD asked), %d rows of 1024 samples per wave (%.3f x 2^26 samples per launch), refresh %d\n",
           cus, blocks, wps, rows, (double)rows * blocks * 4 * 1024 / (1L << 26), refresh);
    printf("cosfir_arith: %ld launches, %.4f ms per launch (HIP events), %.4f ms per 2^26 samples, out[0] = %g\n", launches, ev_ms / launches,
           ev_ms / launches / ((double)rows * blocks * 4 * 1024 / (1L << 26)), h0);
    return 0;
}
