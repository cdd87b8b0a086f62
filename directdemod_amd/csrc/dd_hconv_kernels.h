// The accurate-sync envelope stage as three launches (included by dd_audio.hip only).
//
// decode_noaa.py:852 takes abs(scipy.signal.hilbert(x)) of the FM audio of every search window (N = 118 151 samples at
// 2.048 MS/s).  dd_audio.hip writes that as x + j (x (*) hh) with the length-N circular convolution embedded in a cyclic
// one of length M = 2^18 (see "the envelope as one real convolution" there).  Through the FFT library one batch of 64
// windows was nine passes over [64][M] arrays (pad, three kernels of the real-to-complex transform, the spectrum product,
// three of the complex-to-real transform, hypot): 0.41 ms of a 0.84 ms batch (profiles/r04_noaa_kernel_stats.csv).
// Here the same arithmetic in float64 is three passes:
//   * two windows share one complex transform, z = x_a + j x_b: hh is real, so the convolution leaves the two in the
//     real and the imaginary part -- no real-transform pre/post-processing;
//   * M = 512 x 512 (four-step form, n = 512 n1 + n2, k = k1 + 512 k2):
//       k_hc_cols_fwd   FM angle of the filtered IQ pair straight from the c64 rows (the padded f64 copy is never
//                       written), transform over n1 for eight neighbouring columns per workgroup (the tile goes
//                       through LDS so that global rows are read and written as 128-byte pieces), rows of zeros
//                       beyond the window are not loaded;
//       k_hc_rows       per row k1 (one wave, 8 KB, contiguous): times W_M^{n2 k1}, transform over n2, times the
//                       kernel spectrum (1/M folded in, stored in this [k1][k2] order once per length), inverse
//                       transform over k2, times W_M^{-n2 k1}, in place;
//       k_hc_cols_inv   inverse transform over k1 per column, hypot(x, y) for the n < N outputs only.
//     Traffic per window: 0.95 + 2.1 | 2.1 + 2.1 | 2.1 + 0.95 + 0.95 MB = 11.2 MB against ~29 MB.
//   * a 512-point transform = radix 8 x 8 x 8 on one wave (8 points per lane in registers, two exchanges through 8 KB of LDS,
//     XOR-swizzled so that every 16-lane group of a 16-byte access covers the 64 banks once; no workgroup barrier inside).
//     Twiddles come from two 512-entry tables computed on the host in long double (W_512^j and W_M^j), at most one
//     product of two of them per factor.
#pragma once
#include "dd_common.h"
#include "dd_chain_kernels.h"

#define DD_HC_N 512
#define DD_HC_M (DD_HC_N * DD_HC_N)
#define DD_HC_COLS 8
#define DD_HC_LDS_COLS (DD_HC_COLS * DD_HC_N * 16)        // 65536

__device__ __forceinline__ double2 hc_add(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 hc_sub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 hc_mul(double2 a, double2 b) { return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x)); }
__device__ __forceinline__ double2 hc_mulconj(double2 a, double2 b) { return make_double2(fma(a.x, b.x, a.y * b.y), fma(a.y, b.x, -a.x * b.y)); }
template <bool INV>
__device__ __forceinline__ double2 hc_tw(double2 a, double2 w) { return INV ? hc_mulconj(a, w) : hc_mul(a, w); }
// times -j (forward) / +j (inverse)
template <bool INV>
__device__ __forceinline__ double2 hc_rot(double2 a) { return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x); }

// 8-point DFT, X[k] = sum x[n] e^{-+2 pi j n k / 8}, in place, natural order in and out
template <bool INV>
__device__ __forceinline__ void hc_dft8(double2 (&v)[8]) {
    const double r = 0.70710678118654752440;
    const double2 t0 = hc_add(v[0], v[4]), t4 = hc_sub(v[0], v[4]);
    const double2 t1 = hc_add(v[1], v[5]), d5 = hc_sub(v[1], v[5]);
    const double2 t2 = hc_add(v[2], v[6]), d6 = hc_sub(v[2], v[6]);
    const double2 t3 = hc_add(v[3], v[7]), d7 = hc_sub(v[3], v[7]);
    // d5 W8, d6 W8^2, d7 W8^3 (conjugated for the inverse)
    const double2 t5 = INV ? make_double2((d5.x - d5.y) * r, (d5.x + d5.y) * r) : make_double2((d5.x + d5.y) * r, (d5.y - d5.x) * r);
    const double2 t6 = hc_rot<INV>(d6);
    const double2 t7 = INV ? make_double2((-d7.x - d7.y) * r, (d7.x - d7.y) * r) : make_double2((d7.y - d7.x) * r, (-d7.x - d7.y) * r);
    const double2 u0 = hc_add(t0, t2), u2 = hc_sub(t0, t2), u1 = hc_add(t1, t3), u3 = hc_rot<INV>(hc_sub(t1, t3));
    const double2 w4 = hc_add(t4, t6), w6 = hc_sub(t4, t6), w5 = hc_add(t5, t7), w7 = hc_rot<INV>(hc_sub(t5, t7));
    v[0] = hc_add(u0, u1); v[4] = hc_sub(u0, u1); v[2] = hc_add(u2, u3); v[6] = hc_sub(u2, u3);
    v[1] = hc_add(w4, w5); v[5] = hc_sub(w4, w5); v[3] = hc_add(w6, w7); v[7] = hc_sub(w6, w7);
}

// this lane's stage twiddles: tw1[ka - 1] = W_512^{lane ka}, tw2[kb - 1] = W_64^{(lane & 7) kb}
__device__ __forceinline__ void hc_lane_twiddles(const double2* __restrict__ TA, int lane, double2 (&tw1)[7], double2 (&tw2)[7]) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        tw1[k - 1] = TA[lane * k];
        tw2[k - 1] = TA[8 * (lane & 7) * k];
    }
}

// 512-point transform on one wave.  In: lane l, register a = x[l + 64 a]; out: lane l, register c = X[l + 64 c].
// Index split n = 64 a + 8 b + c, k = ka + 8 kb + 64 kc:
//   W^{nk} = W8^{a ka} . W512^{(8b+c) ka} . W8^{b kb} . W64^{c kb} . W8^{c kc}
// S: 512 elements of LDS owned by this wave (a wave's LDS operations execute in order).
template <bool INV>
__device__ __forceinline__ void hc_fft512(double2 (&v)[8], double2* __restrict__ S, const double2 (&tw1)[7], const double2 (&tw2)[7], int lane) {
    hc_dft8<INV>(v);
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = hc_tw<INV>(v[k], tw1[k - 1]);
    // exchange 1: element (ka, b, c) at 64 ka + ((8 b + c) ^ ((ka & 1) << 3)); lane (b, c) writes its ka, lane (ka, c) reads its b
#pragma unroll
    for (int k = 0; k < 8; ++k) S[64 * k + (lane ^ ((k & 1) << 3))] = v[k];
    const int hi = lane >> 3, c = lane & 7;
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] = S[64 * hi + ((8 * b + c) ^ ((hi & 1) << 3))];
    hc_dft8<INV>(v);
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = hc_tw<INV>(v[k], tw2[k - 1]);
    // exchange 2: element (ka, kb, c) at 64 kb + 8 ka + (c ^ ((ka & 6) | (kb & 1))); lane (ka, c) writes its kb, lane ka + 8 kb reads its c
#pragma unroll
    for (int k = 0; k < 8; ++k) S[64 * k + 8 * hi + (c ^ ((hi & 6) | (k & 1)))] = v[k];
    const int ka = lane & 7, kb = lane >> 3;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = S[64 * kb + 8 * ka + (q ^ ((ka & 6) | (kb & 1)))];
    hc_dft8<INV>(v);
}

// workgroup x of a pass over 64 column tiles -> tile: the eight workgroups that land on one XCD (x mod 8) take eight
// neighbouring tiles, whose 64-byte halves of the c64 / f64 rows then meet in that XCD's L2
__device__ __forceinline__ int hc_tile_of(int bx) { return (bx & 7) * 8 + (bx >> 3); }

// ---- pass 1: columns, forward.  grid (64, pairs), 512 threads, DD_HC_LDS_COLS bytes of dynamic LDS.
// X: filtered IQ, c64 [nwin][L]; window w's audio is x[n] = angle(X[n+1] conj X[n]), n < L2 = L - 1 (demod_fm.py:40-49)
__global__ void __launch_bounds__(512) k_hc_cols_fwd(const float2* __restrict__ X, int64_t L, int64_t L2, int nwin,
                                                      double2* __restrict__ T, const double2* __restrict__ TA) {
    extern __shared__ __attribute__((aligned(16))) char hc_smem[];
    double2* S = reinterpret_cast<double2*>(hc_smem);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, col = t & 7;
    const int c0 = hc_tile_of(blockIdx.x) * DD_HC_COLS, pair = blockIdx.y;
    const bool hasb = 2 * pair + 1 < nwin;
    const float2* xa = X + (int64_t)(2 * pair) * L;
    const float2* xb = X + (int64_t)(2 * pair + (hasb ? 1 : 0)) * L;
    const int R = (int)((L2 + DD_HC_N - 1) / DD_HC_N);                 // rows that hold samples
    for (int row = t >> 3; row < R; row += 64) {
        const int64_t n = (int64_t)row * DD_HC_N + c0 + col;
        double a = 0.0, b = 0.0;
        if (n < L2) {
            a = (double)dd_fm_angle(xa[n + 1], xa[n]);
            if (hasb) b = (double)dd_fm_angle(xb[n + 1], xb[n]);
        }
        S[col * DD_HC_N + (row ^ (col << 1))] = make_double2(a, b);
    }
    __syncthreads();
    double2 tw1[7], tw2[7];
    hc_lane_twiddles(TA, lane, tw1, tw2);
    double2* Sw = S + wv * DD_HC_N;
    double2 v[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int r = lane + 64 * a;
        v[a] = r < R ? Sw[r ^ (wv << 1)] : make_double2(0.0, 0.0);
    }
    hc_fft512<false>(v, Sw, tw1, tw2, lane);
#pragma unroll
    for (int k = 0; k < 8; ++k) Sw[(lane + 64 * k) ^ (wv << 1)] = v[k];
    __syncthreads();
    double2* Tp = T + (int64_t)pair * DD_HC_M + c0 + col;
    for (int row = t >> 3; row < DD_HC_N; row += 64) Tp[(int64_t)row * DD_HC_N] = S[col * DD_HC_N + (row ^ (col << 1))];
}

// ---- pass 2: rows.  grid (128, pairs), 256 threads (one wave per row), 32 KB of LDS
__global__ void __launch_bounds__(256) k_hc_rows(double2* __restrict__ T, const double2* __restrict__ HHp,
                                                  const double2* __restrict__ TA, const double2* __restrict__ TB) {
    __shared__ __attribute__((aligned(16))) double2 S[4 * DD_HC_N];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k1 = blockIdx.x * 4 + wv;
    double2* row = T + ((int64_t)blockIdx.y * DD_HC_N + k1) * DD_HC_N + lane;
    double2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = row[64 * q];
    // W_M^{(lane + 64 q) k1} = W_M^{lane k1} . W_M^{64 q k1}; W_M^{512 h + l} = W_512^h . W_M^l
    double2 tw[8];
    {
        const int i1 = lane * k1;
        const double2 wl = hc_mul(TA[i1 >> 9], TB[i1 & 511]);
        tw[0] = wl;
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            const int i2 = q * k1;
            tw[q] = hc_mul(wl, hc_mul(TA[i2 >> 3], TB[64 * (i2 & 7)]));
        }
    }
    double2 tw1[7], tw2[7];
    hc_lane_twiddles(TA, lane, tw1, tw2);
    double2* Sw = S + wv * DD_HC_N;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = hc_mul(v[q], tw[q]);
    hc_fft512<false>(v, Sw, tw1, tw2, lane);
    const double2* h = HHp + (int64_t)k1 * DD_HC_N + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = hc_mul(v[q], h[64 * q]);
    hc_fft512<true>(v, Sw, tw1, tw2, lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) row[64 * q] = hc_mulconj(v[q], tw[q]);
}

// ---- pass 3: columns, inverse, envelope.  grid (64, pairs), 512 threads, DD_HC_LDS_COLS bytes of dynamic LDS
__global__ void __launch_bounds__(512) k_hc_cols_inv(const double2* __restrict__ T, const float2* __restrict__ X, int64_t L, int64_t L2, int nwin,
                                                      double* __restrict__ ENV, const double2* __restrict__ TA) {
    extern __shared__ __attribute__((aligned(16))) char hc_smem[];
    double2* S = reinterpret_cast<double2*>(hc_smem);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, col = t & 7;
    const int c0 = hc_tile_of(blockIdx.x) * DD_HC_COLS, pair = blockIdx.y;
    const bool hasb = 2 * pair + 1 < nwin;
    const double2* Tp = T + (int64_t)pair * DD_HC_M + c0 + col;
    for (int row = t >> 3; row < DD_HC_N; row += 64) S[col * DD_HC_N + (row ^ (col << 1))] = Tp[(int64_t)row * DD_HC_N];
    __syncthreads();
    double2 tw1[7], tw2[7];
    hc_lane_twiddles(TA, lane, tw1, tw2);
    double2* Sw = S + wv * DD_HC_N;
    double2 v[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) v[a] = Sw[(lane + 64 * a) ^ (wv << 1)];
    hc_fft512<true>(v, Sw, tw1, tw2, lane);
    const int R = (int)((L2 + DD_HC_N - 1) / DD_HC_N);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = lane + 64 * k;
        if (r < R) Sw[r ^ (wv << 1)] = v[k];
    }
    __syncthreads();
    const float2* xa = X + (int64_t)(2 * pair) * L;
    const float2* xb = X + (int64_t)(2 * pair + (hasb ? 1 : 0)) * L;
    double* ea = ENV + (int64_t)(2 * pair) * L2;
    double* eb = ENV + (int64_t)(2 * pair + 1) * L2;
    for (int row = t >> 3; row < R; row += 64) {
        const int64_t n = (int64_t)row * DD_HC_N + c0 + col;
        if (n >= L2) continue;
        const double2 y = S[col * DD_HC_N + (row ^ (col << 1))];
        ea[n] = hypot((double)dd_fm_angle(xa[n + 1], xa[n]), y.x);
        if (hasb) eb[n] = hypot((double)dd_fm_angle(xb[n + 1], xb[n]), y.y);
    }
}

// kernel spectrum in the order k_hc_rows multiplies it: HHp[512 k1 + k2] = HH[k1 + 512 k2], HH the M/2 + 1 bins of a real
// sequence's spectrum (Hermitian extension above M/2)
__global__ void __launch_bounds__(256) k_hc_perm(const double2* __restrict__ HH, double2* __restrict__ HHp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= DD_HC_M) return;
    const int k = (i >> 9) + DD_HC_N * (i & 511);
    const double2 v = k <= DD_HC_M / 2 ? HH[k] : HH[DD_HC_M - k];
    HHp[i] = k <= DD_HC_M / 2 ? v : make_double2(v.x, -v.y);
}
