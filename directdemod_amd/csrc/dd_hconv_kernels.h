// A float64 cyclic convolution of length 2^17 / 2^18 with a fixed kernel spectrum as three launches (included by dd_audio.hip only):
// first user the accurate-sync envelope stage (below), second the chirp-z resampler's convolution (dd_audio.hip, HcCztSrc / HcCztDst).
//
// decode_noaa.py:852 takes abs(scipy.signal.hilbert(x)) of the FM audio of every search window (N = 118 151 samples at
// 2.048 MS/s).  dd_audio.hip writes that as x + j (x (*) hh) with the length-N circular convolution embedded in a cyclic
// one of length M = 2^18 (see "the envelope as one real convolution" there).  Through the FFT library one batch of 64
// windows was nine passes over [64][M] arrays (pad, three kernels of the real-to-complex transform, the spectrum product,
// three of the complex-to-real transform, hypot): 0.41 ms of a 0.84 ms batch (profiles/r04_noaa_kernel_stats.csv).
// Here the same arithmetic in float64 is three passes:
//   * two windows share one complex transform, z = x_a + j x_b: hh is real, so the convolution leaves the two in the
//     real and the imaginary part -- no real-transform pre/post-processing;
//   * M = 512 rows x N2 columns, N2 = 512 (M = 2^18) or 256 (M = 2^17) (four-step form, n = N2 n1 + n2, k = k1 + 512 k2;
//     written out for N2 = 512):
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
//     Rows of 256: a wave takes four of them, sixteen lanes and sixteen registers per row, radix 16 x 16 with one exchange (k_hc_rows256).
//     Twiddles come from small tables computed on the host in long double (W_512^j and W_M^j, j < N2), at most one
//     product of two of them per factor.
//   * the kernels are templates over their source (what element n of image `job` is), their spectrum and their sink (what becomes of
//     element n of the result): HcEnvIO below for the envelope, HcCztSrc / HcCztDst in dd_audio.hip for the resampler's chirp convolution.
#pragma once
#include "dd_common.h"
#include "dd_chain_kernels.h"

#define DD_HC_N 512                 // column length (rows of an image); the row length is 512 (M = 2^18) or 256 (M = 2^17)
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

// 16-point DFT (n = 4 n1 + n0, k = k1 + 4 k0: W16^{nk} = W4^{n1 k1} . W16^{n0 k1} . W4^{n0 k0}), in place, natural order
template <bool INV>
__device__ __forceinline__ void hc_dft4(double2& a, double2& b, double2& c, double2& d) {
    const double2 s0 = hc_add(a, c), d0 = hc_sub(a, c), s1 = hc_add(b, d), d1 = hc_rot<INV>(hc_sub(b, d));
    a = hc_add(s0, s1); c = hc_sub(s0, s1); b = hc_add(d0, d1); d = hc_sub(d0, d1);
}
template <bool INV>
__device__ __forceinline__ void hc_dft16(double2 (&v)[16]) {
    const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, r = 0.70710678118654752440;
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) hc_dft4<INV>(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);       // v[4 k1 + n0] = y[n0][k1]
    // y[n0][k1] W16^{n0 k1}: exponents 1, 2, 3 | 2, 4, 6 | 3, 6, 9
    const double2 w1 = make_double2(c1, -s1), w2 = make_double2(r, -r), w3 = make_double2(s1, -c1);
    const double2 w6 = make_double2(-r, -r), w9 = make_double2(-c1, s1);
    v[4 + 1] = hc_tw<INV>(v[4 + 1], w1); v[8 + 1] = hc_tw<INV>(v[8 + 1], w2); v[12 + 1] = hc_tw<INV>(v[12 + 1], w3);
    v[4 + 2] = hc_tw<INV>(v[4 + 2], w2); v[8 + 2] = hc_rot<INV>(v[8 + 2]);    v[12 + 2] = hc_tw<INV>(v[12 + 2], w6);
    v[4 + 3] = hc_tw<INV>(v[4 + 3], w3); v[8 + 3] = hc_tw<INV>(v[8 + 3], w6); v[12 + 3] = hc_tw<INV>(v[12 + 3], w9);
    // X[k1 + 4 k0] = DFT4 over n0 of y'[n0][k1]: in place on (v[4 k1], v[4 k1 + 1], v[4 k1 + 2], v[4 k1 + 3]) -> k0 = 0..3
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) hc_dft4<INV>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    // v[4 k1 + k0] holds X[k1 + 4 k0]: transpose the 4 x 4 index
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) { const double2 t = v[4 * a + b]; v[4 * a + b] = v[4 * b + a]; v[4 * b + a] = t; }
}

// 256-point transform on 16 lanes (four of them per wave).  In: lane l, register a = x[l + 16 a]; out: lane l, register b = X[l + 16 b].
// n = 16 a + l, k = ka + 16 kb: W^{nk} = W16^{a ka} . W256^{l ka} . W16^{l kb}.  S: 256 elements of LDS owned by this 16-lane group; the
// one exchange stores element (ka, l) at 16 ka + (l ^ ka) (the XOR keeps the transposed read off one bank group).
// ts[ka - 1] = W_256^{l ka}.
template <bool INV>
__device__ __forceinline__ void hc_fft256(double2 (&v)[16], double2* __restrict__ S, const double2 (&ts)[15], int l) {
    hc_dft16<INV>(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = hc_tw<INV>(v[k], ts[k - 1]);
#pragma unroll
    for (int k = 0; k < 16; ++k) S[16 * k + (l ^ k)] = v[k];
#pragma unroll
    for (int b = 0; b < 16; ++b) v[b] = S[16 * l + (b ^ l)];
    hc_dft16<INV>(v);
}

// Geometry of a 512 x N2 image (N2 = 2^LG columns = row length; 512 rows = column length): n = N2 n1 + n2, k = k1 + 512 k2
template <int LG>
struct HcG {
    static constexpr int N2 = 1 << LG;
    static constexpr int64_t M = (int64_t)DD_HC_N << LG;
    static constexpr int TILES = N2 / DD_HC_COLS;
};
// workgroup x of a pass over the column tiles -> tile: the workgroups that land on one XCD (x mod 8) take neighbouring tiles, whose
// 64-byte halves of narrow source rows then meet in that XCD's L2
template <int LG>
__device__ __forceinline__ int hc_tile_of(int bx) { return (bx & 7) * (HcG<LG>::TILES / 8) + (bx >> 3); }

// ---- pass 1: columns, forward.  grid (N2 / 8, jobs), 512 threads, DD_HC_LDS_COLS bytes of dynamic LDS.
// SRC: rows(job) = how many rows of the image hold samples (the rest is zero and is not loaded), at(job, n) = element n.
template <int LG, typename SRC>
__global__ void __launch_bounds__(512) k_hc_cols_fwd(const SRC src, double2* __restrict__ T, const double2* __restrict__ TA) {
    constexpr int N2 = HcG<LG>::N2;
    extern __shared__ __attribute__((aligned(16))) char hc_smem[];
    double2* S = reinterpret_cast<double2*>(hc_smem);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, col = t & 7;
    const int c0 = hc_tile_of<LG>(blockIdx.x) * DD_HC_COLS, job = blockIdx.y;
    const int R = src.rows(job, N2);
    for (int row = t >> 3; row < R; row += 64) S[col * DD_HC_N + (row ^ (col << 1))] = src.at(job, (int64_t)row * N2 + c0 + col);
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
    double2* Tp = T + (int64_t)job * HcG<LG>::M + c0 + col;
    for (int row = t >> 3; row < DD_HC_N; row += 64) Tp[(int64_t)row * N2] = S[col * DD_HC_N + (row ^ (col << 1))];
}

// ---- pass 2: rows of 512 (LG = 9).  grid (128, jobs), 256 threads (one wave per row), 32 KB of LDS.  SPEC: ptr(job) = the kernel
// spectrum in [k1][k2] order with every constant factor folded in
template <typename SPEC>
__global__ void __launch_bounds__(256) k_hc_rows(double2* __restrict__ T, const SPEC spec,
                                                  const double2* __restrict__ TA, const double2* __restrict__ TB) {
    __shared__ __attribute__((aligned(16))) double2 S[4 * DD_HC_N];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k1 = blockIdx.x * 4 + wv;
    double2* row = T + ((int64_t)blockIdx.y * DD_HC_N + k1) * DD_HC_N + lane;
    double2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = row[64 * q];
    // W_M^{(lane + 64 q) k1} = W_M^{lane k1} . W_M^{64 q k1}; W_M^{512 h + l} = W_512^h . W_M^l   (M = 2^18, TB[j] = W_M^j)
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
    const double2* h = spec.ptr(blockIdx.y) + (int64_t)k1 * DD_HC_N + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = hc_mul(v[q], h[64 * q]);
    hc_fft512<true>(v, Sw, tw1, tw2, lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) row[64 * q] = hc_mulconj(v[q], tw[q]);
}

// ---- pass 2: rows of 256 (LG = 8, M = 2^17).  grid (64, jobs), 128 threads: a wave takes four rows, sixteen lanes and sixteen
// registers per row, 4 KB of LDS per row.  TB[j] = W_M^j, j < 256: W_M^{256 h + l} = W_512^h . W_M^l
template <typename SPEC>
__global__ void __launch_bounds__(128) k_hc_rows256(double2* __restrict__ T, const SPEC spec,
                                                     const double2* __restrict__ TA, const double2* __restrict__ TB) {
    __shared__ __attribute__((aligned(16))) double2 S[8 * 256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, l = lane & 15;
    const int k1 = blockIdx.x * 8 + wv * 4 + g;
    double2* row = T + ((int64_t)blockIdx.y * DD_HC_N + k1) * 256 + l;
    double2 v[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) v[a] = row[16 * a];
    double2 tw[16];
    {
        const int i1 = l * k1;
        const double2 wl = hc_mul(TA[i1 >> 8], TB[i1 & 255]);
        tw[0] = wl;
#pragma unroll
        for (int a = 1; a < 16; ++a) {
            const int i2 = a * k1;                                 // W_M^{16 a k1} = W_512^{i2 >> 4} . W_M^{16 (i2 & 15)}
            tw[a] = hc_mul(wl, hc_mul(TA[i2 >> 4], TB[16 * (i2 & 15)]));
        }
    }
    double2 ts[15];
#pragma unroll
    for (int k = 1; k < 16; ++k) ts[k - 1] = TA[2 * l * k];
    double2* Sg = S + (wv * 4 + g) * 256;
#pragma unroll
    for (int a = 0; a < 16; ++a) v[a] = hc_mul(v[a], tw[a]);
    hc_fft256<false>(v, Sg, ts, l);
    const double2* h = spec.ptr(blockIdx.y) + (int64_t)k1 * 256 + l;
#pragma unroll
    for (int b = 0; b < 16; ++b) v[b] = hc_mul(v[b], h[16 * b]);
    hc_fft256<true>(v, Sg, ts, l);
#pragma unroll
    for (int a = 0; a < 16; ++a) row[16 * a] = hc_mulconj(v[a], tw[a]);
}

// ---- pass 3: columns, inverse.  grid (N2 / 8, jobs), 512 threads, DD_HC_LDS_COLS bytes of dynamic LDS.
// DST: rows(job) = how many rows of the result are wanted, put(job, n, y) = what becomes of element n
template <int LG, typename DST>
__global__ void __launch_bounds__(512) k_hc_cols_inv(const double2* __restrict__ T, const DST dst, const double2* __restrict__ TA) {
    constexpr int N2 = HcG<LG>::N2;
    extern __shared__ __attribute__((aligned(16))) char hc_smem[];
    double2* S = reinterpret_cast<double2*>(hc_smem);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, col = t & 7;
    const int c0 = hc_tile_of<LG>(blockIdx.x) * DD_HC_COLS, job = blockIdx.y;
    const double2* Tp = T + (int64_t)job * HcG<LG>::M + c0 + col;
    for (int row = t >> 3; row < DD_HC_N; row += 64) S[col * DD_HC_N + (row ^ (col << 1))] = Tp[(int64_t)row * N2];
    __syncthreads();
    double2 tw1[7], tw2[7];
    hc_lane_twiddles(TA, lane, tw1, tw2);
    double2* Sw = S + wv * DD_HC_N;
    double2 v[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) v[a] = Sw[(lane + 64 * a) ^ (wv << 1)];
    hc_fft512<true>(v, Sw, tw1, tw2, lane);
    const int R = dst.rows(job, N2);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = lane + 64 * k;
        if (r < R) Sw[r ^ (wv << 1)] = v[k];
    }
    __syncthreads();
    for (int row = t >> 3; row < R; row += 64) dst.put(job, (int64_t)row * N2 + c0 + col, S[col * DD_HC_N + (row ^ (col << 1))]);
}

// ---- the accurate-sync envelope's source and sink.  X: filtered IQ, c64 [nwin][L]; window w's audio is x[n] = angle(X[n+1] conj X[n]),
// n < L2 = L - 1 (demod_fm.py:40-49); job p holds windows 2p (real part) and 2p + 1 (imaginary part)
struct HcEnvIO {
    const float2* X;
    int64_t L, L2;
    int nwin;
    double* ENV;          // [nwin][L2] (sink only)
    __device__ int rows(int, int N2) const { return (int)((L2 + N2 - 1) / N2); }
    __device__ double2 at(int pair, int64_t n) const {
        if (n >= L2) return make_double2(0.0, 0.0);
        const float2* xa = X + (int64_t)(2 * pair) * L;
        const double a = (double)dd_fm_angle(xa[n + 1], xa[n]);
        double b = 0.0;
        if (2 * pair + 1 < nwin) b = (double)dd_fm_angle(xa[L + n + 1], xa[L + n]);
        return make_double2(a, b);
    }
    __device__ void put(int pair, int64_t n, double2 y) const {
        if (n >= L2) return;
        const double2 x = at(pair, n);
        ENV[(int64_t)(2 * pair) * L2 + n] = hypot(x.x, y.x);
        if (2 * pair + 1 < nwin) ENV[(int64_t)(2 * pair + 1) * L2 + n] = hypot(x.y, y.y);
    }
};
struct HcOneSpec {
    const double2* p;
    __device__ const double2* ptr(int) const { return p; }
};

// kernel spectrum in the order the row pass multiplies it: out[N2 k1 + k2] = scale . in[k1 + 512 k2]; hermitian: `in` holds the
// M/2 + 1 bins of a real sequence's spectrum (extended by conjugation above M/2), else all M bins
template <int LG>
__global__ void __launch_bounds__(256) k_hc_perm(const double2* __restrict__ in, double2* __restrict__ out, int hermitian, double scale) {
    constexpr int64_t M = HcG<LG>::M;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int64_t k = (i >> LG) + DD_HC_N * (i & (HcG<LG>::N2 - 1));
    double2 v;
    if (hermitian && k > M / 2) { v = in[M - k]; v.y = -v.y; }
    else v = in[k];
    out[i] = make_double2(v.x * scale, v.y * scale);
}

// the three launches for `jobs` images (T: [jobs][512 N2] c128 work buffer; TB: W_M^j for this M)
template <int LG, typename SRC, typename SPEC, typename DST>
static inline void hc_convolve(const SRC& src, const SPEC& spec, const DST& dst, double2* T, int jobs,
                               const double2* TA, const double2* TB, hipStream_t s) {
    hipLaunchKernelGGL((k_hc_cols_fwd<LG, SRC>), dim3(HcG<LG>::TILES, jobs), dim3(512), DD_HC_LDS_COLS, s, src, T, TA);
    if (LG == 9) hipLaunchKernelGGL((k_hc_rows<SPEC>), dim3(DD_HC_N / 4, jobs), dim3(256), 0, s, T, spec, TA, TB);
    else hipLaunchKernelGGL((k_hc_rows256<SPEC>), dim3(DD_HC_N / 8, jobs), dim3(128), 0, s, T, spec, TA, TB);
    hipLaunchKernelGGL((k_hc_cols_inv<LG, DST>), dim3(HcG<LG>::TILES, jobs), dim3(512), DD_HC_LDS_COLS, s, (const double2*)T, dst, TA);
}
template <int LG, typename SRC, typename DST>
static inline hipError_t hc_set_lds_attr() {
    hipError_t e = hipFuncSetAttribute((const void*)k_hc_cols_fwd<LG, SRC>, hipFuncAttributeMaxDynamicSharedMemorySize, DD_HC_LDS_COLS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_hc_cols_inv<LG, DST>, hipFuncAttributeMaxDynamicSharedMemorySize, DD_HC_LDS_COLS);
}
