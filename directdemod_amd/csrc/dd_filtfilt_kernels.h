// Zero-phase FIR (scipy.signal.filtfilt(b, [1], x), filters.py:72-73) device kernels, shared by the
// stand-alone entry points (dd_fir.hip) and the batched accurate-sync chain (dd_audio.hip).
//
//   ext = odd_ext(x, edge = 3K); pass 1: y1[i] = sum_k b[k] ext[max(i-k, 0)], i in [0, N = n + 2 edge)
//   (forward lfilter whose history is the pass's first sample, zi * x0); pass 2 is the same filter run
//   over y1 backwards and cropped: out[m] = sum_k b[k] y1[min(m + edge + k, N-1)], m in [0, n).
//
// Tiled form: a workgroup of 256 lanes produces 2048 consecutive outputs from an LDS image of the
// 2048+K-1 inputs they touch.  A lane owns 8 consecutive outputs and walks the taps 8 at a time, so a
// chunk of 64 multiply-adds reads 15 LDS values (a sliding window held in registers) -- the ratio at
// which the float64 FMA pipes and the LDS port take equally long.  The image is skewed by one element
// per 8 (lane stride 9 elements = 18 banks) so the lanes of a ds_read_b64 fall on distinct banks.
// Every output accumulates its taps in ascending k with one fma each, like the one-lane-per-output form
// it replaces (kept below for 16-byte elements and very long filters): results are bit-identical.
#pragma once
#include "dd_common.h"

template <typename T> struct dd_acc;
template <> struct dd_acc<double> {
    typedef double tap_t;
    __device__ static double zero() { return 0.0; }
    __device__ static double mad(double t, double v, double a) { return fma(t, v, a); }
    __device__ static double oddext(double e, double v) { return 2.0 * e - v; }
};
template <> struct dd_acc<double2> {
    typedef double tap_t;
    __device__ static double2 zero() { return make_double2(0.0, 0.0); }
    __device__ static double2 mad(double t, double2 v, double2 a) { return make_double2(fma(t, v.x, a.x), fma(t, v.y, a.y)); }
    __device__ static double2 oddext(double2 e, double2 v) { return make_double2(2.0 * e.x - v.x, 2.0 * e.y - v.y); }
};
template <> struct dd_acc<float2> {
    typedef float tap_t;
    __device__ static float2 zero() { return make_float2(0.f, 0.f); }
    __device__ static float2 mad(float t, float2 v, float2 a) { return make_float2(fmaf(t, v.x, a.x), fmaf(t, v.y, a.y)); }
    __device__ static float2 oddext(float2 e, float2 v) { return make_float2(2.f * e.x - v.x, 2.f * e.y - v.y); }
};

// ext[i], i in [0, n + 2*edge): odd extension of x about both ends
template <typename T>
__device__ __forceinline__ T dd_ext_at(const T* __restrict__ x, int64_t n, int edge, int64_t i) {
    if (i < edge) return dd_acc<T>::oddext(x[0], x[edge - i]);
    if (i < edge + n) return x[i - edge];
    return dd_acc<T>::oddext(x[n - 1], x[n - 2 - (i - edge - n)]);
}

// ---- one lane per output (any element size, any K)
template <typename T>
__global__ void __launch_bounds__(256) k_filtfilt_fwd(const T* __restrict__ x, T* __restrict__ y1, int64_t n, int edge,
                                                      const double* __restrict__ taps, int K) {
    typedef typename dd_acc<T>::tap_t tap_t;
    const int64_t N = n + 2 * (int64_t)edge;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    T acc = dd_acc<T>::zero();
    for (int k = 0; k < K; ++k) {
        const int64_t j = i - k;
        acc = dd_acc<T>::mad((tap_t)taps[k], dd_ext_at(x, n, edge, j > 0 ? j : 0), acc);
    }
    y1[i] = acc;
}
template <typename T>
__global__ void __launch_bounds__(256) k_filtfilt_bwd(const T* __restrict__ y1, T* __restrict__ out, int64_t n, int edge,
                                                      const double* __restrict__ taps, int K) {
    typedef typename dd_acc<T>::tap_t tap_t;
    const int64_t N = n + 2 * (int64_t)edge;
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    T acc = dd_acc<T>::zero();
    for (int k = 0; k < K; ++k) {
        const int64_t j = m + edge + k;
        acc = dd_acc<T>::mad((tap_t)taps[k], y1[j < N - 1 ? j : N - 1], acc);
    }
    out[m] = acc;
}

// ---- tiled, batched (8-byte elements: double, float2)
#define DD_FF_R 8
#define DD_FF_THREADS 256
#define DD_FF_TILE (DD_FF_R * DD_FF_THREADS)
#define DD_FF_LDS_MAX (64 * 1024)

__host__ __device__ __forceinline__ int dd_ff_phys(int j) { return j + (j >> 3); }
static inline size_t dd_ff_lds_bytes(int K) { return (size_t)(dd_ff_phys(DD_FF_TILE + K - 1) + 1) * 8; }
static inline bool dd_ff_tiled_ok(int K, size_t elem_bytes) { return elem_bytes == 8 && dd_ff_lds_bytes(K) <= DD_FF_LDS_MAX; }

// blockIdx.y = window of the batch; src/dst advance by their strides (elements) per window.
// MODE 0: zero-phase pass 1, MODE 1: zero-phase pass 2, MODE 2: plain causal FIR y[i] = sum_k b[k] x[i-k] whose
// samples before the array come from `hist` (the K-1 inputs that preceded it: filters.py:64-70 with the state kept).
template <typename T, int MODE>
__global__ void __launch_bounds__(DD_FF_THREADS) k_filtfilt_tile(const T* __restrict__ src, T* __restrict__ dst, int64_t n, int edge,
                                                                 const double* __restrict__ taps, int K,
                                                                 int64_t src_stride, int64_t dst_stride, const T* __restrict__ hist = nullptr) {
    constexpr bool BWD = MODE == 1;
    typedef typename dd_acc<T>::tap_t tap_t;
    static_assert(sizeof(T) == 8, "tiled filtfilt: 8-byte elements");
    extern __shared__ double dd_ff_smem[];
    T* s = reinterpret_cast<T*>(dd_ff_smem);
    constexpr int R = DD_FF_R, U = 8;
    const int64_t N = n + 2 * (int64_t)edge;
    const int64_t nout = MODE == 0 ? N : n;
    const T* x = src + (int64_t)blockIdx.y * src_stride;
    T* y = dst + (int64_t)blockIdx.y * dst_stride;
    const int64_t o0 = (int64_t)blockIdx.x * DD_FF_TILE;
    const int W = DD_FF_TILE + K - 1;
    for (int j = threadIdx.x; j < W; j += DD_FF_THREADS) {
        T v;
        if (MODE == 2) {                  // s[j] = x[o0 - (K-1) + j], the carried history before the array
            const int64_t i = o0 - (K - 1) + j;
            v = i >= 0 ? (i < n ? x[i] : dd_acc<T>::zero()) : hist[(K - 1) + i];
        } else if (!BWD) {                // s[j] = ext[max(o0 - (K-1) + j, 0)]
            int64_t i = o0 - (K - 1) + j;
            if (i < 0) i = 0;
            v = i < N ? dd_ext_at(x, n, edge, i) : dd_acc<T>::zero();
        } else {                          // s[j] = y1[min(o0 + edge + j, N-1)]
            int64_t i = o0 + edge + j;
            if (i > N - 1) i = N - 1;
            v = x[i];
        }
        s[dd_ff_phys(j)] = v;
    }
    __syncthreads();
    const int t0 = threadIdx.x * R;
    T acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = dd_acc<T>::zero();
    // fwd: out[t0+r] = sum_k b[k] s[t0 + r + K-1 - k]      bwd: out[t0+r] = sum_k b[k] s[t0 + r + k]
    // Taps in ascending k throughout.  The K % 8 taps that do not fill a chunk go first (fwd) or last (bwd), so
    // that every chunk's window starts at a logical index that is a multiple of 8: its 15 skewed LDS addresses
    // are then one base register plus compile-time offsets c + (c >> 3) -- no address arithmetic per read.
    const int KR = K % U;
    if (!BWD) {
        for (int k = 0; k < KR; ++k) {
            const tap_t b = (tap_t)taps[k];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = dd_acc<T>::mad(b, s[dd_ff_phys(t0 + r + K - 1 - k)], acc[r]);
        }
    }
    const int kbeg = BWD ? 0 : KR, kend = BWD ? K - KR : K;
    for (int k0 = kbeg; k0 < kend; k0 += U) {
        const int lo = BWD ? (t0 + k0) : (t0 + K - 1 - k0 - (U - 1));      // multiple of 8 (t0, k0 - kbeg and K - KR are)
        const T* w = s + dd_ff_phys(lo);
        T win[R + U - 1];
#pragma unroll
        for (int c = 0; c < R + U - 1; ++c) win[c] = w[c + (c >> 3)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const tap_t b = (tap_t)taps[k0 + u];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = dd_acc<T>::mad(b, win[BWD ? (r + u) : (U - 1 + r - u)], acc[r]);
        }
    }
    if (BWD) {
        for (int k = K - KR; k < K; ++k) {
            const tap_t b = (tap_t)taps[k];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = dd_acc<T>::mad(b, s[dd_ff_phys(t0 + r + k)], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t o = o0 + t0 + r;
        if (o < nout) y[o] = acc[r];
    }
}

// both passes for `batch` windows of n samples: in [batch][n] (stride in_stride) -> out (stride out_stride);
// y1 holds batch * (n + 6K) elements.  taps: device, float64.
template <typename T>
static inline void dd_filtfilt_launch(const T* in, int64_t in_stride, T* y1, T* out, int64_t out_stride, int64_t n, int K,
                                      const double* taps_dev, int batch, hipStream_t s) {
    const int edge = 3 * K;
    const int64_t N = n + 2 * (int64_t)edge;
    const size_t lds = dd_ff_lds_bytes(K);
    hipLaunchKernelGGL((k_filtfilt_tile<T, 0>), dim3((unsigned)((N + DD_FF_TILE - 1) / DD_FF_TILE), batch), dim3(DD_FF_THREADS),
                       lds, s, in, y1, n, edge, taps_dev, K, in_stride, N, (const T*)nullptr);
    hipLaunchKernelGGL((k_filtfilt_tile<T, 1>), dim3((unsigned)((n + DD_FF_TILE - 1) / DD_FF_TILE), batch), dim3(DD_FF_THREADS),
                       lds, s, (const T*)y1, out, n, edge, taps_dev, K, N, out_stride, (const T*)nullptr);
}
