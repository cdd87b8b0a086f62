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
#include <vector>

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

// MODE 3's source (float2 only): the samples of a search window are computed where pass 1 stages them -- raw IQ (uint8 pairs minus
// 127.5, source.py:117-118, or complex64) at starts[window] + k times the oscillator whose sample index restarts at 0 in every
// window (comm.py:77 on the window's own commSignal) -- instead of being written out by a kernel of their own and read back
struct DDFrontSrc {
    const void* iq;
    const int64_t* starts;        // device, one per window of the batch
    uint64_t cyc;
    const float2* tbl;
    int u8;
};
__device__ __forceinline__ float2 dd_front_at(const DDFrontSrc& F, int64_t g0, int64_t k) {
    float2 v;
    if (F.u8) {
        const uchar2 u = reinterpret_cast<const uchar2*>(F.iq)[g0 + k];
        v = make_float2((float)u.x - 127.5f, (float)u.y - 127.5f);
    } else {
        v = reinterpret_cast<const float2*>(F.iq)[g0 + k];
    }
    return dd_cmul(v, dd_phasor((uint64_t)k * F.cyc, F.tbl));
}

// blockIdx.y = window of the batch; src/dst advance by their strides (elements) per window.
// MODE 0: zero-phase pass 1, MODE 1: zero-phase pass 2, MODE 2: plain causal FIR y[i] = sum_k b[k] x[i-k] whose
// samples before the array come from `hist` (the K-1 inputs that preceded it: filters.py:64-70 with the state kept),
// MODE 3: pass 1 over samples computed from `front` (src unused).
template <typename T, int MODE>
__global__ void __launch_bounds__(DD_FF_THREADS) k_filtfilt_tile(const T* __restrict__ src, T* __restrict__ dst, int64_t n, int edge,
                                                                 const double* __restrict__ taps, int K,
                                                                 int64_t src_stride, int64_t dst_stride, const T* __restrict__ hist = nullptr,
                                                                 const DDFrontSrc front = DDFrontSrc()) {
    constexpr bool BWD = MODE == 1;
    typedef typename dd_acc<T>::tap_t tap_t;
    static_assert(sizeof(T) == 8, "tiled filtfilt: 8-byte elements");
    extern __shared__ double dd_ff_smem[];
    T* s = reinterpret_cast<T*>(dd_ff_smem);
    constexpr int R = DD_FF_R, U = 8;
    const int64_t N = n + 2 * (int64_t)edge;
    const int64_t nout = (MODE == 0 || MODE == 3) ? N : n;
    const T* x = src + (int64_t)blockIdx.y * src_stride;
    T* y = dst + (int64_t)blockIdx.y * dst_stride;
    const int64_t o0 = (int64_t)blockIdx.x * DD_FF_TILE;
    const int W = DD_FF_TILE + K - 1;
    for (int j = threadIdx.x; j < W; j += DD_FF_THREADS) {
        T v;
        if (MODE == 2) {                  // s[j] = x[o0 - (K-1) + j], the carried history before the array
            const int64_t i = o0 - (K - 1) + j;
            v = i >= 0 ? (i < n ? x[i] : dd_acc<T>::zero()) : hist[(K - 1) + i];
        } else if constexpr (MODE == 3) { // the same extension over computed samples
            int64_t i = o0 - (K - 1) + j;
            if (i < 0) i = 0;
            const int64_t g0 = front.starts[blockIdx.y];
            if (i >= N) v = dd_acc<T>::zero();
            else if (i < edge) v = dd_acc<T>::oddext(dd_front_at(front, g0, 0), dd_front_at(front, g0, edge - i));
            else if (i < edge + n) v = dd_front_at(front, g0, i - edge);
            else v = dd_acc<T>::oddext(dd_front_at(front, g0, n - 1), dd_front_at(front, g0, n - 2 - (i - edge - n)));
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

// both passes over windows whose samples pass 1 computes from raw IQ (MODE 3)
static inline void dd_filtfilt_front_launch(const DDFrontSrc& F, float2* y1, float2* out, int64_t out_stride, int64_t n, int K,
                                            const double* taps_dev, int batch, hipStream_t s) {
    const int edge = 3 * K;
    const int64_t N = n + 2 * (int64_t)edge;
    const size_t lds = dd_ff_lds_bytes(K);
    hipLaunchKernelGGL((k_filtfilt_tile<float2, 3>), dim3((unsigned)((N + DD_FF_TILE - 1) / DD_FF_TILE), batch), dim3(DD_FF_THREADS),
                       lds, s, (const float2*)nullptr, y1, n, edge, taps_dev, K, (int64_t)0, N, (const float2*)nullptr, F);
    hipLaunchKernelGGL((k_filtfilt_tile<float2, 1>), dim3((unsigned)((n + DD_FF_TILE - 1) / DD_FF_TILE), batch), dim3(DD_FF_THREADS),
                       lds, s, (const float2*)y1, out, n, edge, taps_dev, K, N, out_stride, (const float2*)nullptr, DDFrontSrc());
}

// ---- zero-phase FIR whose taps are a short cosine series (real float64 data) ----------------------------------------------
// The windows the reference uses as taps (filters.py:101-226: rollingAverage, hamming, blackmanHarris; decode_noaa.py:677
// pre-filters the envelope with hamming(492)) are  b[k] = sum_q a_q cos(2 pi q k / (K-1)),  q = 0 .. Q <= 3.  For such taps a
// window sum is a handful of prefix-sum differences instead of K multiply-adds:
//     sum_k b[k] s[t + k] = sum_q a_q ( cos(w_q t) A_q[t] + sin(w_q t) B_q[t] ),     w_q = 2 pi q / (K-1),
//     A_q[t] = sum_{j=t}^{t+K-1} s[j] cos(w_q j),   B_q[t] = the same with sin
// (b is symmetric, so the forward pass -- taps running backwards over the window -- is the same expression).  The prefix
// sums are LOCAL to a workgroup's window of 2048 + K - 1 staged samples (origin at the window start: magnitudes stay below
// ~2500 x the signal, the differences carry ~1e-15 relative error), phases are table look-ups at j mod (K-1) (exact period),
// and both passes stage their samples exactly like k_filtfilt_tile.  Hamming(492) over 60 windows of 118 151 samples:
// 2 x 150 us (7.1e9 multiply-adds, 60 % of the f64 vector peak) -> the passes become memory-bound.
#include "dd_cosfit.h"
// (cos, sin)(2 pi q r / (K-1)), q = 1..Q, r = 0..K-2, as the kernel reads it: tab[(q - 1) * (K - 1) + r]
static inline void dd_cos_table(int K, int Q, std::vector<double2>& tab) {
    tab.resize((size_t)Q * (K - 1));
    const long double w = 2.0L * 3.14159265358979323846264338327950288L / (long double)(K - 1);
    for (int q = 1; q <= Q; ++q)
        for (int r = 0; r < K - 1; ++r) {
            const long double ph = w * (long double)((long long)q * r % (K - 1));
            tab[(size_t)(q - 1) * (K - 1) + r] = make_double2((double)cosl(ph), (double)sinl(ph));
        }
}

#define DD_FC_THREADS 512           // (eight waves per workgroup: two workgroups of 69 KB per CU then keep 16 waves in flight)
#define DD_FC_WAVES (DD_FC_THREADS / 64)
#define DD_FC_SEG_MAX 6             // staged samples per lane: ceil((TILE + K - 1) / 512)
template <int Q> struct dd_fc_geom { static constexpr int TILE = Q == 1 ? 2048 : 1024; };
static inline size_t dd_fc_lds_bytes(int K, int Q) {
    const int W = (Q == 1 ? 2048 : 1024) + K - 1;
    return sizeof(double) * ((size_t)(1 + 2 * Q) * (W + 1) + (size_t)(1 + 2 * Q) * DD_FC_WAVES) + sizeof(double2) * (size_t)Q * (K - 1);
}
static inline bool dd_fc_ok(int K, int Q) {
    const int W = (Q == 1 ? 2048 : 1024) + K - 1;
    return Q >= 1 && Q <= 3 && (W + DD_FC_THREADS - 1) / DD_FC_THREADS <= DD_FC_SEG_MAX && dd_fc_lds_bytes(K, Q) <= 150 * 1024;
}

// MODE 0: pass 1 (n + 2 edge outputs), MODE 1: pass 2 (n outputs); blockIdx.y = window of the batch
template <int Q, int MODE>
__global__ void __launch_bounds__(DD_FC_THREADS) k_filtfilt_cos(const double* __restrict__ src, double* __restrict__ dst, int64_t n, int edge,
                                                               const DDCosFit fit, const double2* __restrict__ tab_g, int K,
                                                               int64_t src_stride, int64_t dst_stride) {
    constexpr int NS = 1 + 2 * Q, TILE = dd_fc_geom<Q>::TILE;
    extern __shared__ double dd_fc_smem[];
    const int W = TILE + K - 1, P = K - 1;
    double* C = dd_fc_smem;                               // [NS][W + 1]: C[s][j + 1] = sum of stream s over the window's samples 0..j
    double* wsum = C + (size_t)NS * (W + 1);              // [NS][waves] wave totals
    double2* tab = reinterpret_cast<double2*>(wsum + NS * DD_FC_WAVES);
    const int64_t N = n + 2 * (int64_t)edge;
    const int64_t nout = MODE == 0 ? N : n;
    const double* x = src + (int64_t)blockIdx.y * src_stride;
    double* y = dst + (int64_t)blockIdx.y * dst_stride;
    const int64_t o0 = (int64_t)blockIdx.x * TILE;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int i = t; i < Q * P; i += DD_FC_THREADS) tab[i] = tab_g[i];
    // stage the window's samples (coalesced) into stream 0's row, shifted by one
    for (int j = t; j < W; j += DD_FC_THREADS) {
        double v;
        if (MODE == 0) {                  // ext[max(o0 - (K-1) + j, 0)]
            int64_t i = o0 - (K - 1) + j;
            if (i < 0) i = 0;
            v = i < N ? dd_ext_at(x, n, edge, i) : 0.0;
        } else {                          // y1[min(o0 + edge + j, N-1)]
            int64_t i = o0 + edge + j;
            if (i > N - 1) i = N - 1;
            v = x[i];
        }
        C[j + 1] = v;
    }
    __syncthreads();
    // every lane scans its own run of consecutive samples in registers ...
    const int S = (W + DD_FC_THREADS - 1) / DD_FC_THREADS;
    const int j0 = t * S;
    double run[DD_FC_SEG_MAX][NS];
    double tot[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) tot[s] = 0.0;
    int r = j0 % P;
#pragma unroll
    for (int e = 0; e < DD_FC_SEG_MAX; ++e) {
        if (e < S) {
            const int j = j0 + e;
            const double v = j < W ? C[j + 1] : 0.0;
            tot[0] += v;
            run[e][0] = tot[0];
#pragma unroll
            for (int q = 1; q <= Q; ++q) {
                const double2 cs = tab[(q - 1) * P + r];
                tot[2 * q - 1] = fma(v, cs.x, tot[2 * q - 1]);
                tot[2 * q] = fma(v, cs.y, tot[2 * q]);
                run[e][2 * q - 1] = tot[2 * q - 1];
                run[e][2 * q] = tot[2 * q];
            }
            r = r + 1 == P ? 0 : r + 1;
        }
    }
    __syncthreads();                      // (stream 0's row is rewritten below)
    // ... the runs' totals are scanned across the workgroup ...
    double off[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double inc = tot[s];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double u = __shfl_up(inc, d);
            if (lane >= d) inc += u;
        }
        if (lane == 63) wsum[s * DD_FC_WAVES + wv] = inc;
        off[s] = inc - tot[s];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s)
        for (int w = 0; w < wv; ++w) off[s] += wsum[s * DD_FC_WAVES + w];
    // ... and the prefix sums go to LDS
    if (t == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) C[(size_t)s * (W + 1)] = 0.0;
    }
#pragma unroll
    for (int e = 0; e < DD_FC_SEG_MAX; ++e) {
        if (e < S && j0 + e < W) {
#pragma unroll
            for (int s = 0; s < NS; ++s) C[(size_t)s * (W + 1) + j0 + e + 1] = off[s] + run[e][s];
        }
    }
    __syncthreads();
    // outputs: window [o, o + K - 1] of the staged samples
    for (int o = t; o < TILE; o += DD_FC_THREADS) {
        if (o0 + o >= nout) break;
        double acc = fit.a[0] * (C[o + K] - C[o]);
        const int ro = o % P;
#pragma unroll
        for (int q = 1; q <= Q; ++q) {
            const double2 cs = tab[(q - 1) * P + ro];
            const double A = C[(size_t)(2 * q - 1) * (W + 1) + o + K] - C[(size_t)(2 * q - 1) * (W + 1) + o];
            const double B = C[(size_t)(2 * q) * (W + 1) + o + K] - C[(size_t)(2 * q) * (W + 1) + o];
            acc = fma(fit.a[q], fma(cs.x, A, cs.y * B), acc);
        }
        y[o0 + o] = acc;
    }
}

// both passes (cf. dd_filtfilt_launch); tab_dev: Q * (K - 1) double2 of device memory this call fills from tab_host
template <int Q>
static inline int dd_filtfilt_cos_launch_q(const double* in, int64_t in_stride, double* y1, double* out, int64_t out_stride, int64_t n, int K,
                                           const DDCosFit& fit, const double2* tab_dev, int batch, hipStream_t s) {
    const int edge = 3 * K;
    const int64_t N = n + 2 * (int64_t)edge;
    const size_t lds = dd_fc_lds_bytes(K, Q);
    constexpr int TILE = dd_fc_geom<Q>::TILE;
    static DDOncePerDevice attr;
    if (attr.need()) {
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_filtfilt_cos<Q, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_filtfilt_cos<Q, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr.mark();
    }
    hipLaunchKernelGGL((k_filtfilt_cos<Q, 0>), dim3((unsigned)((N + TILE - 1) / TILE), batch), dim3(DD_FC_THREADS), lds, s,
                       in, y1, n, edge, fit, tab_dev, K, in_stride, N);
    hipLaunchKernelGGL((k_filtfilt_cos<Q, 1>), dim3((unsigned)((n + TILE - 1) / TILE), batch), dim3(DD_FC_THREADS), lds, s,
                       (const double*)y1, out, n, edge, fit, tab_dev, K, N, out_stride);
    return DD_OK;
}
static inline int dd_filtfilt_cos_launch(const double* in, int64_t in_stride, double* y1, double* out, int64_t out_stride, int64_t n, int K,
                                         const DDCosFit& fit, const double2* tab_dev, int batch, hipStream_t s) {
    switch (fit.Q) {
        case 1: return dd_filtfilt_cos_launch_q<1>(in, in_stride, y1, out, out_stride, n, K, fit, tab_dev, batch, s);
        case 2: return dd_filtfilt_cos_launch_q<2>(in, in_stride, y1, out, out_stride, n, K, fit, tab_dev, batch, s);
        case 3: return dd_filtfilt_cos_launch_q<3>(in, in_stride, y1, out, out_stride, n, K, fit, tab_dev, batch, s);
    }
    return DD_ERR_UNSUPPORTED;
}
