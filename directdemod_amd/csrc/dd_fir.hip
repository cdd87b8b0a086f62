// F1/F3 (stand-alone FIR, filters.py:53-75) and F2 (zero-phase filtfilt,
// filters.py:72-73) entry points.
//  - complex64 full-rate data goes through the fused-chain kernels of dd_chain.hip
//    with NCO/FM/decimation disabled (same LDS-tiled direct form / MFMA path);
//  - float64 audio-rate data (NOAA tail): LDS-tiled, register-blocked
//    kernels shared with the zero-phase path (dd_filtfilt_kernels.h); one lane per
//    output remains for filters longer than the tile.
#include "dd_chain_kernels.h"
#include "dd_filtfilt_kernels.h"

__global__ void k_fill_f64(double* p, int n, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

static int fir_f64_state(dd_fir* f, hipStream_t s) {
    if (f->taps_dev) return DD_OK;
    DD_HIP_CHECK(hipMalloc((void**)&f->taps_dev, sizeof(double) * f->K));
    DD_HIP_CHECK(hipMemcpy(f->taps_dev, f->taps.data(), sizeof(double) * f->K, hipMemcpyHostToDevice));
    const int nh = f->K > 1 ? f->K - 1 : 1;
    DD_HIP_CHECK(hipMalloc((void**)&f->hist[0], sizeof(double) * nh));
    DD_HIP_CHECK(hipMalloc((void**)&f->hist[1], sizeof(double) * nh));
    hipLaunchKernelGGL(k_fill_f64, dim3((nh + 255) / 256), dim3(256), 0, s, f->hist[0], nh,
                       f->hist_mode == DD_HIST_ONES ? 1.0 : 0.0);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// float64 side of dd_fir_reset (called from dd_chain.hip)
int dd_fir_reset_f64(dd_fir* f, int mode, const float* hist_host, hipStream_t s) {
    if (!(f->taps_dev || mode == DD_HIST_GIVEN)) return DD_OK;
    int rc = fir_f64_state(f, s);
    if (rc != DD_OK) return rc;
    const int nh = f->K - 1;
    if (nh <= 0) return DD_OK;
    if (mode == DD_HIST_GIVEN) {
        // hist_host holds complex64 pairs; the real path takes the real parts
        std::vector<double> hr(nh);
        for (int i = 0; i < nh; ++i) hr[i] = (double)hist_host[2 * i];
        DD_HIP_CHECK(hipMemcpyAsync(f->hist[f->hpar], hr.data(), sizeof(double) * nh, hipMemcpyHostToDevice, s));
        DD_HIP_CHECK(hipStreamSynchronize(s));
    } else {
        hipLaunchKernelGGL(k_fill_f64, dim3((nh + 255) / 256), dim3(256), 0, s, f->hist[f->hpar], nh,
                           mode == DD_HIST_ONES ? 1.0 : 0.0);
        DD_LAUNCH_CHECK();
    }
    return DD_OK;
}

// float64 history for the real path, given as doubles (lfiltic with initOut values that float32 cannot hold)
extern "C" int dd_fir_reset_hist_f64(dd_fir* f, const double* hist_host, void* stream) {
    DD_REQUIRE(f, "h");
    hipStream_t s = dd_stream(stream);
    const int nh = f->K - 1;
    if (nh <= 0) return DD_OK;
    DD_REQUIRE(hist_host, "hist_host");
    int rc = fir_f64_state(f, s);
    if (rc != DD_OK) return rc;
    DD_HIP_CHECK(hipMemcpyAsync(f->hist[f->hpar], hist_host, sizeof(double) * nh, hipMemcpyHostToDevice, s));
    DD_HIP_CHECK(hipStreamSynchronize(s));
    return DD_OK;
}

// ---------------------------------------------------------------- float64 real FIR
__global__ void __launch_bounds__(256) k_fir_f64(const double* __restrict__ in, double* __restrict__ out, int64_t n,
                                                 const double* __restrict__ taps, int K,
                                                 const double* __restrict__ hist) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
        const int64_t j = i - k;
        const double v = (j >= 0) ? in[j] : hist[(K - 1) + j];
        acc = fma(taps[k], v, acc);
    }
    out[i] = acc;
}
__global__ void k_hist_update_f64(const double* __restrict__ in, int64_t n, int K, const double* __restrict__ hold,
                                  double* __restrict__ hnew) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K - 1) return;
    const int64_t j = n - (K - 1) + i;
    hnew[i] = (j >= 0) ? in[j] : hold[(K - 1) + j];
}

extern "C" int dd_fir_f64(dd_fir* f, const double* in, double* out, int64_t n, int carry, void* stream) {
    DD_REQUIRE(f && n >= 0, "h/n");
    if (n == 0) return DD_OK;
    DD_REQUIRE(in && out, "null buffer");
    hipStream_t s = dd_stream(stream);
    int rc = fir_f64_state(f, s);
    if (rc != DD_OK) return rc;
    if (dd_ff_tiled_ok(f->K, sizeof(double)) && in != out) {
        // LDS-tiled, register-blocked form (dd_filtfilt_kernels.h), same summation order as the plain kernel
        hipLaunchKernelGGL((k_filtfilt_tile<double, 2>), dim3((unsigned)((n + DD_FF_TILE - 1) / DD_FF_TILE), 1), dim3(DD_FF_THREADS),
                           dd_ff_lds_bytes(f->K), s, in, out, n, 0, f->taps_dev, f->K, (int64_t)0, (int64_t)0,
                           (const double*)f->hist[f->hpar]);
    } else {
        hipLaunchKernelGGL(k_fir_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n, f->taps_dev, f->K,
                           f->hist[f->hpar]);
    }
    DD_LAUNCH_CHECK();
    if (carry && f->K > 1) {
        hipLaunchKernelGGL(k_hist_update_f64, dim3((f->K + 254) / 256), dim3(256), 0, s, in, n, f->K,
                           f->hist[f->hpar], f->hist[f->hpar ^ 1]);
        DD_LAUNCH_CHECK();
        f->hpar ^= 1;
    }
    return DD_OK;
}

// ---------------------------------------------------------------- filtfilt
// scipy.signal.filtfilt(b,[1],x): ext = odd_ext(x, 3K); forward lfilter with the
// history = ext[0] (zi * x0), reverse, filter again with history = first sample,
// reverse, crop.  Each pass is a direct FIR whose out-of-range taps read the
// pass's first input sample.
template <typename T>
static int filtfilt_impl(const double* taps_host, int K, const T* in, T* out, int64_t n, hipStream_t s) {
    const int edge = 3 * K;
    if (n <= edge) {
        dd_set_error("The length of the input vector x must be greater than padlen, which is %d.", edge);
        return DD_ERR_INVALID;
    }
    const int64_t N = n + 2 * (int64_t)edge;
    const size_t tb = (sizeof(double) * (size_t)K + 255) & ~(size_t)255;
    DDScratchLock scr;                      // held until this entry point has enqueued everything
    int rcs = scr.get(tb + sizeof(T) * (size_t)N, s);
    char* base = scr.ptr;
    if (rcs != DD_OK) return rcs;
    double* taps = reinterpret_cast<double*>(base);
    T* y1 = reinterpret_cast<T*>(base + tb);
    DD_HIP_CHECK(hipMemcpyAsync(taps, taps_host, sizeof(double) * K, hipMemcpyHostToDevice, s));
    if constexpr (sizeof(T) == 8) {
        if (dd_ff_tiled_ok(K, sizeof(T))) {
            dd_filtfilt_launch<T>(in, n, y1, out, n, n, K, taps, 1, s);
        } else {
            hipLaunchKernelGGL(k_filtfilt_fwd<T>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, in, y1, n, edge, taps, K);
            hipLaunchKernelGGL(k_filtfilt_bwd<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y1, out, n, edge, taps, K);
        }
    } else {
        hipLaunchKernelGGL(k_filtfilt_fwd<T>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, in, y1, n, edge, taps, K);
        hipLaunchKernelGGL(k_filtfilt_bwd<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y1, out, n, edge, taps, K);
    }
    hipError_t le = hipGetLastError();
    hipError_t se = hipStreamSynchronize(s);               // the taps are the caller's host memory
    DD_HIP_CHECK(le);
    DD_HIP_CHECK(se);
    return DD_OK;
}

extern "C" int dd_filtfilt_f64(const double* taps_host, int ntaps, const double* in, double* out,
                               int64_t n, int is_complex, void* stream) {
    DD_REQUIRE(taps_host && ntaps >= 1 && in && out && n >= 0, "arguments");
    if (is_complex) return filtfilt_impl<double2>(taps_host, ntaps, (const double2*)in, (double2*)out, n, dd_stream(stream));
    return filtfilt_impl<double>(taps_host, ntaps, in, out, n, dd_stream(stream));
}

extern "C" int dd_filtfilt_c64(const double* taps_host, int ntaps, const float* in_c64, float* out_c64,
                               int64_t n, void* stream) {
    DD_REQUIRE(taps_host && ntaps >= 1 && in_c64 && out_c64 && n >= 0, "arguments");
    return filtfilt_impl<float2>(taps_host, ntaps, (const float2*)in_c64, (float2*)out_c64, n, dd_stream(stream));
}

// ---------------------------------------------------------------- F4: IIR (butter), float64
// scipy.signal.lfilter's transposed direct form II:
//   y = b0 x + z0 ; z_k = z_{k+1} + b_{k+1} x - a_{k+1} y ; z_{n-2} = b_{n-1} x - a_{n-1} y
// One lane per real component (lane 1 = imaginary part of complex data); the state
// lives in registers for the whole run.  `mode`: 0 state as given, 1 state scaled by the
// pass's first input sample (filtfilt), `rev`: walk the arrays backwards.
#define DD_IIR_MAXN 16
struct dd_iir {
    int n;
    double b[DD_IIR_MAXN], a[DD_IIR_MAXN];
    double zi[DD_IIR_MAXN];
    double* state;          // device: 2 * (n-1) doubles (re, im)
    double* mats;           // device: block-parallel path, [M_hi, M_lo, MG_hi, MG_lo] each IIR_S x IIR_S (see below), short blocks
    double* mats_long;      //         the same for the long block length
    double* scratch;        // device: block / group vectors of the block-parallel path (grow-only)
    size_t scratch_bytes;
};
struct DDIirCoef {
    int n;
    double b[DD_IIR_MAXN], a[DD_IIR_MAXN], zi[DD_IIR_MAXN];
};

__global__ void k_iir_df2t(const double* __restrict__ in, double* __restrict__ out, int64_t n, int ncomp, DDIirCoef C,
                           double* __restrict__ state, int mode, int rev, int save) {
    const int c = threadIdx.x;
    if (c >= ncomp) return;
    const int N = C.n;
    double z[DD_IIR_MAXN];
#pragma unroll
    for (int k = 0; k < DD_IIR_MAXN; ++k) z[k] = 0.0;
    const int64_t first = rev ? n - 1 : 0;
    if (mode == 1) {
        const double x0 = in[first * ncomp + c];
#pragma unroll
        for (int k = 0; k < DD_IIR_MAXN - 1; ++k) if (k < N - 1) z[k] = C.zi[k] * x0;
    } else {
#pragma unroll
        for (int k = 0; k < DD_IIR_MAXN - 1; ++k) if (k < N - 1) z[k] = state[c * (DD_IIR_MAXN - 1) + k];
    }
    for (int64_t i = 0; i < n; ++i) {
        const int64_t idx = (rev ? n - 1 - i : i) * ncomp + c;
        const double x = in[idx];
        const double y = fma(C.b[0], x, z[0]);
#pragma unroll
        for (int k = 0; k < DD_IIR_MAXN - 1; ++k) {
            if (k < N - 1) {
                const double zn = (k + 1 < N - 1) ? z[k + 1] : 0.0;
                z[k] = zn + C.b[k + 1] * x - C.a[k + 1] * y;
            }
        }
        out[idx] = y;
    }
    if (save) {
#pragma unroll
        for (int k = 0; k < DD_IIR_MAXN - 1; ++k) if (k < N - 1) state[c * (DD_IIR_MAXN - 1) + k] = z[k];
    }
}

static void iir_coef(const dd_iir* h, DDIirCoef* C) {
    C->n = h->n;
    for (int k = 0; k < DD_IIR_MAXN; ++k) {
        C->b[k] = h->b[k];
        C->a[k] = h->a[k];
        C->zi[k] = h->zi[k];
    }
}

static int iir_set_state(dd_iir* h, hipStream_t s) {
    double st[2 * (DD_IIR_MAXN - 1)];
    for (int c = 0; c < 2; ++c)
        for (int k = 0; k < DD_IIR_MAXN - 1; ++k) st[c * (DD_IIR_MAXN - 1) + k] = (c == 0 && k < h->n - 1) ? h->zi[k] : 0.0;
    // a real zi applied to complex data seeds the real part only (SciPy casts zi to complex)
    DD_HIP_CHECK(hipMemcpyAsync(h->state, st, sizeof(st), hipMemcpyHostToDevice, s));
    DD_HIP_CHECK(hipStreamSynchronize(s));
    return DD_OK;
}

extern "C" int dd_iir_create(dd_iir** h, const double* b, const double* a, int n, const double* zi_host) {
    DD_REQUIRE(h && b && a, "null argument");
    DD_REQUIRE(n >= 1 && n <= DD_IIR_MAXN, "filter order too high (n <= 16 coefficients)");
    DD_REQUIRE(a[0] != 0.0, "a[0] must be non-zero");
    dd_iir* f = new dd_iir();
    f->n = n;
    for (int k = 0; k < DD_IIR_MAXN; ++k) {
        f->b[k] = k < n ? b[k] / a[0] : 0.0;
        f->a[k] = k < n ? a[k] / a[0] : 0.0;
        f->zi[k] = (zi_host && k < n - 1) ? zi_host[k] : 0.0;
    }
    f->state = nullptr;
    f->mats = nullptr;
    f->mats_long = nullptr;
    f->scratch = nullptr;
    f->scratch_bytes = 0;
    hipError_t e = hipMalloc((void**)&f->state, sizeof(double) * 2 * (DD_IIR_MAXN - 1));
    if (e != hipSuccess) {
        delete f;
        dd_set_error("dd_iir_create: %s", hipGetErrorString(e));
        return e == hipErrorNoDevice ? DD_ERR_NODEVICE : DD_ERR_HIP;
    }
    int rc = iir_set_state(f, nullptr);
    if (rc != DD_OK) {
        (void)hipFree(f->state);
        delete f;
        return rc;
    }
    *h = f;
    return DD_OK;
}

extern "C" int dd_iir_destroy(dd_iir* h) {
    if (h) {
        (void)hipFree(h->state);
        if (h->mats) (void)hipFree(h->mats);
        if (h->mats_long) (void)hipFree(h->mats_long);
        if (h->scratch) (void)hipFree(h->scratch);
        delete h;
    }
    return DD_OK;
}

// ---------------------------------------------------------------- F4 at IQ rate: block-parallel recurrence
// The recurrence is linear in its state: over a block of LB samples, z_end = M z_start + e,
// with M = A^LB (A = the homogeneous DF2T step, a constant S x S matrix, S = n-1) and e = the
// block's end state when started from zero.  So:
//   1. every block's e in parallel (one lane per block and real component, no output);
//   2. the block start states by the same idea one and two levels up (groups of 64 blocks: group end
//      vectors in parallel; if there are more than 128 groups, super-groups of 64 groups likewise; a
//      short sequential sweep over the top level with the matching power of M; then back down, the
//      members of each group in parallel);
//   3. every block again in parallel from its true start state, this time writing y.
// Twice the arithmetic of the sequential form, n / LB lanes wide.  Same float64 recurrence
// per sample.  Conditioning: the DF2T state map is far from normal for narrow-band filters
// (6th-order low-pass at 1 % of Nyquist: |eig| < 0.985 but entries of A^256 up to 1.6e5, with
// M z a cancellation of terms that large), so M must be known to ~1e-24 relative or the block
// recurrence z <- M z + e is unstable.  M is therefore built in __float128 on the host by
// STEPPING the homogeneous recurrence LB times from each unit vector (repeated squaring loses
// the digits again), stored as double-double, and applied in double-double arithmetic; the
// state handed from block to block is a plain double, exactly as in the sequential form.
// block length: 256 samples, or 1024 from 2^25 samples up (measured on 2^24 / 2^26 complex128 samples: 256 ->
// 0.42 / 1.50 ms, 1024 -> 0.92 / 1.37 ms: short blocks keep the block kernels wide, long blocks give each
// block longer contiguous runs)
#ifndef IIR_LB_SHORT
#define IIR_LB_SHORT 256
#endif
#ifndef IIR_LB_LONG
#define IIR_LB_LONG 1024
#endif
#define IIR_LONG_FROM ((int64_t)1 << 25)
#define IIR_G1 32
#define IIR_G2 32
#define IIR_GMAX 32             // >= IIR_G1, IIR_G2: the scan kernels hold a group's vectors in registers
#define IIR_S (DD_IIR_MAXN - 1)
static_assert(IIR_G1 <= IIR_GMAX && IIR_G2 <= IIR_GMAX, "the scan kernels hold a group in registers");
#define IIR_MAT (IIR_S * IIR_S)

__device__ __forceinline__ void dd_iir_step(const DDIirCoef& C, int N, double (&z)[DD_IIR_MAXN], double x, double& y) {
    y = fma(C.b[0], x, z[0]);
#pragma unroll
    for (int k = 0; k < DD_IIR_MAXN - 1; ++k) {
        if (k < N - 1) {
            const double zn = (k + 1 < N - 1) ? z[k + 1] : 0.0;
            z[k] = zn + C.b[k + 1] * x - C.a[k + 1] * y;
        }
    }
}

// pass 1 (write_out = 0): end state of each block from a zero start -> blk[]; pass 3 (write_out = 1):
// run each block from its start state in blk[], write y, last block saves the carried state
__global__ void __launch_bounds__(256) k_iir_blocks(const double* __restrict__ in, double* __restrict__ out, int64_t n, int ncomp,
                                                    DDIirCoef C, double* __restrict__ blk, int64_t nb, int write_out,
                                                    double* __restrict__ state, int save, int lb) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nb * ncomp) return;
    const int64_t b = t / ncomp;
    const int c = (int)(t - b * ncomp);
    const int N = C.n;
    double z[DD_IIR_MAXN];
#pragma unroll
    for (int k = 0; k < DD_IIR_MAXN; ++k) z[k] = 0.0;
    double* slot = blk + t * IIR_S;
    if (write_out) {
#pragma unroll
        for (int k = 0; k < IIR_S; ++k) if (k < N - 1) z[k] = slot[k];
    }
    const int64_t i0 = b * lb;
    const int64_t i1 = i0 + lb < n ? i0 + lb : n;
    // the recurrence is serial, its input is not: 16 samples are requested at once (a lane's reads are a
    // 4 KiB stride apart from its neighbours', so each costs a full memory latency when taken one by one)
    for (int64_t i = i0; i < i1; i += 16) {
        double xs[16], ys[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int64_t q = i + u < i1 ? i + u : i1 - 1;
            xs[u] = in[q * ncomp + c];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (i + u < i1) dd_iir_step(C, N, z, xs[u], ys[u]);
        }
        if (write_out) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (i + u < i1) out[(i + u) * ncomp + c] = ys[u];
            }
        }
    }
    if (!write_out) {
#pragma unroll
        for (int k = 0; k < IIR_S; ++k) if (k < N - 1) slot[k] = z[k];
    } else if (save && b == nb - 1) {
#pragma unroll
        for (int k = 0; k < IIR_S; ++k) if (k < N - 1) state[c * IIR_S + k] = z[k];
    }
}

// The same two passes with the samples staged through LDS.  In the form above a lane walks its own block, so a
// wave's load touches 32-64 different cache lines and uses 16 bytes of each; the lines do get used up over the
// next iterations, but only if they survive in L1 meanwhile (pass 1 ran at 1.5 TB/s).  Here the workgroup's
// 256 chains (128 blocks x re/im, or 256 real blocks) fetch 16 samples per block as whole 16-byte units, a
// block's 256 bytes on 16 adjacent lanes, and park them in LDS rows padded by 16 (8) bytes so that the
// per-chain reads fall on distinct banks; outputs overwrite the inputs in the same LDS slots and leave the same
// way.  The next step's units are requested before the current one is computed and written to the other LDS
// buffer afterwards.  The recurrence itself is the same float64 sequence per sample.
#ifndef IIR_CH
#define IIR_CH 16
#endif
__host__ __device__ __forceinline__ int iir_lds_row(int ncomp) { return IIR_CH * ncomp + (ncomp == 2 ? 2 : 1); }

template <int S>
__device__ __forceinline__ double dd_iir_step_t(const DDIirCoef& C, double (&z)[S], double x) {
    const double y = fma(C.b[0], x, z[0]);
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const double zn = (k + 1 < S) ? z[k + 1] : 0.0;
        z[k] = zn + C.b[k + 1] * x - C.a[k + 1] * y;
    }
    return y;
}

template <int S, bool WRITE>
__global__ void __launch_bounds__(256) k_iir_blocks_t(const double* __restrict__ in, double* __restrict__ out, int64_t n, int ncomp,
                                                      DDIirCoef C, double* __restrict__ blk, int64_t nb,
                                                      double* __restrict__ state, int save, int lb) {
    extern __shared__ double iir_lds[];
    const int nbw = 256 / ncomp, row = iir_lds_row(ncomp), upb = IIR_CH * ncomp / 2;      // blocks per workgroup, row length, 16-byte units per row
    const int nunit = nbw * upb / 256;                                                   // units per lane and step (8)
    const int t = threadIdx.x, bl = t / ncomp, c = t - bl * ncomp;
    const int64_t bw0 = (int64_t)blockIdx.x * nbw, b = bw0 + bl;
    const bool live = b < nb;
    const int64_t total = n * ncomp;
    double z[S];
#pragma unroll
    for (int k = 0; k < S; ++k) z[k] = (WRITE && live) ? blk[(b * ncomp + c) * IIR_S + k] : 0.0;
    const int ilen = live ? (int)((n - b * lb) < lb ? (n - b * lb) : lb) : 0;            // samples of this chain's block
    double2 rg[IIR_CH / 2];
    const double xlast = in[total - 1];
    auto issue = [&](int i) {
#pragma unroll
        for (int k = 0; k < IIR_CH / 2; ++k) {
            if (k < nunit) {
                const int j = t + 256 * k, ub = j / upb, w = j - ub * upb;
                int64_t d = ((bw0 + ub) * lb + i) * ncomp + 2 * w;
                const int64_t dmax = (total - 2) & ~(int64_t)1;
                const bool straggler = d == total - 1;                                   // odd length: the last sample starts a unit
                d = d < dmax ? d : dmax;                                                 // past the end: re-read, never consumed
                rg[k] = *reinterpret_cast<const double2*>(in + d);
                if (straggler) rg[k].x = xlast;
            }
        }
    };
    auto park = [&](double* buf) {
#pragma unroll
        for (int k = 0; k < IIR_CH / 2; ++k) {
            if (k < nunit) {
                const int j = t + 256 * k, ub = j / upb, w = j - ub * upb;
                buf[ub * row + 2 * w] = rg[k].x;
                buf[ub * row + 2 * w + 1] = rg[k].y;
            }
        }
    };
    double* cur = iir_lds;
    double* nxt = iir_lds + nbw * row;
    issue(0);
    park(cur);
    __syncthreads();
    for (int i = 0; i < lb; i += IIR_CH) {
        if (i + IIR_CH < lb) issue(i + IIR_CH);
        double* mine = cur + bl * row + c;
#pragma unroll
        for (int u = 0; u < IIR_CH; ++u) {
            if (i + u < ilen) {
                const double y = dd_iir_step_t<S>(C, z, mine[u * ncomp]);
                if (WRITE) mine[u * ncomp] = y;
            }
        }
        if (WRITE) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < IIR_CH / 2; ++k) {
                if (k < nunit) {
                    const int j = t + 256 * k, ub = j / upb, w = j - ub * upb;
                    const int64_t bb = bw0 + ub;
                    const int64_t d = (bb * lb + i) * ncomp + 2 * w;
                    const int64_t dend = (bb + 1) * lb * ncomp < total ? (bb + 1) * lb * ncomp : total;   // end of this block's data
                    if (bb < nb && d + 1 < dend) *reinterpret_cast<double2*>(out + d) = make_double2(cur[ub * row + 2 * w], cur[ub * row + 2 * w + 1]);
                    else if (bb < nb && d < dend) out[d] = cur[ub * row + 2 * w];
                }
            }
        }
        if (i + IIR_CH < lb) park(nxt);
        __syncthreads();
        double* tmp = cur; cur = nxt; nxt = tmp;
    }
    if (!WRITE) {
        if (live) {
#pragma unroll
            for (int k = 0; k < S; ++k) blk[(b * ncomp + c) * IIR_S + k] = z[k];
        }
    } else if (save && live && b == nb - 1) {
#pragma unroll
        for (int k = 0; k < S; ++k) state[c * IIR_S + k] = z[k];
    }
}

// The block passes for complex128 input as ONE WAVE per workgroup, its samples brought in by LDS-DMA
// (global_load_lds_dwordx4: 64 lanes x 16 bytes = 1 KiB per instruction, no registers).
// Why: the passes are bound by the bytes a CU keeps in flight, not by the access pattern (longer contiguous pieces
// and non-power-of-two block strides changed nothing): k_iir_blocks_t holds one 32 KB step per 70 KB workgroup in
// flight = 64 KB per CU, and at the ~4 us a request takes under load that is 3.3-3.5 TB/s -- what it measures.
// LDS-DMA needs no staging registers, so the whole LDS is the prefetch queue: a wave owns 32 blocks (lane = block,
// re/im), a step is 32 samples = 512 contiguous bytes from each block (one DMA instruction per PAIR of blocks), and
// a ring of IIR_W_NB step buffers (3 x 16.6 KB, three waves per CU) keeps two steps per wave = 96 KB per CU on
// the wire while the third is computed.  The lane walks its row exactly as before (same float64 recurrence per
// sample); outputs overwrite the inputs in LDS and leave as 16-byte stores.  One wave: no barriers, only counted
// waits (loads, DMAs and stores retire in order on vmcnt).
// s_waitcnt vmcnt(n) for a wave-uniform n that is a multiple of 8 (the counter holds 63: anything above waits for 63 -- one retirement more than asked)
__device__ __forceinline__ void iir_wait_vmcnt(int n) {
    switch (n >> 3) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
}
typedef double iir_v2d __attribute__((ext_vector_type(2)));      // (a register pair an asm statement can name)
#define IIR_W_BLOCKS 32
#define IIR_W_CH 32
#define IIR_W_NB 3
#define IIR_W_PAIR (2 * IIR_W_CH * 16 + 16)              // two 512-byte rows, then 16 bytes of padding
#define IIR_W_BUF ((IIR_W_BLOCKS / 2) * IIR_W_PAIR)
template <int S, bool WRITE>
__global__ void __launch_bounds__(64) k_iir_blocks_w(const double2* __restrict__ in, double2* __restrict__ out, int64_t n, DDIirCoef C,
                                                     double* __restrict__ blk, int64_t nb, double* __restrict__ state, int save, int lb) {
    extern __shared__ __attribute__((aligned(16))) char iir_w_lds[];
    const int lane = threadIdx.x, bl = lane >> 1, c = lane & 1;
#ifndef IIR_WRITE_FORWARD
    // the pass that writes takes the workgroups' blocks from the END of the input -- what the read pass touched last is what the memory-side cache (256 MB)
    // still holds -- and its stores are non-temporal, so that the 16 bytes written per sample do not push the 8 still to be read out of it (round 6, same call:
    // write pass 360 -> 327 us, the call 0.644 -> 0.591 ms; reversed alone 347 us, non-temporal alone 351 us)
    const int64_t b0 = (int64_t)(WRITE ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * IIR_W_BLOCKS, b = b0 + bl;
#else
    const int64_t b0 = (int64_t)blockIdx.x * IIR_W_BLOCKS, b = b0 + bl;
#endif
    const bool live = b < nb;
    double z[S];
#pragma unroll
    for (int k = 0; k < S; ++k) z[k] = (WRITE && live) ? blk[(b * 2 + c) * IIR_S + k] : 0.0;
    if (WRITE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // (the states: before anything below is counted)
    const int ilen = live ? (int)((n - b * lb) < lb ? (n - b * lb) : lb) : 0;            // samples of this chain's block
    const int nsteps = lb / IIR_W_CH;
    // the counted waits of the write pass assume that every step issues all 16 of its (predicated) stores: true for a
    // workgroup whose 32 blocks all exist and are whole; the last workgroup of a call waits for everything instead
    const bool partial = (b0 + IIR_W_BLOCKS > nb) || ((b0 + IIR_W_BLOCKS) * (int64_t)lb > n);
    // sample this lane moves in a DMA / a store of pair r: lanes 0..31 the first block of the pair, 32..63 the second
    const int half = lane >> 5, l32 = lane & 31;
    auto issue = [&](int step) {
        char* buf = iir_w_lds + (step % IIR_W_NB) * IIR_W_BUF;
#pragma unroll
        for (int r = 0; r < IIR_W_BLOCKS / 2; ++r) {
            int64_t idx = (b0 + 2 * r + half) * lb + (int64_t)step * IIR_W_CH + l32;
            idx = idx < n ? idx : n - 1;                                                  // past the end: re-read, never consumed
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + idx),
                                             (__attribute__((address_space(3))) void*)(buf + r * IIR_W_PAIR), 16, 0, 0);
        }
    };
#pragma unroll
    for (int k = 0; k < IIR_W_NB - 1; ++k)
        if (k < nsteps) issue(k);
    for (int st = 0; st < nsteps; ++st) {
        if (st + IIR_W_NB - 1 < nsteps) issue(st + IIR_W_NB - 1);
        // step st must have landed.  Younger than its DMAs: the DMA batches of the steps after it and (write pass) the
        // store batches of the iterations since -- 16 instructions each
        const int ahead = nsteps - 1 - st < IIR_W_NB - 1 ? nsteps - 1 - st : IIR_W_NB - 1;       // DMA batches in flight behind step st
        // ... and (write pass) the store batches issued since step st's DMAs were: those of the two steps before this one
        if (WRITE && partial) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // fewer than 16 stores per step may have issued: nothing to count on
        else iir_wait_vmcnt(16 * ahead + (WRITE ? 16 * (st < IIR_W_NB - 1 ? st : IIR_W_NB - 1) : 0));              // (2 x 16 DMAs + 2 x 16 stores = 64: one more than the counter holds -> 63)
        char* cur = iir_w_lds + (st % IIR_W_NB) * IIR_W_BUF;
        double* mine = reinterpret_cast<double*>(cur + (bl >> 1) * IIR_W_PAIR + (bl & 1) * (IIR_W_CH * 16)) + c;
        const int left = ilen - st * IIR_W_CH;
        if (!partial) {                                     // (no per-sample guard where every block exists and is whole: k_iir_blocks_w32)
#pragma unroll 1
            for (int u0 = 0; u0 < IIR_W_CH; u0 += 8) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = mine[2 * (u0 + u)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double y = dd_iir_step_t<S>(C, z, x[u]);
                    if (WRITE) mine[2 * (u0 + u)] = y;
                }
            }
        } else {
#pragma unroll 1
            for (int u0 = 0; u0 < IIR_W_CH; u0 += 8) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = mine[2 * (u0 + u)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u0 + u < left) {
                        const double y = dd_iir_step_t<S>(C, z, x[u]);
                        if (WRITE) mine[2 * (u0 + u)] = y;
                    }
                }
            }
        }
        if (WRITE) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            // the rows hold the outputs
#ifndef IIR_W_PLAIN_READS
            // the rows are read back by instructions the compiler cannot see into: it puts s_waitcnt vmcnt(0) in front of every LDS read that
            // might touch what a DMA in flight writes -- sixteen drains of the whole queue per step, each store waiting for the one before it
            // (round 6, the ISA: profiles/r06_iir_notes.txt).  The rows of THIS step landed before the loop above read them.
            const uint32_t rows = (uint32_t)(uintptr_t)cur + (uint32_t)lane * 16u;          // (LDS: the low 32 bits of the generic address are the byte offset)
#pragma unroll
            for (int r0 = 0; r0 < IIR_W_BLOCKS / 2; r0 += 4) {
                iir_v2d v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[q]) : "v"(rows), "n"((r0 + q) * IIR_W_PAIR));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t bb = b0 + 2 * (r0 + q) + half;
                    const int64_t idx = bb * lb + (int64_t)st * IIR_W_CH + l32;
                    if (bb < nb && idx < n) {
#ifndef IIR_WRITE_PLAIN
                        __builtin_nontemporal_store(v[q], reinterpret_cast<iir_v2d*>(out + idx));
#else
                        *reinterpret_cast<iir_v2d*>(out + idx) = v[q];
#endif
                    }
                }
            }
#else
#pragma unroll
            for (int r = 0; r < IIR_W_BLOCKS / 2; ++r) {
                const int64_t bb = b0 + 2 * r + half;
                const int64_t pos = (int64_t)st * IIR_W_CH + l32;
                const int64_t idx = bb * lb + pos;
                const double2 v = *reinterpret_cast<const double2*>(cur + r * IIR_W_PAIR + lane * 16);
                if (bb < nb && idx < n) out[idx] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            // rows read before a later DMA overwrites them
#endif
        }
    }
    if (!WRITE) {
        if (live) {
#pragma unroll
            for (int k = 0; k < S; ++k) blk[(b * 2 + c) * IIR_S + k] = z[k];
        }
    } else if (save && live && b == nb - 1) {
#pragma unroll
        for (int k = 0; k < S; ++k) state[c * IIR_S + k] = z[k];
    }
}

// The same passes for COMPLEX64 input (round 6: decode_funcube.py:160 / decode_meteorm2.py:157 low-pass the IQ stream as the source hands it
// over -- complex64; lfilter's output is complex128).  The samples come in as they are -- half the bytes of the widened copy the class route
// made first (8 + 16 B per sample for the copy, then 16 per pass) -- and are widened where a lane picks them up.  One DMA instruction
// (64 lanes x 16 bytes) now carries a step of FOUR blocks: lane L brings samples 2 (L >> 2), 2 (L >> 2) + 1 of block 4 r + (L & 3), so a
// block's 32 samples sit in 16-byte pairs 64 bytes apart in the 1 KiB chunk (chunks 64 bytes of padding apart: the rows of chunks r and r + 4
// share banks -- two-way, on one 4-byte read per 14 float64 operations).  Outputs go through a tile of their
// own (the rows of k_iir_blocks_w: 512 bytes per block and step) and leave as 16-byte stores.  Ring of three input buffers + the tile =
// 42.8 KB per wave, three waves per CU.  A 16-byte-aligned 16-byte load never crosses a page: the pair that holds the last sample of an odd-length
// input reads 8 bytes past it, inside the page of that sample, and nothing looks at them.
#ifndef IIR_W32_NB
#define IIR_W32_NB 3
#endif
#ifndef IIR_W32_READ_CH
#define IIR_W32_READ_CH 64
#endif
// CH = samples per block and step: 32 (a DMA instruction carries a step of four blocks, 256 contiguous bytes each) or -- the pass that only
// reads -- 64 (two blocks, 512 bytes each: the rate of these passes follows the length of the contiguous pieces, 2.9 TB/s at 256 bytes,
// 3.9 at the 512 of the complex128 kernel; the write pass keeps 32: its output tile would double)
template <int CH> struct IirW32 {
    static constexpr int BPC = 128 / CH;                       // blocks per DMA instruction (64 lanes x 2 samples)
    static constexpr int NDMA = IIR_W_BLOCKS / BPC;            // DMA instructions per step
    static constexpr int CHUNK = 1024 + (CH == 32 ? 64 : 32);  // its kilobyte + padding (the rows of chunks a bank period apart share banks: two-way)
    static constexpr int IN = NDMA * CHUNK;                    // one step of the wave
};
#define IIR_W32_LDS(CH, WR) (IIR_W32_NB * IirW32<CH>::IN + ((WR) ? IIR_W_BUF : 0))
template <int S, bool WRITE, int CH>
__global__ void __launch_bounds__(64) k_iir_blocks_w32(const float2* __restrict__ in, double2* __restrict__ out, int64_t n, DDIirCoef C,
                                                       double* __restrict__ blk, int64_t nb, double* __restrict__ state, int save, int lb) {
    static_assert(CH == 32 || (CH == 64 && !WRITE), "the output tile holds 32 samples per block");
    typedef IirW32<CH> G;
    extern __shared__ __attribute__((aligned(16))) char iir_w_lds[];
    char* const tile = iir_w_lds + IIR_W32_NB * G::IN;
    const int lane = threadIdx.x, bl = lane >> 1, c = lane & 1;
#ifndef IIR_WRITE_FORWARD
    // the pass that writes takes the workgroups' blocks from the END of the input -- what the read pass touched last is what the memory-side cache (256 MB)
    // still holds -- and its stores are non-temporal, so that the 16 bytes written per sample do not push the 8 still to be read out of it (round 6, same call:
    // write pass 360 -> 327 us, the call 0.644 -> 0.591 ms; reversed alone 347 us, non-temporal alone 351 us)
    const int64_t b0 = (int64_t)(WRITE ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * IIR_W_BLOCKS, b = b0 + bl;
#else
    const int64_t b0 = (int64_t)blockIdx.x * IIR_W_BLOCKS, b = b0 + bl;
#endif
    const bool live = b < nb;
    double z[S];
#pragma unroll
    for (int k = 0; k < S; ++k) z[k] = (WRITE && live) ? blk[(b * 2 + c) * IIR_S + k] : 0.0;
    if (WRITE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // (the states: before anything below is counted)
    const int ilen = live ? (int)((n - b * lb) < lb ? (n - b * lb) : lb) : 0;
    const int nsteps = lb / CH;
    const bool partial = (b0 + IIR_W_BLOCKS > nb) || ((b0 + IIR_W_BLOCKS) * (int64_t)lb > n);
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t last_pair = (n - 1) & ~(int64_t)1;
    auto issue = [&](int step) {
        char* buf = iir_w_lds + (step % IIR_W32_NB) * G::IN;
#pragma unroll
        for (int r = 0; r < G::NDMA; ++r) {
            int64_t idx = (b0 + G::BPC * r + (lane & (G::BPC - 1))) * lb + (int64_t)step * CH + 2 * (lane / G::BPC);
            idx = idx < n ? idx : last_pair;                                              // past the end: re-read, never consumed
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + idx),
                                             (__attribute__((address_space(3))) void*)(buf + r * G::CHUNK), 16, 0, 0);
        }
    };
#pragma unroll
    for (int k = 0; k < IIR_W32_NB - 1; ++k)
        if (k < nsteps) issue(k);
    double* const mo = reinterpret_cast<double*>(tile + (bl >> 1) * IIR_W_PAIR + (bl & 1) * (IIR_W_CH * 16)) + c;
    const uint32_t rows = (uint32_t)(uintptr_t)tile + (uint32_t)lane * 16u;
    for (int st = 0; st < nsteps; ++st) {
        if (st + IIR_W32_NB - 1 < nsteps) issue(st + IIR_W32_NB - 1);
        // step st must have landed: younger than its DMAs are the DMA batches (NDMA) of the steps after it and (write pass) the store batches
        // (16) of the NB - 1 steps before this one
        const int ahead = nsteps - 1 - st < IIR_W32_NB - 1 ? nsteps - 1 - st : IIR_W32_NB - 1;
        if (WRITE && partial) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else iir_wait_vmcnt(G::NDMA * ahead + (WRITE ? 16 * (st < IIR_W32_NB - 1 ? st : IIR_W32_NB - 1) : 0));
        const char* cur = iir_w_lds + (st % IIR_W32_NB) * G::IN;
        const float* mine = reinterpret_cast<const float*>(cur + (bl / G::BPC) * G::CHUNK + (bl & (G::BPC - 1)) * 16) + c;
        constexpr int PS = 4 * G::BPC;                                                    // floats from a pair of a block to its next one
        const int left = ilen - st * CH;
        if (!partial) {
            // every block of this wave exists and is whole (all workgroups but the last): no per-sample guard -- a compare, an exec-mask
            // save / restore and a branch per sample beside fifteen float64 operations, and nothing could move across them
#pragma unroll 1
            for (int u0 = 0; u0 < CH; u0 += 8) {
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = mine[PS * ((u0 + u) >> 1) + 2 * (u & 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double y = dd_iir_step_t<S>(C, z, (double)x[u]);
                    if (WRITE) mo[2 * (u0 + u)] = y;
                }
            }
        } else {
#pragma unroll 1
            for (int u0 = 0; u0 < CH; u0 += 8) {
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = mine[PS * ((u0 + u) >> 1) + 2 * (u & 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u0 + u < left) {
                        const double y = dd_iir_step_t<S>(C, z, (double)x[u]);
                        if (WRITE) mo[2 * (u0 + u)] = y;
                    }
                }
            }
        }
        if (WRITE) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            // the tile holds the outputs
#pragma unroll
            for (int r0 = 0; r0 < IIR_W_BLOCKS / 2; r0 += 4) {
                iir_v2d v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[q]) : "v"(rows), "n"((r0 + q) * IIR_W_PAIR));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t bb = b0 + 2 * (r0 + q) + half;
                    const int64_t idx = bb * lb + (int64_t)st * IIR_W_CH + l32;
                    if (bb < nb && idx < n) {
#ifndef IIR_WRITE_PLAIN
                        __builtin_nontemporal_store(v[q], reinterpret_cast<iir_v2d*>(out + idx));
#else
                        *reinterpret_cast<iir_v2d*>(out + idx) = v[q];
#endif
                    }
                }
            }
        }
    }
    if (!WRITE) {
        if (live) {
#pragma unroll
            for (int k = 0; k < S; ++k) blk[(b * 2 + c) * IIR_S + k] = z[k];
        }
    } else if (save && live && b == nb - 1) {
#pragma unroll
        for (int k = 0; k < S; ++k) state[c * IIR_S + k] = z[k];
    }
}

// u <- M u + e with M = hi + lo (double-double), products and sum carried in double-double.
// One chain (a group of blocks, or the sweep over the groups) is spread over RP = 8 or 16 adjacent lanes,
// lane r owning row r of M in registers and component r of u: a step is S shuffles and S double-double
// multiply-adds per lane instead of S*S in one lane (the serial form kept a single wave per chain busy for
// ~2500 cycles per step: 720 dependent-ish f64 operations at 4 cycles each).  Row sums accumulate in the
// same order as before, so the states are bit-identical.
template <int S>
struct IirRows {
    static constexpr int RP = S <= 8 ? 8 : 16;
};
// every element of x[] in its register before anything after this statement starts (an empty asm that names them all as read-write operands)
template <int S>
__device__ __forceinline__ void iir_tie(double (&x)[S]) {
    if constexpr (S == 1) asm volatile("" : "+v"(x[0]));
    else if constexpr (S == 2) asm volatile("" : "+v"(x[0]), "+v"(x[1]));
    else if constexpr (S == 3) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
    else if constexpr (S == 4) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    else if constexpr (S == 5) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]));
    else if constexpr (S == 6) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
    else if constexpr (S == 7) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]));
    else if constexpr (S == 8) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    else if constexpr (S == 9) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]));
    else if constexpr (S == 10) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]));
    else if constexpr (S == 11) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]));
    else if constexpr (S == 12) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]));
    else if constexpr (S == 13) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]));
    else if constexpr (S == 14) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]));
    else if constexpr (S == 15) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]));
}
template <int S>
__device__ __forceinline__ double dd_iir_affine_row(const double (&mh)[S], const double (&ml)[S], double u, double e, int base) {
#pragma clang fp contract(off)      // error-free transformations below: no fusing of their multiplies and adds
    // all S components first (round 6: fetched one by one inside the loop below every shuffle was followed by a wait -- S LDS round trips per
    // step of the chain, 720 of a step's 1080 cycles)
    double xs[S];
#pragma unroll
    for (int q = 0; q < S; ++q) xs[q] = __shfl(u, base + q);
    iir_tie<S>(xs);                 // (left to itself the compiler still issues half of them one by one between the sums)
    double ah = e, al = 0.0;
#pragma unroll
    for (int q = 0; q < S; ++q) {
        const double x = xs[q];
        const double m = mh[q];
        const double p = m * x;
        const double pe = fma(m, x, -p) + ml[q] * x;                   // exact product tail + low limb
        const double sh = ah + p;                                      // two-sum
        const double bb = sh - ah;
        const double se = (ah - (sh - bb)) + (p - bb);
        ah = sh;
        al += se + pe;
    }
    return ah + al;
}

// phase 0: group end vectors from zero (grp[]); phase 2: block start states written over blk[]
template <int S>
__global__ void __launch_bounds__(64, 1) k_iir_groups(double* __restrict__ blk, double* __restrict__ grp, int64_t nb, int ncomp,
                                                   const double* __restrict__ mats, int phase, int G) {
    constexpr int RP = IirRows<S>::RP;
    const int64_t ng = (nb + G - 1) / G;
    const int lane = threadIdx.x, r = lane % RP, base = lane - r;
    int64_t t = (int64_t)blockIdx.x * (64 / RP) + lane / RP;           // chain = (group, component)
    const bool live = t < ng * ncomp;
    if (!live) t = ng * ncomp - 1;                                     // idle chains shadow the last one (shuffles stay convergent)
    const bool row = r < S;
    const int rr = row ? r : 0;
    const int64_t g = t / ncomp;
    const int c = (int)(t - g * ncomp);
    double mh[S], ml[S];
#pragma unroll
    for (int q = 0; q < S; ++q) { mh[q] = mats[rr * IIR_S + q]; ml[q] = mats[IIR_MAT + rr * IIR_S + q]; }
    double u = (phase == 2) ? grp[t * IIR_S + rr] : 0.0;
    const int64_t b0 = g * G, b1 = b0 + G < nb ? b0 + G : nb;
    // the chain u <- M u + e is serial; the e vectors are not: ALL of the group's (G <= IIR_GMAX) are requested before the first step.
    // (Round 6: with one fetched a step ahead a step took ~0.5 us -- a memory round trip -- for ~200 cycles of arithmetic; 32 steps 16 -> 4 us.)
    double ev[IIR_GMAX];
#pragma unroll
    for (int i = 0; i < IIR_GMAX; ++i) ev[i] = b0 + i < b1 ? blk[((b0 + i) * ncomp + c) * IIR_S + rr] : 0.0;
#pragma unroll
    for (int i = 0; i < IIR_GMAX; ++i) {
        const bool in = b0 + i < b1;                                   // (false only in the last, short group: every lane takes every step, the result is dropped)
        if (phase == 2 && live && row && in) blk[((b0 + i) * ncomp + c) * IIR_S + r] = u;      // this block's start state
        const double un = dd_iir_affine_row<S>(mh, ml, u, ev[i], base);
        u = in ? un : u;
    }
    if (phase == 0 && live && row) grp[t * IIR_S + r] = u;
}

// phase 1: sequential sweep over the groups (one chain per component): grp[g] <- start state of group g
template <int S>
__global__ void __launch_bounds__(64, 1) k_iir_group_sweep(double* __restrict__ grp, int64_t ng, int ncomp, const double* __restrict__ mats,
                                                        const double* __restrict__ state, int zero_state) {
    constexpr int RP = IirRows<S>::RP;
    const int lane = threadIdx.x, r = lane % RP, base = lane - r;
    int c = lane / RP;
    const bool live = c < ncomp;
    if (!live) c = ncomp - 1;
    const bool row = r < S;
    const int rr = row ? r : 0;
    double mh[S], ml[S];
#pragma unroll
    for (int q = 0; q < S; ++q) { mh[q] = mats[rr * IIR_S + q]; ml[q] = mats[IIR_MAT + rr * IIR_S + q]; }
    double u = zero_state ? 0.0 : state[c * IIR_S + rr];
    for (int64_t g0 = 0; g0 < ng; g0 += IIR_GMAX) {                    // (the e vectors IIR_GMAX at a time, all requested before the first of their steps)
        double ev[IIR_GMAX];
#pragma unroll
        for (int i = 0; i < IIR_GMAX; ++i) ev[i] = g0 + i < ng ? grp[((g0 + i) * ncomp + c) * IIR_S + rr] : 0.0;
#pragma unroll
        for (int i = 0; i < IIR_GMAX; ++i) {
            if (g0 + i < ng) {                                         // (the same for every lane)
                if (live && row) grp[((g0 + i) * ncomp + c) * IIR_S + r] = u;
                u = dd_iir_affine_row<S>(mh, ml, u, ev[i], base);
            }
        }
    }
}

// M = A^LB by stepping the homogeneous DF2T recurrence (z0' = z1 - a1 z0, ...) from each unit
// vector, M1 = M^G1 by stepping the block map, M2 = M1^G2 by stepping the group map; all in
// __float128, split into double-double: out = [M hi, M lo, M1 hi, M1 lo, M2 hi, M2 lo].
static void iir_block_matrices(const dd_iir* h, int lb, double* out /* 6 * IIR_MAT */) {
    const int S = h->n - 1;
    typedef __float128 q_t;
    q_t M[3][IIR_S][IIR_S];
    for (int j = 0; j < S; ++j) {
        q_t z[IIR_S + 1];
        for (int k = 0; k <= IIR_S; ++k) z[k] = 0;
        z[j] = 1;
        for (int t = 0; t < lb; ++t) {
            const q_t y = z[0];
            for (int k = 0; k < S; ++k) z[k] = (k + 1 < S ? z[k + 1] : (q_t)0) - (q_t)h->a[k + 1] * y;
        }
        for (int k = 0; k < S; ++k) M[0][k][j] = z[k];
    }
    const int steps[2] = {IIR_G1, IIR_G2};
    for (int lv = 1; lv < 3; ++lv) {
        for (int j = 0; j < S; ++j) {
            q_t u[IIR_S], v[IIR_S];
            for (int k = 0; k < S; ++k) u[k] = (k == j) ? 1 : 0;
            for (int t = 0; t < steps[lv - 1]; ++t) {
                for (int r = 0; r < S; ++r) {
                    q_t acc = 0;
                    for (int q = 0; q < S; ++q) acc += M[lv - 1][r][q] * u[q];
                    v[r] = acc;
                }
                for (int r = 0; r < S; ++r) u[r] = v[r];
            }
            for (int k = 0; k < S; ++k) M[lv][k][j] = u[k];
        }
    }
    for (int i = 0; i < 6 * IIR_MAT; ++i) out[i] = 0.0;
    for (int lv = 0; lv < 3; ++lv)
        for (int r = 0; r < S; ++r)
            for (int c = 0; c < S; ++c) {
                const double mh = (double)M[lv][r][c];
                out[(2 * lv) * IIR_MAT + r * IIR_S + c] = mh;
                out[(2 * lv + 1) * IIR_MAT + r * IIR_S + c] = (double)(M[lv][r][c] - (q_t)mh);
            }
}

// in32: `in` is complex64 (ncomp == 2, buffers 16-byte aligned: the caller has checked) -- the one-wave kernels k_iir_blocks_w32
static int iir_parallel(dd_iir* h, const double* in, double* out, int64_t n, int ncomp, int carry, hipStream_t s, bool in32 = false) {
    const int S = h->n - 1;
    int lb = n >= IIR_LONG_FROM ? IIR_LB_LONG : IIR_LB_SHORT;
    if (const char* e = DD_TUNE_ENV("DD_IIR_LB")) lb = atoi(e) == IIR_LB_LONG ? IIR_LB_LONG : IIR_LB_SHORT;      // A/B switch
    // block start states: blocks -> groups of G1 -> (if there are many groups) super-groups of G2 -> one short serial sweep
    const int64_t nb = (n + lb - 1) / lb, ng = (nb + IIR_G1 - 1) / IIR_G1;
    const bool three = ng > 2 * IIR_G2;
    const int64_t ns = three ? (ng + IIR_G2 - 1) / IIR_G2 : 0;
    double*& mats = (lb == IIR_LB_LONG) ? h->mats_long : h->mats;
    if (!mats) {                                            // first input of this length class on this handle
        double hm[6 * IIR_MAT];
        iir_block_matrices(h, lb, hm);
        DD_HIP_CHECK(hipMalloc((void**)&mats, sizeof(hm)));
        DD_HIP_CHECK(hipMemcpy(mats, hm, sizeof(hm), hipMemcpyHostToDevice));
    }
    // block and group vectors live in a scratch buffer kept on the handle (allocation and release cost ~0.4 ms per call)
    const size_t blk_bytes = (sizeof(double) * IIR_S * nb * ncomp + 255) & ~(size_t)255;
    const size_t grp_bytes = (sizeof(double) * IIR_S * ng * ncomp + 255) & ~(size_t)255;
    const size_t need = blk_bytes + grp_bytes + sizeof(double) * IIR_S * (ns + 1) * ncomp;
    if (h->scratch_bytes < need) {
        if (h->scratch) DD_HIP_CHECK(hipFree(h->scratch));
        h->scratch = nullptr;
        h->scratch_bytes = 0;
        DD_HIP_CHECK(hipMalloc((void**)&h->scratch, need));
        h->scratch_bytes = need;
    }
    double* blk = h->scratch;
    double* grp = (double*)((char*)h->scratch + blk_bytes);
    double* sup = (double*)((char*)h->scratch + blk_bytes + grp_bytes);
    DDIirCoef C;
    iir_coef(h, &C);
    const unsigned gb = (unsigned)((nb * ncomp + 255) / 256);
    // LDS-staged block kernels need 16-byte aligned buffers (always true for whole device arrays)
    const bool staged = !(((uintptr_t)in | (uintptr_t)out) & 15) && !DD_TUNE_ENV("DD_IIR_UNSTAGED");
    const unsigned gbt = (unsigned)((nb + 256 / ncomp - 1) / (256 / ncomp));
    const size_t lds_t = sizeof(double) * 2 * (256 / ncomp) * iir_lds_row(ncomp);
#define DD_IIR_BLOCKS(SS, WR, SAVE)                                                                                  \
    case SS: {                                                                                                       \
        static DDOncePerDevice attr_set;                                                                             \
        if (attr_set.need()) {                                                                                       \
            DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_iir_blocks_t<SS, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t)); \
            attr_set.mark();                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((k_iir_blocks_t<SS, WR>), dim3(gbt), dim3(256), lds_t, s, in, out, n, ncomp, C, blk, nb, h->state, SAVE, lb); \
    } break;
#define DD_IIR_BLOCKS_ALL(WR, SAVE)                                                                                  \
    switch (S) {                                                                                                     \
        DD_IIR_BLOCKS(1, WR, SAVE) DD_IIR_BLOCKS(2, WR, SAVE) DD_IIR_BLOCKS(3, WR, SAVE) DD_IIR_BLOCKS(4, WR, SAVE)   \
        DD_IIR_BLOCKS(5, WR, SAVE) DD_IIR_BLOCKS(6, WR, SAVE) DD_IIR_BLOCKS(7, WR, SAVE) DD_IIR_BLOCKS(8, WR, SAVE)   \
        DD_IIR_BLOCKS(9, WR, SAVE) DD_IIR_BLOCKS(10, WR, SAVE) DD_IIR_BLOCKS(11, WR, SAVE) DD_IIR_BLOCKS(12, WR, SAVE) \
        DD_IIR_BLOCKS(13, WR, SAVE) DD_IIR_BLOCKS(14, WR, SAVE) DD_IIR_BLOCKS(15, WR, SAVE)                           \
        default: break;                                                                                              \
    }
    // complex128 input: the one-wave LDS-DMA form (DD_IIR_WAVE=0 keeps the 256-thread staged kernels: A/B switch)
    static const bool wave_env = !(DD_TUNE_ENV("DD_IIR_WAVE") && atoi(DD_TUNE_ENV("DD_IIR_WAVE")) == 0);
    const bool wave = staged && ncomp == 2 && wave_env && (lb % IIR_W_CH) == 0;
    const unsigned gbw = (unsigned)((nb + IIR_W_BLOCKS - 1) / IIR_W_BLOCKS);
    const size_t lds_w = (size_t)IIR_W_NB * IIR_W_BUF;
#define DD_IIR_BLOCKS_W(SS, WR, SAVE)                                                                                \
    case SS: {                                                                                                       \
        static DDOncePerDevice attr_w;                                                                               \
        if (attr_w.need()) {                                                                                         \
            DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_iir_blocks_w<SS, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w)); \
            attr_w.mark();                                                                                           \
        }                                                                                                            \
        hipLaunchKernelGGL((k_iir_blocks_w<SS, WR>), dim3(gbw), dim3(64), lds_w, s, reinterpret_cast<const double2*>(in), \
                           reinterpret_cast<double2*>(out), n, C, blk, nb, h->state, SAVE, lb);                      \
    } break;
#define DD_IIR_BLOCKS_W_ALL(WR, SAVE)                                                                                \
    switch (S) {                                                                                                     \
        DD_IIR_BLOCKS_W(1, WR, SAVE) DD_IIR_BLOCKS_W(2, WR, SAVE) DD_IIR_BLOCKS_W(3, WR, SAVE) DD_IIR_BLOCKS_W(4, WR, SAVE)   \
        DD_IIR_BLOCKS_W(5, WR, SAVE) DD_IIR_BLOCKS_W(6, WR, SAVE) DD_IIR_BLOCKS_W(7, WR, SAVE) DD_IIR_BLOCKS_W(8, WR, SAVE)   \
        DD_IIR_BLOCKS_W(9, WR, SAVE) DD_IIR_BLOCKS_W(10, WR, SAVE) DD_IIR_BLOCKS_W(11, WR, SAVE) DD_IIR_BLOCKS_W(12, WR, SAVE) \
        DD_IIR_BLOCKS_W(13, WR, SAVE) DD_IIR_BLOCKS_W(14, WR, SAVE) DD_IIR_BLOCKS_W(15, WR, SAVE)                           \
        default: break;                                                                                              \
    }
    // (the pass that only reads has no output tile: 26 KB per wave, six waves per CU instead of three)
#define DD_IIR_BLOCKS_W32(SS, WR, SAVE)                                                                              \
    case SS: {                                                                                                       \
        constexpr int CHW = (WR) ? 32 : IIR_W32_READ_CH;                                                             \
        const size_t lds_w32 = (size_t)IIR_W32_LDS(CHW, WR);                                                         \
        static DDOncePerDevice attr_w32;                                                                             \
        if (attr_w32.need()) {                                                                                       \
            DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_iir_blocks_w32<SS, WR, CHW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w32)); \
            attr_w32.mark();                                                                                         \
        }                                                                                                            \
        hipLaunchKernelGGL((k_iir_blocks_w32<SS, WR, CHW>), dim3(gbw), dim3(64), lds_w32, s, reinterpret_cast<const float2*>(in), \
                           reinterpret_cast<double2*>(out), n, C, blk, nb, h->state, SAVE, lb);                      \
    } break;
#define DD_IIR_BLOCKS_W32_ALL(WR, SAVE)                                                                              \
    switch (S) {                                                                                                     \
        DD_IIR_BLOCKS_W32(1, WR, SAVE) DD_IIR_BLOCKS_W32(2, WR, SAVE) DD_IIR_BLOCKS_W32(3, WR, SAVE) DD_IIR_BLOCKS_W32(4, WR, SAVE)   \
        DD_IIR_BLOCKS_W32(5, WR, SAVE) DD_IIR_BLOCKS_W32(6, WR, SAVE) DD_IIR_BLOCKS_W32(7, WR, SAVE) DD_IIR_BLOCKS_W32(8, WR, SAVE)   \
        DD_IIR_BLOCKS_W32(9, WR, SAVE) DD_IIR_BLOCKS_W32(10, WR, SAVE) DD_IIR_BLOCKS_W32(11, WR, SAVE) DD_IIR_BLOCKS_W32(12, WR, SAVE) \
        DD_IIR_BLOCKS_W32(13, WR, SAVE) DD_IIR_BLOCKS_W32(14, WR, SAVE) DD_IIR_BLOCKS_W32(15, WR, SAVE)                           \
        default: break;                                                                                              \
    }
    if (in32) { DD_IIR_BLOCKS_W32_ALL(false, 0) }
    else if (wave) { DD_IIR_BLOCKS_W_ALL(false, 0) }
    else if (staged) { DD_IIR_BLOCKS_ALL(false, 0) }
    else hipLaunchKernelGGL(k_iir_blocks, dim3(gb), dim3(256), 0, s, in, out, n, ncomp, C, blk, nb, 0, h->state, 0, lb);
    // the state size is a compile-time constant of the scan kernels: with a run-time S the unrolled
    // double-double loops kept all 15 x 15 predicated products (~4 us per block step)
#define DD_IIR_SCAN(SS)                                                                                              \
    case SS: {                                                                                                       \
        const unsigned cpw = 64 / IirRows<SS>::RP;                                                                   \
        const unsigned gg = (unsigned)((ng * ncomp + cpw - 1) / cpw), gs = (unsigned)((ns * ncomp + cpw - 1) / cpw); \
        const int zero = carry ? 0 : 1;                                                                              \
        hipLaunchKernelGGL(k_iir_groups<SS>, dim3(gg), dim3(64), 0, s, blk, grp, nb, ncomp, mats, 0, IIR_G1);     \
        if (three) {                                                                                                 \
            hipLaunchKernelGGL(k_iir_groups<SS>, dim3(gs), dim3(64), 0, s, grp, sup, ng, ncomp, mats + 2 * IIR_MAT, 0, IIR_G2); \
            hipLaunchKernelGGL(k_iir_group_sweep<SS>, dim3(1), dim3(64), 0, s, sup, ns, ncomp, mats + 4 * IIR_MAT, h->state, zero); \
            hipLaunchKernelGGL(k_iir_groups<SS>, dim3(gs), dim3(64), 0, s, grp, sup, ng, ncomp, mats + 2 * IIR_MAT, 2, IIR_G2); \
        } else {                                                                                                     \
            hipLaunchKernelGGL(k_iir_group_sweep<SS>, dim3(1), dim3(64), 0, s, grp, ng, ncomp, mats + 2 * IIR_MAT, h->state, zero); \
        }                                                                                                            \
        hipLaunchKernelGGL(k_iir_groups<SS>, dim3(gg), dim3(64), 0, s, blk, grp, nb, ncomp, mats, 2, IIR_G1);     \
    } break;
    switch (S) {
        DD_IIR_SCAN(1) DD_IIR_SCAN(2) DD_IIR_SCAN(3) DD_IIR_SCAN(4) DD_IIR_SCAN(5) DD_IIR_SCAN(6) DD_IIR_SCAN(7) DD_IIR_SCAN(8)
        DD_IIR_SCAN(9) DD_IIR_SCAN(10) DD_IIR_SCAN(11) DD_IIR_SCAN(12) DD_IIR_SCAN(13) DD_IIR_SCAN(14) DD_IIR_SCAN(15)
        default: break;
    }
#undef DD_IIR_SCAN
    if (in32) { DD_IIR_BLOCKS_W32_ALL(true, carry ? 1 : 0) }
    else if (wave) { DD_IIR_BLOCKS_W_ALL(true, carry ? 1 : 0) }
    else if (staged) { DD_IIR_BLOCKS_ALL(true, carry ? 1 : 0) }
    else hipLaunchKernelGGL(k_iir_blocks, dim3(gb), dim3(256), 0, s, in, out, n, ncomp, C, blk, nb, 1, h->state, carry ? 1 : 0, lb);
#undef DD_IIR_BLOCKS_W32_ALL
#undef DD_IIR_BLOCKS_W32
#undef DD_IIR_BLOCKS_W_ALL
#undef DD_IIR_BLOCKS_W
#undef DD_IIR_BLOCKS_ALL
#undef DD_IIR_BLOCKS
    DD_LAUNCH_CHECK();
    return DD_OK;
}

extern "C" int dd_iir_f64(dd_iir* h, const double* in, double* out, int64_t n, int is_complex, int carry, void* stream) {
    DD_REQUIRE(h && n >= 0, "h/n");
    if (n == 0) return DD_OK;
    DD_REQUIRE(in && out, "null buffer");
    if (n >= 16 * IIR_LB_SHORT && h->n >= 2 && in != out) return iir_parallel(h, in, out, n, is_complex ? 2 : 1, carry, dd_stream(stream));
    DDIirCoef C;
    iir_coef(h, &C);
    if (!carry) {        // plain lfilter: zero state, nothing kept (filters.py:75)
        for (int k = 0; k < DD_IIR_MAXN; ++k) C.zi[k] = 0.0;
        hipLaunchKernelGGL(k_iir_df2t, dim3(1), dim3(64), 0, dd_stream(stream), in, out, n, is_complex ? 2 : 1, C, h->state, 1, 0, 0);
    } else {
        hipLaunchKernelGGL(k_iir_df2t, dim3(1), dim3(64), 0, dd_stream(stream), in, out, n, is_complex ? 2 : 1, C, h->state, 0, 0, 1);
    }
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// F4 on complex64 input (round 6): filters.py:75 on what the sources hand over -- decode_funcube.py:160, decode_meteorm2.py:157 low-pass the IQ
// stream itself.  lfilter gives complex128 for complex64 input (float64 coefficients): `out` is complex128.  Long inputs: the block-parallel
// passes read the complex64 samples as they are (k_iir_blocks_w32); short or unaligned ones are widened into `out` and filtered there in place.
__global__ void __launch_bounds__(256) k_iir_widen_c64(const float2* __restrict__ in, double2* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float2 v = in[i]; out[i] = make_double2((double)v.x, (double)v.y); }
}
extern "C" int dd_iir_c64(dd_iir* h, const void* in_c64, double* out_c128, int64_t n, int carry, void* stream) {
    DD_REQUIRE(h && n >= 0, "h/n");
    if (n == 0) return DD_OK;
    DD_REQUIRE(in_c64 && out_c128 && (const void*)in_c64 != (const void*)out_c128, "buffers");
    hipStream_t s = dd_stream(stream);
    const bool aligned = !(((uintptr_t)in_c64 | (uintptr_t)out_c128) & 15);
    if (n >= 16 * IIR_LB_SHORT && h->n >= 2 && aligned && !DD_TUNE_ENV("DD_IIR_WIDEN"))
        return iir_parallel(h, (const double*)in_c64, out_c128, n, 2, carry, s, true);
    hipLaunchKernelGGL(k_iir_widen_c64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float2*)in_c64, (double2*)out_c128, n);
    DDIirCoef C;
    iir_coef(h, &C);
    if (!carry) {
        for (int k = 0; k < DD_IIR_MAXN; ++k) C.zi[k] = 0.0;
        hipLaunchKernelGGL(k_iir_df2t, dim3(1), dim3(64), 0, s, out_c128, out_c128, n, 2, C, h->state, 1, 0, 0);
    } else {
        hipLaunchKernelGGL(k_iir_df2t, dim3(1), dim3(64), 0, s, out_c128, out_c128, n, 2, C, h->state, 0, 0, 1);
    }
    DD_LAUNCH_CHECK();
    return DD_OK;
}

template <typename T>
__global__ void __launch_bounds__(256) k_odd_ext(const T* __restrict__ x, T* __restrict__ ext, int64_t n, int edge) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n + 2 * (int64_t)edge) ext[i] = dd_ext_at(x, n, edge, i);
}

extern "C" int dd_iir_filtfilt_f64(dd_iir* h, const double* in, double* out, int64_t n, int is_complex, void* stream) {
    DD_REQUIRE(h && in && out && n >= 0, "arguments");
    const int edge = 3 * h->n;
    if (n <= edge) {
        dd_set_error("The length of the input vector x must be greater than padlen, which is %d.", edge);
        return DD_ERR_INVALID;
    }
    hipStream_t s = dd_stream(stream);
    const int nc = is_complex ? 2 : 1;
    const int64_t N = n + 2 * (int64_t)edge;
    double *ext = nullptr, *y1 = nullptr;
    DD_HIP_CHECK(hipMalloc((void**)&ext, sizeof(double) * N * nc));
    hipError_t e = hipMalloc((void**)&y1, sizeof(double) * N * nc);
    if (e != hipSuccess) {
        (void)hipFree(ext);
        dd_set_error("hipMalloc: %s", hipGetErrorString(e));
        return DD_ERR_NOMEM;
    }
    DDIirCoef C;
    iir_coef(h, &C);
    if (is_complex) hipLaunchKernelGGL(k_odd_ext<double2>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, (const double2*)in, (double2*)ext, n, edge);
    else hipLaunchKernelGGL(k_odd_ext<double>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, in, ext, n, edge);
    // forward pass with zi * ext[0]; backward pass over y1 with zi * y1[N-1], written in place order
    hipLaunchKernelGGL(k_iir_df2t, dim3(1), dim3(64), 0, s, ext, y1, N, nc, C, h->state, 1, 0, 0);
    hipLaunchKernelGGL(k_iir_df2t, dim3(1), dim3(64), 0, s, y1, ext, N, nc, C, h->state, 1, 1, 0);
    hipError_t le = hipGetLastError();
    hipError_t ce = hipMemcpyAsync(out, ext + (int64_t)edge * nc, sizeof(double) * n * nc, hipMemcpyDeviceToDevice, s);
    hipError_t se = hipStreamSynchronize(s);
    (void)hipFree(ext);
    (void)hipFree(y1);
    DD_HIP_CHECK(le);
    DD_HIP_CHECK(ce);
    DD_HIP_CHECK(se);
    return DD_OK;
}

// dd_code_warmup (dd_runtime.hip): the runtime loads a translation unit's code object when one of its kernels is first named
int dd_code_touch_fir(void) {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, (const void*)k_fill_f64) == hipSuccess ? DD_OK : DD_ERR_HIP;
}
