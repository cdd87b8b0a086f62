// Audio-rate rows of the NOAA tail (R2, A1, X1, X2), float64 on the
// device so that sync index picks stay bit-exact (H7).  Sizes here are 1e5..1e6
// samples -- far from any roofline; the transforms are plain library FFTs (hipFFT)
// with hand-written pre/post kernels, the correlation is a direct f64 kernel and the
// peak pick uses device sort / stream compaction (hipCUB) + a tiny sequential grouping.
#include "dd_common.h"
#include <hipfft/hipfft.h>
#include <hipcub/hipcub.hpp>
#include <map>
#include <mutex>
#include <vector>
#include <algorithm>

#define DD_FFT_CHECK(expr)                                                     \
    do {                                                                       \
        hipfftResult _r = (expr);                                              \
        if (_r != HIPFFT_SUCCESS) {                                            \
            dd_set_error("%s failed: hipfft error %d", #expr, (int)_r);        \
            return DD_ERR_HIP;                                                 \
        }                                                                      \
    } while (0)

// ---------------------------------------------------------------- plan cache
struct PlanKey {
    int dev;
    int type;
    int64_t n;
    int batch;
    bool operator<(const PlanKey& o) const {
        if (dev != o.dev) return dev < o.dev;
        if (type != o.type) return type < o.type;
        if (n != o.n) return n < o.n;
        return batch < o.batch;
    }
};
static std::mutex g_plan_mu;
static std::map<PlanKey, hipfftHandle> g_plans;

static int get_plan(hipfftHandle* out, hipfftType type, int64_t n, int batch, hipStream_t s) {
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    DD_REQUIRE(n >= 1 && n < (1ll << 31), "FFT length");
    std::lock_guard<std::mutex> lk(g_plan_mu);
    PlanKey k{dev, (int)type, n, batch};
    auto it = g_plans.find(k);
    if (it == g_plans.end()) {
        hipfftHandle h;
        DD_FFT_CHECK(hipfftPlan1d(&h, (int)n, type, batch));
        it = g_plans.emplace(k, h).first;
    }
    DD_FFT_CHECK(hipfftSetStream(it->second, s));
    *out = it->second;
    return DD_OK;
}

static inline unsigned grid1(int64_t n) { return (unsigned)((n + 255) / 256); }

// ---------------------------------------------------------------- A1: abs(hilbert(x)) per block
// scipy.signal.hilbert: Xf = fft(x); h[0] = 1, h[1..(N-1)/2 or N/2-1] = 2, h[N/2] = 1 (N even),
// 0 elsewhere; ifft(Xf * h); demod_am takes the magnitude (demod_am.py:29).
__global__ void __launch_bounds__(256) k_real_to_cplx(const double* __restrict__ in, double2* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_double2(in[i], 0.0);
}
__global__ void __launch_bounds__(256) k_hilbert_mask(double2* __restrict__ X, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double h;
    if ((n & 1) == 0) h = (i == 0 || i == n / 2) ? 1.0 : (i < n / 2 ? 2.0 : 0.0);
    else h = (i == 0) ? 1.0 : (i < (n + 1) / 2 ? 2.0 : 0.0);
    X[i] = make_double2(X[i].x * h, X[i].y * h);
}
__global__ void __launch_bounds__(256) k_cplx_abs(const double2* __restrict__ in, double* __restrict__ out, int64_t n, double inv_n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = hypot(in[i].x * inv_n, in[i].y * inv_n);
}

static int envelope_block(const double* in, double* out, int64_t n, double2* work, hipStream_t s) {
    hipfftHandle plan;
    int rc = get_plan(&plan, HIPFFT_Z2Z, n, 1, s);
    if (rc != DD_OK) return rc;
    hipLaunchKernelGGL(k_real_to_cplx, dim3(grid1(n)), dim3(256), 0, s, in, work, n);
    DD_FFT_CHECK(hipfftExecZ2Z(plan, (hipfftDoubleComplex*)work, (hipfftDoubleComplex*)work, HIPFFT_FORWARD));
    hipLaunchKernelGGL(k_hilbert_mask, dim3(grid1(n)), dim3(256), 0, s, work, n);
    DD_FFT_CHECK(hipfftExecZ2Z(plan, (hipfftDoubleComplex*)work, (hipfftDoubleComplex*)work, HIPFFT_BACKWARD));
    hipLaunchKernelGGL(k_cplx_abs, dim3(grid1(n)), dim3(256), 0, s, work, out, n, 1.0 / (double)n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

extern "C" int dd_am_envelope_f64(const double* in, double* out, int64_t n, int64_t block, void* stream) {
    DD_REQUIRE(n >= 0 && block >= 1, "n/block");
    if (n == 0) return DD_OK;
    DD_REQUIRE(in && out, "null buffer");
    hipStream_t s = dd_stream(stream);
    // block list by the chunker rule (decode_noaa.py:644-653 via chunker.py:36-45)
    std::vector<std::pair<int64_t, int64_t>> blocks;
    int64_t i = 0;
    while (i + block < n) {
        blocks.push_back({i, i + block});
        i += block;
    }
    if (blocks.empty()) blocks.push_back({0, n});
    else if (blocks.back().second != n) blocks.push_back({blocks.back().second, n});
    int64_t maxlen = 0;
    for (auto& b : blocks) maxlen = std::max(maxlen, b.second - b.first);
    double2* work = nullptr;
    DD_HIP_CHECK(hipMalloc((void**)&work, sizeof(double2) * maxlen));
    int rc = DD_OK;
    for (auto& b : blocks) {
        rc = envelope_block(in + b.first, out + b.first, b.second - b.first, work, s);
        if (rc != DD_OK) break;
    }
    hipError_t e = hipStreamSynchronize(s);
    hipFree(work);
    if (rc != DD_OK) return rc;
    DD_HIP_CHECK(e);
    return DD_OK;
}

// ---------------------------------------------------------------- R2: scipy.signal.resample (real input)
// X = rfft(x); Y[:nyq] = X[:nyq] (nyq = min(num,Nx)/2 + 1), Nyquist bin doubled when
// down-sampling / halved when up-sampling an even N; y = irfft(Y, num) * num / Nx.
__global__ void __launch_bounds__(256) k_resample_bins(const double2* __restrict__ X, double2* __restrict__ Y, int64_t nx_bins,
                                                       int64_t ny_bins, int64_t N, int64_t num, int64_t Nx) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= ny_bins) return;
    const int64_t nyq = N / 2 + 1;
    double2 v = make_double2(0.0, 0.0);
    if (k < nyq && k < nx_bins) v = X[k];
    if ((N & 1) == 0 && k == N / 2) {
        if (num < Nx) { v.x *= 2.0; v.y *= 2.0; }
        else if (Nx < num) { v.x *= 0.5; v.y *= 0.5; }
    }
    Y[k] = v;
}
__global__ void __launch_bounds__(256) k_scale_f64(double* __restrict__ y, int64_t n, double f) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] *= f;
}

extern "C" int dd_resample_fft_f64(const double* in, double* out, int64_t n, int64_t num, void* stream) {
    DD_REQUIRE(n >= 1 && num >= 1, "n/num");
    DD_REQUIRE(in && out, "null buffer");
    hipStream_t s = dd_stream(stream);
    hipfftHandle pf, pb;
    int rc = get_plan(&pf, HIPFFT_D2Z, n, 1, s);
    if (rc != DD_OK) return rc;
    rc = get_plan(&pb, HIPFFT_Z2D, num, 1, s);
    if (rc != DD_OK) return rc;
    const int64_t nxb = n / 2 + 1, nyb = num / 2 + 1;
    double2 *X = nullptr, *Y = nullptr;
    double* tmp = nullptr;       // D2Z may overwrite its input: work on a copy
    DD_HIP_CHECK(hipMalloc((void**)&X, sizeof(double2) * nxb));
    DD_HIP_CHECK(hipMalloc((void**)&Y, sizeof(double2) * nyb));
    DD_HIP_CHECK(hipMalloc((void**)&tmp, sizeof(double) * n));
    DD_HIP_CHECK(hipMemcpyAsync(tmp, in, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    hipfftResult r1 = hipfftExecD2Z(pf, tmp, (hipfftDoubleComplex*)X);
    const int64_t N = num < n ? num : n;
    hipLaunchKernelGGL(k_resample_bins, dim3(grid1(nyb)), dim3(256), 0, s, X, Y, nxb, nyb, N, num, n);
    hipfftResult r2 = hipfftExecZ2D(pb, (hipfftDoubleComplex*)Y, out);
    hipLaunchKernelGGL(k_scale_f64, dim3(grid1(num)), dim3(256), 0, s, out, num, 1.0 / (double)n);
    hipError_t e = hipStreamSynchronize(s);
    hipFree(X);
    hipFree(Y);
    hipFree(tmp);
    if (r1 != HIPFFT_SUCCESS || r2 != HIPFFT_SUCCESS) {
        dd_set_error("hipfft exec failed (%d, %d)", (int)r1, (int)r2);
        return DD_ERR_HIP;
    }
    DD_HIP_CHECK(e);
    return DD_OK;
}

// ---------------------------------------------------------------- X1: normalised correlation
// cor = correlate(h, needle, 'same'); sums = convolve(h*h, ones(m), 'same');
// out = cor / sqrt(sums * sum(needle^2))  (decode_noaa.py:671-673).  Both windows are
// h[k-(m-1) .. k], k = i + (m-1)/2, so one pass computes both (float64, direct form).
__global__ void __launch_bounds__(256) k_xcorr_norm(const double* __restrict__ h, int64_t n, const double* __restrict__ v, int m,
                                                    double vv, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t k = i + (m - 1) / 2;
    const int64_t a0 = k - (m - 1);
    double c = 0.0, e = 0.0;
    int t0 = a0 < 0 ? (int)(-a0) : 0;
    int t1 = (a0 + m > n) ? (int)(n - a0) : m;
    for (int t = t0; t < t1; ++t) {
        const double x = h[a0 + t];
        c = fma(v[t], x, c);
        e = fma(x, x, e);
    }
    out[i] = c / sqrt(e * vv);
}

// Run-length form.  The sync needles are np.repeat(bits, rep) * 233 + 11) / 255 (decode_noaa.py:690-694):
// 40 bits held for rep samples each, i.e. a dozen constant runs over 560 (crude) or 19 680 (accurate)
// samples.  Over a run the correlation is value * (window sum of h), so with prefix sums P of h and Q of
// h^2 an output costs two lookups per run and two for the energy instead of m multiply-adds: the accurate
// window went from 2.1 ms (2.3e9 MAC) to tens of microseconds.  float64 prefix sums over <= 1e6 values
// of O(1): the window differences carry ~1e-13 relative error -- the size of the difference between the
// direct sum and SciPy's FFT method, and well inside the 1e-9 of the stage.
#define DD_XCORR_MAX_RUNS 64
struct DDRuns {
    int nr;
    int start[DD_XCORR_MAX_RUNS + 1];
    double val[DD_XCORR_MAX_RUNS];
};
struct SqOp {
    __host__ __device__ double operator()(const double& x) const { return x * x; }
};

__global__ void __launch_bounds__(256) k_xcorr_runs(const double* __restrict__ P, const double* __restrict__ Q, int64_t n, int m,
                                                    const DDRuns R, double vv, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t a0 = i + (m - 1) / 2 - (m - 1);           // window h[a0 .. a0+m-1], zero outside [0, n)
    auto at = [&](const double* S, int64_t x) { return S[x < 0 ? 0 : (x > n ? n : x)]; };
    double c = 0.0;
    double lo = at(P, a0);
    for (int r = 0; r < R.nr; ++r) {
        const double hi = at(P, a0 + R.start[r + 1]);
        c = fma(R.val[r], hi - lo, c);
        lo = hi;
    }
    double e = at(Q, a0 + m) - at(Q, a0);
    if (!(e > 1e-13 * Q[n])) { c = 0.0; e = 0.0; }         // an all-zero window: 0/0 like the direct form
    out[i] = c / sqrt(e * vv);
}

static int xcorr_runs(const double* h, int64_t n, const double* needle_host, int m, double vv, const DDRuns& R, double* out, hipStream_t s) {
    double *P = nullptr, *Q = nullptr;
    void* tmp = nullptr;
    size_t tb1 = 0, tb2 = 0;
    hipcub::TransformInputIterator<double, SqOp, const double*> h2(h, SqOp());
    DD_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, tb1, h, P, (int)n, s));
    DD_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, tb2, h2, Q, (int)n, s));
    const size_t tb = tb1 > tb2 ? tb1 : tb2;
    DD_HIP_CHECK(hipMalloc((void**)&P, sizeof(double) * (2 * (n + 1))));
    Q = P + (n + 1);
    hipError_t e = hipMalloc(&tmp, tb ? tb : 16);
    if (e != hipSuccess) {
        hipFree(P);
        dd_set_error("hipMalloc: %s", hipGetErrorString(e));
        return DD_ERR_NOMEM;
    }
    hipError_t e1 = hipMemsetAsync(P, 0, sizeof(double), s);
    hipError_t e2 = hipMemsetAsync(Q, 0, sizeof(double), s);
    size_t t1 = tb, t2 = tb;
    hipError_t e3 = hipcub::DeviceScan::InclusiveSum(tmp, t1, h, P + 1, (int)n, s);
    hipError_t e4 = hipcub::DeviceScan::InclusiveSum(tmp, t2, h2, Q + 1, (int)n, s);
    hipLaunchKernelGGL(k_xcorr_runs, dim3(grid1(n)), dim3(256), 0, s, P, Q, n, m, R, vv, out);
    hipError_t le = hipGetLastError();
    hipError_t se = hipStreamSynchronize(s);
    hipFree(tmp);
    hipFree(P);
    (void)needle_host;
    DD_HIP_CHECK(e1); DD_HIP_CHECK(e2); DD_HIP_CHECK(e3); DD_HIP_CHECK(e4); DD_HIP_CHECK(le); DD_HIP_CHECK(se);
    return DD_OK;
}

extern "C" int dd_xcorr_norm_f64(const double* h, int64_t n, const double* needle_host, int m, double* out, void* stream) {
    DD_REQUIRE(n >= 1 && m >= 1 && m <= n, "n/m");
    DD_REQUIRE(h && needle_host && out, "null buffer");
    hipStream_t s = dd_stream(stream);
    {
        // piecewise-constant needle with few runs -> prefix-sum form
        DDRuns R;
        R.nr = 0;
        bool ok = n < (int64_t)1 << 31;
        R.start[0] = 0;
        for (int t = 0; t < m && ok; ++t) {
            if (t == 0 || needle_host[t] != needle_host[t - 1]) {
                if (R.nr == DD_XCORR_MAX_RUNS) { ok = false; break; }
                R.start[R.nr] = t;
                R.val[R.nr] = needle_host[t];
                ++R.nr;
            }
        }
        if (ok && m >= 16 * R.nr) {
            R.start[R.nr] = m;
            double vv = 0.0;
            for (int t = 0; t < m; ++t) vv += needle_host[t] * needle_host[t];
            return xcorr_runs(h, n, needle_host, m, vv, R, out, s);
        }
    }
    double* v = nullptr;
    DD_HIP_CHECK(hipMalloc((void**)&v, sizeof(double) * m));
    DD_HIP_CHECK(hipMemcpyAsync(v, needle_host, sizeof(double) * m, hipMemcpyHostToDevice, s));
    double vv = 0.0;
    for (int t = 0; t < m; ++t) vv += needle_host[t] * needle_host[t];
    hipLaunchKernelGGL(k_xcorr_norm, dim3(grid1(n)), dim3(256), 0, s, h, n, v, m, vv, out);
    hipError_t le = hipGetLastError();
    hipError_t e = hipStreamSynchronize(s);
    hipFree(v);
    DD_HIP_CHECK(le);
    DD_HIP_CHECK(e);
    return DD_OK;
}

// ---------------------------------------------------------------- X2: peak pick (decode_noaa.py:713-751)
struct GtThr {
    const double* cor;
    double thr;
    __host__ __device__ bool operator()(const int64_t& i) const { return cor[i] > thr; }
};

extern "C" int dd_find_peaks_f64(const double* cor, int64_t n, double samp_rate, int needle_len,
                                 int64_t* peaks_host, int max_peaks, int* n_peaks, void* stream) {
    DD_REQUIRE(cor && n >= 1 && samp_rate > 0 && peaks_host && n_peaks && max_peaks >= 1, "arguments");
    hipStream_t s = dd_stream(stream);
    const int K = (int)(2 * ((double)n / samp_rate)) + 2;                 // expectedPeaks (:714)
    DD_REQUIRE(K <= n, "signal shorter than the expected peak count");
    // ---- mean of the K largest and K smallest values (argpartition, :717-723): device sort
    double* sorted = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    DD_HIP_CHECK(hipMalloc((void**)&sorted, sizeof(double) * n));
    DD_HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, cor, sorted, (int)n, 0, 64, s));
    DD_HIP_CHECK(hipMalloc(&tmp, tmp_bytes));
    DD_HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, cor, sorted, (int)n, 0, 64, s));
    std::vector<double> lo(K), hi(K);
    DD_HIP_CHECK(hipMemcpyAsync(lo.data(), sorted, sizeof(double) * K, hipMemcpyDeviceToHost, s));
    DD_HIP_CHECK(hipMemcpyAsync(hi.data(), sorted + (n - K), sizeof(double) * K, hipMemcpyDeviceToHost, s));
    DD_HIP_CHECK(hipStreamSynchronize(s));
    hipFree(tmp);
    hipFree(sorted);
    double sum_hi = 0.0, sum_lo = 0.0;
    for (int i = 0; i < K; ++i) { sum_hi += hi[i]; sum_lo += lo[i]; }
    double avgpk = sum_hi / K;
    avgpk -= 0.25 * (avgpk - sum_lo / K);                                 // NOAA_PEAKHEIGHTWIGGLE (:723)
    // ---- candidates cor > threshold, ascending index (:726): device stream compaction
    int64_t* cand = nullptr;
    int* d_count = nullptr;
    DD_HIP_CHECK(hipMalloc((void**)&cand, sizeof(int64_t) * n));
    DD_HIP_CHECK(hipMalloc((void**)&d_count, sizeof(int)));
    hipcub::CountingInputIterator<int64_t> idx(0);
    GtThr pred{cor, avgpk};
    tmp = nullptr;
    tmp_bytes = 0;
    DD_HIP_CHECK(hipcub::DeviceSelect::If(nullptr, tmp_bytes, idx, cand, d_count, (int)n, pred, s));
    DD_HIP_CHECK(hipMalloc(&tmp, tmp_bytes));
    DD_HIP_CHECK(hipcub::DeviceSelect::If(tmp, tmp_bytes, idx, cand, d_count, (int)n, pred, s));
    int count = 0;
    DD_HIP_CHECK(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, s));
    DD_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<int64_t> ci(count);
    std::vector<double> cv(count);
    if (count > 0) {
        DD_HIP_CHECK(hipMemcpy(ci.data(), cand, sizeof(int64_t) * count, hipMemcpyDeviceToHost));
        // candidate heights: gather on the host side from a device copy of the few values
        std::vector<double> all;
        // the candidates are few (a handful of samples around each sync): fetch them one run at a time
        int64_t run_start = 0;
        while (run_start < count) {
            int64_t run_end = run_start;
            while (run_end + 1 < count && ci[run_end + 1] == ci[run_end] + 1) ++run_end;
            DD_HIP_CHECK(hipMemcpy(cv.data() + run_start, cor + ci[run_start], sizeof(double) * (run_end - run_start + 1),
                                   hipMemcpyDeviceToHost));
            run_start = run_end + 1;
        }
    }
    hipFree(tmp);
    hipFree(cand);
    hipFree(d_count);
    // ---- group by >= 0.45 s from the running maximum, first maximum wins (:729-746)
    const double min_dist = 0.45 * samp_rate;                             // NOAA_MINPEAKDIST
    std::vector<int64_t> peaks;
    bool have = false;
    double cur_max = 0.0;
    int64_t cur_idx = 0;
    for (int q = 0; q < count; ++q) {
        if (have && (double)(ci[q] - cur_idx) >= min_dist) {
            peaks.push_back(cur_idx);
            have = false;
        }
        if (!have || cur_max < cv[q]) {
            cur_max = cv[q];
            cur_idx = ci[q];
            have = true;
        }
    }
    if (have) peaks.push_back(cur_idx);
    // the reference appends currentMaxIndex even when there was no candidate (None): an
    // empty candidate list cannot happen (the maximum itself exceeds the threshold)
    const int shift = needle_len / 2;                                     // int(len(sync)/2) (:749)
    for (auto& p : peaks) p -= shift;
    std::sort(peaks.begin(), peaks.end());
    if ((int)peaks.size() > max_peaks) {
        dd_set_error("dd_find_peaks_f64: %d peaks found, buffer holds %d", (int)peaks.size(), max_peaks);
        return DD_ERR_INVALID;
    }
    for (size_t i = 0; i < peaks.size(); ++i) peaks_host[i] = peaks[i];
    *n_peaks = (int)peaks.size();
    return DD_OK;
}
