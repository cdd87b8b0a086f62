// Audio-rate rows of the NOAA tail (R2, A1, X1, X2), float64 on the
// device so that sync index picks stay bit-exact (H7).  Sizes here are 1e5..1e6
// samples -- far from any roofline; the transforms are plain library FFTs (hipFFT)
// with hand-written pre/post kernels, the correlation is a direct f64 kernel and the
// peak pick uses device sort / stream compaction (hipCUB) + a tiny sequential grouping.
#include "dd_common.h"
#include <hipfft/hipfft.h>
#include <map>
#include <mutex>
#include <vector>
#include <algorithm>
#include <chrono>
#include <complex>

#define DD_FFT_CHECK(expr)                                                     \
    do {                                                                       \
        hipfftResult _r = (expr);                                              \
        if (_r != HIPFFT_SUCCESS) {                                            \
            dd_set_error("%s failed: hipfft error %d", #expr, (int)_r);        \
            return DD_ERR_HIP;                                                 \
        }                                                                      \
    } while (0)

// ---------------------------------------------------------------- plan cache
// (a plan is bound to the stream it is used on -- hipfftSetStream -- and owns a work area: one plan per stream, so two
// host threads on two streams never re-bind or share one)
struct PlanKey {
    int dev;
    int type;
    int64_t n;
    int batch;
    hipStream_t stream;
    bool operator<(const PlanKey& o) const {
        if (dev != o.dev) return dev < o.dev;
        if (type != o.type) return type < o.type;
        if (n != o.n) return n < o.n;
        if (batch != o.batch) return batch < o.batch;
        return stream < o.stream;
    }
};
static std::mutex g_plan_mu;
static std::map<PlanKey, hipfftHandle> g_plans;

static int get_plan(hipfftHandle* out, hipfftType type, int64_t n, int batch, hipStream_t s) {
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    DD_REQUIRE(n >= 1 && n < (1ll << 31), "FFT length");
    std::lock_guard<std::mutex> lk(g_plan_mu);
    PlanKey k{dev, (int)type, n, batch, s};
    auto it = g_plans.find(k);
    if (it == g_plans.end()) {
        hipfftHandle h;
        DD_FFT_CHECK(hipfftPlan1d(&h, (int)n, type, batch));
        DD_FFT_CHECK(hipfftSetStream(h, s));
        it = g_plans.emplace(k, h).first;
    }
    *out = it->second;
    return DD_OK;
}

static inline unsigned grid1(int64_t n) { return (unsigned)((n + 255) / 256); }

// An entry point that enqueues kernels on buffers another call may free or reuse (the Hilbert-kernel spectra of g_hilb, the pageable
// staging vectors of the uploads) and then returns EARLY on an error must not leave that work in flight: this guard synchronises the
// stream when the function is left without having done so itself.
struct DDSyncOnExit {
    hipStream_t s;
    bool armed;
    explicit DDSyncOnExit(hipStream_t st) : s(st), armed(true) {}
    void done() { armed = false; }
    ~DDSyncOnExit() { if (armed) (void)hipStreamSynchronize(s); }
};

// ---------------------------------------------------------------- own float64 cyclic convolution of length 2^17 / 2^18 (dd_hconv_kernels.h)
#include "dd_hconv_kernels.h"
static std::mutex g_hc_mu;
static double2* g_hc_tab[64] = {nullptr};                        // device -> W_512^j (512) | W_{2^18}^j (512) | W_{2^17}^j (256)
static bool hc_length_ok(int64_t M) { return M == ((int64_t)1 << 17) || M == ((int64_t)1 << 18); }
// lg: 9 (M = 2^18) or 8 (M = 2^17)
static int hc_tables(int lg, const double2** TA, const double2** TB) {
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    DD_REQUIRE(dev >= 0 && dev < 64, "device index");
    std::lock_guard<std::mutex> lk(g_hc_mu);
    if (!g_hc_tab[dev]) {
        std::vector<double2> h(2 * DD_HC_N + 256);
        const long double tp = 6.283185307179586476925286766559L;
        for (int j = 0; j < DD_HC_N; ++j) {
            h[j] = make_double2((double)cosl(tp * j / DD_HC_N), (double)-sinl(tp * j / DD_HC_N));
            h[DD_HC_N + j] = make_double2((double)cosl(tp * j / 262144.0L), (double)-sinl(tp * j / 262144.0L));
        }
        for (int j = 0; j < 256; ++j) h[2 * DD_HC_N + j] = make_double2((double)cosl(tp * j / 131072.0L), (double)-sinl(tp * j / 131072.0L));
        double2* d = nullptr;
        DD_HIP_CHECK(hipMalloc((void**)&d, sizeof(double2) * h.size()));
        hipError_t e = hipMemcpy(d, h.data(), sizeof(double2) * h.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); dd_set_error("twiddle table upload: %s", hipGetErrorString(e)); return DD_ERR_HIP; }
        g_hc_tab[dev] = d;
    }
    *TA = g_hc_tab[dev];
    *TB = g_hc_tab[dev] + (lg == 9 ? DD_HC_N : 2 * DD_HC_N);
    return DD_OK;
}
// the column passes take 64 KB of dynamic LDS: once per device and instantiation
template <int LG, typename SRC, typename DST>
static int hc_ready() {
    static DDOncePerDevice once;
    if (once.need()) {
        DD_HIP_CHECK((hc_set_lds_attr<LG, SRC, DST>()));
        once.mark();
    }
    return DD_OK;
}
// envelope of `nwin` windows (two per image) through the three launches; HHp: the kernel spectrum in row-pass order
static int hc_envelope(int64_t M, const float2* X, int64_t L, int nwin, const double2* HHp, double2* W, double* ENV, hipStream_t s) {
    const int lg = M == ((int64_t)1 << 18) ? 9 : 8;
    const double2 *TA = nullptr, *TB = nullptr;
    int rc = hc_tables(lg, &TA, &TB);
    if (rc != DD_OK) return rc;
    const HcEnvIO io = {X, L, L - 1, nwin, ENV};
    const HcOneSpec sp = {HHp};
    const int pairs = (nwin + 1) / 2;
    if (lg == 9) {
        rc = hc_ready<9, HcEnvIO, HcEnvIO>();
        if (rc == DD_OK) hc_convolve<9>(io, sp, io, W, pairs, TA, TB, s);
    } else {
        rc = hc_ready<8, HcEnvIO, HcEnvIO>();
        if (rc == DD_OK) hc_convolve<8>(io, sp, io, W, pairs, TA, TB, s);
    }
    return rc;
}

struct DDCztKey;
static void czt_forget_stream(int dev, hipStream_t s);
// the stream is about to be destroyed (dd_stream_destroy, after it has been synchronised): its plans (with their work areas) and
// its chirp-z tables go with it -- a later stream that happens to get the same address must not inherit a plan bound to a dead
// stream, and dead entries must not fill the 64-entry chirp-z cache
void dd_audio_forget_stream(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        for (auto it = g_plans.begin(); it != g_plans.end();) {
            if (it->first.dev == dev && it->first.stream == s) {
                (void)hipfftDestroy(it->second);
                it = g_plans.erase(it);
            } else {
                ++it;
            }
        }
    }
    czt_forget_stream(dev, s);
}

// ---------------------------------------------------------------- A1: abs(hilbert(x)) per block
// scipy.signal.hilbert: Xf = fft(x); h[0] = 1, h[1..(N-1)/2 or N/2-1] = 2, h[N/2] = 1 (N even),
// 0 elsewhere; ifft(Xf * h); demod_am takes the magnitude (demod_am.py:29).
__global__ void __launch_bounds__(256) k_real_to_cplx(const double* __restrict__ in, double2* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_double2(in[i], 0.0);
}
// blockIdx.y = block (or window) of the batch
__global__ void __launch_bounds__(256) k_hilbert_mask_b(double2* __restrict__ X, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double h;
    if ((n & 1) == 0) h = (i == 0 || i == n / 2) ? 1.0 : (i < n / 2 ? 2.0 : 0.0);
    else h = (i == 0) ? 1.0 : (i < (n + 1) / 2 ? 2.0 : 0.0);
    double2* p = X + (int64_t)blockIdx.y * n + i;
    *p = make_double2(p->x * h, p->y * h);
}
__global__ void __launch_bounds__(256) k_cplx_abs_b(const double2* __restrict__ in, double* __restrict__ out, int64_t n, double inv_n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double2 v = in[(int64_t)blockIdx.y * n + i];
    out[(int64_t)blockIdx.y * n + i] = hypot(v.x * inv_n, v.y * inv_n);
}

// (defined with the accurate-sync code below: the blocks' envelopes through the own float64 transform, no FFT-library plan)
static std::mutex g_sync_mu;
static int64_t hc_block_len(int64_t N, bool* split);
static int hc_block_envelope(const double* x, double* env, int64_t N, int jobs, bool split, int64_t M, double2* T, hipStream_t s);

// `batch` consecutive blocks of n samples each: one batched transform pair
static int envelope_blocks(const double* in, double* out, int64_t n, int batch, double2* work, hipStream_t s) {
    hipfftHandle plan;
    int rc = get_plan(&plan, HIPFFT_Z2Z, n, batch, s);
    if (rc != DD_OK) return rc;
    hipLaunchKernelGGL(k_real_to_cplx, dim3(grid1(n * batch)), dim3(256), 0, s, in, work, n * batch);
    DD_FFT_CHECK(hipfftExecZ2Z(plan, (hipfftDoubleComplex*)work, (hipfftDoubleComplex*)work, HIPFFT_FORWARD));
    hipLaunchKernelGGL(k_hilbert_mask_b, dim3(grid1(n), batch), dim3(256), 0, s, work, n);
    DD_FFT_CHECK(hipfftExecZ2Z(plan, (hipfftDoubleComplex*)work, (hipfftDoubleComplex*)work, HIPFFT_BACKWARD));
    hipLaunchKernelGGL(k_cplx_abs_b, dim3(grid1(n), batch), dim3(256), 0, s, work, out, n, 1.0 / (double)n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

extern "C" int dd_am_envelope_f64(const double* in, double* out, int64_t n, int64_t block, void* stream) {
    DD_REQUIRE(n >= 0 && block >= 1, "n/block");
    if (n == 0) return DD_OK;
    DD_REQUIRE(in && out, "null buffer");
    hipStream_t s = dd_stream(stream);
    // block list by the chunker rule (decode_noaa.py:644-653 via chunker.py:36-45): full blocks while one more
    // fits strictly inside, then the remainder (a full-size last block when n is an exact multiple)
    int64_t nfull = 0;
    while ((nfull + 1) * block < n) ++nfull;
    const int64_t rem = n - nfull * block;                  // 1 .. block
    const int GB = 16;                                       // full blocks per batched transform
    const int64_t gb = nfull < GB ? nfull : GB;
    // Round 5: blocks that fit the own float64 transform go through it (hc_block_envelope: a 240 000-sample block by the even / odd split
    // of the Hilbert kernel) -- no FFT-library plan, whose creation costs a process's first call 0.9 s.  DD_AM_HILBERT=lib: the library.
    static const char* amh_env = getenv("DD_AM_HILBERT");
    const bool own_ok = !(amh_env && !strcmp(amh_env, "lib"));
    bool split_b = false, split_r = false;
    const int64_t Mb = (own_ok && nfull > 0) ? hc_block_len(block, &split_b) : 0;
    const int64_t Mr = own_ok ? hc_block_len(rem, &split_r) : 0;
    const int64_t wlen = nfull ? std::max<int64_t>(Mb ? (split_b ? gb * Mb : Mb) : 0, Mb ? 0 : gb * block) : 0;
    const int64_t wrem = Mr ? Mr : rem;
    DDScratchLock scr;                      // held until this entry point has enqueued everything
    int rc = scr.get(sizeof(double2) * (size_t)(wlen > wrem ? wlen : wrem), s);
    char* base = scr.ptr;
    if (rc != DD_OK) return rc;
    double2* work = reinterpret_cast<double2*>(base);
    if (Mb) {
        std::lock_guard<std::mutex> lk(g_sync_mu);           // (the kernel-spectrum cache)
        const int per = split_b ? (int)gb : 1;
        for (int64_t b0 = 0; b0 < nfull && rc == DD_OK; b0 += per)
            rc = hc_block_envelope(in + b0 * block, out + b0 * block, block, (int)(nfull - b0 < per ? nfull - b0 : per), split_b, Mb, work, s);
    } else {
        for (int64_t b0 = 0; b0 < nfull && rc == DD_OK; b0 += GB) {
            const int nbk = (int)(nfull - b0 < GB ? nfull - b0 : GB);
            rc = envelope_blocks(in + b0 * block, out + b0 * block, block, nbk, work, s);
        }
    }
    if (rc == DD_OK) {
        if (Mr) {
            std::lock_guard<std::mutex> lk(g_sync_mu);
            rc = hc_block_envelope(in + nfull * block, out + nfull * block, rem, 1, split_r, Mr, work, s);
        } else {
            rc = envelope_blocks(in + nfull * block, out + nfull * block, rem, 1, work, s);
        }
    }
    return rc;
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

// ---------------------------------------------------------------- polyphase rational resampler (stream form)
// y[j] = sum_k hp[k] xu[(j + npr) down - k], xu = the input with up-1 zeros stuffed between samples (SciPy's
// resample_poly / upfirdn definition; hp = front-padded, up-scaled low-pass).  Only k = k0 + q up contribute
// (k0 = t mod up, t = (j + npr) down), pairing hp[k0 + q up] with x[i0 - q], i0 = (t - k0) / up: one lane per
// output walks its polyphase branch.  Inputs before the chunk come from the carried history (the last `nh`
// inputs of the stream), inputs past `n_total` (only when flushing) are zeros.
struct dd_rpoly {
    int up, down, ntaps, q;          // q = inputs an output can reach back: ceil(ntaps / up)
    int64_t npr;
    double* taps;                    // device, ntaps
    double* hist[2];                 // device, q each (oldest first), ping-pong
    int hpar, nh;                    // valid history samples
    int64_t n_in, j_next;            // inputs consumed, next output index
};

__global__ void __launch_bounds__(256) k_rpoly(const double* __restrict__ in, int64_t n, int64_t a, const double* __restrict__ hist, int nh,
                                               const double* __restrict__ taps, int ntaps, int up, int down, int64_t npr,
                                               int64_t j0, int64_t n_out, double* __restrict__ out) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    const int64_t t = (j0 + o + npr) * (int64_t)down;
    const int k0 = (int)(t % up);
    const int64_t i0 = (t - k0) / up;
    double acc = 0.0;
    int64_t i = i0;
    for (int k = k0; k < ntaps; k += up, --i) {
        if (i < a - nh) break;                             // older than anything kept: zeros from here on (stream start)
        if (i >= a + n) continue;                          // past the end of the stream (flush): zero
        const double x = i >= a ? in[i - a] : hist[nh - (a - i)];
        acc = fma(taps[k], x, acc);
    }
    out[o] = acc;
}
// new history = the last q samples of (old history ++ chunk)
__global__ void k_rpoly_hist(const double* __restrict__ in, int64_t n, const double* __restrict__ hold, int nh, double* __restrict__ hnew, int nh_new) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nh_new) return;
    const int64_t src = (int64_t)nh + n - nh_new + i;      // index into old history ++ chunk
    hnew[i] = src < nh ? hold[src] : in[src - nh];
}

extern "C" int dd_rpoly_create(dd_rpoly** h, const double* taps_host, int ntaps, int up, int down, int64_t n_pre_remove) {
    DD_REQUIRE(h && taps_host && ntaps >= 1 && up >= 1 && down >= 1 && n_pre_remove >= 0, "arguments");
    dd_rpoly* r = new dd_rpoly();
    r->up = up; r->down = down; r->ntaps = ntaps; r->npr = n_pre_remove;
    r->q = (ntaps + up - 1) / up;
    r->taps = nullptr; r->hist[0] = r->hist[1] = nullptr;
    r->hpar = 0; r->nh = 0; r->n_in = 0; r->j_next = 0;
    hipError_t e = hipMalloc((void**)&r->taps, sizeof(double) * ntaps);
    if (e == hipSuccess) e = hipMemcpy(r->taps, taps_host, sizeof(double) * ntaps, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&r->hist[0], sizeof(double) * (r->q > 0 ? r->q : 1));
    if (e == hipSuccess) e = hipMalloc((void**)&r->hist[1], sizeof(double) * (r->q > 0 ? r->q : 1));
    if (e != hipSuccess) {
        (void)hipFree(r->taps); (void)hipFree(r->hist[0]); (void)hipFree(r->hist[1]);
        delete r;
        dd_set_error("dd_rpoly_create: %s", hipGetErrorString(e));
        return e == hipErrorNoDevice ? DD_ERR_NODEVICE : DD_ERR_HIP;
    }
    *h = r;
    return DD_OK;
}
extern "C" int dd_rpoly_destroy(dd_rpoly* r) {
    if (r) { (void)hipFree(r->taps); (void)hipFree(r->hist[0]); (void)hipFree(r->hist[1]); delete r; }
    return DD_OK;
}
extern "C" int dd_rpoly_reset(dd_rpoly* r) {
    DD_REQUIRE(r, "h");
    r->nh = 0; r->n_in = 0; r->j_next = 0;
    return DD_OK;
}
// outputs the next dd_rpoly_process(n, flush) call will write
extern "C" int64_t dd_rpoly_out_count(const dd_rpoly* r, int64_t n, int flush) {
    if (!r || n < 0) return DD_ERR_INVALID;
    const int64_t tot = r->n_in + n;
    int64_t j_last;
    if (flush) j_last = (tot * r->up + r->down - 1) / r->down - 1;         // ceil(tot up / down) outputs in all
    else j_last = tot > 0 ? (tot * r->up - 1) / r->down - r->npr : -1;      // every input the output needs has arrived
    const int64_t c = j_last - r->j_next + 1;
    return c > 0 ? c : 0;
}
extern "C" int dd_rpoly_process(dd_rpoly* r, const double* in, int64_t n, int flush, double* out, int64_t* n_out, void* stream) {
    DD_REQUIRE(r && n >= 0, "h/n");
    DD_REQUIRE(in || n == 0, "in");
    hipStream_t s = dd_stream(stream);
    const int64_t cnt = dd_rpoly_out_count(r, n, flush);
    if (n_out) *n_out = cnt;
    if (cnt > 0) {
        DD_REQUIRE(out, "out");
        hipLaunchKernelGGL(k_rpoly, dim3(grid1(cnt)), dim3(256), 0, s, in, n, r->n_in, r->hist[r->hpar], r->nh, r->taps, r->ntaps,
                           r->up, r->down, r->npr, r->j_next, cnt, out);
        DD_LAUNCH_CHECK();
        r->j_next += cnt;
    }
    if (n > 0) {
        const int64_t have = (int64_t)r->nh + n;
        const int nh_new = (int)(have < r->q ? have : r->q);
        hipLaunchKernelGGL(k_rpoly_hist, dim3((nh_new + 255) / 256), dim3(256), 0, s, in, n, r->hist[r->hpar], r->nh, r->hist[r->hpar ^ 1], nh_new);
        DD_LAUNCH_CHECK();
        r->hpar ^= 1;
        r->nh = nh_new;
        r->n_in += n;
    }
    return DD_OK;
}

// The FFT resampler's intermediates come from the per-stream scratch (DDScratchLock): the call neither allocates, frees nor synchronises (in the
// C3 chunk loop -- one call per 2^22-sample chunk -- those were 88 of the 140 us a chunk cost the host).
// ---- R2 when the chunk length has a large prime factor (C3: 83 886 = 2.3.11.31.41, 83 887 = 149.563): the library's length-n
// transform is then Bluestein's chirp-z at >= 2n - 1 points (175 616 for these) in some 26 launches.  Downsampling needs only
// the K = num/2 + 1 lowest bins, and a chirp-z for K bins needs a cyclic convolution of only n + K - 1 points:
//   X[k] = w[k] . sum_m (x[m] w[m]) conj(w[k - m]),   w[m] = exp(-i pi m^2 / n)   (m^2 reduced mod 2n in integers: exact phase)
// = pre-multiply | forward transform of length L (7-smooth, >= n + K - 1) | times the chirp's spectrum | inverse | post-multiply,
// and the chirp tables depend on (n, K) only, so chunks of DIFFERENT lengths share one batch (the chunk loop of config 3
// alternates 83 886 / 83 887): five launches + two library transforms for the whole chunk list.
struct DDCztKey {
    int dev;
    hipStream_t s;
    int64_t n, K, L;
    bool operator<(const DDCztKey& o) const {
        if (dev != o.dev) return dev < o.dev;
        if (s != o.s) return s < o.s;
        if (n != o.n) return n < o.n;
        if (K != o.K) return K < o.K;
        return L < o.L;
    }
};
struct DDCztTab { double2* w; double2* bspec; double2* bspec_p; };     // bspec_p: bspec / L in the row-pass order of dd_hconv_kernels.h (L = 2^17, 2^18), else null
static std::mutex g_czt_mu;
static std::map<DDCztKey, DDCztTab> g_czt;
static void czt_forget_stream(int dev, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_czt_mu);
    for (auto it = g_czt.begin(); it != g_czt.end();) {
        if (it->first.dev == dev && it->first.s == s) {
            (void)hipFree(it->second.w);
            (void)hipFree(it->second.bspec);
            if (it->second.bspec_p) (void)hipFree(it->second.bspec_p);
            it = g_czt.erase(it);
        } else {
            ++it;
        }
    }
}

__global__ void __launch_bounds__(256) k_czt_tables(double2* __restrict__ w, double2* __restrict__ bt, int64_t n, int64_t K, int64_t L) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int64_t r = (i * i) % (2 * n);
        double sn, cs;
        sincospi((double)r / (double)n, &sn, &cs);
        w[i] = make_double2(cs, -sn);
    }
    if (i < L) {
        // conj(w[m]) at m = i (0 <= m < K) and at m = i - L (-(n-1) <= m < 0); zero in between (L >= n + K - 1)
        const int64_t m = i < K ? i : (i > L - n ? L - i : -1);
        double2 v = make_double2(0.0, 0.0);
        if (m >= 0) {
            const int64_t r = (m * m) % (2 * n);
            double sn, cs;
            sincospi((double)r / (double)n, &sn, &cs);
            v = make_double2(cs, sn);
        }
        bt[i] = v;
    }
}

static int64_t largest_prime_factor(int64_t n) {
    int64_t best = 1;
    for (int64_t p = 2; p * p <= n; ++p)
        while (n % p == 0) { best = p; n /= p; }
    return n > 1 ? n : best;
}
// the chirp-z route pays when the library would run Bluestein itself (radices up to 17 are native) and few bins are kept
static bool czt_wanted(int64_t n, int64_t num) {
    static const char* env = DD_TUNE_ENV("DD_RESAMPLE_CZT");    // tools / tests: 0 = never, 1 = whenever downsampling
    if (env && atoi(env) == 0) return false;
    if (!(num < n && n >= 256)) return false;
    if (env && atoi(env) == 1) return true;
    return largest_prime_factor(n) > 17 && 4 * (num / 2 + 1) <= n;
}

static int czt_tables(int64_t n, int64_t K, int64_t L, hipStream_t s, DDCztTab* out) {
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_czt_mu);
    const DDCztKey key{dev, s, n, K, L};
    auto it = g_czt.find(key);
    if (it == g_czt.end()) {
        // a chunk loop has one or two lengths.  Tables are never freed (another thread may be using them): past 64 of them the
        // caller takes the library's own transform instead
        if (g_czt.size() >= 64) return 1;
        DDCztTab t{nullptr, nullptr, nullptr};
        DD_HIP_CHECK(hipMalloc((void**)&t.w, sizeof(double2) * (size_t)n));
        hipError_t e = hipMalloc((void**)&t.bspec, sizeof(double2) * (size_t)L);
        if (e != hipSuccess) { (void)hipFree(t.w); DD_HIP_CHECK(e); }
        hipfftHandle pl;
        int rc = get_plan(&pl, HIPFFT_Z2Z, L, 1, s);
        if (rc != DD_OK) { (void)hipFree(t.w); (void)hipFree(t.bspec); return rc; }
        hipLaunchKernelGGL(k_czt_tables, dim3(grid1(n > L ? n : L)), dim3(256), 0, s, t.w, t.bspec, n, K, L);
        if (hipfftExecZ2Z(pl, (hipfftDoubleComplex*)t.bspec, (hipfftDoubleComplex*)t.bspec, HIPFFT_FORWARD) != HIPFFT_SUCCESS) {
            (void)hipFree(t.w); (void)hipFree(t.bspec);
            dd_set_error("hipfft exec failed (chirp spectrum)");
            return DD_ERR_HIP;
        }
        if (hc_length_ok(L)) {
            e = hipMalloc((void**)&t.bspec_p, sizeof(double2) * (size_t)L);
            if (e != hipSuccess) { (void)hipFree(t.w); (void)hipFree(t.bspec); DD_HIP_CHECK(e); }
            if (L == ((int64_t)1 << 18)) hipLaunchKernelGGL(k_hc_perm<9>, dim3((unsigned)(L / 256)), dim3(256), 0, s, t.bspec, t.bspec_p, 0, 1.0 / (double)L);
            else hipLaunchKernelGGL(k_hc_perm<8>, dim3((unsigned)(L / 256)), dim3(256), 0, s, t.bspec, t.bspec_p, 0, 1.0 / (double)L);
        }
        it = g_czt.emplace(key, t).first;
    }
    *out = it->second;
    return DD_OK;
}

struct DDCztJob {
    int64_t in_off, out_off, n;
    const double2* w;
    const double2* bspec;                                  // (bspec_p when the convolution runs through dd_hconv_kernels.h)
    double scale;                                          // 1 / n
};
#define DD_CZT_MAXB 16
struct DDCztJobs { DDCztJob j[DD_CZT_MAXB]; };             // passed by value: no upload per call
// source, spectrum and sink of the chirp convolution as three launches of dd_hconv_kernels.h: a[m] = x[m] w[m] (m < n, zero
// beyond), times the chirp's spectrum (1 / L folded in), and of the result the K lowest elements times w[k] -- k_czt_pre, k_czt_mul
// and k_czt_bins inside the column and row passes, the library's two length-L transforms replaced
struct HcCztIO {
    const void* in;
    int in_is_f32;
    DDCztJobs jobs;
    double2* Y;                  // sink: [jobs][K]
    int64_t K, num;
};
struct HcCztSrc : HcCztIO {
    __device__ int rows(int job, int N2) const { return (int)((jobs.j[job].n + N2 - 1) / N2); }
    __device__ double2 at(int job, int64_t m) const {
        const DDCztJob& j = jobs.j[job];
        if (m >= j.n) return make_double2(0.0, 0.0);
        const double x = in_is_f32 ? (double)reinterpret_cast<const float*>(in)[j.in_off + m] : reinterpret_cast<const double*>(in)[j.in_off + m];
        const double2 w = j.w[m];
        return make_double2(x * w.x, x * w.y);
    }
};
struct HcCztDst : HcCztIO {
    __device__ int rows(int, int N2) const { return (int)((K + N2 - 1) / N2); }
    __device__ void put(int job, int64_t k, double2 c) const {
        if (k >= K) return;
        const double2 w = jobs.j[job].w[k];
        double2 v = make_double2(c.x * w.x - c.y * w.y, c.x * w.y + c.y * w.x);
        if ((num & 1) == 0 && k == num / 2) { v.x *= 2.0; v.y *= 2.0; }
        Y[(int64_t)job * K + k] = v;
    }
};
struct HcCztSpec {
    DDCztJobs jobs;
    __device__ const double2* ptr(int job) const { return jobs.j[job].bspec; }
};
template <typename T>
__global__ void __launch_bounds__(256) k_czt_pre(const T* __restrict__ in, const DDCztJobs jobs, int64_t L, double2* __restrict__ A) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= L) return;
    const DDCztJob& j = jobs.j[blockIdx.y];
    double2 v = make_double2(0.0, 0.0);
    if (m < j.n) {
        const double x = (double)in[j.in_off + m];
        const double2 w = j.w[m];
        v = make_double2(x * w.x, x * w.y);
    }
    A[(int64_t)blockIdx.y * L + m] = v;
}
__global__ void __launch_bounds__(256) k_czt_mul(double2* __restrict__ A, const DDCztJobs jobs, int64_t L) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= L) return;
    const double2 b = jobs.j[blockIdx.y].bspec[m];
    double2* p = A + (int64_t)blockIdx.y * L + m;
    const double2 a = *p;
    *p = make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// bins 0 .. num/2 of the length-n transform -> the half spectrum the length-num inverse takes (scipy.signal.resample, real
// input, downsampling: the kept Nyquist bin of an even num collects both halves; same rule as k_rs_bins_b)
__global__ void __launch_bounds__(256) k_czt_bins(const double2* __restrict__ A, const DDCztJobs jobs, int64_t L, double2* __restrict__ Y, int64_t ny_bins,
                                                  int64_t num) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= ny_bins) return;
    const DDCztJob& j = jobs.j[blockIdx.y];
    const double2 c = A[(int64_t)blockIdx.y * L + k];
    const double2 w = j.w[k];
    const double il = 1.0 / (double)L;
    double2 v = make_double2((c.x * w.x - c.y * w.y) * il, (c.x * w.y + c.y * w.x) * il);
    if ((num & 1) == 0 && k == num / 2) { v.x *= 2.0; v.y *= 2.0; }
    Y[(int64_t)blockIdx.y * ny_bins + k] = v;
}
__global__ void __launch_bounds__(256) k_czt_scatter(const double* __restrict__ src, const DDCztJobs jobs, int64_t num, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < num) out[jobs.j[blockIdx.y].out_off + i] = src[(int64_t)blockIdx.y * num + i] * jobs.j[blockIdx.y].scale;
}

// chunks idx[0..B) (all with target length num, every one wanted by czt_wanted), at most DD_CZT_MAXB per batch
static int resample_czt_batch(const void* in, int in_is_f32, const int64_t* in_off, const int64_t* n_host, double* out, const int64_t* out_off,
                              int64_t num, const std::vector<int>& idx_all, hipStream_t s) {
    const int64_t K = num / 2 + 1;
    int64_t nmax = 0;
    for (int j : idx_all) nmax = n_host[j] > nmax ? n_host[j] : nmax;
    // convolution length: the smaller of the next 2^a and 3.2^a (measured for config 3, need 86 199, ms per 16 chunks: 98 304 =
    // 3.2^15 0.103, 131 072 0.112, 114 688 = 7.2^14 0.116, 86 400 = the smallest 7-smooth multiple of 16 0.140, 90 112 = 11.2^13
    // 0.147: the library's power-of-two passes beat less data).  DD_CZT_LEN=<n> (tools) forces a length
    static const char* lenv = DD_TUNE_ENV("DD_CZT_LEN");
    const char* oenv = getenv("DD_CZT_OWN");                // tools / tests: 0 = the library's transforms at any length
    int64_t L = 1;
    while (L < nmax + K - 1) L <<= 1;
    // 2^17 / 2^18: the convolution as three launches of our own float64 transform (dd_hconv_kernels.h) instead of pre-multiply +
    // library transform + multiply + library transform + post-multiply (config 3: ten launches -> three)
    const bool own = hc_length_ok(L) && !(oenv && atoi(oenv) == 0) && !lenv;
    if (!own && L >= 4 && 3 * (L / 4) >= nmax + K - 1) L = 3 * (L / 4);
    if (lenv && atoll(lenv) >= nmax + K - 1) L = atoll(lenv);
    for (int j : idx_all) {                                  // every table first: 1 = not taken, nothing enqueued yet
        DDCztTab t;
        const int rc = czt_tables(n_host[j], K, L, s, &t);
        if (rc != DD_OK) return rc;
    }
    for (size_t at = 0; at < idx_all.size(); at += DD_CZT_MAXB) {
        const int B = (int)std::min<size_t>(DD_CZT_MAXB, idx_all.size() - at);
        DDCztJobs jobs;
        memset(&jobs, 0, sizeof(jobs));
        for (int b = 0; b < B; ++b) {
            const int j = idx_all[at + b];
            DDCztTab t;
            int rc = czt_tables(n_host[j], K, L, s, &t);
            if (rc != DD_OK) return rc;
            jobs.j[b] = DDCztJob{in_off[j], out_off[j], n_host[j], t.w, own ? t.bspec_p : t.bspec, 1.0 / (double)n_host[j]};
        }
        hipfftHandle pz = nullptr, pb;
        int rc = own ? DD_OK : get_plan(&pz, HIPFFT_Z2Z, L, B, s);
        if (rc != DD_OK) return rc;
        rc = get_plan(&pb, HIPFFT_Z2D, num, B, s);
        if (rc != DD_OK) return rc;
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_y = al(sizeof(double2) * (size_t)L * B), o_r = o_y + al(sizeof(double2) * (size_t)K * B);
        const size_t need = o_r + al(sizeof(double) * (size_t)num * B);
        DDScratchLock scr;
        rc = scr.get(need, s);
        if (rc != DD_OK) return rc;
        double2* A = reinterpret_cast<double2*>(scr.ptr);
        double2* Y = reinterpret_cast<double2*>(scr.ptr + o_y);
        double* res = reinterpret_cast<double*>(scr.ptr + o_r);
        hipfftResult r1 = HIPFFT_SUCCESS, r2 = HIPFFT_SUCCESS;
        if (own) {
            HcCztSrc src; HcCztDst dst; HcCztSpec sp;
            src.in = in; src.in_is_f32 = in_is_f32; src.jobs = jobs; src.Y = Y; src.K = K; src.num = num;
            static_cast<HcCztIO&>(dst) = static_cast<const HcCztIO&>(src);
            sp.jobs = jobs;
            const double2 *TA = nullptr, *TB = nullptr;
            const int lg = L == ((int64_t)1 << 18) ? 9 : 8;
            rc = hc_tables(lg, &TA, &TB);
            if (rc == DD_OK) rc = lg == 9 ? hc_ready<9, HcCztSrc, HcCztDst>() : hc_ready<8, HcCztSrc, HcCztDst>();
            if (rc != DD_OK) return rc;
            if (lg == 9) hc_convolve<9>(src, sp, dst, A, B, TA, TB, s);
            else hc_convolve<8>(src, sp, dst, A, B, TA, TB, s);
        } else {
            if (in_is_f32) hipLaunchKernelGGL(k_czt_pre<float>, dim3(grid1(L), B), dim3(256), 0, s, (const float*)in, jobs, L, A);
            else hipLaunchKernelGGL(k_czt_pre<double>, dim3(grid1(L), B), dim3(256), 0, s, (const double*)in, jobs, L, A);
            r1 = hipfftExecZ2Z(pz, (hipfftDoubleComplex*)A, (hipfftDoubleComplex*)A, HIPFFT_FORWARD);
            hipLaunchKernelGGL(k_czt_mul, dim3(grid1(L), B), dim3(256), 0, s, A, jobs, L);
            r2 = hipfftExecZ2Z(pz, (hipfftDoubleComplex*)A, (hipfftDoubleComplex*)A, HIPFFT_BACKWARD);
            hipLaunchKernelGGL(k_czt_bins, dim3(grid1(K), B), dim3(256), 0, s, A, jobs, L, Y, K, num);
        }
        const hipfftResult r3 = hipfftExecZ2D(pb, (hipfftDoubleComplex*)Y, res);
        hipLaunchKernelGGL(k_czt_scatter, dim3(grid1(num), B), dim3(256), 0, s, res, jobs, num, out);
        if (r1 != HIPFFT_SUCCESS || r2 != HIPFFT_SUCCESS || r3 != HIPFFT_SUCCESS) {
            dd_set_error("hipfft exec failed (%d, %d, %d)", (int)r1, (int)r2, (int)r3);
            return DD_ERR_HIP;
        }
        DD_LAUNCH_CHECK();
    }
    return DD_OK;
}

extern "C" int dd_resample_fft_f64(const double* in, double* out, int64_t n, int64_t num, void* stream) {
    DD_REQUIRE(n >= 1 && num >= 1, "n/num");
    DD_REQUIRE(in && out, "null buffer");
    hipStream_t s = dd_stream(stream);
    if (czt_wanted(n, num)) {
        const int64_t zero = 0;
        const int rc = resample_czt_batch(in, 0, &zero, &n, out, &zero, num, std::vector<int>{0}, s);
        if (rc != 1) return rc;
    }
    hipfftHandle pf, pb;
    int rc = get_plan(&pf, HIPFFT_D2Z, n, 1, s);
    if (rc != DD_OK) return rc;
    rc = get_plan(&pb, HIPFFT_Z2D, num, 1, s);
    if (rc != DD_OK) return rc;
    const int64_t nxb = n / 2 + 1, nyb = num / 2 + 1;
    const size_t bx = (sizeof(double2) * nxb + 255) & ~(size_t)255, by = (sizeof(double2) * nyb + 255) & ~(size_t)255;
    const size_t need = bx + by + sizeof(double) * n;
    char* base = nullptr;
    DDScratchLock scr;                      // held until this entry point has enqueued everything
    rc = scr.get(need, s);
    base = scr.ptr;
    if (rc != DD_OK) return rc;
    double2* X = reinterpret_cast<double2*>(base);
    double2* Y = reinterpret_cast<double2*>(base + bx);
    double* tmp = reinterpret_cast<double*>(base + bx + by);               // D2Z may overwrite its input: work on a copy
    DD_HIP_CHECK(hipMemcpyAsync(tmp, in, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    hipfftResult r1 = hipfftExecD2Z(pf, tmp, (hipfftDoubleComplex*)X);
    const int64_t N = num < n ? num : n;
    hipLaunchKernelGGL(k_resample_bins, dim3(grid1(nyb)), dim3(256), 0, s, X, Y, nxb, nyb, N, num, n);
    hipfftResult r2 = hipfftExecZ2D(pb, (hipfftDoubleComplex*)Y, out);
    hipLaunchKernelGGL(k_scale_f64, dim3(grid1(num)), dim3(256), 0, s, out, num, 1.0 / (double)n);
    if (r1 != HIPFFT_SUCCESS || r2 != HIPFFT_SUCCESS) {
        dd_set_error("hipfft exec failed (%d, %d)", (int)r1, (int)r2);
        return DD_ERR_HIP;
    }
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// ---- R2 over a chunk list: the per-chunk FFT resample of a chunk loop (decode_fm.py:54-70: every 2^22-sample chunk ends
// in bwLim(strict) = scipy.signal.resample of ITS outputs) for all chunks at once.  Chunks of equal (length, target
// length) share a batched plan: gather (f32 or f64 -> f64) | batched D2Z | spectrum bins | batched Z2D | scale + scatter.
struct DDRsJob { int64_t in_off, out_off; };
template <typename T>
__global__ void __launch_bounds__(256) k_rs_gather(const T* __restrict__ in, const DDRsJob* __restrict__ jobs, int64_t n, double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[(int64_t)blockIdx.y * n + i] = (double)in[jobs[blockIdx.y].in_off + i];
}
__global__ void __launch_bounds__(256) k_rs_bins_b(const double2* __restrict__ X, double2* __restrict__ Y, int64_t nx_bins, int64_t ny_bins, int64_t N, int64_t num, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= ny_bins) return;
    X += (int64_t)blockIdx.y * nx_bins;
    Y += (int64_t)blockIdx.y * ny_bins;
    // scipy.signal.resample for real input (rfft route): keep the first N/2+1 bins; the Nyquist bin of the SHORTER length is
    // halved when downsampling / doubled... same rule as k_resample_bins
    double2 v = make_double2(0.0, 0.0);
    const int64_t nyq = N / 2;
    if (k < nx_bins && k <= nyq) {
        v = X[k];
        if (N % 2 == 0 && k == nyq) {
            if (num < n) { v.x *= 2.0; v.y *= 2.0; }      // downsampling: the kept Nyquist bin collects both halves
            else if (num > n) { v.x *= 0.5; v.y *= 0.5; }
        }
    }
    Y[k] = v;
}
__global__ void __launch_bounds__(256) k_rs_scatter(const double* __restrict__ src, const DDRsJob* __restrict__ jobs, int64_t num, double scale, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < num) out[jobs[blockIdx.y].out_off + i] = src[(int64_t)blockIdx.y * num + i] * scale;
}

extern "C" int dd_resample_fft_chunks(const void* in, int in_is_f32, const int64_t* in_off_host, const int64_t* n_host, double* out,
                                      const int64_t* out_off_host, const int64_t* num_host, int count, void* stream) {
    DD_REQUIRE(in && out && in_off_host && n_host && out_off_host && num_host && count >= 0, "arguments");
    hipStream_t s = dd_stream(stream);
    std::vector<char> done(count, 0);
    for (int j = 0; j < count; ++j) DD_REQUIRE(n_host[j] >= 1 && num_host[j] >= 1, "n/num");
    // chunks whose length the library would transform by Bluestein: one chirp-z batch per target length, whatever the lengths
    for (int first = 0; first < count; ++first) {
        if (done[first] || !czt_wanted(n_host[first], num_host[first])) continue;
        std::vector<int> idx;
        for (int j = first; j < count; ++j)
            if (!done[j] && num_host[j] == num_host[first] && czt_wanted(n_host[j], num_host[j])) idx.push_back(j);
        const int rc = resample_czt_batch(in, in_is_f32, in_off_host, n_host, out, out_off_host, num_host[first], idx, s);
        if (rc == 1) break;                                  // table cache full: the groups below take everything that is left
        if (rc != DD_OK) return rc;
        for (int j : idx) done[j] = 1;
    }
    for (int first = 0; first < count; ++first) {
        if (done[first]) continue;
        const int64_t n = n_host[first], num = num_host[first];
        std::vector<DDRsJob> jobs;
        for (int j = first; j < count; ++j)
            if (!done[j] && n_host[j] == n && num_host[j] == num) { jobs.push_back({in_off_host[j], out_off_host[j]}); done[j] = 1; }
        const int B = (int)jobs.size();
        hipfftHandle pf, pb;
        int rc = get_plan(&pf, HIPFFT_D2Z, n, B, s);
        if (rc != DD_OK) return rc;
        rc = get_plan(&pb, HIPFFT_Z2D, num, B, s);
        if (rc != DD_OK) return rc;
        const int64_t nxb = n / 2 + 1, nyb = num / 2 + 1;
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_x = al(sizeof(DDRsJob) * B), o_y = o_x + al(sizeof(double2) * nxb * B), o_t = o_y + al(sizeof(double2) * nyb * B);
        const size_t o_r = o_t + al(sizeof(double) * n * B), need = o_r + al(sizeof(double) * num * B);
        DDScratchLock scr;
        rc = scr.get(need, s);
        if (rc != DD_OK) return rc;
        DDRsJob* dj = reinterpret_cast<DDRsJob*>(scr.ptr);
        double2* X = reinterpret_cast<double2*>(scr.ptr + o_x);
        double2* Y = reinterpret_cast<double2*>(scr.ptr + o_y);
        double* tmp = reinterpret_cast<double*>(scr.ptr + o_t);
        double* res = reinterpret_cast<double*>(scr.ptr + o_r);
        DD_HIP_CHECK(hipMemcpyAsync(dj, jobs.data(), sizeof(DDRsJob) * B, hipMemcpyHostToDevice, s));     // (pageable source: staged before the call returns)
        if (in_is_f32) hipLaunchKernelGGL(k_rs_gather<float>, dim3(grid1(n), B), dim3(256), 0, s, (const float*)in, dj, n, tmp);
        else hipLaunchKernelGGL(k_rs_gather<double>, dim3(grid1(n), B), dim3(256), 0, s, (const double*)in, dj, n, tmp);
        hipfftResult r1 = hipfftExecD2Z(pf, tmp, (hipfftDoubleComplex*)X);
        const int64_t N = num < n ? num : n;
        hipLaunchKernelGGL(k_rs_bins_b, dim3(grid1(nyb), B), dim3(256), 0, s, X, Y, nxb, nyb, N, num, n);
        hipfftResult r2 = hipfftExecZ2D(pb, (hipfftDoubleComplex*)Y, res);
        hipLaunchKernelGGL(k_rs_scatter, dim3(grid1(num), B), dim3(256), 0, s, res, dj, num, 1.0 / (double)n, out);
        if (r1 != HIPFFT_SUCCESS || r2 != HIPFFT_SUCCESS) {
            dd_set_error("hipfft exec failed (%d, %d)", (int)r1, (int)r2);
            return DD_ERR_HIP;
        }
        DD_LAUNCH_CHECK();
    }
    return DD_OK;
}

// grow-only scratch per device for the audio-rate entry points' intermediates (no allocation in the steady state:
// a hipMalloc/hipFree pair costs 50-100 us, a dozen of them were half of a correlate + peak-pick call)
static void* g_sync_scratch[64] = {nullptr};
static size_t g_sync_scratch_bytes[64] = {0};

static int sync_scratch(size_t bytes, char** out) {
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    DD_REQUIRE(dev >= 0 && dev < 64, "device index");
    if (g_sync_scratch_bytes[dev] < bytes) {
        if (g_sync_scratch[dev]) DD_HIP_CHECK(hipFree(g_sync_scratch[dev]));
        g_sync_scratch[dev] = nullptr;
        g_sync_scratch_bytes[dev] = 0;
        DD_HIP_CHECK(hipMalloc(&g_sync_scratch[dev], bytes));
        g_sync_scratch_bytes[dev] = bytes;
    }
    *out = (char*)g_sync_scratch[dev];
    return DD_OK;
}

// pinned host staging per device for the entry points' one copy back (grow-only; callers hold g_sync_mu)
static void* g_pin[64] = {nullptr};
static size_t g_pin_bytes[64] = {0};
static int sync_pinned(size_t bytes, char** out) {
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    DD_REQUIRE(dev >= 0 && dev < 64, "device index");
    if (g_pin_bytes[dev] < bytes) {
        if (g_pin[dev]) DD_HIP_CHECK(hipHostFree(g_pin[dev]));
        g_pin[dev] = nullptr;
        g_pin_bytes[dev] = 0;
        DD_HIP_CHECK(hipHostMalloc(&g_pin[dev], bytes, hipHostMallocDefault));
        g_pin_bytes[dev] = bytes;
    }
    *out = (char*)g_pin[dev];
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
#define DD_CS_MAXNEEDLES 2               // needles (sync words) one call correlates
struct DDRuns2 { DDRuns r[DD_CS_MAXNEEDLES]; double vv[DD_CS_MAXNEEDLES]; };

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

// (defined with the batched accurate-sync chain further down: prefix sums of h and h^2 over tiles of 2048 samples)
#define DD_SCAN_TILE 2048
__global__ void k_scan_part(const double* __restrict__ h, int64_t n, int tiles, double2* __restrict__ part);
__global__ void k_scan_final(const double* __restrict__ h, int64_t n, int tiles, const double2* __restrict__ part, double* __restrict__ P, double* __restrict__ Q);

static int xcorr_runs(const double* h, int64_t n, const double* needle_host, int m, double vv, const DDRuns& R, double* out, hipStream_t s) {
    // P[i] = sum h[0..i), Q[i] = sum h^2[0..i): the two-launch tile scan of the batched chain, batch of one
    const int tiles = (int)((n + DD_SCAN_TILE - 1) / DD_SCAN_TILE);
    std::lock_guard<std::mutex> lk(g_sync_mu);
    char* base = nullptr;
    const size_t pq_bytes = (sizeof(double) * (2 * (n + 1)) + 255) & ~(size_t)255;
    int rc = sync_scratch(pq_bytes + sizeof(double2) * (size_t)tiles, &base);
    if (rc != DD_OK) return rc;
    double* P = (double*)base;
    double* Q = P + (n + 1);
    double2* part = (double2*)(base + pq_bytes);
    hipLaunchKernelGGL(k_scan_part, dim3(tiles, 1), dim3(256), 0, s, h, n, tiles, part);
    hipLaunchKernelGGL(k_scan_final, dim3(tiles, 1), dim3(256), 0, s, h, n, tiles, part, P, Q);
    hipLaunchKernelGGL(k_xcorr_runs, dim3(grid1(n)), dim3(256), 0, s, P, Q, n, m, R, vv, out);
    hipError_t le = hipGetLastError();
    hipError_t se = hipStreamSynchronize(s);
    (void)needle_host;
    DD_HIP_CHECK(le); DD_HIP_CHECK(se);
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
    DDScratchLock scr;                      // held until this entry point has enqueued everything
    int rcs = scr.get(sizeof(double) * (size_t)m, s);
    char* base = scr.ptr;
    if (rcs != DD_OK) return rcs;
    double* v = reinterpret_cast<double*>(base);
    DD_HIP_CHECK(hipMemcpyAsync(v, needle_host, sizeof(double) * m, hipMemcpyHostToDevice, s));
    double vv = 0.0;
    for (int t = 0; t < m; ++t) vv += needle_host[t] * needle_host[t];
    hipLaunchKernelGGL(k_xcorr_norm, dim3(grid1(n)), dim3(256), 0, s, h, n, v, m, vv, out);
    hipError_t le = hipGetLastError();
    hipError_t e = hipStreamSynchronize(s);                 // the needle is the caller's host memory
    DD_HIP_CHECK(le);
    DD_HIP_CHECK(e);
    return DD_OK;
}

// ---------------------------------------------------------------- X2: peak pick (decode_noaa.py:713-751)
// The reference takes the means of the K largest and K smallest correlation values with np.argpartition (:717-723; K is
// two per second of signal) and then every index whose value exceeds a threshold between them (:726).  No sort of the
// whole array is needed for that: a radix SELECT finds the K-th largest and K-th smallest value exactly -- eight
// passes over the data, one byte of the order-preserving 64-bit key per pass, histograms in LDS (16 interleaved copies,
// so that the many samples of one bin do not serialise on one address), a one-workgroup kernel between passes that
// picks the bin and narrows the prefix -- and the values beyond them (fewer than K each) are appended to a small
// buffer; the host sorts those 2K values and sums them in ascending order.  Candidates: per-tile counts, a scan of the
// counts, a second pass that writes the indices in ascending order.
__device__ __forceinline__ unsigned long long dd_key_f64(double x) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);          // ascending in x, total order (-0 < +0, NaNs at the ends)
}
struct DDSelState {
    unsigned long long prefix[2];      // selected high bytes so far: [0] K-th largest, [1] K-th smallest
    unsigned int remaining[2];         // rank still to find inside the prefix
    unsigned int beyond[2];            // values strictly beyond the final key (above / below)
    unsigned int hist[2][256];
    unsigned int n_out[2];             // appended values
};
__global__ void __launch_bounds__(256) k_sel_hist(const double* __restrict__ x, int64_t n, int pass, DDSelState* __restrict__ st) {
    __shared__ unsigned int h[2][16][256];
    for (int i = threadIdx.x; i < 2 * 16 * 256; i += 256) (&h[0][0][0])[i] = 0;
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const unsigned long long p0 = st->prefix[0], p1 = st->prefix[1];
    const int copy = threadIdx.x & 15;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned long long k = dd_key_f64(x[i]);
        const unsigned long long hi = pass ? (k >> (shift + 8)) : 0;
        const unsigned int d = (unsigned int)(k >> shift) & 255u;
        if (hi == p0) atomicAdd(&h[0][copy][d], 1u);
        if (hi == p1) atomicAdd(&h[1][copy][d], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        unsigned int c = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) c += h[i >> 8][k][i & 255];
        if (c) atomicAdd(&st->hist[i >> 8][i & 255], c);
    }
}
// one wave: pick the byte of this pass for both selections, clear the histograms
__global__ void __launch_bounds__(64) k_sel_pick(DDSelState* __restrict__ st) {
    if (threadIdx.x == 0) {
        unsigned int r = st->remaining[0], c = 0;
        int d = 255;
        for (; d > 0; --d) { if (c + st->hist[0][d] >= r) break; c += st->hist[0][d]; }      // from the top
        st->prefix[0] = (st->prefix[0] << 8) | (unsigned long long)d;
        st->remaining[0] = r - c;
        st->beyond[0] += c;
    }
    if (threadIdx.x == 1) {
        unsigned int r = st->remaining[1], c = 0;
        int d = 0;
        for (; d < 255; ++d) { if (c + st->hist[1][d] >= r) break; c += st->hist[1][d]; }      // from the bottom
        st->prefix[1] = (st->prefix[1] << 8) | (unsigned long long)d;
        st->remaining[1] = r - c;
        st->beyond[1] += c;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) st->hist[i >> 8][i & 255] = 0;
}
// values strictly above the K-th largest / strictly below the K-th smallest (fewer than K each), any order
__global__ void __launch_bounds__(256) k_sel_collect(const double* __restrict__ x, int64_t n, DDSelState* __restrict__ st, double* __restrict__ above,
                                                     double* __restrict__ below, unsigned int cap) {
    const unsigned long long khi = st->prefix[0], klo = st->prefix[1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = x[i];
        const unsigned long long k = dd_key_f64(v);
        if (k > khi) { const unsigned int o = atomicAdd(&st->n_out[0], 1u); if (o < cap) above[o] = v; }
        if (k < klo) { const unsigned int o = atomicAdd(&st->n_out[1], 1u); if (o < cap) below[o] = v; }
    }
}
// candidates cor > thr: per tile of 2048 values the count ...
#define DD_CAND_TILE 2048
__global__ void __launch_bounds__(256) k_cand_count(const double* __restrict__ cor, int64_t n, double thr, unsigned int* __restrict__ cnt) {
    __shared__ unsigned int sw[4];
    const int64_t i0 = (int64_t)blockIdx.x * DD_CAND_TILE + 8 * threadIdx.x;
    unsigned int c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) c += (i0 + j < n && cor[i0 + j] > thr) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}
// ... exclusive scan of the tile counts (one workgroup; cnt[tiles] receives the total) ...
__global__ void __launch_bounds__(256) k_cand_scan(unsigned int* __restrict__ cnt, int tiles) {
    __shared__ unsigned int carry, sw[4];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b = 0; b < tiles; b += 256) {
        const int i = b + threadIdx.x;
        const unsigned int v = i < tiles ? cnt[i] : 0u;
        unsigned int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned int u = __shfl_up(incl, d); if ((int)(threadIdx.x & 63) >= d) incl += u; }
        if ((threadIdx.x & 63) == 63) sw[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned int off = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += sw[w];
        if (i < tiles) cnt[i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) cnt[tiles] = carry;
}
// ... and the indices (with their heights), ascending
__global__ void __launch_bounds__(256) k_cand_write(const double* __restrict__ cor, int64_t n, double thr, const unsigned int* __restrict__ off,
                                                    int64_t* __restrict__ idx, double* __restrict__ val) {
    __shared__ unsigned int sw[4];
    const int64_t i0 = (int64_t)blockIdx.x * DD_CAND_TILE + 8 * threadIdx.x;
    unsigned int c = 0;
    bool f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { f[j] = i0 + j < n && cor[i0 + j] > thr; c += f[j] ? 1u : 0u; }
    unsigned int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned int u = __shfl_up(incl, d); if ((int)(threadIdx.x & 63) >= d) incl += u; }
    if ((threadIdx.x & 63) == 63) sw[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned int o = off[blockIdx.x] + incl - c;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) o += sw[w];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (f[j]) { idx[o] = i0 + j; val[o] = cor[i0 + j]; ++o; }
}

extern "C" int dd_find_peaks_f64(const double* cor, int64_t n, double samp_rate, int needle_len,
                                 int64_t* peaks_host, int max_peaks, int* n_peaks, void* stream) {
    DD_REQUIRE(cor && n >= 1 && samp_rate > 0 && peaks_host && n_peaks && max_peaks >= 1, "arguments");
    hipStream_t s = dd_stream(stream);
    const int K = (int)(2 * ((double)n / samp_rate)) + 2;                 // expectedPeaks (:714)
    DD_REQUIRE(K <= n, "signal shorter than the expected peak count");
    // ---- all intermediates from the scratch arena: [select state | above K | below K | tile counts | cand idx n | cand val n]
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int tiles = (int)((n + DD_CAND_TILE - 1) / DD_CAND_TILE);
    const size_t o_above = al(sizeof(DDSelState)), o_below = o_above + al(sizeof(double) * K), o_cnt = o_below + al(sizeof(double) * K);
    const size_t o_cand = o_cnt + al(sizeof(unsigned int) * (tiles + 1)), o_cv = o_cand + al(sizeof(int64_t) * n);
    std::lock_guard<std::mutex> lk(g_sync_mu);
    char* base = nullptr;
    int rc = sync_scratch(o_cv + al(sizeof(double) * n), &base);
    if (rc != DD_OK) return rc;
    DDSelState* st = (DDSelState*)base;
    double* d_above = (double*)(base + o_above);
    double* d_below = (double*)(base + o_below);
    unsigned int* d_cnt = (unsigned int*)(base + o_cnt);
    int64_t* cand = (int64_t*)(base + o_cand);
    double* d_cv = (double*)(base + o_cv);
    // ---- mean of the K largest and K smallest values (argpartition, :717-723): radix select
    DDSelState h0;
    memset(&h0, 0, sizeof(h0));
    h0.remaining[0] = h0.remaining[1] = (unsigned int)K;
    DD_HIP_CHECK(hipMemcpyAsync(st, &h0, sizeof(h0), hipMemcpyHostToDevice, s));
    const unsigned int sel_grid = (unsigned int)(grid1(n) < 1024 ? grid1(n) : 1024);
    for (int pass = 0; pass < 8; ++pass) {
        hipLaunchKernelGGL(k_sel_hist, dim3(sel_grid), dim3(256), 0, s, cor, n, pass, st);
        hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(64), 0, s, st);
    }
    hipLaunchKernelGGL(k_sel_collect, dim3(sel_grid), dim3(256), 0, s, cor, n, st, d_above, d_below, (unsigned int)K);
    DD_LAUNCH_CHECK();
    DDSelState h1;
    std::vector<double> hi(K), lo(K);
    DD_HIP_CHECK(hipMemcpyAsync(&h1, st, sizeof(h1), hipMemcpyDeviceToHost, s));
    DD_HIP_CHECK(hipMemcpyAsync(hi.data(), d_above, sizeof(double) * K, hipMemcpyDeviceToHost, s));
    DD_HIP_CHECK(hipMemcpyAsync(lo.data(), d_below, sizeof(double) * K, hipMemcpyDeviceToHost, s));
    DD_HIP_CHECK(hipStreamSynchronize(s));
    auto unkey = [](unsigned long long k) {
        const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        double d;
        memcpy(&d, &u, sizeof(d));
        return d;
    };
    DD_REQUIRE(h1.n_out[0] == h1.beyond[0] && h1.n_out[1] == h1.beyond[1] && h1.n_out[0] < (unsigned int)K && h1.n_out[1] < (unsigned int)K,
               "dd_find_peaks_f64: selection bookkeeping (internal)");
    {
        const double vhi = unkey(h1.prefix[0]), vlo = unkey(h1.prefix[1]);
        for (unsigned int i = h1.n_out[0]; i < (unsigned int)K; ++i) hi[i] = vhi;     // the K-th largest itself and its ties
        for (unsigned int i = h1.n_out[1]; i < (unsigned int)K; ++i) lo[i] = vlo;
        std::sort(hi.begin(), hi.end());
        std::sort(lo.begin(), lo.end());
    }
    double sum_hi = 0.0, sum_lo = 0.0;
    for (int i = 0; i < K; ++i) { sum_hi += hi[i]; sum_lo += lo[i]; }     // ascending, like the sums over the sorted array they replace
    double avgpk = sum_hi / K;
    avgpk -= 0.25 * (avgpk - sum_lo / K);                                 // NOAA_PEAKHEIGHTWIGGLE (:723)
    // ---- candidates cor > threshold, ascending index (:726), with their heights
    hipLaunchKernelGGL(k_cand_count, dim3(tiles), dim3(256), 0, s, cor, n, avgpk, d_cnt);
    hipLaunchKernelGGL(k_cand_scan, dim3(1), dim3(256), 0, s, d_cnt, tiles);
    hipLaunchKernelGGL(k_cand_write, dim3(tiles), dim3(256), 0, s, cor, n, avgpk, d_cnt, cand, d_cv);
    DD_LAUNCH_CHECK();
    unsigned int ucount = 0;
    DD_HIP_CHECK(hipMemcpyAsync(&ucount, d_cnt + tiles, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    DD_HIP_CHECK(hipStreamSynchronize(s));
    const int count = (int)ucount;
    std::vector<int64_t> ci(count);
    std::vector<double> cv(count);
    if (count > 0) {
        DD_HIP_CHECK(hipMemcpyAsync(ci.data(), cand, sizeof(int64_t) * count, hipMemcpyDeviceToHost, s));
        DD_HIP_CHECK(hipMemcpyAsync(cv.data(), d_cv, sizeof(double) * count, hipMemcpyDeviceToHost, s));
        DD_HIP_CHECK(hipStreamSynchronize(s));
    }
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

// ---------------------------------------------------------------- 8f-2: accurate-sync windows, batched
// getAccurateSync (decode_noaa.py:808-880) cuts one +-width window of IQ samples around every crude sync
// and runs, per window:  offsetFreq -> filter(blackmanHarris(151, zeroPhase)) -> demod_fm -> demod_am
// (:852) and then __correlateAndFindPeaks with the zero-phase hamming(492) pre-filter (:677-767, :853).
// The windows are independent and equally long, so the whole chain runs once over [windows][samples]
// arrays: a dozen launches per batch instead of ~40 launches, ~25 allocations and 8 host round trips per
// window.  Each stage is the arithmetic of the per-window entry points (same kernels or the same
// device functions); only the prefix sums and the batched FFT plan may
// round differently, at the 1e-13 level of the correlation.
#include "dd_chain_kernels.h"
#include "dd_filtfilt_kernels.h"

template <bool U8>
__global__ void __launch_bounds__(256) k_sync_front(const void* __restrict__ iq, const int64_t* __restrict__ starts, int64_t L,
                                                    uint64_t cyc, const float2* __restrict__ tbl, float2* __restrict__ X) {
    // four samples per lane: four loads in flight, 32 contiguous bytes stored
    const int64_t i0 = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x);
    if (i0 >= L) return;
    const int64_t g = starts[blockIdx.y] + i0;
    float2 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t ge = i0 + e < L ? g + e : g;
        if (U8) {
            const uchar2 u = reinterpret_cast<const uchar2*>(iq)[ge];
            v[e] = make_float2((float)u.x - 127.5f, (float)u.y - 127.5f);
        } else {
            v[e] = reinterpret_cast<const float2*>(iq)[ge];
        }
    }
    float2* out = X + (int64_t)blockIdx.y * L + i0;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = dd_cmul(v[e], dd_phasor((uint64_t)(i0 + e) * cyc, tbl));      // sample index restarts per window (Q5)
    if (i0 + 3 < L && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
        reinterpret_cast<float4*>(out)[0] = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
        reinterpret_cast<float4*>(out)[1] = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (i0 + e < L) out[e] = v[e];
    }
}

// demod_fm (stateless) straight into the FFT buffer: W[b][j] = (angle(Y[j+1] conj Y[j]), 0)
__global__ void __launch_bounds__(256) k_sync_fm(const float2* __restrict__ Y, int64_t L, double2* __restrict__ W) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= L - 1) return;
    const float2* y = Y + (int64_t)blockIdx.y * L;
    W[(int64_t)blockIdx.y * (L - 1) + j] = make_double2((double)dd_fm_angle(y[j + 1], y[j]), 0.0);
}

// P[b][i] = sum h[b][0..i), Q likewise of h^2, in two launches over tiles of 2048 samples: tile sums, then each
// tile adds the sums of the tiles before it (ascending) to its own scan -- every tile of every window in parallel
// (a lane scans 8 consecutive samples, but the tile is fetched -- and the prefix sums are written -- with lanes on consecutive
// addresses, through an LDS image skewed by one element per 8: read lane by lane, 64-byte runs at a 64-byte stride, these
// kernels moved 2 TB/s)
#define DD_SCAN_LDS (DD_SCAN_TILE + DD_SCAN_TILE / 8)
__device__ __forceinline__ void dd_scan_tile_load(const double* __restrict__ h, int64_t n, int64_t tile0, int t, double* __restrict__ lds,
                                                  double (&p)[8], double (&q)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int e = t + 256 * r;
        lds[e + (e >> 3)] = (tile0 + e < n) ? h[tile0 + e] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double v = lds[9 * t + j];
        p[j] = j ? p[j - 1] + v : v;
        q[j] = j ? q[j - 1] + v * v : v * v;
    }
}
// out[tile0 + 1 + e] = v[e] for the tile's 2048 values held 8 per lane (lane t: e = 8 t .. 8 t + 7), stored coalesced
__device__ __forceinline__ void dd_scan_tile_store(double* __restrict__ out, int64_t n, int64_t tile0, int t, double* __restrict__ lds, const double (&v)[8]) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) lds[9 * t + j] = v[j];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int e = t + 256 * r;
        if (tile0 + e < n) out[tile0 + e + 1] = lds[e + (e >> 3)];
    }
}
__global__ void __launch_bounds__(256) k_scan_part(const double* __restrict__ h, int64_t n, int tiles, double2* __restrict__ part) {
    __shared__ double sp[4], sq[4];
    __shared__ double lds[DD_SCAN_LDS];
    h += (int64_t)blockIdx.y * n;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    double p[8], q[8];
    dd_scan_tile_load(h, n, (int64_t)blockIdx.x * DD_SCAN_TILE, t, lds, p, q);
    double tp = p[7], tq = q[7];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { tp += __shfl_down(tp, d); tq += __shfl_down(tq, d); }
    if (lane == 0) { sp[wv] = tp; sq[wv] = tq; }
    __syncthreads();
    if (t == 0) part[(int64_t)blockIdx.y * tiles + blockIdx.x] = make_double2(((sp[0] + sp[1]) + sp[2]) + sp[3], ((sq[0] + sq[1]) + sq[2]) + sq[3]);
}
__global__ void __launch_bounds__(256) k_scan_final(const double* __restrict__ h, int64_t n, int tiles, const double2* __restrict__ part,
                                                    double* __restrict__ P, double* __restrict__ Q) {
    __shared__ double sp[4], sq[4];
    h += (int64_t)blockIdx.y * n;
    P += (int64_t)blockIdx.y * (n + 1);
    Q += (int64_t)blockIdx.y * (n + 1);
    part += (int64_t)blockIdx.y * tiles;
    __shared__ double lds[DD_SCAN_LDS];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t tile0 = (int64_t)blockIdx.x * DD_SCAN_TILE;
    double p[8], q[8];
    dd_scan_tile_load(h, n, tile0, t, lds, p, q);
    // sums of the tiles before this one: every lane takes the tiles t, t + 256, ..., the workgroup adds them up (one lane
    // walking all of them was 77 us of the accurate windows' 1.1 ms per batch)
    __shared__ double bp[4], bq[4];
    double cp = 0.0, cq = 0.0;
    for (int k = t; k < (int)blockIdx.x; k += 256) { const double2 v = part[k]; cp += v.x; cq += v.y; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { cp += __shfl_down(cp, d); cq += __shfl_down(cq, d); }
    if (lane == 0) { bp[wv] = cp; bq[wv] = cq; }
    double tp = p[7], tq = q[7];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double a = __shfl_up(tp, d), c = __shfl_up(tq, d);
        if (lane >= d) { tp += a; tq += c; }
    }
    if (lane == 63) { sp[wv] = tp; sq[wv] = tq; }
    double ep = __shfl_up(tp, 1), eq = __shfl_up(tq, 1);
    if (lane == 0) { ep = 0.0; eq = 0.0; }
    __syncthreads();
    cp = ((bp[0] + bp[1]) + bp[2]) + bp[3];
    cq = ((bq[0] + bq[1]) + bq[2]) + bq[3];
    for (int w = 0; w < wv; ++w) { cp += sp[w]; cq += sq[w]; }
    ep += cp;
    eq += cq;
    if (blockIdx.x == 0 && t == 0) { P[0] = 0.0; Q[0] = 0.0; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[j] += ep; q[j] += eq; }
    dd_scan_tile_store(P, n, tile0, t, lds, p);
    dd_scan_tile_store(Q, n, tile0, t, lds, q);
}

// Peak pick of one window (decode_noaa.py:713-762) when the window is shorter than the 0.45 s group
// distance: expectedPeaks K = 2, every candidate falls in one group, and the pick is the first index of
// the maximum provided it exceeds the threshold.  Also the two "extras": peak height and the mean of the
// next needle-length of the envelope.  The correlation values are reduced where they are produced (first
// maximum, two largest, two smallest per tile of 1024 outputs); the correlation array itself is never stored.
struct DDPk {
    double m1, m2, l1, l2;
    int64_t i1;
    int nan;
};
__device__ __forceinline__ DDPk dd_pk_merge(const DDPk& a, const DDPk& b) {
    DDPk r;
    if (b.m1 > a.m1 || (b.m1 == a.m1 && b.i1 < a.i1)) {
        r.m1 = b.m1; r.i1 = b.i1; r.m2 = fmax(a.m1, b.m2);
    } else {
        r.m1 = a.m1; r.i1 = a.i1; r.m2 = fmax(a.m2, b.m1);
    }
    if (b.l1 < a.l1) { r.l1 = b.l1; r.l2 = fmin(a.l1, b.l2); }
    else { r.l1 = a.l1; r.l2 = fmin(a.l2, b.l1); }
    r.nan = a.nan | b.nan;
    return r;
}
__device__ __forceinline__ DDPk dd_pk_shfl(const DDPk& a, int d) {
    DDPk r;
    r.m1 = __shfl_down(a.m1, d); r.m2 = __shfl_down(a.m2, d);
    r.l1 = __shfl_down(a.l1, d); r.l2 = __shfl_down(a.l2, d);
    r.i1 = __shfl_down(a.i1, d); r.nan = __shfl_down(a.nan, d);
    return r;
}
__device__ __forceinline__ DDPk dd_pk_empty() {
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    DDPk a = {-inf, -inf, inf, inf, INT64_MAX, 0};
    return a;
}

// Normalised correlation in the run-length form (k_xcorr_runs) of a batch of windows, reduced per tile.
// Workgroups are dealt to the XCDs window by window (dispatch is round-robin over the 8 XCDs), so the ~14
// reads of every prefix-sum element come out of one XCD's L2.
#define DD_XC_TILE 1024
// (Tried in round 4 and not kept, same call, 64 windows: the run table in scalar registers with the loop unrolled -- all 64
// look-ups of a lane in flight, 169 registers, 2 waves per SIMD -- 129 us; the look-ups staged in LDS along the comb of the
// needle's run-boundary grid (984 / 492 samples: 1.5-2.7 loads from L2 per output instead of 16, but 32 KB of LDS per wave =
// 5 waves per CU) 107-162 us; this loop, 8 waves per SIMD walking the runs in step so that neighbouring workgroups read
// neighbouring prefix sums at the same time: 80 us.  Two runs' look-ups in flight (66 registers, 7 waves): 79-82 against 81-86, noise;
// fewer workgroups per CU (so that one XCD's workgroups stay inside one window's prefix sums): 87 us at 7 per CU, 98 at 4, 146 at 2.
// profiles/r04_noaa_timeline.txt)
__global__ void __launch_bounds__(256) k_xcorr_runs_pk(const double* __restrict__ P, const double* __restrict__ Q, int64_t n, int m,
                                                       const DDRuns2 R2, const int* __restrict__ group, int tiles, int nwin, DDPk* __restrict__ part) {
    __shared__ DDPk sw[4];
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int win = (k / tiles) * 8 + xcd, tile = k % tiles;
    if (win >= nwin) return;
    const int gsel = group ? __builtin_amdgcn_readfirstlane(group[win]) : 0;        // which needle this window is searched for
    const DDRuns& R = R2.r[gsel];
    const double vv = R2.vv[gsel];
    P += (int64_t)win * (n + 1);
    Q += (int64_t)win * (n + 1);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    auto at = [&](const double* S, int64_t x) { return S[x < 0 ? 0 : (x > n ? n : x)]; };
    const double qn = 1e-13 * Q[n];
    // four outputs per lane, the run loop outermost: the four lookups of a run boundary are independent loads
    constexpr int NJ = DD_XC_TILE / 256;
    int64_t a0[NJ];
    double c[NJ], lo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        a0[j] = (int64_t)tile * DD_XC_TILE + j * 256 + t + (m - 1) / 2 - (m - 1);
        c[j] = 0.0;
        lo[j] = at(P, a0[j]);
    }
    for (int r = 0; r < R.nr; ++r) {
        double hi[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) hi[j] = at(P, a0[j] + R.start[r + 1]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) { c[j] = fma(R.val[r], hi[j] - lo[j], c[j]); lo[j] = hi[j]; }
    }
    DDPk a = dd_pk_empty();
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int64_t i = (int64_t)tile * DD_XC_TILE + j * 256 + t;
        double e = at(Q, a0[j] + m) - at(Q, a0[j]);
        double cc = c[j];
        if (!(e > qn)) { cc = 0.0; e = 0.0; }
        const double x = cc / sqrt(e * vv);
        if (i < n) {
            DDPk bq = {x, -inf, x, inf, i, (x != x) ? 1 : 0};
            a = dd_pk_merge(a, bq);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a = dd_pk_merge(a, dd_pk_shfl(a, d));
    if (lane == 0) sw[wv] = a;
    __syncthreads();
    if (t == 0) part[(int64_t)win * tiles + tile] = dd_pk_merge(dd_pk_merge(sw[0], sw[1]), dd_pk_merge(sw[2], sw[3]));
}

__global__ void __launch_bounds__(256) k_sync_peak(const DDPk* __restrict__ part, int tiles, const double* __restrict__ env, int64_t n, int m,
                                                   int64_t* __restrict__ peak, double* __restrict__ height, double* __restrict__ tsync) {
    __shared__ DDPk sw[4];
    __shared__ double ssum[4];
    const double* ev = env + (int64_t)blockIdx.x * n;
    part += (int64_t)blockIdx.x * tiles;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    DDPk a = dd_pk_empty();
    for (int k = t; k < tiles; k += 256) a = dd_pk_merge(a, part[k]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) a = dd_pk_merge(a, dd_pk_shfl(a, d));
    if (lane == 0) sw[wv] = a;
    __syncthreads();
    a = dd_pk_merge(dd_pk_merge(sw[0], sw[1]), dd_pk_merge(sw[2], sw[3]));
    double avgpk = (0.0 + a.m2 + a.m1) / 2.0;                       // mean of the K = 2 largest (:717-721)
    avgpk -= 0.25 * (avgpk - (0.0 + a.l1 + a.l2) / 2.0);            // NOAA_PEAKHEIGHTWIGGLE (:723)
    const bool found = !a.nan && a.m1 > avgpk;
    const int64_t i = a.i1 - m / 2;                                 // :749
    const bool tail = found && i + 2 * (int64_t)m < n;              // :755
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (tail) {
        int64_t j = i + m + t;
        const int64_t end = i + 2 * (int64_t)m;
        for (; j + 768 < end; j += 1024) { s0 += ev[j]; s1 += ev[j + 256]; s2 += ev[j + 512]; s3 += ev[j + 768]; }
        for (; j < end; j += 256) s0 += ev[j];
    }
    double sacc = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sacc += __shfl_down(sacc, d);
    if (lane == 0) ssum[wv] = sacc;
    __syncthreads();
    if (t == 0) {
        const double tot = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        peak[blockIdx.x] = found ? i : INT64_MIN;
        height[blockIdx.x] = found ? a.m1 : __longlong_as_double(0x7ff8000000000000ll);
        tsync[blockIdx.x] = tail ? tot / (double)m : __longlong_as_double(0x7ff8000000000000ll);
    }
}

// ---- the envelope as one real convolution.  abs(hilbert(x)) = |x + j (x (*) hh)| where (*) is the length-N
// circular convolution and hh = imag(ifft(h)) the Hilbert kernel of scipy's spectrum mask h (the real part
// of ifft(h) is the unit impulse).  The window length N = 118 151 has a large prime factor, so the library's
// length-N transforms are Bluestein chirp-z: two padded power-of-two transforms each way, complex.  The
// circular convolution needs only outputs [0, N), which a length-M >= 2N-1 cyclic convolution with the kernel
// laid out at offsets -(N-1)..N-1 gives without wrap-around: one real-to-complex and one complex-to-real
// power-of-two transform per window, a quarter of the work.  The kernel spectrum is built once per length
// from the closed form of hh.
__global__ void __launch_bounds__(256) k_sync_fm_pad(const float2* __restrict__ Y, int64_t L, double* __restrict__ XR, int64_t M) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const float2* y = Y + (int64_t)blockIdx.y * L;
    XR[(int64_t)blockIdx.y * M + j] = j < L - 1 ? (double)dd_fm_angle(y[j + 1], y[j]) : 0.0;
}
__global__ void __launch_bounds__(256) k_spec_mul(double2* __restrict__ S, const double2* __restrict__ HH, int64_t nb) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nb) return;
    double2* p = S + (int64_t)blockIdx.y * nb + k;
    const double2 a = *p, h = HH[k];
    *p = make_double2(a.x * h.x - a.y * h.y, a.x * h.y + a.y * h.x);
}
__global__ void __launch_bounds__(256) k_env_hypot(const double* __restrict__ XR, const double* __restrict__ YR, int64_t M, int64_t n,
                                                   double* __restrict__ env) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    env[(int64_t)blockIdx.y * n + i] = hypot(XR[(int64_t)blockIdx.y * M + i], YR[(int64_t)blockIdx.y * M + i]);
}

static std::map<std::pair<int, int64_t>, double2*> g_hilb;      // (device, N) -> spectrum of the padded kernel / M
static std::vector<std::pair<int, int64_t>> g_hilb_order;
// sin(pi num / den) for integers num >= 0, den > 0: the argument is reduced to [0, pi/2] exactly in integers first
static double dd_sinpi_frac(int64_t num, int64_t den) {
    int64_t r = num % (2 * den);
    double sg = 1.0;
    if (r >= den) { r -= den; sg = -1.0; }
    if (2 * r > den) r = den - r;
    return sg * sin(3.14159265358979323846 * (double)r / (double)den);
}

// In-place radix-2 transform of a power-of-two length on the HOST, float64, twiddles from one table (once per Hilbert-kernel
// spectrum: 2^18 points take a few milliseconds).  Round 5: the kernel spectra no longer go through the FFT library -- its first plan of
// a process costs hundreds of milliseconds, and the reference decodes one file per process (main.py:208-270).
static void host_fft_pow2(std::vector<std::complex<double>>& v) {
    // (plain arrays and spelt-out complex arithmetic: std::complex's operator* goes through a NaN-checking library call)
    const size_t n = v.size();
    double* a = reinterpret_cast<double*>(v.data());
    std::vector<double> wr(n / 2), wi(n / 2);
    const double step0 = -6.283185307179586476925286766559 / (double)n;
    // one octant by the library, the rest by symmetry of the unit circle (k -> n/4 - k, then k -> k + n/4)
    const size_t q = n / 4;
    for (size_t k = 0; k <= q / 2 && k < n / 2; ++k) {
        const double c = cos(step0 * (double)k), sn = sin(step0 * (double)k);
        wr[k] = c; wi[k] = sn;
        if (q >= k && q - k < n / 2) { wr[q - k] = -sn; wi[q - k] = -c; }
    }
    for (size_t k = 0; k < q && k + q < n / 2; ++k) { wr[k + q] = wi[k]; wi[k + q] = -wr[k]; }
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(a[2 * i], a[2 * j]); std::swap(a[2 * i + 1], a[2 * j + 1]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len / 2, step = n / len;
        for (size_t i = 0; i < n; i += len) {
            double* lo = a + 2 * i;
            double* hi = a + 2 * (i + half);
            for (size_t k = 0; k < half; ++k) {
                const double c = wr[k * step], sn = wi[k * step];
                const double xr = hi[2 * k] * c - hi[2 * k + 1] * sn, xi = hi[2 * k] * sn + hi[2 * k + 1] * c;
                const double ur = lo[2 * k], ui = lo[2 * k + 1];
                lo[2 * k] = ur + xr; lo[2 * k + 1] = ui + xi;
                hi[2 * k] = ur - xr; hi[2 * k + 1] = ui - xi;
            }
        }
    }
}
// the spectrum of a real kernel image of length M (divided by M): the M/2 + 1 bins the library's real transforms multiply, and -- for the
// lengths of dd_hconv_kernels.h -- once more behind them in the order its row pass reads (out[N2 k1 + k2] = bin k1 + 512 k2).  One allocation.
static int kernel_spectrum_upload(const std::vector<double>& img, int64_t M, double2** out, hipStream_t s) {
    std::vector<std::complex<double>> v((size_t)M);
    for (int64_t i = 0; i < M; ++i) v[(size_t)i] = std::complex<double>(img[(size_t)i], 0.0);
    host_fft_pow2(v);
    const int64_t nb = M / 2 + 1;
    const bool own = hc_length_ok(M);
    std::vector<double2> h((size_t)(nb + (own ? M : 0)));
    const double sc = 1.0 / (double)M;
    for (int64_t k = 0; k < nb; ++k) h[(size_t)k] = make_double2(v[(size_t)k].real() * sc, v[(size_t)k].imag() * sc);
    if (own) {
        const int lg = M == ((int64_t)1 << 18) ? 9 : 8;
        for (int64_t i = 0; i < M; ++i) {
            const int64_t k = (i >> lg) + DD_HC_N * (i & (((int64_t)1 << lg) - 1));
            h[(size_t)(nb + i)] = make_double2(v[(size_t)k].real() * sc, v[(size_t)k].imag() * sc);
        }
    }
    double2* HH = nullptr;
    DD_HIP_CHECK(hipMalloc((void**)&HH, sizeof(double2) * h.size()));
    hipError_t e = hipMemcpyAsync(HH, h.data(), sizeof(double2) * h.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);                          // (the staging vector dies with this call)
    if (e != hipSuccess) { (void)hipFree(HH); dd_set_error("Hilbert kernel spectrum: %s", hipGetErrorString(e)); return DD_ERR_HIP; }
    *out = HH;
    return DD_OK;
}
static void hilb_cache_put(std::pair<int, int64_t> key, double2* HH) {
    // (one spectrum per length: up to 8 MB each; a process that walks through recordings of many different lengths keeps the
    // eight most recently built -- the callers hold g_sync_mu and leave nothing in flight when they return (DDSyncOnExit))
    g_hilb_order.push_back(key);
    while (g_hilb_order.size() > 8) {
        auto old = g_hilb.find(g_hilb_order.front());
        if (old != g_hilb.end()) { (void)hipDeviceSynchronize(); (void)hipFree(old->second); g_hilb.erase(old); }      // (dd_am_envelope_f64 returns with its kernels in flight)
        g_hilb_order.erase(g_hilb_order.begin());
    }
    g_hilb[key] = HH;
}

// hh[n] = imag(ifft(h))[n] = (2/N) sum_{k=1..m} sin(2 pi k n / N), m = the number of doubled bins of scipy's mask
// ((N-1)/2 for odd N, N/2 - 1 for even N) = (2/N) sin(pi m n/N) sin(pi (m+1) n/N) / sin(pi n/N): a closed form, so no
// length-N (Bluestein) plan is ever built for it; accurate to a few 1e-17 (checked against a long-double sum).
static int hilbert_kernel_spectrum(int64_t n, int64_t M, const double2** out, hipStream_t s) {
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    int lgM = 0;
    while (((int64_t)1 << lgM) < M) ++lgM;
    auto key = std::make_pair(dev, (n << 6) | lgM);            // (length and cyclic length)
    auto it = g_hilb.find(key);
    if (it != g_hilb.end()) { *out = it->second; return DD_OK; }
    const int64_t m = (n & 1) ? (n - 1) / 2 : n / 2 - 1;
    std::vector<double> host((size_t)M, 0.0);                 // buf[j mod M] = hh[j mod N], j in [-(N-1), N-1]
    for (int64_t j = 1; j < n; ++j) {
        const double v = (2.0 / (double)n) * dd_sinpi_frac(m * j, n) * dd_sinpi_frac((m + 1) * j, n) / dd_sinpi_frac(j, n);
        host[(size_t)j] = v;
        host[(size_t)(M - n + j)] = v;
    }
    double2* HH = nullptr;
    const int rc = kernel_spectrum_upload(host, M, &HH, s);
    if (rc != DD_OK) return rc;
    hilb_cache_put(key, HH);
    *out = HH;
    return DD_OK;
}

// The Hilbert kernel of an EVEN length N is zero at even lags, hh[2j] = 0, hh[2j+1] = (2/N) cot(pi (2j+1) / N) =: g[j]: the length-N circular
// convolution falls apart into two of length N/2 with the same kernel,
//     H(x)[2m+1] = (g (*) x_even)[m]        H(x)[2m] = (g (*) x_odd)[m-1]        (indices mod N/2)
// and z = x_even + j x_odd carries both through ONE complex convolution.  decode_noaa.py:647-653 takes the envelope in blocks of 240 000
// samples: two length-120 000 convolutions fit the cyclic length 2^18 of dd_hconv_kernels.h (>= 2 (N/2) - 1), the block itself does not
// (it would need 2^19).  This is g's spectrum for that image -- g[j mod N/2] at lags j in [-(N/2 - 1), N/2 - 1] -- in row-pass order.
static int hilbert_split_spectrum(int64_t N, int64_t M, const double2** out_perm, hipStream_t s) {
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    int lgM = 0;
    while (((int64_t)1 << lgM) < M) ++lgM;
    auto key = std::make_pair(dev, -((N << 6) | lgM));         // (negative: the split kernel of length N, beside the full ones)
    auto it = g_hilb.find(key);
    if (it != g_hilb.end()) { *out_perm = it->second + (M / 2 + 1); return DD_OK; }
    const int64_t N2 = N / 2;
    std::vector<double> host((size_t)M, 0.0);
    auto g = [&](int64_t j) {                                  // (2/N) cot(pi (2j+1) / N), arguments reduced in integers
        const int64_t k = 2 * j + 1;
        return (2.0 / (double)N) * dd_sinpi_frac(2 * k + N, 2 * N) / dd_sinpi_frac(k, N);      // cos(pi k / N) = sin(pi (2k + N) / (2N))
    };
    for (int64_t j = 0; j < N2; ++j) {
        const double v = g(j);
        host[(size_t)j] = v;                                   // lag +j
        if (j > 0) host[(size_t)(M - N2 + j)] = v;             // lag j - N/2 (the same circular index)
    }
    double2* HH = nullptr;
    const int rc = kernel_spectrum_upload(host, M, &HH, s);
    if (rc != DD_OK) return rc;
    hilb_cache_put(key, HH);
    *out_perm = HH + (M / 2 + 1);
    return DD_OK;
}

// a block of real float64 audio as the source and its envelope as the sink of the three launches; job = block.
// Split form (even block length N): element n of the image = (x[2n], x[2n+1]), n < N/2; result element m = (H(x)[2m+1], H(x)[2(m+1)]).
struct HcBlkSplitIO {
    const double* x;
    double* env;
    int64_t N, N2;
    __device__ int rows(int, int cols) const { return (int)((N2 + cols - 1) / cols); }
    __device__ double2 at(int job, int64_t n) const {
        if (n >= N2) return make_double2(0.0, 0.0);
        const double* p = x + (int64_t)job * N + 2 * n;
        return make_double2(p[0], p[1]);
    }
    __device__ void put(int job, int64_t m, double2 y) const {
        if (m >= N2) return;
        const double* p = x + (int64_t)job * N;
        double* e = env + (int64_t)job * N;
        e[2 * m + 1] = hypot(p[2 * m + 1], y.x);
        const int64_t m1 = m + 1 == N2 ? 0 : m + 1;
        e[2 * m1] = hypot(p[2 * m1], y.y);
    }
};
// Plain form (any length n with 2 n + 2 <= M): element i = (x[i], 0); result element i = (H(x)[i], -)
struct HcBlkRealIO {
    const double* x;
    double* env;
    int64_t n;
    __device__ int rows(int, int cols) const { return (int)((n + cols - 1) / cols); }
    __device__ double2 at(int, int64_t i) const { return i < n ? make_double2(x[i], 0.0) : make_double2(0.0, 0.0); }
    __device__ void put(int, int64_t i, double2 y) const { if (i < n) env[i] = hypot(x[i], y.x); }
};
// envelope of `jobs` blocks of N samples each (x + job N) through dd_hconv_kernels.h; T: [jobs][M] c128 work buffer.  split: the
// even / odd form above (N even, N - 1 <= M); else the plain form (one block, 2 N + 2 <= M).  DD_ERR_UNSUPPORTED: the caller's other route.
static int hc_block_envelope(const double* x, double* env, int64_t N, int jobs, bool split, int64_t M, double2* T, hipStream_t s) {
    if (!hc_length_ok(M)) return DD_ERR_UNSUPPORTED;
    const int lg = M == ((int64_t)1 << 18) ? 9 : 8;
    const double2 *TA = nullptr, *TB = nullptr;
    int rc = hc_tables(lg, &TA, &TB);
    if (rc != DD_OK) return rc;
    const double2* HHp = nullptr;
    if (split) {
        rc = hilbert_split_spectrum(N, M, &HHp, s);
        if (rc != DD_OK) return rc;
        const HcBlkSplitIO io = {x, env, N, N / 2};
        const HcOneSpec sp = {HHp};
        if (lg == 9) { rc = hc_ready<9, HcBlkSplitIO, HcBlkSplitIO>(); if (rc == DD_OK) hc_convolve<9>(io, sp, io, T, jobs, TA, TB, s); }
        else { rc = hc_ready<8, HcBlkSplitIO, HcBlkSplitIO>(); if (rc == DD_OK) hc_convolve<8>(io, sp, io, T, jobs, TA, TB, s); }
    } else {
        const double2* HH = nullptr;
        rc = hilbert_kernel_spectrum(N, M, &HH, s);
        if (rc != DD_OK) return rc;
        HHp = HH + (M / 2 + 1);
        const HcBlkRealIO io = {x, env, N};
        const HcOneSpec sp = {HHp};
        if (lg == 9) { rc = hc_ready<9, HcBlkRealIO, HcBlkRealIO>(); if (rc == DD_OK) hc_convolve<9>(io, sp, io, T, 1, TA, TB, s); }
        else { rc = hc_ready<8, HcBlkRealIO, HcBlkRealIO>(); if (rc == DD_OK) hc_convolve<8>(io, sp, io, T, 1, TA, TB, s); }
    }
    return rc;
}

// cyclic length of dd_hconv_kernels.h for the envelope of a block of N real samples, 0 = not on this route; *split: the even / odd form
static int64_t hc_block_len(int64_t N, bool* split) {
    if (N < 2) return 0;
    if ((N & 1) == 0 && N - 1 <= ((int64_t)1 << 18)) { *split = true; return N - 1 <= ((int64_t)1 << 17) ? (int64_t)1 << 17 : (int64_t)1 << 18; }
    *split = false;
    if (2 * N + 2 <= ((int64_t)1 << 17)) return (int64_t)1 << 17;
    if (2 * N + 2 <= ((int64_t)1 << 18)) return (int64_t)1 << 18;
    return 0;
}
// What dd_noaa_crude_tail will need for `n` audio samples in blocks of `block` -- the Hilbert-kernel spectra of the block and of the ragged
// last block (host transforms: ~20 ms) and the transform's twiddle tables -- built ahead of time.  noaa_sync calls this from a thread of
// its own when the decoder object is created, so that it overlaps the upload of the recording and the audio chain; the result sits in the
// cache the crude tail looks in.  Harmless when the lengths turn out different (the crude tail builds what it needs).
extern "C" int dd_noaa_prepare(int64_t n, int64_t block, void* stream) {
    DD_REQUIRE(n >= 1 && block >= 2, "arguments");
    hipStream_t s = dd_stream(stream);
    int64_t nfull = 0;
    while ((nfull + 1) * block < n) ++nfull;
    const int64_t rem = n - nfull * block;
    std::lock_guard<std::mutex> lk(g_sync_mu);
    const int64_t lens[2] = {nfull > 0 ? block : 0, rem};
    for (int i = 0; i < 2; ++i) {
        bool split = false;
        const int64_t M = hc_block_len(lens[i], &split);
        if (!M) continue;
        const double2 *TA = nullptr, *TB = nullptr, *sp = nullptr;
        int rc = hc_tables(M == ((int64_t)1 << 18) ? 9 : 8, &TA, &TB);
        if (rc == DD_OK) rc = split ? hilbert_split_spectrum(lens[i], M, &sp, s) : hilbert_kernel_spectrum(lens[i], M, &sp, s);
        if (rc != DD_OK) return rc;
    }
    return DD_OK;
}

// diagnostic (no GPU needed): is this tap set a cosine series b[k] = sum_q a[q] cos(2 pi q k / (K - 1)), q <= 3, as the windows of
// filters.py:101-226 are?  Returns 1 and fills a[0..3], *Q (highest harmonic) when the zero-phase filter of the accurate-sync
// windows takes the prefix-sum form for it, 0 when it keeps the tiled direct form.
extern "C" int dd_debug_cos_fit(const double* taps_host, int K, double* a_out, int* Q_out) {
    DD_REQUIRE(taps_host && K >= 1 && a_out && Q_out, "arguments");
    DDCosFit f;
    if (!dd_cos_fit(taps_host, K, &f) || !dd_fc_ok(K, f.Q)) return 0;
    for (int q = 0; q < 4; ++q) a_out[q] = f.a[q];
    *Q_out = f.Q;
    return 1;
}

// diagnostic: the envelope stage of dd_noaa_sync_windows alone.  X: device c64 [nwin][L] (what the zero-phase FIR leaves),
// env: device f64 [nwin][L - 1] = abs(hilbert(angle(X[n+1] conj X[n]))).  route 0: dd_hconv_kernels.h (needs the padded length
// 2^17 or 2^18, i.e. 32 768 < L <= 131 072; DD_ERR_INVALID otherwise), route 1: the library's padded real transforms.  Synchronises.
extern "C" int dd_debug_sync_envelope(const void* X_dev, int64_t L, int nwin, int route, double* env_dev, void* stream) {
    DD_REQUIRE(X_dev && env_dev && nwin >= 1 && L >= 4 && L < ((int64_t)1 << 30) && (route == 0 || route == 1), "arguments");
    const int64_t L2 = L - 1;
    int64_t M = 1;
    while (M < 2 * L2 + 2) M <<= 1;
    DD_REQUIRE(route == 1 || hc_length_ok(M), "route 0 needs 32768 < L <= 131072");
    const int64_t nb = M / 2 + 1;
    hipStream_t s = dd_stream(stream);
    std::lock_guard<std::mutex> lk(g_sync_mu);
    const double2* HH = nullptr;
    int rc = hilbert_kernel_spectrum(L2, M, &HH, s);
    if (rc != DD_OK) return rc;
    const float2* X = (const float2*)X_dev;
    const int pairs = (nwin + 1) / 2;
    char* buf = nullptr;
    const size_t bW = sizeof(double2) * (size_t)pairs * M, bSP = sizeof(double2) * (size_t)nwin * nb, bYR = sizeof(double) * (size_t)nwin * M;
    DD_HIP_CHECK(hipMalloc((void**)&buf, bW + (route ? bSP + bYR : 0)));
    if (route == 0) {
        rc = hc_envelope(M, X, L, nwin, HH + nb, (double2*)buf, env_dev, s);
    } else {
        double* XR = (double*)buf;
        double2* SP = (double2*)(buf + bW);
        double* YR = (double*)(buf + bW + bSP);
        hipfftHandle pf, pb;
        rc = get_plan(&pf, HIPFFT_D2Z, M, nwin, s);
        if (rc == DD_OK) rc = get_plan(&pb, HIPFFT_Z2D, M, nwin, s);
        if (rc == DD_OK) {
            hipLaunchKernelGGL(k_sync_fm_pad, dim3(grid1(M), nwin), dim3(256), 0, s, X, L, XR, M);
            hipfftResult r1 = hipfftExecD2Z(pf, XR, (hipfftDoubleComplex*)SP);
            hipLaunchKernelGGL(k_spec_mul, dim3(grid1(nb), nwin), dim3(256), 0, s, SP, HH, nb);
            hipfftResult r2 = hipfftExecZ2D(pb, (hipfftDoubleComplex*)SP, YR);
            hipLaunchKernelGGL(k_env_hypot, dim3(grid1(L2), nwin), dim3(256), 0, s, XR, YR, M, L2, env_dev);
            if (r1 != HIPFFT_SUCCESS || r2 != HIPFFT_SUCCESS) { dd_set_error("hipfft exec failed (%d, %d)", (int)r1, (int)r2); rc = DD_ERR_HIP; }
        }
    }
    hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(s);
    (void)hipFree(buf);
    if (rc != DD_OK) return rc;
    if (e1 != hipSuccess || e2 != hipSuccess) { dd_set_error("dd_debug_sync_envelope: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); return DD_ERR_HIP; }
    return DD_OK;
}

extern "C" int dd_noaa_sync_windows(const void* iq, int iq_kind, const int64_t* starts_host, int n_windows, int64_t win_len,
                                    uint64_t cycles_q64, const double* fir_taps_host, int fir_ntaps,
                                    const double* pre_taps_host, int pre_ntaps, const double* needle_host, int needle_len,
                                    double samp_rate, int64_t* peak_host, double* height_host, double* tsync_host,
                                    void* stream) {
    return dd_noaa_sync_windows_multi(iq, iq_kind, starts_host, nullptr, n_windows, win_len, cycles_q64, fir_taps_host, fir_ntaps,
                                      pre_taps_host, pre_ntaps, needle_host, needle_len, 1, samp_rate, peak_host, height_host, tsync_host, stream);
}

// The windows of several sync words in one call (getAccurateSync searches sync A around the crude A positions and sync B around
// the crude B positions, decode_noaa.py:828-835: two window lists, one chain, two needles of one length): needle_of_window_host[w]
// says which of the n_needles needles window w is correlated with (NULL: needle 0).  One upload, batches that mix the lists, one
// copy back, one synchronisation -- the second call's host work no longer sits between the two lists' kernels.
extern "C" int dd_noaa_sync_windows_multi(const void* iq, int iq_kind, const int64_t* starts_host, const int* needle_of_window_host,
                                          int n_windows, int64_t win_len, uint64_t cycles_q64, const double* fir_taps_host, int fir_ntaps,
                                          const double* pre_taps_host, int pre_ntaps, const double* needle_host, int needle_len, int n_needles,
                                          double samp_rate, int64_t* peak_host, double* height_host, double* tsync_host,
                                          void* stream) {
    DD_REQUIRE(n_windows >= 0, "n_windows");
    if (n_windows == 0) return DD_OK;
    static const char* tenv = getenv("DD_SYNC_TRACE");               // tools: host-side time stamps inside the call, to stderr
    const bool trace = tenv && atoi(tenv);
    auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tt0 = now_us();
    DD_REQUIRE(iq && starts_host && peak_host && height_host && tsync_host, "null buffer");
    DD_REQUIRE(iq_kind == 0 || iq_kind == 1, "iq_kind (0 complex64, 1 uint8 pairs)");
    DD_REQUIRE(fir_taps_host && fir_ntaps >= 1 && pre_ntaps >= 0 && (pre_taps_host || pre_ntaps == 0), "taps");
    DD_REQUIRE(needle_host && needle_len >= 1 && samp_rate > 0, "needle/samp_rate");
    DD_REQUIRE(n_needles >= 1 && n_needles <= DD_CS_MAXNEEDLES, "n_needles (1 or 2)");
    if (needle_of_window_host)
        for (int w = 0; w < n_windows; ++w) DD_REQUIRE(needle_of_window_host[w] >= 0 && needle_of_window_host[w] < n_needles, "needle_of_window");
    const int64_t L = win_len, L2 = win_len - 1;
    DD_REQUIRE(L2 >= 2 && needle_len <= L2 && L < ((int64_t)1 << 30), "window length");
    if (!((double)L2 < 0.45 * samp_rate)) {
        dd_set_error("dd_noaa_sync_windows: windows of %lld samples are not shorter than the 0.45 s peak distance; "
                     "use the per-window entry points", (long long)L);
        return DD_ERR_INVALID;
    }
    if (L <= 3 * fir_ntaps || (pre_ntaps && L2 <= 3 * pre_ntaps)) {
        dd_set_error("The length of the input vector x must be greater than padlen, which is %d.",
                     L <= 3 * fir_ntaps ? 3 * fir_ntaps : 3 * pre_ntaps);
        return DD_ERR_INVALID;
    }
    if (!dd_ff_tiled_ok(fir_ntaps, 8) || (pre_ntaps && !dd_ff_tiled_ok(pre_ntaps, 8))) {
        dd_set_error("dd_noaa_sync_windows: filter too long for the tiled zero-phase kernel");
        return DD_ERR_INVALID;
    }
    // piecewise-constant needles -> runs
    DDRuns2 R2;
    for (int d = 0; d < n_needles; ++d) {
        const double* nh = needle_host + (size_t)d * needle_len;
        DDRuns& R = R2.r[d];
        R.nr = 0;
        for (int t = 0; t < needle_len; ++t) {
            if (t == 0 || nh[t] != nh[t - 1]) {
                if (R.nr == DD_XCORR_MAX_RUNS) {
                    dd_set_error("dd_noaa_sync_windows: the needle has more than %d constant runs", DD_XCORR_MAX_RUNS);
                    return DD_ERR_INVALID;
                }
                R.start[R.nr] = t;
                R.val[R.nr] = nh[t];
                ++R.nr;
            }
        }
        R.start[R.nr] = needle_len;
        double vv = 0.0;
        for (int t = 0; t < needle_len; ++t) vv += nh[t] * nh[t];
        R2.vv[d] = vv;
    }
    for (int d = n_needles; d < DD_CS_MAXNEEDLES; ++d) { R2.r[d] = R2.r[0]; R2.vv[d] = R2.vv[0]; }

    hipStream_t s = dd_stream(stream);
    const float2* tbl = dd_nco_table();
    if (!tbl) {
        dd_set_error("NCO table initialisation failed (no GPU?)");
        return DD_ERR_NODEVICE;
    }
    // (the DD_SYNC_* switches below are read on every call on purpose: the test suite and tools/ change routes inside one process)
    const char* fr_env = getenv("DD_SYNC_FRONT");                     // tools / tests: "kernel" = the front end as a launch of its own
    const bool front_fused = !(fr_env && !strcmp(fr_env, "kernel"));
    int bmax = 64;
    if (const char* e = getenv("DD_SYNC_BATCH")) bmax = atoi(e) > 0 ? atoi(e) : bmax;
    const int B = n_windows < bmax ? n_windows : bmax;
    const int64_t N1 = L + 6 * (int64_t)fir_ntaps, N2 = L2 + 6 * (int64_t)pre_ntaps;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // layout (per batch of B windows)
    const size_t o_starts = 0;
    const size_t o_group = o_starts + al(sizeof(int64_t) * n_windows);
    const size_t o_taps1 = o_group + al(sizeof(int) * n_windows);
    const size_t o_taps2 = o_taps1 + al(sizeof(double) * fir_ntaps);
    const size_t o_tab = o_taps2 + al(sizeof(double) * (pre_ntaps ? pre_ntaps : 1));
    const size_t o_res = o_tab + al(sizeof(double2) * 3 * (size_t)(pre_ntaps ? pre_ntaps : 1));
    const size_t o_X = o_res + al(24 * (size_t)n_windows);
    const size_t o_Y1 = o_X + al(sizeof(float2) * B * L);                 // X: c64 [B][L]; later the filtered IQ again
    const size_t o_W = o_Y1 + al(sizeof(float2) * B * N1);                // Y1: c64 [B][N1]
    int64_t M = 1;
    while (M < 2 * L2 + 2) M <<= 1;                                       // cyclic convolution length of the envelope stage
    const int64_t nb = M / 2 + 1;
    const char* hm = getenv("DD_SYNC_HILBERT");
    const bool hilbert_fft = hm && !strcmp(hm, "fft");                    // A/B switches: the library's length-N transforms ("fft"),
    const bool hilbert_own = !hilbert_fft && hc_length_ok(M) && !(hm && !strcmp(hm, "lib"));   // its padded real transforms ("lib"); dd_hconv_kernels.h
    const size_t o_SP = o_W + al(sizeof(double) * (B + (B & 1)) * M);     // (two windows share one complex [M] image in dd_hconv_kernels.h)                 // W/XR: f64 [B][M] (or c128 [B][L2]); later P, Q: f64 [B][L2+1] each
    const size_t o_YR = o_SP + al(sizeof(double2) * B * nb);              // SP: c128 [B][M/2+1]
    const size_t o_ENV = o_YR + al(sizeof(double) * B * M);               // YR: f64 [B][M]
    const size_t o_F1 = o_ENV + al(sizeof(double) * B * L2);              // ENV f64 [B][L2]
    const size_t o_H = o_F1 + al(sizeof(double) * B * N2);                // F1: f64 [B][N2]; later the scan tile sums and per-tile peak records
    const size_t total = o_H + al(sizeof(double) * B * L2);               // H: f64 [B][L2]
    char* base = nullptr;
    std::lock_guard<std::mutex> lk(g_sync_mu);
    int rc = sync_scratch(total + 4096, &base);
    if (rc != DD_OK) return rc;
    DDSyncOnExit sync_guard(s);                       // (an early error return below leaves nothing in flight)
    int64_t* d_starts = (int64_t*)(base + o_starts);
    const int* d_group = needle_of_window_host ? (const int*)(base + o_group) : nullptr;
    double* d_taps1 = (double*)(base + o_taps1);
    double* d_taps2 = (double*)(base + o_taps2);
    int64_t* d_peak = (int64_t*)(base + o_res);
    double* d_height = (double*)(base + o_res + 8 * (size_t)n_windows);
    double* d_tsync = (double*)(base + o_res + 16 * (size_t)n_windows);
    float2* X = (float2*)(base + o_X);
    float2* Y1 = (float2*)(base + o_Y1);
    double2* W = (double2*)(base + o_W);
    double* XR = (double*)(base + o_W);
    double2* SP = (double2*)(base + o_SP);
    double* YR = (double*)(base + o_YR);
    const double2* HH = nullptr;
    if (!hilbert_fft) {
        rc = hilbert_kernel_spectrum(L2, M, &HH, s);
        if (rc != DD_OK) return rc;
    }
    double* ENV = (double*)(base + o_ENV);
    double* F1 = (double*)(base + o_F1);
    double* H = (double*)(base + o_H);
    // window starts, both tap sets and the cosine table go up as ONE copy (they are neighbours in the layout)
    std::vector<char> up(o_res, 0);
    memcpy(up.data() + o_starts, starts_host, sizeof(int64_t) * n_windows);
    if (needle_of_window_host) memcpy(up.data() + o_group, needle_of_window_host, sizeof(int) * n_windows);
    memcpy(up.data() + o_taps1, fir_taps_host, sizeof(double) * fir_ntaps);
    if (pre_ntaps) memcpy(up.data() + o_taps2, pre_taps_host, sizeof(double) * pre_ntaps);
    // the envelope's pre-filter is hamming(492) (decode_noaa.py:677): a two-term cosine series -- prefix-sum form
    // (dd_filtfilt_kernels.h; DD_SYNC_DIRECT_FIR=1, tools: the 492 multiply-adds per sample of the tiled direct form)
    DDCosFit fit2;
    static const char* direct_env = DD_TUNE_ENV("DD_SYNC_DIRECT_FIR");
    const bool cos2 = pre_ntaps && !(direct_env && atoi(direct_env)) && dd_cos_fit_cached(pre_taps_host, pre_ntaps, &fit2) && dd_fc_ok(pre_ntaps, fit2.Q);
    double2* d_tab = (double2*)(base + o_tab);
    if (cos2) {
        // (the table of one tap set is kept on the host between calls; the copy's pageable source is staged before the call returns)
        static std::mutex tab_mu;
        static std::vector<double2> tabh;
        static int tab_K = 0, tab_Q = 0;
        std::lock_guard<std::mutex> tl(tab_mu);
        if (tab_K != pre_ntaps || tab_Q != fit2.Q) { dd_cos_table(pre_ntaps, fit2.Q, tabh); tab_K = pre_ntaps; tab_Q = fit2.Q; }
        memcpy(up.data() + o_tab, tabh.data(), sizeof(double2) * tabh.size());
    }
    const double tt_up0 = now_us() - tt0;
    DD_HIP_CHECK(hipMemcpyAsync(base, up.data(), o_res, hipMemcpyHostToDevice, s));          // (pageable source: staged before the call returns)
    const double tt_up1 = now_us() - tt0;
    for (int w0 = 0; w0 < n_windows; w0 += B) {
        const int b = n_windows - w0 < B ? n_windows - w0 : B;
        const dim3 gL(grid1(L), b), gL2(grid1(L2), b), gL4(grid1((L + 3) / 4), b);
        if (front_fused) {
            // X <- filtfilt(oscillator x raw IQ): pass 1 computes the samples where it stages them
            const DDFrontSrc F = {iq, d_starts + w0, cycles_q64, tbl, iq_kind};
            dd_filtfilt_front_launch(F, Y1, X, L, L, fir_ntaps, d_taps1, b, s);
        } else {
            if (iq_kind == 1) hipLaunchKernelGGL(k_sync_front<true>, gL4, dim3(256), 0, s, iq, d_starts + w0, L, cycles_q64, tbl, X);
            else hipLaunchKernelGGL(k_sync_front<false>, gL4, dim3(256), 0, s, iq, d_starts + w0, L, cycles_q64, tbl, X);
            dd_filtfilt_launch<float2>(X, L, Y1, X, L, L, fir_ntaps, d_taps1, b, s);        // X <- filtfilt(X): pass 2 reads only Y1
        }
        if (hilbert_fft) {
            hipLaunchKernelGGL(k_sync_fm, gL2, dim3(256), 0, s, X, L, W);
            hipfftHandle plan;
            rc = get_plan(&plan, HIPFFT_Z2Z, L2, b, s);
            if (rc != DD_OK) return rc;
            DD_FFT_CHECK(hipfftExecZ2Z(plan, (hipfftDoubleComplex*)W, (hipfftDoubleComplex*)W, HIPFFT_FORWARD));
            hipLaunchKernelGGL(k_hilbert_mask_b, gL2, dim3(256), 0, s, W, L2);
            DD_FFT_CHECK(hipfftExecZ2Z(plan, (hipfftDoubleComplex*)W, (hipfftDoubleComplex*)W, HIPFFT_BACKWARD));
            hipLaunchKernelGGL(k_cplx_abs_b, gL2, dim3(256), 0, s, W, ENV, L2, 1.0 / (double)L2);
        } else if (hilbert_own) {
            rc = hc_envelope(M, X, L, b, HH + nb, W, ENV, s);
            if (rc != DD_OK) return rc;
        } else {
            hipfftHandle pf, pb;
            rc = get_plan(&pf, HIPFFT_D2Z, M, b, s);
            if (rc == DD_OK) rc = get_plan(&pb, HIPFFT_Z2D, M, b, s);
            if (rc != DD_OK) return rc;
            hipLaunchKernelGGL(k_sync_fm_pad, dim3(grid1(M), b), dim3(256), 0, s, X, L, XR, M);
            DD_FFT_CHECK(hipfftExecD2Z(pf, XR, (hipfftDoubleComplex*)SP));
            hipLaunchKernelGGL(k_spec_mul, dim3(grid1(nb), b), dim3(256), 0, s, SP, HH, nb);
            DD_FFT_CHECK(hipfftExecZ2D(pb, (hipfftDoubleComplex*)SP, YR));
            hipLaunchKernelGGL(k_env_hypot, gL2, dim3(256), 0, s, XR, YR, M, L2, ENV);
        }
        const double* hay = ENV;
        if (pre_ntaps) {
            if (cos2) {
                rc = dd_filtfilt_cos_launch(ENV, L2, F1, H, L2, L2, pre_ntaps, fit2, d_tab, b, s);
                if (rc != DD_OK) return rc;
            } else {
                dd_filtfilt_launch<double>(ENV, L2, F1, H, L2, L2, pre_ntaps, d_taps2, b, s);
            }
            hay = H;
        }
        double* P = (double*)W;                                            // prefix sums into the (now free) FFT buffer
        double* Q = P + (size_t)b * (L2 + 1);
        const int stiles = (int)((L2 + DD_SCAN_TILE - 1) / DD_SCAN_TILE), xtiles = (int)((L2 + DD_XC_TILE - 1) / DD_XC_TILE);
        double2* spart = (double2*)F1;                                     // tile sums, then the per-tile peak records:
        DDPk* ppart = (DDPk*)(F1 + 2 * (size_t)b * stiles);                // both in the pre-filter's (now free) work buffer
        hipLaunchKernelGGL(k_scan_part, dim3(stiles, b), dim3(256), 0, s, hay, L2, stiles, spart);
        hipLaunchKernelGGL(k_scan_final, dim3(stiles, b), dim3(256), 0, s, hay, L2, stiles, spart, P, Q);
        hipLaunchKernelGGL(k_xcorr_runs_pk, dim3(8 * ((b + 7) / 8) * xtiles), dim3(256), 0, s, P, Q, L2, needle_len, R2, d_group ? d_group + w0 : nullptr, xtiles, b, ppart);
        hipLaunchKernelGGL(k_sync_peak, dim3(b), dim3(256), 0, s, ppart, xtiles, ENV, L2, needle_len, d_peak + w0, d_height + w0, d_tsync + w0);
        DD_LAUNCH_CHECK();
    }
    char* down = nullptr;                                                                        // the three result arrays, one copy (pinned)
    rc = sync_pinned(24 * (size_t)n_windows, &down);
    if (rc != DD_OK) return rc;
    const double tt_enq = now_us() - tt0;
    DD_HIP_CHECK(hipMemcpyAsync(down, base + o_res, 24 * (size_t)n_windows, hipMemcpyDeviceToHost, s));
    DD_HIP_CHECK(hipStreamSynchronize(s));
    sync_guard.done();
    { const int sr = dd_seam_poll_all(); if (sr != DD_OK) return sr; }
    if (trace) fprintf(stderr, "sync windows host us (%d windows): upload starts %.0f, upload enqueued %.0f, batches enqueued %.0f, synchronised %.0f\n",
                       n_windows, tt_up0, tt_up1, tt_enq, now_us() - tt0);
    memcpy(peak_host, down, 8 * (size_t)n_windows);
    memcpy(height_host, down + 8 * (size_t)n_windows, 8 * (size_t)n_windows);
    memcpy(tsync_host, down + 16 * (size_t)n_windows, 8 * (size_t)n_windows);
    return DD_OK;
}

// ---------------------------------------------------------------- getCrudeSync's audio-rate tail in ONE host call
// decode_noaa.py:781-790: envelope of the FM audio in 240 000-sample blocks (__getAM :631-657 -> demod_am.py:29), then for sync A
// and sync B the normalised correlation (:659-675) and the peak pick (:713-751).  Stage by stage through the entry points
// above that was ~70 launches, a dozen host round trips and -- measured at 60 s of recording -- 2.0 of the 2.2 ms of the crude
// sync (profiles/r03_side_benchmarks.txt); the samples themselves are 3.6 M doubles.  Here:
//   * envelope = hypot(x, H x) with H x from a real-to-complex / complex-to-real transform pair per block (bin k of the
//     spectrum times -j for 0 < k < N/2, zero at DC and Nyquist: the imaginary part of scipy.signal.hilbert's analytic
//     signal) -- half the transform work of the complex pair, batched over the full blocks;
//   * prefix sums of the envelope and its square ONCE, both needles correlated in one launch (blockIdx.y);
//   * the means of the K largest / K smallest correlation values, the threshold and the candidate list of BOTH needles in
//     eleven launches that never come back to the host: eight radix-select passes (one byte of the order-preserving key
//     each; every workgroup re-derives the bins picked so far from the earlier passes' global histograms, so no pick
//     kernel sits between them), the collection of the values beyond the K-th, their sort and ascending summation
//     (one workgroup per needle), the candidates by atomic append -- instead of 2 x 19 dependent launches and 2 x 2 host
//     round trips.  (Tried first: all of it as ONE persistent launch with grid-wide barriers.  It measured 0.56-0.76 ms:
//     ten barriers of 2 x 128..512 workgroups polling one word each cost more than the launch boundaries they replaced,
//     profiles/r04_noaa_stages.txt);
//   * one host synchronisation at the end (the grouping by 0.45 s of :729-746 runs on the host over a few thousand candidates).
// Results: the index lists are those of the staged route and of the reference (tests/golden/noaa_c4*.npz); the envelope agrees
// with the complex-transform form to ~1e-15 relative.
#define DD_CS_WG 512                  // workgroups per needle and selection launch
#define DD_CS_COPIES 8                // interleaved LDS histograms per selection
#define DD_CS_KMAX 2048               // largest K (two per second of audio + 2) the in-kernel sort holds
struct DDCrudeSel {
    unsigned int hist[8][2][256];     // per pass: [K-th largest | K-th smallest]
    unsigned int n_beyond[2];         // values appended above / below
    unsigned int n_cand;              // candidates appended
    unsigned int pad;
    unsigned int beyond_cnt[2];       // bookkeeping: how many values lie strictly beyond the final keys
    unsigned long long key[2];
    double thr, sum_hi, sum_lo;
};

__global__ void __launch_bounds__(256) k_cvt_f32_f64(const float* __restrict__ in, double* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}
// spectrum of a real block -> spectrum of its Hilbert transform (blockIdx.y = block of the batch; nb = N/2 + 1 bins)
__global__ void __launch_bounds__(256) k_hilb_bins(double2* __restrict__ S, int64_t nb, int64_t N) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nb) return;
    double2* p = S + (int64_t)blockIdx.y * nb + k;
    const double2 v = *p;
    const bool zero = k == 0 || (2 * k == N);
    *p = zero ? make_double2(0.0, 0.0) : make_double2(v.y, -v.x);          // -j X[k]
}
__global__ void __launch_bounds__(256) k_env_hypot_flat(const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ env, int64_t n, double inv_n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) env[i] = hypot(x[i], y[i] * inv_n);
}
__global__ void __launch_bounds__(256) k_pad_f64(const double* __restrict__ x, int64_t n, double* __restrict__ XR, int64_t M) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < M) XR[j] = j < n ? x[j] : 0.0;
}
// exclusive scan of the tile sums (one workgroup), so that the final pass adds one number per tile instead of walking all
// the tiles before it (1765 of them for a minute of audio)
__global__ void __launch_bounds__(256) k_scan_mid(double2* __restrict__ part, int tiles) {
    __shared__ double sp[4], sq[4];
    __shared__ double cp, cq;
    if (threadIdx.x == 0) { cp = 0.0; cq = 0.0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int b = 0; b < tiles; b += 256) {
        const int i = b + threadIdx.x;
        const double2 v = i < tiles ? part[i] : make_double2(0.0, 0.0);
        double ip = v.x, iq = v.y;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double a = __shfl_up(ip, d), c = __shfl_up(iq, d);
            if (lane >= d) { ip += a; iq += c; }
        }
        if (lane == 63) { sp[wv] = ip; sq[wv] = iq; }
        __syncthreads();
        double op = cp, oq = cq;
        for (int w = 0; w < wv; ++w) { op += sp[w]; oq += sq[w]; }
        if (i < tiles) part[i] = make_double2(op + ip - v.x, oq + iq - v.y);
        __syncthreads();
        if (threadIdx.x == 255) { cp = op + ip; cq = oq + iq; }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_scan_final_x(const double* __restrict__ h, int64_t n, const double2* __restrict__ partx,
                                                      double* __restrict__ P, double* __restrict__ Q) {
    __shared__ double sp[4], sq[4];
    __shared__ double lds[DD_SCAN_LDS];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t tile0 = (int64_t)blockIdx.x * DD_SCAN_TILE;
    double p[8], q[8];
    dd_scan_tile_load(h, n, tile0, t, lds, p, q);
    const double2 base = partx[blockIdx.x];
    double cp = base.x, cq = base.y;
    double tp = p[7], tq = q[7];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double a = __shfl_up(tp, d), c = __shfl_up(tq, d);
        if (lane >= d) { tp += a; tq += c; }
    }
    if (lane == 63) { sp[wv] = tp; sq[wv] = tq; }
    double ep = __shfl_up(tp, 1), eq = __shfl_up(tq, 1);
    if (lane == 0) { ep = 0.0; eq = 0.0; }
    __syncthreads();
    for (int w = 0; w < wv; ++w) { cp += sp[w]; cq += sq[w]; }
    ep += cp;
    eq += cq;
    if (blockIdx.x == 0 && t == 0) { P[0] = 0.0; Q[0] = 0.0; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[j] += ep; q[j] += eq; }
    dd_scan_tile_store(P, n, tile0, t, lds, p);
    dd_scan_tile_store(Q, n, tile0, t, lds, q);
}
// k_xcorr_runs for up to two needles of equal length at once (blockIdx.y = needle; out[needle][n])
// The 256 outputs of a workgroup read P at a0 + start[r], r = 0 .. nr: 256 + m + 1 consecutive prefix sums, each wanted by
// ~nr outputs.  They are staged in LDS once (when they fit: 817 doubles for the crude needles) -- straight from L2 the kernel
// ran at the L2's bandwidth, 108 us for 2 x 3.6 M outputs.
// A lane owns outputs t, t + 256, t + 512, t + 768 of a 1024-output tile: four independent chains per run boundary (one
// output per lane was a chain of ~15 dependent LDS reads per wave: 93 us for 2 x 3.6 M outputs, latency bound).
#define DD_XC_LDS_MAX 4096
#define DD_XCN_TILE 1024
// (round 4: the run table comes out of LDS instead of one scalar load from the kernel arguments per run and the loop is
// unrolled by four -- the loop used to wait for that load, then for its four reads, run after run; the energy look-ups of
// the four outputs are issued together.  Same operations in the same order per output.)
template <bool STAGED>
__global__ void __launch_bounds__(256) k_xcorr_runs_n(const double* __restrict__ P, const double* __restrict__ Q, int64_t n, int m,
                                                      const DDRuns2 R2, double* __restrict__ out) {
    __shared__ double sP[STAGED ? DD_XC_LDS_MAX : 1];
    __shared__ double sval[DD_XCORR_MAX_RUNS];
    __shared__ int sst[DD_XCORR_MAX_RUNS + 4];
    const DDRuns& R = R2.r[blockIdx.y];
    const int nr = R.nr;
    const int64_t i0 = (int64_t)blockIdx.x * DD_XCN_TILE;
    const int64_t base = i0 + (m - 1) / 2 - (m - 1);             // window of output i: P[base + (i - i0) + start[r]]
    auto at = [&](const double* S, int64_t x) { return S[x < 0 ? 0 : (x > n ? n : x)]; };
    if (threadIdx.x < DD_XCORR_MAX_RUNS) {
        const int r = threadIdx.x;
        sst[r] = r < nr ? R.start[r + 1] : 0;                     // sst[r] = end of run r
        sval[r] = r < nr ? R.val[r] : 0.0;
    }
    if (STAGED)
        for (int k = threadIdx.x; k < DD_XCN_TILE + m + 1; k += 256) sP[k] = at(P, base + k);
    // energy window ends of this lane's four outputs (independent of the loop below: in flight across it)
    double qa[4], qb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t a0 = base + threadIdx.x + 256 * u;
        qa[u] = at(Q, a0);
        qb[u] = at(Q, a0 + m);
    }
    __syncthreads();
    auto look = [&](int u, int st) -> double {
        return STAGED ? sP[threadIdx.x + 256 * u + st] : at(P, base + threadIdx.x + 256 * u + st);
    };
    double c[4] = {0.0, 0.0, 0.0, 0.0}, lo[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) lo[u] = look(u, 0);
    int r = 0;
    for (; r + 4 <= nr; r += 4) {
        int st[4];
        double v[4], hi[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { st[k] = sst[r + k]; v[k] = sval[r + k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) hi[k][u] = look(u, st[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) { c[u] = fma(v[k], hi[k][u] - lo[u], c[u]); lo[u] = hi[k][u]; }
    }
    for (; r < nr; ++r) {
        const int st = sst[r];
        const double v = sval[r];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const double hi = look(u, st); c[u] = fma(v, hi - lo[u], c[u]); lo[u] = hi; }
    }
    const double qn = 1e-13 * Q[n], vv = R2.vv[blockIdx.y];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + threadIdx.x + 256 * u;
        double e = qb[u] - qa[u];
        double cc = c[u];
        if (!(e > qn)) { cc = 0.0; e = 0.0; }
        if (i < n) out[(int64_t)blockIdx.y * n + i] = cc / sqrt(e * vv);
    }
}

__device__ __forceinline__ double dd_aload_f64(const double* p) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    return __longlong_as_double((long long)__hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// one wave: the bin that holds rank `r` counted from the top (TOP) or the bottom of a 256-bin histogram, and how many values
// lie in the bins beyond it.  Lane l owns bins 4 l .. 4 l + 3.
template <bool TOP>
__device__ __forceinline__ void dd_pick_bin(const unsigned int* gh, unsigned int r, int lane, int* bin, unsigned int* beyond) {
    unsigned int c[4], tot = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = gh[4 * lane + j]; tot += c[j]; }
    unsigned int incl = tot;                          // TOP: sum over lanes >= l; else lanes <= l
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int u = TOP ? __shfl_down(incl, d) : __shfl_up(incl, d);
        if (TOP ? (lane + d < 64) : (lane >= d)) incl += u;
    }
    unsigned int before = incl - tot;                 // values in the lanes beyond this one
    int found = -1;
    unsigned int fb = 0;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = TOP ? 3 - jj : jj;
        if (found < 0 && before + c[j] >= r) { found = 4 * lane + j; fb = before; }
        before += c[j];
    }
    // the first lane from the far end that finds it is the one; broadcast
    const unsigned long long m = __ballot(found >= 0);
    const int src = m ? (TOP ? (63 - __builtin_clzll(m)) : __builtin_ctzll(m)) : 0;
    *bin = __shfl(found, src);
    *beyond = __shfl(fb, src);
    if (!m) { *bin = TOP ? 0 : 255; *beyond = 0; }
}
// The selections' state after passes 0 .. upto-1, recomputed from the global histograms of those passes (complete: they were
// filled by earlier launches) by waves 0 (K-th largest) and 1 (K-th smallest) of every workgroup, and handed to all lanes.
struct DDCsState { unsigned long long prefix[2]; unsigned int remaining[2], beyond[2]; };
__device__ __forceinline__ DDCsState dd_cs_state(const DDCrudeSel* S, int upto, int K, DDCsState* lds_tmp) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv < 2) {
        unsigned long long prefix = 0ull;
        unsigned int remaining = (unsigned int)K, beyond = 0u;
        for (int p = 0; p < upto; ++p) {
            int bin;
            unsigned int by;
            if (wv == 0) dd_pick_bin<true>(S->hist[p][0], remaining, lane, &bin, &by);
            else dd_pick_bin<false>(S->hist[p][1], remaining, lane, &bin, &by);
            prefix = (prefix << 8) | (unsigned long long)bin;
            remaining -= by;
            beyond += by;
        }
        if (lane == 0) { lds_tmp->prefix[wv] = prefix; lds_tmp->remaining[wv] = remaining; lds_tmp->beyond[wv] = beyond; }
    }
    __syncthreads();
    const DDCsState st = *lds_tmp;
    __syncthreads();
    return st;
}

// pass `pass` of the radix select (one byte of the key): histogram of the values whose higher bytes equal the prefix so far.
// grid (G, needles); the launch boundary is the barrier between passes.
__global__ void __launch_bounds__(256) k_cs_hist(const double* __restrict__ cor_all, int64_t n, int K, int pass, DDCrudeSel* __restrict__ sel_all) {
    __shared__ unsigned int h[2][DD_CS_COPIES][256];
    __shared__ DDCsState tmp;
    const int nd = blockIdx.y, g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
    const double* cor = cor_all + (int64_t)nd * n;
    DDCrudeSel* S = sel_all + nd;
    for (int i = t; i < 2 * DD_CS_COPIES * 256; i += 256) (&h[0][0][0])[i] = 0;
    const DDCsState st = dd_cs_state(S, pass, K, &tmp);          // (its barriers also cover the clearing above)
    const int64_t i_lo = n * g / G, i_hi = n * (g + 1) / G;
    const int shift = 56 - 8 * pass;
    const int copy = t & (DD_CS_COPIES - 1);
    for (int64_t i = i_lo + t; i < i_hi; i += 1024) {             // four loads in flight per lane
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i + 256 * u < i_hi) ? cor[i + 256 * u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + 256 * u >= i_hi) break;
            const unsigned long long k = dd_key_f64(v[u]);
            const unsigned long long hi = pass ? (k >> (shift + 8)) : 0;
            const unsigned int d = (unsigned int)(k >> shift) & 255u;
            if (hi == st.prefix[0]) atomicAdd(&h[0][copy][d], 1u);
            if (hi == st.prefix[1]) atomicAdd(&h[1][copy][d], 1u);
        }
    }
    __syncthreads();
    for (int i = t; i < 512; i += 256) {
        unsigned int c = 0;
#pragma unroll
        for (int k = 0; k < DD_CS_COPIES; ++k) c += h[i >> 8][k][i & 255];
        if (c) atomicAdd(&S->hist[pass][i >> 8][i & 255], c);
    }
}
// the values strictly beyond the two final keys (fewer than K each), any order
__global__ void __launch_bounds__(256) k_cs_collect(const double* __restrict__ cor_all, int64_t n, int K, DDCrudeSel* __restrict__ sel_all, double* __restrict__ beyond_all) {
    __shared__ DDCsState tmp;
    const int nd = blockIdx.y, g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
    const double* cor = cor_all + (int64_t)nd * n;
    DDCrudeSel* S = sel_all + nd;
    double* above = beyond_all + (size_t)nd * 2 * DD_CS_KMAX;
    double* below = above + DD_CS_KMAX;
    const DDCsState st = dd_cs_state(S, 8, K, &tmp);
    const int64_t i_lo = n * g / G, i_hi = n * (g + 1) / G;
    for (int64_t i = i_lo + t; i < i_hi; i += 256) {
        const double v = cor[i];
        const unsigned long long k = dd_key_f64(v);
        if (k > st.prefix[0]) { const unsigned int o = atomicAdd(&S->n_beyond[0], 1u); if (o < DD_CS_KMAX) above[o] = v; }
        if (k < st.prefix[1]) { const unsigned int o = atomicAdd(&S->n_beyond[1], 1u); if (o < DD_CS_KMAX) below[o] = v; }
    }
}
// one workgroup per needle: the K largest (then the K smallest) sorted ascending and summed in that order (the sums over the
// sorted array that np.argpartition's slices stand for, :717-723), threshold
__global__ void __launch_bounds__(256) k_cs_threshold(int K, DDCrudeSel* __restrict__ sel_all, const double* __restrict__ beyond_all) {
    __shared__ double srt[DD_CS_KMAX];
    __shared__ DDCsState tmp;
    const int nd = blockIdx.x, t = threadIdx.x;
    DDCrudeSel* S = sel_all + nd;
    const double* above = beyond_all + (size_t)nd * 2 * DD_CS_KMAX;
    const double* below = above + DD_CS_KMAX;
    const DDCsState st = dd_cs_state(S, 8, K, &tmp);
    double sums[2] = {0.0, 0.0};
    for (int w = 0; w < 2; ++w) {
        const unsigned int nb = st.beyond[w];
        const unsigned long long kk = st.prefix[w];
        const unsigned long long u = (kk >> 63) ? (kk & 0x7fffffffffffffffull) : ~kk;
        const double kth = __longlong_as_double((long long)u);
        const double* src = w ? below : above;
        int np2 = 1;
        while (np2 < K) np2 <<= 1;
        const double inf = __longlong_as_double(0x7ff0000000000000ll);
        for (int i = t; i < np2; i += 256) srt[i] = i < (int)nb ? src[i] : (i < K ? kth : inf);
        __syncthreads();
        for (int k2 = 2; k2 <= np2; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int i = t; i < np2; i += 256) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const double a = srt[i], b = srt[ixj];
                        const bool up = (i & k2) == 0;
                        if (up ? (a > b) : (a < b)) { srt[i] = b; srt[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        if (t == 0) {
            double acc = 0.0;
            for (int i = 0; i < K; ++i) acc += srt[i];
            sums[w] = acc;
            if (w == 0) S->sum_hi = acc; else S->sum_lo = acc;
        }
        __syncthreads();
    }
    if (t == 0) {
        double avgpk = sums[0] / K;
        avgpk -= 0.25 * (avgpk - sums[1] / K);                             // NOAA_PEAKHEIGHTWIGGLE (:723)
        S->thr = avgpk;
        S->key[0] = st.prefix[0]; S->key[1] = st.prefix[1];
        S->beyond_cnt[0] = st.beyond[0]; S->beyond_cnt[1] = st.beyond[1];
    }
}
// candidates cor > threshold (:726) with their heights, IN INDEX ORDER (the grouping of :729-746 walks them in that order; appended
// by atomics they came out shuffled and the host sorted 5 000 + 17 000 of them for the 60 s recording: 0.45 ms of a 1.0 ms call).
// Two launches: every wave counts the candidates of its contiguous stretch, then -- its offset = the counts of the waves before
// it -- writes them where they belong (ballot + prefix count, no barrier).  The first DD_CS_HEAD of a needle go into the block
// the host fetches in its one copy (behind the counters), later ones into the overflow arrays
#define DD_CS_HEAD 24576
#define DD_CS_WAVES (DD_CS_WG * 4)
struct DDCand { int64_t idx; double val; };
struct DDCrudeHead { unsigned int n_cand, n_beyond[2], beyond_cnt[2], pad[3]; };      // 32 bytes per needle, then DDCand[needles][DD_CS_HEAD]
__device__ __forceinline__ void dd_cs_stretch(int64_t n, int wave, int64_t* lo, int64_t* hi) {
    *lo = n * wave / DD_CS_WAVES;
    *hi = n * (wave + 1) / DD_CS_WAVES;
}
__global__ void __launch_bounds__(256) k_cs_cand_count(const double* __restrict__ cor_all, int64_t n, DDCrudeSel* __restrict__ sel_all,
                                                       unsigned int* __restrict__ cnt_all) {
    const int nd = blockIdx.y, lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const double* cor = cor_all + (int64_t)nd * n;
    DDCrudeSel* S = sel_all + nd;
    const double thr = S->thr;
    int64_t lo, hi;
    dd_cs_stretch(n, wave, &lo, &hi);
    unsigned int c = 0;
    for (int64_t i = lo + lane; i < hi; i += 64) c += cor[i] > thr ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d);
    if (lane == 0) {
        cnt_all[(size_t)nd * DD_CS_WAVES + wave] = c;
        if (c) atomicAdd(&S->n_cand, c);
    }
}
__global__ void __launch_bounds__(256) k_cs_cand_write(const double* __restrict__ cor_all, int64_t n, const DDCrudeSel* __restrict__ sel_all,
                                                       const unsigned int* __restrict__ cnt_all, DDCand* __restrict__ head_all,
                                                       int64_t* __restrict__ cidx_all, double* __restrict__ cval_all, unsigned int cap) {
    const int nd = blockIdx.y, lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const double* cor = cor_all + (int64_t)nd * n;
    const unsigned int* cnt = cnt_all + (size_t)nd * DD_CS_WAVES;
    DDCand* head = head_all + (size_t)nd * DD_CS_HEAD;
    int64_t* cidx = cidx_all + (size_t)nd * cap;
    double* cval = cval_all + (size_t)nd * cap;
    const double thr = sel_all[nd].thr;
    if (cnt[wave] == 0) return;                                    // (wave uniform)
    unsigned int off = 0;
    for (int w = lane; w < wave; w += 64) off += cnt[w];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) off += __shfl_xor(off, d);
    int64_t lo, hi;
    dd_cs_stretch(n, wave, &lo, &hi);
    for (int64_t i0 = lo; i0 < hi; i0 += 64) {
        const int64_t i = i0 + lane;
        const double v = i < hi ? cor[i] : 0.0;
        const bool take = i < hi && v > thr;
        const unsigned long long mask = __ballot(take);
        if (take) {
            const unsigned int o = off + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
            if (o < DD_CS_HEAD) head[o] = DDCand{i, v};
            else if (o < cap) { cidx[o] = i; cval[o] = v; }
        }
        off += (unsigned int)__popcll(mask);
    }
}
__global__ void k_cs_head(const DDCrudeSel* __restrict__ sel, DDCrudeHead* __restrict__ hdr, int n_needles) {
    const int d = threadIdx.x;
    if (d >= n_needles) return;
    DDCrudeHead h = {sel[d].n_cand, {sel[d].n_beyond[0], sel[d].n_beyond[1]}, {sel[d].beyond_cnt[0], sel[d].beyond_cnt[1]}, {0, 0, 0}};
    hdr[d] = h;
}


extern "C" int dd_noaa_crude_tail(const void* audio, int audio_is_f32, int64_t n, double samp_rate, int64_t block,
                                  const double* needles_host, int m, int n_needles, double* env_out,
                                  int64_t* peaks_host, int max_peaks, int* n_peaks, void* stream) {
    DD_REQUIRE(audio && n >= 1 && samp_rate > 0 && block >= 1 && needles_host && m >= 1 && m <= n, "arguments");
    DD_REQUIRE(n_needles >= 1 && n_needles <= DD_CS_MAXNEEDLES && peaks_host && n_peaks && max_peaks >= 1, "arguments");
    hipStream_t s = dd_stream(stream);
    const int K = (int)(2 * ((double)n / samp_rate)) + 2;                 // expectedPeaks (:714)
    DD_REQUIRE(K <= n, "signal shorter than the expected peak count");
    if (K > DD_CS_KMAX || n >= ((int64_t)1 << 31)) return DD_ERR_UNSUPPORTED;          // (the caller takes the staged route)
    DDRuns2 R2;
    for (int d = 0; d < n_needles; ++d) {
        const double* nh = needles_host + (size_t)d * m;
        DDRuns& R = R2.r[d];
        R.nr = 0;
        R.start[0] = 0;
        for (int t = 0; t < m; ++t) {
            if (t == 0 || nh[t] != nh[t - 1]) {
                if (R.nr == DD_XCORR_MAX_RUNS) return DD_ERR_UNSUPPORTED;
                R.start[R.nr] = t;
                R.val[R.nr] = nh[t];
                ++R.nr;
            }
        }
        R.start[R.nr] = m;
        double vv = 0.0;
        for (int t = 0; t < m; ++t) vv += nh[t] * nh[t];
        R2.vv[d] = vv;
    }
    for (int d = n_needles; d < DD_CS_MAXNEEDLES; ++d) { R2.r[d] = R2.r[0]; R2.vv[d] = R2.vv[0]; }
    // block list by the chunker rule (decode_noaa.py:644-653 via chunker.py:36-45)
    int64_t nfull = 0;
    while ((nfull + 1) * block < n) ++nfull;
    const int64_t rem = n - nfull * block;
    const int GB = 16;
    const int64_t gb = nfull < GB ? nfull : GB;
    const int64_t nbins_b = block / 2 + 1, nbins_r = rem / 2 + 1;
    const int tiles = (int)((n + DD_SCAN_TILE - 1) / DD_SCAN_TILE);
    const unsigned int cap = 1u << 16;                                    // candidates per needle held on the device
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t o_x = take(audio_is_f32 ? sizeof(double) * (size_t)n : 0);
    const size_t o_env = take(env_out ? 0 : sizeof(double) * (size_t)n);
    // the ragged last block: a length with a large prime factor (14 100 = 2^2 3 5^2 47 for a minute of audio) makes the library
    // run Bluestein's algorithm -- twenty launches for 14 100 samples.  Its envelope then goes through the zero-padded cyclic
    // convolution with the Hilbert kernel that the accurate-sync windows use (hilbert_kernel_spectrum): four launches and two
    // power-of-two transforms.
    int64_t Mr = 0;
    if (rem >= 2 && largest_prime_factor(rem) > 17) { Mr = 1; while (Mr < 2 * rem + 2) Mr <<= 1; }
    const size_t spec_r = (size_t)(Mr ? Mr / 2 + 1 : nbins_r);
    const size_t spec_elems = (size_t)(gb * nbins_b) > spec_r ? (size_t)(gb * nbins_b) : spec_r;
    const size_t o_spec = take(sizeof(double2) * spec_elems);
    const size_t y_r = (size_t)(Mr ? 2 * Mr : rem);
    const size_t o_y = take(sizeof(double) * ((size_t)(gb * block) > y_r ? (size_t)(gb * block) : y_r));
    // Round 5: the blocks' envelopes through the own float64 transform (hc_block_envelope: the even / odd split of the Hilbert kernel puts a
    // 240 000-sample block on the cyclic length 2^18) -- no FFT-library plan on this path, whose creation was 0.9 s of a process's first call.
    // DD_AM_HILBERT=lib (tools / tests) keeps the library's transforms.
    static const char* amh_env = getenv("DD_AM_HILBERT");
    const bool own_ok = !(amh_env && !strcmp(amh_env, "lib"));
    bool split_b = false, split_r = false;
    const int64_t Mb_own = (own_ok && nfull > 0) ? hc_block_len(block, &split_b) : 0;
    const int64_t Mr_own = (own_ok && rem >= 2) ? hc_block_len(rem, &split_r) : 0;
    const int64_t T_elems = std::max<int64_t>(Mb_own && split_b ? gb * Mb_own : (Mb_own ? Mb_own : 0), Mr_own);
    const size_t o_T = take(sizeof(double2) * (size_t)T_elems);
    const size_t o_P = take(sizeof(double) * (size_t)(n + 1)), o_Q = take(sizeof(double) * (size_t)(n + 1));
    const size_t o_part = take(sizeof(double2) * (size_t)tiles);
    const size_t o_cor = take(sizeof(double) * (size_t)n * n_needles);
    const size_t o_sel = take(sizeof(DDCrudeSel) * n_needles);
    const size_t o_bey = take(sizeof(double) * 2 * DD_CS_KMAX * n_needles);
    const size_t o_ci = take(sizeof(int64_t) * (size_t)cap * n_needles), o_cv = take(sizeof(double) * (size_t)cap * n_needles);
    const size_t head_bytes = sizeof(DDCrudeHead) * DD_CS_MAXNEEDLES + sizeof(DDCand) * (size_t)DD_CS_HEAD * n_needles;
    const size_t o_head = take(head_bytes);
    const size_t o_cnt = take(sizeof(unsigned int) * DD_CS_WAVES * n_needles);
    std::lock_guard<std::mutex> lk(g_sync_mu);
    char* base = nullptr;
    int rc = sync_scratch(off, &base);
    if (rc != DD_OK) return rc;
    DDSyncOnExit sync_guard(s);                       // (an early error return below leaves nothing in flight)
    const double* x = audio_is_f32 ? (const double*)(base + o_x) : (const double*)audio;
    double* env = env_out ? env_out : (double*)(base + o_env);
    double2* spec = (double2*)(base + o_spec);
    double* y = (double*)(base + o_y);
    double* P = (double*)(base + o_P);
    double* Q = (double*)(base + o_Q);
    double2* part = (double2*)(base + o_part);
    double* cor = (double*)(base + o_cor);
    DDCrudeSel* sel = (DDCrudeSel*)(base + o_sel);
    static const char* tenv = getenv("DD_CRUDE_TRACE");
    const bool trace = tenv && atoi(tenv);
    auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tt0 = now_us();
    double tt[6] = {0, 0, 0, 0, 0, 0};
    if (audio_is_f32) hipLaunchKernelGGL(k_cvt_f32_f64, dim3(grid1(n)), dim3(256), 0, s, (const float*)audio, (double*)(base + o_x), n);
    // ---- envelope
    auto env_blocks = [&](int64_t first, int64_t N, int batch) -> int {
        hipfftHandle pf, pb;
        int r = get_plan(&pf, HIPFFT_D2Z, N, batch, s);
        if (r == DD_OK) r = get_plan(&pb, HIPFFT_Z2D, N, batch, s);
        if (r != DD_OK) return r;
        const int64_t nb = N / 2 + 1;
        DD_FFT_CHECK(hipfftExecD2Z(pf, (hipfftDoubleReal*)(x + first), (hipfftDoubleComplex*)spec));
        hipLaunchKernelGGL(k_hilb_bins, dim3(grid1(nb), batch), dim3(256), 0, s, spec, nb, N);
        DD_FFT_CHECK(hipfftExecZ2D(pb, (hipfftDoubleComplex*)spec, (hipfftDoubleReal*)y));
        hipLaunchKernelGGL(k_env_hypot_flat, dim3(grid1(N * batch)), dim3(256), 0, s, x + first, y, env + first, N * batch, 1.0 / (double)N);
        return DD_OK;
    };
    double2* Tw = (double2*)(base + o_T);
    if (Mb_own) {
        // (plain form: one block per call; split form: a batch of blocks, one complex image each)
        const int per = split_b ? (int)gb : 1;
        for (int64_t b0 = 0; b0 < nfull && rc == DD_OK; b0 += per)
            rc = hc_block_envelope(x + b0 * block, env + b0 * block, block, (int)(nfull - b0 < per ? nfull - b0 : per), split_b, Mb_own, Tw, s);
    } else {
        for (int64_t b0 = 0; b0 < nfull && rc == DD_OK; b0 += GB) rc = env_blocks(b0 * block, block, (int)(nfull - b0 < GB ? nfull - b0 : GB));
    }
    tt[0] = now_us() - tt0;
    if (rc == DD_OK && Mr_own) {
        rc = hc_block_envelope(x + nfull * block, env + nfull * block, rem, 1, split_r, Mr_own, Tw, s);
    } else if (rc == DD_OK && Mr) {
        const double2* HH = nullptr;
        rc = hilbert_kernel_spectrum(rem, Mr, &HH, s);
        hipfftHandle pf, pb;
        if (rc == DD_OK) rc = get_plan(&pf, HIPFFT_D2Z, Mr, 1, s);
        if (rc == DD_OK) rc = get_plan(&pb, HIPFFT_Z2D, Mr, 1, s);
        if (rc == DD_OK) {
            double* XR = y, *YR = y + Mr;
            const int64_t nb = Mr / 2 + 1;
            hipLaunchKernelGGL(k_pad_f64, dim3(grid1(Mr)), dim3(256), 0, s, x + nfull * block, rem, XR, Mr);
            DD_FFT_CHECK(hipfftExecD2Z(pf, XR, (hipfftDoubleComplex*)spec));
            hipLaunchKernelGGL(k_spec_mul, dim3(grid1(nb), 1), dim3(256), 0, s, spec, HH, nb);
            DD_FFT_CHECK(hipfftExecZ2D(pb, (hipfftDoubleComplex*)spec, YR));
            hipLaunchKernelGGL(k_env_hypot, dim3(grid1(rem), 1), dim3(256), 0, s, XR, YR, Mr, rem, env + nfull * block);
        }
    } else if (rc == DD_OK) {
        rc = env_blocks(nfull * block, rem, 1);
    }
    if (rc != DD_OK) return rc;
    tt[1] = now_us() - tt0;
    // ---- prefix sums once, both correlations in one launch; selection, threshold, candidates of both needles: twelve launches,
    // nothing comes back to the host in between.  (These eighteen launches of our own kernels were also replayed as ONE captured
    // graph launch -- 300 calls with identical results -- for no gain, 0.995 against 0.985 ms per call: DD_CRUDE_TRACE=1 shows the
    // host done enqueueing the whole call after 0.12 ms of the 0.53 ms the device needs.  What the call did lose was 0.45 ms on the
    // host AFTER the synchronisation, sorting candidates: k_cs_cand_write.  profiles/r04_noaa_timeline.txt)
    auto enqueue_tail = [&](hipStream_t s) -> int {
    hipLaunchKernelGGL(k_scan_part, dim3(tiles, 1), dim3(256), 0, s, env, n, tiles, part);
    hipLaunchKernelGGL(k_scan_mid, dim3(1), dim3(256), 0, s, part, tiles);
    hipLaunchKernelGGL(k_scan_final_x, dim3(tiles), dim3(256), 0, s, env, n, part, P, Q);
    if (DD_XCN_TILE + m + 1 <= DD_XC_LDS_MAX)
        hipLaunchKernelGGL(k_xcorr_runs_n<true>, dim3((unsigned)((n + DD_XCN_TILE - 1) / DD_XCN_TILE), n_needles), dim3(256), 0, s, P, Q, n, m, R2, cor);
    else
        hipLaunchKernelGGL(k_xcorr_runs_n<false>, dim3((unsigned)((n + DD_XCN_TILE - 1) / DD_XCN_TILE), n_needles), dim3(256), 0, s, P, Q, n, m, R2, cor);
    DD_HIP_CHECK(hipMemsetAsync(sel, 0, sizeof(DDCrudeSel) * n_needles, s));
    for (int pass = 0; pass < 8; ++pass) hipLaunchKernelGGL(k_cs_hist, dim3(DD_CS_WG, n_needles), dim3(256), 0, s, cor, n, K, pass, sel);
    hipLaunchKernelGGL(k_cs_collect, dim3(DD_CS_WG, n_needles), dim3(256), 0, s, cor, n, K, sel, (double*)(base + o_bey));
    hipLaunchKernelGGL(k_cs_threshold, dim3(n_needles), dim3(256), 0, s, K, sel, (const double*)(base + o_bey));
    DDCrudeHead* d_hdr = (DDCrudeHead*)(base + o_head);
    DDCand* d_head = (DDCand*)(base + o_head + sizeof(DDCrudeHead) * DD_CS_MAXNEEDLES);
    hipLaunchKernelGGL(k_cs_cand_count, dim3(DD_CS_WG, n_needles), dim3(256), 0, s, cor, n, sel, (unsigned int*)(base + o_cnt));
    hipLaunchKernelGGL(k_cs_cand_write, dim3(DD_CS_WG, n_needles), dim3(256), 0, s, cor, n, (const DDCrudeSel*)sel, (const unsigned int*)(base + o_cnt), d_head,
                       (int64_t*)(base + o_ci), (double*)(base + o_cv), cap);
    hipLaunchKernelGGL(k_cs_head, dim3(1), dim3(64), 0, s, sel, d_hdr, n_needles);
    DD_LAUNCH_CHECK();
    return DD_OK;
    };
    rc = enqueue_tail(s);
    if (rc != DD_OK) return rc;
    tt[2] = now_us() - tt0;
    char* pin = nullptr;
    rc = sync_pinned(head_bytes, &pin);
    if (rc != DD_OK) return rc;
    DD_HIP_CHECK(hipMemcpyAsync(pin, base + o_head, head_bytes, hipMemcpyDeviceToHost, s));
    tt[3] = now_us() - tt0;
    DD_HIP_CHECK(hipStreamSynchronize(s));
    sync_guard.done();
    { const int sr = dd_seam_poll_all(); if (sr != DD_OK) return sr; }        // (the audio may come from a chunk-list launch on this stream)
    tt[4] = now_us() - tt0;
    if (trace) fprintf(stderr, "crude tail host us: blocks enqueued %.0f, remainder %.0f, tail enqueued %.0f, copy enqueued %.0f, synchronised %.0f\n", tt[0], tt[1], tt[2], tt[3], tt[4]);
    const DDCrudeHead* hs = (const DDCrudeHead*)pin;
    const DDCand* hc = (const DDCand*)(pin + sizeof(DDCrudeHead) * DD_CS_MAXNEEDLES);
    const unsigned int first_n = DD_CS_HEAD;
    for (int d = 0; d < n_needles; ++d) {
        const DDCrudeHead& h1 = hs[d];
        DD_REQUIRE(h1.n_beyond[0] == h1.beyond_cnt[0] && h1.n_beyond[1] == h1.beyond_cnt[1] && h1.n_beyond[0] < (unsigned int)K && h1.n_beyond[1] < (unsigned int)K,
                   "dd_noaa_crude_tail: selection bookkeeping (internal)");
        const unsigned int count = h1.n_cand;
        if (count > cap) return DD_ERR_UNSUPPORTED;                       // (a threshold that lets > 65 536 values through: staged route)
        std::vector<std::pair<int64_t, double>> cand(count);
        for (unsigned int i = 0; i < count && i < first_n; ++i) cand[i] = {hc[(size_t)d * first_n + i].idx, hc[(size_t)d * first_n + i].val};
        if (count > first_n) {
            const unsigned int more = count - first_n;
            std::vector<int64_t> ci(more);
            std::vector<double> cv(more);
            DD_HIP_CHECK(hipMemcpyAsync(ci.data(), (int64_t*)(base + o_ci) + (size_t)d * cap + first_n, sizeof(int64_t) * more, hipMemcpyDeviceToHost, s));
            DD_HIP_CHECK(hipMemcpyAsync(cv.data(), (double*)(base + o_cv) + (size_t)d * cap + first_n, sizeof(double) * more, hipMemcpyDeviceToHost, s));
            DD_HIP_CHECK(hipStreamSynchronize(s));
            for (unsigned int i = 0; i < more; ++i) cand[first_n + i] = {ci[i], cv[i]};
        }
        // (the candidates arrive in index order: k_cs_cand_write)
        // group by >= 0.45 s from the running maximum, first maximum wins (:729-746)
        const double min_dist = 0.45 * samp_rate;
        std::vector<int64_t> peaks;
        bool have = false;
        double cur_max = 0.0;
        int64_t cur_idx = 0;
        for (unsigned int q = 0; q < count; ++q) {
            if (have && (double)(cand[q].first - cur_idx) >= min_dist) { peaks.push_back(cur_idx); have = false; }
            if (!have || cur_max < cand[q].second) { cur_max = cand[q].second; cur_idx = cand[q].first; have = true; }
        }
        if (have) peaks.push_back(cur_idx);
        const int shift = m / 2;                                          // int(len(sync)/2) (:749)
        for (auto& p : peaks) p -= shift;
        std::sort(peaks.begin(), peaks.end());
        if ((int)peaks.size() > max_peaks) {
            dd_set_error("dd_noaa_crude_tail: %d peaks found, buffer holds %d", (int)peaks.size(), max_peaks);
            return DD_ERR_INVALID;
        }
        for (size_t i = 0; i < peaks.size(); ++i) peaks_host[(size_t)d * max_peaks + i] = peaks[i];
        n_peaks[d] = (int)peaks.size();
        if (trace) fprintf(stderr, "   needle %d: %u candidates, %d peaks, done at %.0f us\n", d, count, (int)peaks.size(), now_us() - tt0);
    }
    return DD_OK;
}

