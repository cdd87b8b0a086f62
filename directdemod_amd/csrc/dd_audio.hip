// Audio-rate rows of the NOAA tail (R2, A1, X1, X2, and 8f-2 / P: the accurate-sync windows and the crude tail), float64 on the
// device so that sync index picks stay bit-exact (H7).  Sizes here are 1e5 .. 1e7 samples.  This file holds what the families share -- the FFT
// library's plan cache (the stand-alone class routes only), the own float64 cyclic convolution of 2^17 / 2^18 points (dd_hconv_kernels.h) and
// its tables -- and includes the five parts (dd_audio_envelope.h, dd_audio_resample.h, dd_audio_xcorr.h, dd_audio_sync.h, dd_audio_crude.h):
// one translation unit, split by entry-point family in round 6.
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

// ---------------------------------------------------------------- the entry-point families (round 6: one file each, one translation unit)
#include "dd_audio_envelope.h"      // A1   dd_am_envelope_f64
#include "dd_audio_resample.h"      // R2   dd_resample_fft_f64 / _chunks, dd_rpoly_*
#include "dd_audio_xcorr.h"         // X1, X2   dd_xcorr_norm_f64, dd_find_peaks_f64
#include "dd_audio_sync.h"          // 8f-2  dd_noaa_sync_windows(_multi), dd_noaa_prepare
#include "dd_audio_crude.h"         // P     dd_noaa_crude_tail

// dd_code_warmup (dd_runtime.hip): the runtime loads a translation unit's code object when one of its kernels is first named
int dd_code_touch_audio(void) {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, (const void*)k_cvt_f32_f64) == hipSuccess ? DD_OK : DD_ERR_HIP;
}
