// R2: commSignal.bwLim(strict) = scipy.signal.resample (comm.py:110-116): dd_resample_fft_f64, dd_resample_fft_chunks, the chirp-z form, and the polyphase extension dd_rpoly_*
// One of the five parts of dd_audio.hip (round 6: the 2600-line unit split along its entry-point families; still ONE translation unit --
// the parts share the plan cache, the float64 transform and the scratch buffers of dd_audio.hip and are included there, in this order).
// Internal; not a stand-alone header.
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
