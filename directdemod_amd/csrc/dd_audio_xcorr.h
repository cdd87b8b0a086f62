// X1 / X2: decode_noaa.__correlate and __correlateAndFindPeaks (decode_noaa.py:659-767): dd_xcorr_norm_f64, dd_find_peaks_f64, and the audio-rate entry points' scratch buffers
// One of the five parts of dd_audio.hip (round 6: the 2600-line unit split along its entry-point families; still ONE translation unit --
// the parts share the plan cache, the float64 transform and the scratch buffers of dd_audio.hip and are included there, in this order).
// Internal; not a stand-alone header.
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
