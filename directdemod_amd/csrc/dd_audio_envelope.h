// A1: demod_am.demod -- abs(hilbert(x)) per block (demod_am.py:18-29, decode_noaa.py:644-653): dd_am_envelope_f64
// One of the five parts of dd_audio.hip (round 6: the 2600-line unit split along its entry-point families; still ONE translation unit --
// the parts share the plan cache, the float64 transform and the scratch buffers of dd_audio.hip and are included there, in this order).
// Internal; not a stand-alone header.
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
