// getAccurateSync's windows, batched (decode_noaa.py:808-880): dd_noaa_sync_windows(_multi), dd_noaa_prepare, the Hilbert envelope as a float64 convolution
// One of the five parts of dd_audio.hip (round 6: the 2600-line unit split along its entry-point families; still ONE translation unit --
// the parts share the plan cache, the float64 transform and the scratch buffers of dd_audio.hip and are included there, in this order).
// Internal; not a stand-alone header.
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
// lengths of dd_hconv_kernels.h -- once more behind them in the order its row pass reads (out[N2 k1 + k2] = bin k1 + 512 k2).  Host part:
// no device call (dd_noaa_prepare runs it while the runtime is still busy with the process's first copy).
static void kernel_spectrum_host(const std::vector<double>& img, int64_t M, std::vector<double2>& h) {
    std::vector<std::complex<double>> v((size_t)M);
    for (int64_t i = 0; i < M; ++i) v[(size_t)i] = std::complex<double>(img[(size_t)i], 0.0);
    host_fft_pow2(v);
    const int64_t nb = M / 2 + 1;
    const bool own = hc_length_ok(M);
    h.resize((size_t)(nb + (own ? M : 0)));
    const double sc = 1.0 / (double)M;
    for (int64_t k = 0; k < nb; ++k) h[(size_t)k] = make_double2(v[(size_t)k].real() * sc, v[(size_t)k].imag() * sc);
    if (own) {
        const int lg = M == ((int64_t)1 << 18) ? 9 : 8;
        for (int64_t i = 0; i < M; ++i) {
            const int64_t k = (i >> lg) + DD_HC_N * (i & (((int64_t)1 << lg) - 1));
            h[(size_t)(nb + i)] = make_double2(v[(size_t)k].real() * sc, v[(size_t)k].imag() * sc);
        }
    }
}
// device part: one allocation, one copy
static int kernel_spectrum_put(const std::vector<double2>& h, double2** out, hipStream_t s) {
    double2* HH = nullptr;
    DD_HIP_CHECK(hipMalloc((void**)&HH, sizeof(double2) * h.size()));
    hipError_t e = hipMemcpyAsync(HH, h.data(), sizeof(double2) * h.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);                          // (the staging vector dies with the caller)
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
// The spectra on the HOST, kept for the life of the process (at most eight, 3-8 MB each).  Two reasons: dd_noaa_prepare builds them ahead
// without a device call (its thread runs beside the runtime's first copy and the recording's upload; the call that needs one uploads it,
// 0.3 ms), and a pageable staging vector of a megabyte or more must not be FREED after its copy: the runtime pins such a source in place, and
// returning pinned pages to the system (free -> munmap of a large block) stalled the next device operation of the process by 25-40 ms
// (tools/debug/cold_c4_trace.py: the upload of the second spectrum after the first one's vector had died, in the thread that did it or in
// any other).  Own mutex: prepare fills it without g_sync_mu, the other callers hold g_sync_mu.
static std::mutex g_hilb_host_mu;
static std::map<std::pair<int, int64_t>, std::vector<double2>*> g_hilb_host;
static const std::vector<double2>* hilb_host_find(std::pair<int, int64_t> key) {
    std::lock_guard<std::mutex> lk(g_hilb_host_mu);
    auto it = g_hilb_host.find(key);
    return it == g_hilb_host.end() ? nullptr : it->second;
}
static const std::vector<double2>* hilb_host_keep(std::pair<int, int64_t> key, std::vector<double2>& h) {
    std::lock_guard<std::mutex> lk(g_hilb_host_mu);
    auto it = g_hilb_host.find(key);
    if (it != g_hilb_host.end()) return it->second;               // (another thread was first: its copy stays, ours is dropped unused -- never pinned)
    if (g_hilb_host.size() >= 8) return nullptr;                  // (a process walking through many lengths: the ninth and later are not kept)
    std::vector<double2>* p = new std::vector<double2>();
    p->swap(h);
    g_hilb_host[key] = p;
    return p;
}
static int lg_of(int64_t M) {
    int lgM = 0;
    while (((int64_t)1 << lgM) < M) ++lgM;
    return lgM;
}

// hh[n] = imag(ifft(h))[n] = (2/N) sum_{k=1..m} sin(2 pi k n / N), m = the number of doubled bins of scipy's mask
// ((N-1)/2 for odd N, N/2 - 1 for even N) = (2/N) sin(pi m n/N) sin(pi (m+1) n/N) / sin(pi n/N): a closed form, so no
// length-N (Bluestein) plan is ever built for it; accurate to a few 1e-17 (checked against a long-double sum).
static std::pair<int, int64_t> hilbert_kernel_key(int dev, int64_t n, int64_t M) { return std::make_pair(dev, (n << 6) | lg_of(M)); }      // (length and cyclic length)
static void hilbert_kernel_host(int64_t n, int64_t M, std::vector<double2>& h) {
    const int64_t m = (n & 1) ? (n - 1) / 2 : n / 2 - 1;
    std::vector<double> host((size_t)M, 0.0);                 // buf[j mod M] = hh[j mod N], j in [-(N-1), N-1]
    for (int64_t j = 1; j < n; ++j) {
        const double v = (2.0 / (double)n) * dd_sinpi_frac(m * j, n) * dd_sinpi_frac((m + 1) * j, n) / dd_sinpi_frac(j, n);
        host[(size_t)j] = v;
        host[(size_t)(M - n + j)] = v;
    }
    kernel_spectrum_host(host, M, h);
}
static int hilbert_kernel_spectrum(int64_t n, int64_t M, const double2** out, hipStream_t s) {
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    const auto key = hilbert_kernel_key(dev, n, M);
    auto it = g_hilb.find(key);
    if (it != g_hilb.end()) { *out = it->second; return DD_OK; }
    std::vector<double2> h;
    const std::vector<double2>* hp = hilb_host_find(key);
    if (!hp) { hilbert_kernel_host(n, M, h); hp = hilb_host_keep(key, h); }
    double2* HH = nullptr;
    const int rc = kernel_spectrum_put(hp ? *hp : h, &HH, s);
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
static std::pair<int, int64_t> hilbert_split_key(int dev, int64_t N, int64_t M) { return std::make_pair(dev, -((N << 6) | lg_of(M))); }    // (negative: the split kernel of length N, beside the full ones)
static void hilbert_split_host(int64_t N, int64_t M, std::vector<double2>& h) {
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
    kernel_spectrum_host(host, M, h);
}
static int hilbert_split_spectrum(int64_t N, int64_t M, const double2** out_perm, hipStream_t s) {
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    const auto key = hilbert_split_key(dev, N, M);
    auto it = g_hilb.find(key);
    if (it != g_hilb.end()) { *out_perm = it->second + (M / 2 + 1); return DD_OK; }
    std::vector<double2> h;
    const std::vector<double2>* hp = hilb_host_find(key);
    if (!hp) { hilbert_split_host(N, M, h); hp = hilb_host_keep(key, h); }
    double2* HH = nullptr;
    const int rc = kernel_spectrum_put(hp ? *hp : h, &HH, s);
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
extern "C" int dd_noaa_prepare(int64_t n, int64_t block, int64_t window, void* stream) {
    DD_REQUIRE(n >= 0 && block >= 2 && window >= 0, "arguments");
    (void)stream;
    static const char* tenv = getenv("DD_CRUDE_TRACE");              // tools: host-side time stamps, to stderr
    const auto t0 = std::chrono::steady_clock::now();
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    struct Job { std::pair<int, int64_t> key; int64_t len, M; bool split; };
    std::vector<Job> jobs;
    if (n >= 1) {
        int64_t nfull = 0;
        while ((nfull + 1) * block < n) ++nfull;
        const int64_t lens[2] = {nfull > 0 ? block : 0, n - nfull * block};
        for (int i = 0; i < 2; ++i) {
            bool split = false;
            const int64_t M = hc_block_len(lens[i], &split);
            if (M) jobs.push_back(Job{split ? hilbert_split_key(dev, lens[i], M) : hilbert_kernel_key(dev, lens[i], M), lens[i], M, split});
        }
    }
    if (window >= 4) {                                               // dd_noaa_sync_windows: L2 = window - 1 angles, cyclic length >= 2 L2 + 2
        const int64_t L2 = window - 1;
        int64_t M = 1;
        while (M < 2 * L2 + 2) M <<= 1;
        if (hc_length_ok(M)) jobs.push_back(Job{hilbert_kernel_key(dev, L2, M), L2, M, false});
    }
    int built = 0;
    for (const Job& j : jobs) {
        {
            std::lock_guard<std::mutex> lk(g_sync_mu);
            if (g_hilb.count(j.key)) continue;                       // on the device already
        }
        if (hilb_host_find(j.key)) continue;
        // closed forms and a radix-2 transform, outside every lock, no device call: in a fresh process the runtime is still busy with its
        // first copy (~100 ms, _hip.require_gpu's warm-up thread) and this thread beside it
        std::vector<double2> h;
        if (j.split) hilbert_split_host(j.len, j.M, h); else hilbert_kernel_host(j.len, j.M, h);
        (void)hilb_host_keep(j.key, h);
        ++built;
    }
    if (tenv) fprintf(stderr, "noaa prepare host us (n %ld, window %ld): %d spectra built in %ld\n", (long)n, (long)window, built,
                      (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count());
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
