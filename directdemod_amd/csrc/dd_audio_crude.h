// getCrudeSync's audio-rate tail in one host call (decode_noaa.py:769-806): dd_noaa_crude_tail
// One of the five parts of dd_audio.hip (round 6: the 2600-line unit split along its entry-point families; still ONE translation unit --
// the parts share the plan cache, the float64 transform and the scratch buffers of dd_audio.hip and are included there, in this order).
// Internal; not a stand-alone header.
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
