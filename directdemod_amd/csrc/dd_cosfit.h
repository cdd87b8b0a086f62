// Cosine-series windows (filters.py:101-226): taps b[k] = sum_q a_q cos(2 pi q k / (K-1)), q = 0 .. Q <= 3 -- the fit and its cache.
// Shared by the zero-phase kernels (dd_filtfilt_kernels.h) and the running-sum M = 1 chain kernel (dd_cosfir.hip).  Host only.
#pragma once
#include <math.h>
#include <string.h>
#include <vector>
struct DDCosFit {
    int Q;              // highest harmonic with a non-zero coefficient
    double a[4];
};
// taps == sum_q a_q cos(2 pi q k / (K-1)) to 1e-13 of the largest tap?  (least squares over q = 0..3, long double)
static inline bool dd_cos_fit(const double* taps, int K, DDCosFit* f) {
    if (K < 64) return false;
    const int NQ = 4;
    long double G[NQ][NQ + 1];
    const long double w = 2.0L * 3.14159265358979323846264338327950288L / (long double)(K - 1);
    for (int p = 0; p < NQ; ++p) {
        for (int q = 0; q < NQ; ++q) {
            long double acc = 0.0L;
            for (int k = 0; k < K; ++k) acc += cosl(w * (long double)((long long)p * k % (K - 1))) * cosl(w * (long double)((long long)q * k % (K - 1)));
            G[p][q] = acc;
        }
        long double acc = 0.0L;
        for (int k = 0; k < K; ++k) acc += cosl(w * (long double)((long long)p * k % (K - 1))) * (long double)taps[k];
        G[p][NQ] = acc;
    }
    for (int c = 0; c < NQ; ++c) {                       // Gauss-Jordan with partial pivoting
        int piv = c;
        for (int r = c + 1; r < NQ; ++r) if (fabsl(G[r][c]) > fabsl(G[piv][c])) piv = r;
        if (fabsl(G[piv][c]) < 1e-12L) return false;
        for (int j = 0; j <= NQ; ++j) { const long double tmp = G[c][j]; G[c][j] = G[piv][j]; G[piv][j] = tmp; }
        for (int r = 0; r < NQ; ++r) {
            if (r == c) continue;
            const long double m = G[r][c] / G[c][c];
            for (int j = c; j <= NQ; ++j) G[r][j] -= m * G[c][j];
        }
    }
    long double a[NQ], peak = 0.0L, res = 0.0L;
    for (int q = 0; q < NQ; ++q) a[q] = G[q][NQ] / G[q][q];
    for (int k = 0; k < K; ++k) {
        long double v = 0.0L;
        for (int q = 0; q < NQ; ++q) v += a[q] * cosl(w * (long double)((long long)q * k % (K - 1)));
        const long double d = fabsl(v - (long double)taps[k]);
        if (d > res) res = d;
        if (fabsl((long double)taps[k]) > peak) peak = fabsl((long double)taps[k]);
    }
    if (!(res <= 1e-13L * peak)) return false;
    f->Q = 0;
    for (int q = 0; q < NQ; ++q) {
        f->a[q] = (double)a[q];
        if (fabsl(a[q]) > 1e-14L * peak) f->Q = q;
    }
    for (int q = f->Q + 1; q < NQ; ++q) f->a[q] = 0.0;
    return f->Q >= 1;                                    // (a rolling average alone is not worth a kernel of its own)
}
// the fit of a tap set is looked up before it is computed (a few thousand cosl calls: ~1.5 ms on the host, as much as the
// whole accurate-sync batch it was meant to speed up)
#include <mutex>
struct DDCosFitEntry { std::vector<double> taps; bool ok; DDCosFit fit; };
static inline bool dd_cos_fit_cached(const double* taps, int K, DDCosFit* f) {
    static std::mutex mu;
    static std::vector<DDCosFitEntry> cache;
    std::lock_guard<std::mutex> lk(mu);
    for (const DDCosFitEntry& e : cache)
        if ((int)e.taps.size() == K && memcmp(e.taps.data(), taps, sizeof(double) * K) == 0) {
            if (e.ok) *f = e.fit;
            return e.ok;
        }
    DDCosFitEntry e;
    e.taps.assign(taps, taps + K);
    e.ok = dd_cos_fit(taps, K, &e.fit);
    if (cache.size() >= 16) cache.erase(cache.begin());
    cache.push_back(e);
    if (e.ok) *f = e.fit;
    return e.ok;
}
