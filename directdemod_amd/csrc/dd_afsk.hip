// AFSK1200 front end: the reference's pure-Python correlator loop
// (decode_afsk1200.py:126-141) and bit-edge detector (:147-156) as two float64 kernels.
// Audio rate (22 050 S/s): the point is removing a minutes-long Python double loop, not a
// roofline.  Arithmetic order follows the reference (product, then add, sub = 0..bs-1;
// ((mi^2 + mq^2) - si^2) - sq^2) with explicitly rounded operations, so the result is the
// reference's float64 value and sign(binary_filter) -- which feeds integer bit decisions --
// is bit-exact.
#include "dd_common.h"

#pragma clang fp contract(off)      // (also -ffp-contract=off for this unit in __graft_entry__.py)

#define DD_AFSK_MAX_BS 64

struct DDAfskTables {
    double t[4][DD_AFSK_MAX_BS];
};

__global__ void __launch_bounds__(256) k_afsk_binary(const double* __restrict__ sig, int64_t n, int bs,
                                                     const DDAfskTables tb, double* __restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    if (s >= n - bs) { out[s] = 0.0; return; }          // the reference's loop stops at len - buffer_size (:126)
    double mi = 0.0, mq = 0.0, si = 0.0, sq = 0.0;
    for (int k = 0; k < bs; ++k) {
        const double x = sig[s + k];
        mi = __dadd_rn(mi, __dmul_rn(x, tb.t[0][k]));
        mq = __dadd_rn(mq, __dmul_rn(x, tb.t[1][k]));
        si = __dadd_rn(si, __dmul_rn(x, tb.t[2][k]));
        sq = __dadd_rn(sq, __dmul_rn(x, tb.t[3][k]));
    }
    double r = __dadd_rn(__dmul_rn(mi, mi), __dmul_rn(mq, mq));
    r = __dadd_rn(r, -__dmul_rn(si, si));
    r = __dadd_rn(r, -__dmul_rn(sq, sq));
    out[s] = r;
}

// np.correlate(sign(bf), kernel, 'same') / spb with kernel = [-1]*(spb//2) + [1]*(spb - spb//2):
// out[i] = sum_j sign(bf[i + j - spb/2]) * kernel[j] / spb, zero outside (NumPy 'same': n_left = M/2)
__global__ void __launch_bounds__(256) k_afsk_edges(const double* __restrict__ bf, int64_t n, int spb, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int half = spb / 2;
    int acc = 0;
    for (int j = 0; j < spb; ++j) {
        const int64_t q = i + j - half;
        if (q < 0 || q >= n) continue;
        const double v = bf[q];
        const int sg = (v > 0.0) - (v < 0.0);            // np.sign (NaN does not occur: bf is a finite polynomial of the audio)
        acc += (j < half) ? -sg : sg;
    }
    out[i] = (double)acc / (double)spb;
}

extern "C" int dd_afsk_binary_filter_f64(const double* sig, int64_t n, const double* tables_host, int bs,
                                         double* out, void* stream) {
    DD_REQUIRE(n >= 0 && bs >= 1 && bs <= DD_AFSK_MAX_BS, "dd_afsk_binary_filter_f64: need 1 <= buffer_size <= 64");
    DD_REQUIRE(tables_host != nullptr, "dd_afsk_binary_filter_f64: tables");
    if (n == 0) return DD_OK;
    DD_REQUIRE(sig != nullptr && out != nullptr, "dd_afsk_binary_filter_f64: null buffer");
    DDAfskTables tb;
    memset(&tb, 0, sizeof(tb));
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < bs; ++k) tb.t[c][k] = tables_host[(size_t)c * bs + k];
    hipLaunchKernelGGL(k_afsk_binary, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, dd_stream(stream), sig, n, bs, tb, out);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

extern "C" int dd_afsk_edges_f64(const double* binary_filter, int64_t n, int spb, double* out, void* stream) {
    DD_REQUIRE(n >= 0 && spb >= 1 && spb <= 4096, "dd_afsk_edges_f64: samples per baud");
    DD_REQUIRE(n == 0 || n >= spb, "dd_afsk_edges_f64: signal shorter than one baud");
    if (n == 0) return DD_OK;
    DD_REQUIRE(binary_filter != nullptr && out != nullptr, "dd_afsk_edges_f64: null buffer");
    hipLaunchKernelGGL(k_afsk_edges, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, dd_stream(stream), binary_filter, n, spb, out);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// dd_code_warmup (dd_runtime.hip): the runtime loads a translation unit's code object when one of its kernels is first named
int dd_code_touch_afsk(void) {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, (const void*)k_afsk_binary) == hipSuccess ? DD_OK : DD_ERR_HIP;
}
