// Wave-autonomous decimating chain kernel (dd_decimw.hip).  Internal.
#pragma once
#include "dd_common.h"

struct DDChainParams;
// kept with a filter: the taps as a PAD launch meets them (zeros over the gaps of the padded LDS image), for one (M, tap shift) at a time
#define DD_DECIMW_TAPS_CAP 640
struct DDDecimWTaps {
    float* dev;
    int key;
    float host[DD_DECIMW_TAPS_CAP];      // the copy's source: lives as long as the filter (the copy is asynchronous)
};
// even M in [8, 64], 2 <= K <= 256, complex64 (8-byte aligned) or raw u8 (2-byte aligned) input, FM or complex64 output
int dd_decimw_supported(int K, int M, int flags, const void* in);
// the WHOLE chunk in one launch (stream start, chunk end and the carried state included): P as dd_fused_launch fills it;
// taps_g0 = the reversed taps g[j] = h[K-1-j] on the device, at least one zero in front of g[0] and 38 behind g[K-1]; taps_host = h[0 .. K-1]
// *kernel_id (may be null): DD_KERNEL_DECIM_BLOCKS (k_chain_decim_b, K <= 8 M) or DD_KERNEL_DECIM_WAVE (k_chain_decim_w)
int dd_decimw_launch(const DDChainParams& P, const float* taps_g0, const double* taps_host, DDDecimWTaps* cache, hipStream_t stream, int* kernel_id = nullptr);
