// Wave-autonomous decimating chain kernel (dd_decimw.hip).  Internal.
#pragma once
#include "dd_chain_kernels.h"

// even M in [8, 64], 2 <= K <= 256, complex64 (8-byte aligned) or raw u8 (2-byte aligned) input, FM or complex64 output
int dd_decimw_supported(int K, int M, int flags, const void* in);
// the WHOLE chunk in one launch (stream start, chunk end and the carried state included): P as dd_fused_launch fills it;
// taps_g0 = the reversed taps g[j] = h[K-1-j] on the device, at least one zero in front of g[0] and 22 behind g[K-1]
int dd_decimw_launch(const DDChainParams& P, const float* taps_g0, hipStream_t stream);
