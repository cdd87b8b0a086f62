// Wave-autonomous decimating chain kernel (dd_decimw.hip).  Internal.
#pragma once
#include "dd_chain_kernels.h"

// even M in [8, 64], 2 <= K <= 256, complex64 (8-byte aligned) or raw u8 (2-byte aligned) input, FM or complex64 output
int dd_decimw_supported(int K, int M, int flags, const void* in);
// the WHOLE chunk in one launch (stream start, chunk end and the carried state included): P as dd_fused_launch fills it;
// taps_g0 = the reversed taps g[j] = h[K-1-j] on the device, at least one zero in front of g[0] and 22 behind g[K-1]
int dd_decimw_launch(const DDChainParams& P, const float* taps_g0, hipStream_t stream);
// the launch geometry (host arithmetic, no GPU): out[0..7] = {R0, nrows, phi, HP, e, K16, waves per CU, run_rows}
int dd_decimw_plan(int64_t abs0, int64_t L, int64_t Ld, int K, int M, int off, int u8, uintptr_t in_addr, int ncu, int64_t* out);
