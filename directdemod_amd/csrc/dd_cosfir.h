// Running-sum (cosine-series window) form of the fused M == 1 FM chain (dd_cosfir.hip).  Internal.
#pragma once
#include "dd_chain_kernels.h"

// 255 taps of the form a0 + a1 cos(2 pi k / 254) (filters.hamming, filters.py:199), M = 1, FM output; complex64 or raw u8 input
int dd_cos1k_supported(const double* taps, int K, int M, int flags);
int dd_cos1k_create(void** st, const double* taps, int K);
void dd_cos1k_destroy(void* st);
// the WHOLE chunk in one launch (stream start, chunk end and the carried state included): P as dd_fused_launch fills it
int dd_cos1k_launch(void* st, const DDChainParams& P, hipStream_t stream);
