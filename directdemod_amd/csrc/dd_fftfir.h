// f32 overlap-save FFT form of the fused M == 1 FM chain (dd_fftfir.hip).  Internal.
#pragma once
#include "dd_chain_kernels.h"

// taps up to 256, M = 1, FM output
int dd_fft1k_supported(int K, int M, int flags);
int dd_fft_create(void** st, const double* taps, int K);
void dd_fft_destroy(void* st);
// one wave per 1024-point block, the WHOLE chunk in one launch (stream start, chunk end and the carried state included):
// P as dd_fused_launch fills it
int dd_fft1k_launch(void* st, const DDChainParams& P, hipStream_t stream);
