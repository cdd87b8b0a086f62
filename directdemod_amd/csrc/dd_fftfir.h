// f32 overlap-save FFT form of the fused M == 1 chain (dd_fftfir.hip).  Internal.
#pragma once
#include "dd_chain_kernels.h"

int dd_fft_supported(int K, int M, int flags);
int dd_fft_create(void** st, const double* taps, int K);
void dd_fft_destroy(void* st);
// FM angles of the FIR outputs [p_a, p_b) of the chunk described by P (interior run: every input sample the outputs
// depend on, and the 256 before p_a, lie inside the chunk; p_a - s >= 0)
int dd_fft_launch(void* st, const DDChainParams& P, int64_t p_a, int64_t p_b, hipStream_t stream);
