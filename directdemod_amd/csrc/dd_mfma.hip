// placeholder: MFMA path hooks (filled in by the f16-split Toeplitz kernel)
#include "dd_chain_kernels.h"
int dd_mfma_supported(int K, int M, int flags) { (void)K; (void)M; (void)flags; return 0; }
int dd_mfma_create(void** st, const double* taps, int K) { (void)st; (void)taps; (void)K; return DD_ERR_UNSUPPORTED; }
void dd_mfma_destroy(void* st) { (void)st; }
int dd_mfma_launch(void* st, const DDChainParams& P, hipStream_t s) { (void)st; (void)P; (void)s; return DD_ERR_UNSUPPORTED; }
