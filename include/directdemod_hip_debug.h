/*
 * directdemod_hip_debug.h -- diagnostic entry points of libdirectdemod_hip.so (round 6: moved out of the public header
 * directdemod_hip.h).  Used by the test suite and the tools under tools/ only: forcing one of several kernels
 * that compute the same thing (A/B runs, parity of every kernel against the oracle), filling LDS with a pattern, withholding an
 * in-launch hand-over, and the host arithmetic of launch plans (checked without a GPU).  No reference counterpart; not part of
 * the drop-in boundary.
 */
#ifndef DIRECTDEMOD_HIP_DEBUG_H
#define DIRECTDEMOD_HIP_DEBUG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* diagnostic: fill the LDS of every compute unit with `pattern` (LDS is not cleared between workgroups).  The parity
 * suite runs the chain kernels after a NaN fill and after a zero fill and requires bit-identical outputs. */
int  dd_debug_fill_lds(uint32_t pattern, void* stream);
/* diagnostic: force one of the M = 1 chain kernels for every later launch of this process -- "ab" (k_chain_mfma_ab), "fft1k" (k_chain_fft1k, wherever it applies), "cos1k" (k_chain_cos1k where it applies -- 255 taps of a
 * two-term cosine series, FM output -- and the choice by tap class elsewhere), "decimp" (M > 1: the tile kernels of rounds 1-4 instead of
 * k_chain_decim_b / k_chain_decim_w), "auto" / NULL (by tap class, the default).  The parity,
 * full-size and determinism suites run both FM kernels this way; the environment variable DD_MFMA_KERNEL seeds the choice
 * once per process.  No reference counterpart. */
int  dd_debug_select_kernel(const char* name);
/* diagnostic: the NEXT chunk-list launch withholds the hand-over flag of chunk `withhold_chunk` (>= 0; -1: none) and every
 * later one bounds its in-launch waits by 2^spin_log2 polls (0: the default, 2^19 = about a tenth of a second per wait).  Lets a test see DD_ERR_TIMEOUT
 * instead of silently wrong samples.  No reference counterpart (the reference's chunk loop is sequential, decode_noaa.py:619-624). */
int  dd_debug_seam(int withhold_chunk, int spin_log2);
/* diagnostics that need no GPU (host arithmetic of two launch paths, checked by the CPU test suite):
 * dd_debug_fft1k_plan -- the block grid and the block -> wave map of a k_chain_fft1k launch over a chunk of L samples (s = 1: a
 *   stream start, no angle for the first output; out_align_elems: how many elements `out` sits behind a 64-byte line; ncu compute
 *   units; rounds: 0 = default).  out[0..6] = base, nblk, grid, nwaves, K, b, 32, then r0[32], r1[32], wstart[32].
 * dd_debug_cos_fit -- 1 when the taps are a cosine series of at most four terms for which the zero-phase filter of the
 *   accurate-sync windows takes its prefix-sum form (a[0..3], *Q filled), else 0.
 * dd_debug_sync_envelope -- the envelope stage of dd_noaa_sync_windows alone: X_dev c64 [nwin][L] (device) -> env_dev f64
 *   [nwin][L - 1] = abs(hilbert(angle(X[n+1] conj X[n]))) (decode_noaa.py:852 -> demod_am.py:29).  route 0: the three-launch
 *   float64 transform of csrc/dd_hconv_kernels.h (512 x 512 for 65 536 < L <= 131 072, 512 x 256 for 32 768 < L <= 65 536), route 1: the FFT
 *   library's padded real transforms. */
int  dd_debug_fft1k_plan(int64_t L, int s, int out_align_elems, int ncu, int rounds, int* out);
/* dd_debug_cos1k_plan -- the row grid of a k_chain_cos1k launch (same arguments): out[0..3] = base (first sample of row 0; row q covers
 *   samples [base + 1024 q, +1024)), rows, workgroups, waves (wave w takes rows [rows w / waves, rows (w + 1) / waves)). */
int  dd_debug_cos1k_plan(int64_t L, int s, int out_align_elems, int ncu, int* out);
/* dd_debug_decimw_plan -- the row grid of a k_chain_decim_w / k_chain_decim_b launch over a chunk that starts at absolute sample abs0 with
 *   decimation phase off and keeps Ld samples: out[0..11] = R0 (absolute index of the first row: row R is the block of 2048 samples
 *   [2048 R, 2048 (R + 1))), rows, phi = (abs0 + off) mod M, samples kept in LDS in front of a block, start shift (0 / 1), taps per lane
 *   rounded up to 16 (window form), waves per CU, rows per run, form (1: block sums, k_chain_decim_b, K <= 8 M; 0: one window per lane,
 *   k_chain_decim_w), partial sums per output ceil(K / M), first block sample of the second accumulator set, samples of the LDS image.
 *   DD_ERR_UNSUPPORTED when the kernels do not take (K, M). */
int  dd_debug_decimw_plan(int64_t abs0, int64_t Ld, int K, int M, int off, int ncu, int64_t* out);
int  dd_debug_cos_fit(const double* taps_host, int K, double* a_out, int* Q_out);
int  dd_debug_sync_envelope(const void* X_dev, int64_t L, int nwin, int route, double* env_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIRECTDEMOD_HIP_DEBUG_H */
