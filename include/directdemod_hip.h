/*
 * directdemod_hip.h -- C-ABI of the MI355X (gfx950) IQ-sample DSP hot path that
 * replaces DirectDemod's NumPy/SciPy per-sample path.
 *
 * The reference is pure Python (no FFI of its own); every entry point below
 * replaces one SciPy/NumPy call site of the reference, cited as file:line
 * relative to the reference tree.  The Python classes in directdemod_amd/
 * (commSignal, filters.*, demod_fm, demod_am, chunker -- same names/signatures
 * as the reference's) bind these symbols with ctypes; INTEGRATION.md shows the
 * stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; all buffers are DEVICE pointers unless the parameter
 *     name ends in _host;  complex64 = interleaved float {re,im}.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *     compute entry points are asynchronous on that stream; carried state
 *     lives in device memory inside the handles, so chunk loops need no
 *     host<->device synchronisation.
 *   - return value: 0 = DD_OK, <0 = error (never throws); dd_last_error()
 *     returns a thread-local message.
 *   - one caller thread per handle (the reference is single-threaded and its
 *     state carry makes chunk order significant).
 */
#ifndef DIRECTDEMOD_HIP_H
#define DIRECTDEMOD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DD_OK               0
#define DD_ERR_INVALID     -1   /* bad argument (maps to ValueError/TypeError) */
#define DD_ERR_HIP         -2   /* HIP runtime error */
#define DD_ERR_NOMEM       -3
#define DD_ERR_UNSUPPORTED -4
#define DD_ERR_NODEVICE    -5   /* no MI355X visible: the product path fails loudly */
#define DD_ERR_TIMEOUT     -6   /* a chunk-list launch (dd_*_process_chunks) gave up waiting for the carried state of the previous
                                 * chunk inside the launch: the outputs of THAT call are invalid.  Reported by the next chunk-list
                                 * call on the same filter or by dd_stream_sync, whichever comes first */

/* FIR history initialisation (filters.py:44-48, 64-70) */
#define DD_HIST_ZEROS 0     /* plain lfilter / lfiltic with no past inputs */
#define DD_HIST_ONES  1     /* lfilter_zi(b,[1]) unscaled == delay line of 1.0+0j (quirk Q1) */
#define DD_HIST_GIVEN 2     /* explicit past inputs, oldest first */

/* dd_chain_create flags */
#define DD_CHAIN_NCO        1   /* apply commSignal.offsetFreq        (comm.py:63-78)   */
#define DD_CHAIN_FM         2   /* apply demod_fm.demod on the output (demod_fm.py:29-51) */
#define DD_CHAIN_U8_INPUT   4   /* input is interleaved u8 I,Q; source.read fused (source.py:117-118) */
#define DD_CHAIN_FORCE_DIRECT 8 /* disable the MFMA fast path (f32 direct form only) */
#define DD_CHAIN_TIGHT       16 /* M = 1 chains: keep the running-sum kernel (k_chain_cos1k, filters.hamming(255)) off -- its FIR output is
                                 * a0 R + a1 C with R, C the rectangular-window sums, which cancel to the window's -50 dB in the stop band: FM
                                 * angles of a signal the filter rejects are good to 1e-3 rad there, against 3e-4 for the transform kernel
                                 * this flag selects instead (1.4 x the time).  Also a flag of dd_fused_process(_chunks);
                                 * Python: filters.filter.tight = True.  No reference counterpart (float64 there, filters.py:64-70) */

/* ---- runtime ---------------------------------------------------------------- */
const char* dd_last_error(void);
const char* dd_version(void);
int  dd_device_count(int* count);
int  dd_set_device(int device);
/* the calling thread's current device (HIP keeps one per host thread, default 0): a helper thread started on behalf of a caller
 * takes the caller's device with dd_set_device(dd_get_device()) before it touches the GPU */
int  dd_get_device(int* device);
int  dd_device_name(char* buf, int buflen);
int  dd_malloc(void** dptr, size_t bytes);
int  dd_free(void* dptr);
int  dd_memset(void* dptr, int value, size_t bytes, void* stream);
int  dd_host_alloc_pinned(void** hptr, size_t bytes);
int  dd_host_free_pinned(void* hptr);
/* pin / unpin a range of caller-owned host memory in place (hipHostRegister): lets source.IQwav's memmap or an
 * in-memory recording feed hipMemcpyAsync directly (BASELINE north_star: "pinned-host ring buffers with
 * hipMemcpyAsync overlapped on a side stream").  DD_ERR_UNSUPPORTED when the range cannot be pinned. */
int  dd_host_register(void* hptr, size_t bytes);
int  dd_host_unregister(void* hptr);
/* (diagnostic entry points -- dd_debug_*: kernel selection for A/B runs, LDS fills, launch-plan arithmetic for the CPU test suite -- are
 * declared in directdemod_hip_debug.h; none of them has a reference counterpart, none is needed to use the library) */
int  dd_memcpy_h2d(void* dst, const void* src_host, size_t bytes, void* stream);
int  dd_memcpy_d2h(void* dst_host, const void* src, size_t bytes, void* stream);
int  dd_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
/* a 4 KB and a 1 MB synchronous host-to-device copy on the calling thread's device (the process's first copy pays ~90 ms of runtime set-up
 * whatever its size, its first large one another ~7 ms): _hip.py makes them on a helper thread when the GPU is first touched.  No reference counterpart. */
int  dd_copy_warmup(void);
/* names one kernel of every translation unit of the library, so that the runtime loads the code objects now (1-4 ms each) and not inside the
 * caller's first launches: _hip.py calls it on the same helper thread, after dd_copy_warmup.  No reference counterpart. */
int  dd_code_warmup(void);
int  dd_stream_create(void** stream);
int  dd_stream_destroy(void* stream);
int  dd_stream_sync(void* stream);
int  dd_event_create(void** ev);
int  dd_event_destroy(void* ev);
int  dd_event_record(void* ev, void* stream);
int  dd_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);   /* syncs on ev_stop */
int  dd_event_sync(void* ev);
int  dd_stream_wait_event(void* stream, void* ev);   /* later work on `stream` waits for `ev` (no host sync) */

/* ---- S1: source.IQwav/IQdat/IQwavAlt.read (source.py:117-118,209-210,303-304) -- */
/* interleaved uint8 I,Q  ->  complex64 minus (127.5+127.5j) */
int dd_u8iq_to_c64(const uint8_t* in_iq, float* out_c64, int64_t n, void* stream);

/* ---- N1: commSignal.offsetFreq (comm.py:63-78) ------------------------------ */
/* out[i] = in[i] * exp(-j 2 pi f (start_index+i)/fs).  `cycles_q64` is
 * frac(f/fs) * 2^64 (two's complement wrap), computed exactly on the host; the
 * phase is exact modular integer arithmetic for any start_index.
 * In place allowed (in == out). */
int dd_nco_c64(const float* in_c64, float* out_c64, int64_t n, uint64_t cycles_q64,
               int64_t start_index, void* stream);
/* the same with a per-sample frequency array (device, float64 Hz): comm.py:77 with an array
 * freqOffset (Doppler correction, decode_funcube.py:228); phase formed and reduced in float64 */
int dd_nco_c64_freqs(const float* in_c64, float* out_c64, int64_t n, const double* freqs_hz, double samp_rate,
                     int64_t start_index, void* stream);

/* ---- F1/F3: filters.filter.applyOn, FIR (a=[1]) (filters.py:21-75) ------------ */
typedef struct dd_fir dd_fir;
int dd_fir_create(dd_fir** h, const double* taps, int ntaps);
int dd_fir_destroy(dd_fir* h);
/* (re)initialise the carried history: DD_HIST_ZEROS / DD_HIST_ONES / DD_HIST_GIVEN
 * (hist_host: ntaps-1 complex64 (or float for the real path), oldest first). */
int dd_fir_reset(dd_fir* h, int mode, const float* hist_host, void* stream);
/* history of the float64 real path (dd_fir_f64) given as ntaps-1 doubles, oldest first: lfiltic(b,[1],x,initOut)
 * (filters.py:47-48,66-67) with initOut values that are not float32 numbers.  Call after dd_fir_reset. */
int dd_fir_reset_hist_f64(dd_fir* h, const double* hist_host, void* stream);
/* causal FIR, complex64 in -> complex64 out, same length, history carried
 * (storeState) or not (carry=0: history read but left untouched). */
int dd_fir_c64(dd_fir* h, const float* in_c64, float* out_c64, int64_t n, int carry, void* stream);
/* real float64 path used at audio rate (NOAA tail keeps float64) */
int dd_fir_f64(dd_fir* h, const double* in, double* out, int64_t n, int carry, void* stream);

/* ---- F2: filters.filter zeroPhase -> scipy.signal.filtfilt (filters.py:72-73) -- */
/* odd extension 3*ntaps, forward/backward with zi*x0; stateless.
 * is_complex: 0 = float64 real, 1 = complex128 interleaved doubles.  n > 3*ntaps. */
int dd_filtfilt_f64(const double* taps_host, int ntaps, const double* in, double* out,
                    int64_t n, int is_complex, void* stream);
/* complex64 in/out, float32 arithmetic (full-rate IQ windows, decode_noaa.py:852) */
int dd_filtfilt_c64(const double* taps_host, int ntaps, const float* in_c64, float* out_c64,
                    int64_t n, void* stream);

/* ---- F4: filters.butter -> scipy.signal.lfilter with a != [1] (filters.py:232-273) ---
 * Transposed direct form II recurrence, float64, state carried on the device.
 * Short inputs: one lane per real component.  From 4096 samples up: block-parallel
 * (block end states from zero, two- or three-level scan of the block start states with
 * the block map A^256 held in double-double, blocks re-run from their true start states;
 * asynchronous on `stream`, intermediates in a scratch buffer owned by the handle), which
 * is what full-rate IQ through a butter takes (decode_funcube.py:160,230; SURVEY.md
 * 8f-3).  b, a: `n` coefficients each
 * (pad the shorter with zeros), a[0] != 0.  zi_host: n-1 initial state values
 * (scipy.signal.lfilter_zi(b, a), unscaled like filters.py:45) or NULL for zeros. */
typedef struct dd_iir dd_iir;
int dd_iir_create(dd_iir** h, const double* b, const double* a, int n, const double* zi_host);
int dd_iir_destroy(dd_iir* h);
/* is_complex: 0 = float64, 1 = complex128 (interleaved); carry: keep the final state */
int dd_iir_f64(dd_iir* h, const double* in, double* out, int64_t n, int is_complex, int carry, void* stream);
/* the same on complex64 input, complex128 output (what scipy.signal.lfilter returns for a complex64 signal and float64 coefficients): the IQ
 * stream as source.py hands it over, low-passed before any decimation (decode_funcube.py:160, decode_meteorm2.py:157 -> filters.py:75).  From 4096
 * samples up the block-parallel passes read the complex64 samples as they are (8 + 8 + 16 bytes per sample move instead of 72 through a widened
 * copy); shorter or unaligned inputs are widened into `out` and filtered there.  in_c64 != out_c128. */
int dd_iir_c64(dd_iir* h, const void* in_c64, double* out_c128, int64_t n, int carry, void* stream);
/* scipy.signal.filtfilt(b, a, x): odd extension 3*n, zi scaled by the first sample of each pass */
int dd_iir_filtfilt_f64(dd_iir* h, const double* in, double* out, int64_t n, int is_complex, void* stream);

/* ---- R1: commSignal.bwLim non-strict (comm.py:118-130) ---------------------- */
/* out[i] = in[offset + i*m]; elem_bytes in {4,8,16}; n_out = ceil((n-offset)/m) */
int dd_decimate(const void* in, void* out, int64_t n, int m, int offset, int elem_bytes,
                int64_t* n_out, void* stream);

/* ---- D1: demod_fm.demod (demod_fm.py:29-51) --------------------------------- */
typedef struct dd_fm dd_fm;
int dd_fm_create(dd_fm** h);
int dd_fm_destroy(dd_fm* h);
int dd_fm_reset(dd_fm* h);
/* angle(x[i]*conj(x[i-1])); first call with carry writes n-1 outputs, later calls n
 * (quirk Q3); carry=0 always n-1.  *n_out receives the count written. */
int dd_fm_discrim_c64(dd_fm* h, const float* in_c64, float* out, int64_t n, int carry,
                      int64_t* n_out, void* stream);

/* ---- fused hot path: offsetFreq -> filter(FIR) -> bwLim(M) -> demod_fm -------
 * (decode_noaa.py:623, decode_fm.py:64-68, decode_afsk1200.py:79-94,
 *  tutorial/3_chunking.py:24-38).  One kernel per chunk; all carried state (NCO
 *  sample index, FIR history, decimation phase, last FM sample) is derived from
 *  the absolute sample index or kept on the device. */
/* Object-model form: state lives where the reference keeps it -- FIR history in the
 * filter object (dd_fir), last FM sample in the demodulator (dd_fm, NULL = no FM,
 * output is complex64), NCO sample index and decimation phase are the chunker
 * variables "freqoffset"/"bwlim" passed by value (chunker.py:54-84, comm.py:73-76,
 * 121-125).  nco: 0/1.  offset: chunk-relative index of the first kept sample.
 * flags: DD_CHAIN_U8_INPUT | DD_CHAIN_FORCE_DIRECT.  carry: storeState of the filter. */
int dd_fused_process(dd_fir* fir, dd_fm* fm, const void* in, void* out, int64_t n,
                     int nco, uint64_t cycles_q64, int64_t start_index, int decim, int offset,
                     int flags, int carry, int64_t* n_out, void* stream);

/* A chunk LIST through the object-model form: `in` points at the first chunk's first sample, chunk i is
 * bounds_host[i+1] - bounds_host[i] samples long (nchunks + 1 ascending offsets), start_index / offset are the chunker
 * variables of the FIRST chunk (the later chunks' follow by the reference's carry rules, comm.py:75-76, 123-125), the
 * filter and the demodulator carry their state (storeState) exactly as over nchunks dd_fused_process calls: outputs
 * concatenated at `out`, bit-identical to that loop, counts in n_out_host; ONE launch for a decimating chain.  The drop-in
 * classes use it when a chunk loop's chunks are consecutive views of one device-resident recording (comm.py). */
int dd_fused_process_chunks(dd_fir* fir, dd_fm* fm, const void* in, void* out, const int64_t* bounds_host, int nchunks,
                            int nco, uint64_t cycles_q64, int64_t start_index, int decim, int offset, int flags,
                            int64_t* n_out_host, void* stream);

/* Convenience handle bundling one filter, one FM demodulator and the chunker
 * variables of one stream (used by bench.py and the sharded multi-GPU driver). */
typedef struct dd_chain dd_chain;
int dd_chain_create(dd_chain** h, const double* taps, int ntaps, uint64_t cycles_q64,
                    int decim, int flags);
int dd_chain_destroy(dd_chain* h);
/* start a new stream (new chunker object): abs index 0, history ones, no FM state */
int dd_chain_reset(dd_chain* h, void* stream);
/* Start a stream position without touching the device: the next chunk is taken to begin at
 * absolute sample `abs_index` with an all-zero FIR history and no previous FM sample (index 0:
 * the stream start, history of ones like dd_chain_reset).  A shard can then be run in ONE call
 * over [start - lead, stop) with lead >= ntaps-1+decim and the lead-in's outputs discarded --
 * the same outputs as dd_chain_prime + dd_chain_process (the halo is re-filtered
 * locally, no exchange), one launch instead of two. */
int dd_chain_seek(dd_chain* h, int64_t abs_index, void* stream);

/* multi-GPU / shard start: establish state as if samples [0, abs_index) had been
 * processed, from the `n_halo` raw input samples that precede abs_index
 * (n_halo >= ntaps-1+decim; fewer only if abs_index == n_halo i.e. stream start). */
int dd_chain_prime(dd_chain* h, const void* halo_in, int64_t n_halo, int64_t abs_index, void* stream);
/* number of outputs the next dd_chain_process(n) will write (pure host arithmetic) */
int64_t dd_chain_out_count(const dd_chain* h, int64_t n);
/* process one chunk of n input samples (complex64, or u8 pairs with DD_CHAIN_U8_INPUT).
 * out: float32 radians when DD_CHAIN_FM, else complex64. */
int dd_chain_process(dd_chain* h, const void* in, void* out, int64_t n, int64_t* n_out, void* stream);
/* Several chunks in one call: chunk i is the samples [bounds_host[i], bounds_host[i+1]) of `in` (nchunks + 1 ascending
 * offsets).  Same outputs, bit for bit, as nchunks dd_chain_process calls in that order -- concatenated at `out`, counts
 * per chunk in n_out_host (may be NULL) -- and the same carried state afterwards; for a decimating chain (the reference's
 * chunk loops: decode_fm.py:54-70 with 2^22-sample chunks, decode_noaa.py:614-624) ONE kernel launch instead of one per
 * chunk (persistent workgroups walk every chunk's tiles, the state a chunk hands to the next travels through device
 * memory inside the launch).  bench.py: C3's sixteen chunks 0.23 ms as a loop, 0.12 ms here. */
int dd_chain_process_chunks(dd_chain* h, const void* in, void* out, const int64_t* bounds_host, int nchunks,
                            int64_t* n_out_host, void* stream);
/* which kernel the chain dispatches to: 0 = f32 direct form, 1 = f16-split MFMA Toeplitz */
int dd_chain_path(const dd_chain* h);
/* the kernel the last dd_chain_process call launched (one launch per call): lets tests and benchmarks assert
 * that the intended kernel -- not a slower sibling with the same results -- produced the output */
#define DD_KERNEL_NONE 0             /* nothing launched yet (or a chunk without a kept sample) */
#define DD_KERNEL_DENSE_F32 1        /* k_chain_dense: M = 1, f32 direct form */
#define DD_KERNEL_DECIM_TILES 2      /* k_chain_decim: M > 1, one workgroup per tile (short or unaligned chunks) */
#define DD_KERNEL_DECIM_PERSISTENT 3 /* k_chain_decim_p: M > 1, persistent interior run + edge tiles in the same launch */
#define DD_KERNEL_MFMA_WS 4          /* k_chain_mfma_ws: M = 1, wave-specialised MFMA kernel (round 1); not in the product library since round 6 (-DDD_WITH_WS builds only) */
#define DD_KERNEL_MFMA_TILES 5       /* k_chain_mfma_edge: M = 1, MFMA, one workgroup per tile */
#define DD_KERNEL_MFMA_AB 6          /* k_chain_mfma_ab: M = 1, FM or complex64 output, two alternating matrix-wave sets + edge tiles in the same launch */
#define DD_KERNEL_FFT_OS 7           /* k_chain_fft1k: M = 1, FM output, 162..256 taps: f32 overlap-save FFT convolution, one wave per 1024-point block, NCO commuted into the tap spectrum, whole chunk in one launch */
#define DD_KERNEL_DECIM_MULTI 8      /* k_chain_decim_multi: M > 1, every chunk of a dd_chain_process_chunks call in one launch */
#define DD_KERNEL_COS_RS 9           /* k_chain_cos1k: M = 1, FM or complex64 output, 255 taps a0 + a1 cos(2 pi k / 254) (filters.hamming): the FIR as three running sums, one wave per run of 1024-sample rows, whole chunk in one launch */
#define DD_KERNEL_DECIM_WAVE 10      /* k_chain_decim_w: M > 1 (even, 8..64), up to 256 taps, FM or complex64 output, complex64 or raw u8 input: one wave per row of 64 kept outputs on the absolute decimation grid, no barrier; a chunk list is ONE launch of it.  Since round 6 only where K > 8 M */
#define DD_KERNEL_DECIM_BLOCKS 11    /* k_chain_decim_b (round 6): the same rows and grids for K <= 8 M (every K for M >= 32: the reference's /34 and /50) -- every staged sample is read once: a lane forms the ceil(K / M) block sums of the M samples that end at its kept sample on the matrix pipe (v_mfma_f32_4x4x1, exact f32), the sums travel up the lanes */
int dd_chain_last_kernel(const dd_chain* h);
int dd_fir_last_kernel(const dd_fir* h);       /* the same for a filter object driven through dd_fused_process (the drop-in classes) */
long long dd_fir_launch_count(const dd_fir* h); /* fused kernel launches through this filter since it was created or last reset (a chunk list in one launch counts once): tests tell "one launch" from "a launch per chunk" by it */
/* HIP-event timing of the last dd_chain_process main kernel is up to the caller. */

/* ---- R2: commSignal.bwLim strict -> scipy.signal.resample (comm.py:110-116) --- */
/* Fourier-domain resample of one chunk, float64 real: n -> num samples.  Asynchronous on `stream`; the intermediates
 * live in a grow-only buffer kept by the library (no allocation, no synchronisation in the steady state). */
int dd_resample_fft_f64(const double* in, double* out, int64_t n, int64_t num, void* stream);
/* The same per-chunk resample for a whole chunk list in one call (a chunk loop ends every chunk in bwLim(strict):
 * decode_fm.py:54-70): chunk i is the n_host[i] samples at in + in_off_host[i] (float32 when in_is_f32 -- the FM output of
 * dd_chain_process_chunks -- else float64) and becomes num_host[i] float64 samples at out + out_off_host[i]; offsets in
 * elements.  Chunks of equal (n, num) share batched hipFFT plans: two transforms per group instead of two per chunk. */
int dd_resample_fft_chunks(const void* in, int in_is_f32, const int64_t* in_off_host, const int64_t* n_host, double* out,
                           const int64_t* out_off_host, const int64_t* num_host, int count, void* stream);

/* ---- polyphase rational resampler (BASELINE north_star, config 3 "polyphase resample to 11.025 kS/s").
 * The reference has NO counterpart (its only resampler is the FFT one above, comm.py:110-116): a build-defined
 * stage following SciPy's published scipy.signal.resample_poly (upfirdn of the front-padded, up-scaled low-pass
 * `taps`; n_pre_remove outputs dropped), float64 real, in STREAM form: state (the last ceil(ntaps/up) inputs, the
 * input and output positions) is carried from call to call, a call emits the outputs whose inputs have all
 * arrived, flush = 1 emits the tail (inputs past the end are zeros) -- concatenated, the outputs equal
 * resample_poly of the concatenated input.  One caller thread per handle. */
typedef struct dd_rpoly dd_rpoly;
int dd_rpoly_create(dd_rpoly** h, const double* taps, int ntaps, int up, int down, int64_t n_pre_remove);
int dd_rpoly_destroy(dd_rpoly* h);
int dd_rpoly_reset(dd_rpoly* h);
int64_t dd_rpoly_out_count(const dd_rpoly* h, int64_t n, int flush);
int dd_rpoly_process(dd_rpoly* h, const double* in, int64_t n, int flush, double* out, int64_t* n_out, void* stream);

/* ---- A1: demod_am.demod = abs(hilbert(x)) (demod_am.py:18-29) in fixed blocks
 *      (decode_noaa.py:647-653: 240 000-sample blocks, chunker rule).  Asynchronous on `stream` like
 *      dd_resample_fft_f64: intermediates come from a grow-only buffer the library keeps per (device, stream) -- no
 *      allocation and no synchronisation in the steady state; callers on different streams never share a buffer,
 *      two host threads on one stream take turns enqueuing ----------- */
int dd_am_envelope_f64(const double* in, double* out, int64_t n, int64_t block, void* stream);

/* ---- X1: decode_noaa.__correlate (decode_noaa.py:659-675) --------------------- */
/* out[i] = corr_same(h,needle)[i] / sqrt(winenergy_same(h,m)[i] * sum(needle^2)) */
int dd_xcorr_norm_f64(const double* h, int64_t n, const double* needle_host, int m,
                      double* out, void* stream);

/* ---- X2: peak pick of decode_noaa.__correlateAndFindPeaks (decode_noaa.py:713-751) */
/* cor: n float64 on device.  Writes up to max_peaks int64 indices (already shifted
 * by -needle_len/2, sorted) to peaks_host and their count to n_peaks.  Synchronous. */
int dd_find_peaks_f64(const double* cor, int64_t n, double samp_rate, int needle_len,
                      int64_t* peaks_host, int max_peaks, int* n_peaks, void* stream);

/* ---- accurate-sync windows, batched (decode_noaa.getAccurateSync, decode_noaa.py:808-880;
 *      SURVEY.md 8f-2).  For each of n_windows windows of win_len IQ samples starting at
 *      iq[starts_host[w]] (iq on the device; iq_kind 0 = complex64, 1 = interleaved uint8 pairs
 *      minus 127.5, source.py:117-118) runs the per-window chain of :852-853 --
 *      offsetFreq (sample index restarting at 0, cycles_q64 as in dd_nco_c64) -> zero-phase FIR
 *      fir_taps (blackmanHarris(151)) -> demod_fm (stateless) -> demod_am -> zero-phase FIR
 *      pre_taps on the envelope (hamming(492), the default argument of :677; pre_ntaps 0 = none)
 *      -> normalised correlation with the needle (:659-675) -> peak pick (:713-751) -- and
 *      returns per window: peak_host = picked index in the window, already minus needle_len/2
 *      (INT64_MIN if no sample exceeds the threshold); height_host = correlation at the peak
 *      (:762); tsync_host = mean of the envelope over the needle length following the sync
 *      (:755-757), NaN where the reference appends None.  Windows must be shorter than the
 *      0.45 s peak-group distance (true for getAccurateSync's windows at any sample rate);
 *      longer ones are DD_ERR_INVALID -- run them through the per-stage entry points.
 *      All taps / needle pointers are host float64.  Synchronous. */
int dd_noaa_sync_windows(const void* iq, int iq_kind, const int64_t* starts_host, int n_windows,
                         int64_t win_len, uint64_t cycles_q64,
                         const double* fir_taps_host, int fir_ntaps,
                         const double* pre_taps_host, int pre_ntaps,
                         const double* needle_host, int needle_len, double samp_rate,
                         int64_t* peak_host, double* height_host, double* tsync_host, void* stream);
/* ... the window lists of several sync words in one call (getAccurateSync, decode_noaa.py:828-835, searches sync A around the
 *      crude A positions and sync B around the crude B positions: two lists, one chain, two needles of one length).
 *      needle_host holds n_needles (1 or 2) needles of needle_len values back to back; needle_of_window_host[w] names the
 *      needle window w is correlated with (NULL: needle 0).  Results in window order.  One upload, one copy back. */
int dd_noaa_sync_windows_multi(const void* iq, int iq_kind, const int64_t* starts_host, const int* needle_of_window_host,
                               int n_windows, int64_t win_len, uint64_t cycles_q64,
                               const double* fir_taps_host, int fir_ntaps,
                               const double* pre_taps_host, int pre_ntaps,
                               const double* needle_host, int needle_len, int n_needles, double samp_rate,
                               int64_t* peak_host, double* height_host, double* tsync_host, void* stream);

/* dd_noaa_prepare -- builds ahead of time, ON THE HOST, what dd_noaa_crude_tail needs for `n` audio samples in blocks of `block` (0: nothing;
 * decode_noaa.py:647-653) and what dd_noaa_sync_windows needs for windows of `window` IQ samples (0: nothing; decode_noaa.py:823-825): the
 * Hilbert-kernel spectra of the block, of the ragged last block and of the window (closed forms and host transforms, ~12 ms each; no device call --
 * the call that needs a spectrum uploads it, 0.3 ms).  Optional: noaa_sync calls it from threads of their own when the decoder object is created,
 * beside the runtime's first copy, the upload and the audio chain.  `stream` is unused.  No reference counterpart (scipy.signal.hilbert plans
 * nothing ahead). */
int  dd_noaa_prepare(int64_t n, int64_t block, int64_t window, void* stream);
/* P -- getCrudeSync's audio-rate tail (decode_noaa.py:781-790) in one host call: the envelope of `audio` (device float32 or
 *      float64, n samples at samp_rate) in `block`-sample blocks by the chunker rule (__getAM :631-657 -> demod_am.py:29),
 *      then for each of n_needles (1 or 2: sync A and sync B, :786 and :790) piecewise-constant needles of m samples
 *      (needles_host[needle][m], host float64) the normalised correlation (:659-675) and the peak pick (:713-751).  Per
 *      needle d the picked indices -- ascending, already minus m / 2 -- land in peaks_host[d * max_peaks ...] and their
 *      number in n_peaks[d]; env_out (device float64[n], may be NULL) receives the envelope.  The index lists are those of
 *      dd_am_envelope_f64 + dd_xcorr_norm_f64 + dd_find_peaks_f64 called stage by stage.  DD_ERR_UNSUPPORTED (nothing done)
 *      for needles with more than 64 runs, more than 2048 expected peaks (~17 min of audio) or more than 65 536 samples
 *      above the threshold: take the staged route.  Synchronous. */
int dd_noaa_crude_tail(const void* audio, int audio_is_f32, int64_t n, double samp_rate, int64_t block,
                       const double* needles_host, int m, int n_needles, double* env_out,
                       int64_t* peaks_host, int max_peaks, int* n_peaks, void* stream);

/* ---- AFSK1200 correlators (decode_afsk1200.py:99-158) ------------ */
/* binary_filter[s] = mi^2 + mq^2 - si^2 - sq^2, the four sums over buffer_size samples
 * from s against tables_host[4][bs] = mark cos/sin, space cos/sin (:110-123); entries
 * s >= n - bs are 0 (:126).  Same operation order as the reference: float64 bit-exact. */
int dd_afsk_binary_filter_f64(const double* sig, int64_t n, const double* tables_host, int bs,
                              double* out, void* stream);
/* out = np.correlate(np.sign(binary_filter), [-1]*(spb/2) + [1]*(spb - spb/2), 'same') / spb (:147-156) */
int dd_afsk_edges_f64(const double* binary_filter, int64_t n, int spb, double* out, void* stream);

/* np.abs for demod_am.demod_amFLT (demod_am.py:35-62: butter low-pass of |sig|).
 * kind 0: float64, 1: complex128, 2: complex64 input; float64 output (hypot). */
int dd_abs_f64(const void* in, int kind, double* out, int64_t n, void* stream);

/* float32 -> float64 / complex64 -> complex128 widening (audio-rate hand-over) */
int dd_f32_to_f64(const float* in, double* out, int64_t n, void* stream);
int dd_f64_to_f32(const double* in, float* out, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIRECTDEMOD_HIP_H */
