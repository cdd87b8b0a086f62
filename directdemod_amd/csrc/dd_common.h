// Internal helpers shared by the gfx950 translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "../../include/directdemod_hip.h"
#include "../../include/directdemod_hip_debug.h"

void dd_set_error(const char* fmt, ...);
int dd_seam_poll_all(void);         // (dd_chain.hip) chunk-list launches whose in-launch hand-over timed out: DD_OK or DD_ERR_TIMEOUT

#define DD_HIP_CHECK(expr)                                                        \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            dd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                         __FILE__, __LINE__);                                     \
            return (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice)        \
                       ? DD_ERR_NODEVICE                                          \
                       : (_e == hipErrorOutOfMemory ? DD_ERR_NOMEM : DD_ERR_HIP); \
        }                                                                         \
    } while (0)

#define DD_REQUIRE(cond, msg)                                 \
    do {                                                      \
        if (!(cond)) {                                        \
            dd_set_error("invalid argument: %s", msg);        \
            return DD_ERR_INVALID;                            \
        }                                                     \
    } while (0)

#define DD_LAUNCH_CHECK() DD_HIP_CHECK(hipGetLastError())

static inline hipStream_t dd_stream(void* s) { return (hipStream_t)s; }

// Tuning knobs whose sweeps are finished and recorded (tools/README.md names each with its record under profiles/): the product library
// holds the chosen values as constants and reads none of them from the environment; a -DDD_TUNING build (tools/mkvariant.sh N <unit>
// -DDD_TUNING, loaded through DD_LIB_PATH) reads them again for a new sweep.  What the product still reads: DD_LIB_PATH (Python side),
// DD_MFMA_KERNEL (seeds dd_debug_select_kernel once), the route switches the test suite compares (DD_SYNC_HILBERT, DD_SYNC_FRONT,
// DD_SYNC_BATCH, DD_CZT_OWN, DD_AM_HILBERT), the host-side time stamps DD_CRUDE_TRACE / DD_SYNC_TRACE, DD_POOL_BYTES / DD_RESIDENT_BYTES.
#ifdef DD_TUNING
#include <stdlib.h>
#define DD_TUNE_ENV(name) getenv(name)
#else
#define DD_TUNE_ENV(name) ((const char*)nullptr)
#endif

// Grow-only scratch buffer per (device, stream) for an entry point's intermediates.  An entry point takes the lock,
// enqueues its copies and kernels on that stream and drops the lock when it returns: the buffer is then protected by
// stream order (a later call on the same stream is enqueued behind this one; a call on another stream has a buffer of
// its own), and two host threads that use the same stream take turns enqueuing.  Neither allocates nor synchronises in
// the steady state.
struct DDScratchLock {
    void* entry = nullptr;
    char* ptr = nullptr;
    int get(size_t bytes, hipStream_t s);      // DD_OK or an error code; ptr is valid until this object dies
    ~DDScratchLock();
    DDScratchLock() = default;
    DDScratchLock(const DDScratchLock&) = delete;
    DDScratchLock& operator=(const DDScratchLock&) = delete;
};
void dd_scratch_forget_stream(hipStream_t s);  // the stream is about to be destroyed: free its buffers
void dd_audio_forget_stream(hipStream_t s);    // ... and its hipFFT plans and chirp-z tables (dd_audio.hip)


// Per-device one-time work (hipFuncSetAttribute and friends apply to the current device only; the C-ABI has
// dd_set_device, so a process may drive several GPUs, from several threads).  One bit per device ordinal;
// racing first calls on one device both do the idempotent work.
#ifdef __cplusplus
#include <atomic>
struct DDOncePerDevice {
    std::atomic<unsigned long long> done{0};
    static int dev() { int d = 0; return hipGetDevice(&d) == hipSuccess ? (d & 63) : 0; }
    bool need() const { return !((done.load(std::memory_order_acquire) >> dev()) & 1ull); }
    void mark() { done.fetch_or(1ull << dev(), std::memory_order_release); }
};
// compute units of the current device (256 on MI355X); queried once per device
static inline int dd_cu_count() {
    static std::atomic<int> n[64];
    const int d = DDOncePerDevice::dev();
    int v = n[d].load(std::memory_order_relaxed);
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
        n[d].store(v, std::memory_order_relaxed);
    }
    return v;
}

// one per translation unit: names one of its kernels, so that the runtime loads the unit's code object (dd_code_warmup)
int dd_code_touch_decimw(void);
int dd_code_touch_audio(void);
int dd_code_touch_chain(void);
int dd_code_touch_cosfir(void);
int dd_code_touch_fftfir(void);
int dd_code_touch_mfma(void);
int dd_code_touch_fir(void);
int dd_code_touch_afsk(void);

#endif

// ---------------------------------------------------------------------------
// NCO phase arithmetic (comm.py:77).  phase(n) = frac(n * f/fs) is carried as a
// 64-bit binary fraction: phase64 = n * cycles_q64 (mod 2^64), exact for any n.
// exp(-j 2 pi phase) = T[top 12 bits] * exp(-j theta), theta = 2 pi * low bits
// < 1.6e-3 rad (2-term series).  T is a 4096-entry float2 table computed in
// float64 on the device at library initialisation.
// ---------------------------------------------------------------------------
#define DD_NCO_TBITS 12
#define DD_NCO_TSIZE (1 << DD_NCO_TBITS)

const float2* dd_nco_table(void);   // device pointer, lazily initialised (host side)

__device__ __forceinline__ float2 dd_cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

__device__ __forceinline__ float2 dd_phasor(uint64_t phase64, const float2* __restrict__ tbl) {
    const uint32_t k = (uint32_t)(phase64 >> (64 - DD_NCO_TBITS));
    const uint32_t lo = (uint32_t)(phase64 >> (64 - DD_NCO_TBITS - 32));   // next 32 bits
    // theta = 2*pi * lo * 2^-(12+32)
    const float theta = (float)lo * (6.283185307179586f * 5.684341886080802e-14f);   // 2^-44
    const float t2 = theta * theta;
    const float c = fmaf(-0.5f, t2, 1.0f);
    const float s = theta * fmaf(-0.16666667f, t2, 1.0f);
    const float2 T = tbl[k];                     // (cos, -sin)(2 pi k/4096)
    // (Tc + j Ts') * (c - j s), Ts' = -sin
    return make_float2(fmaf(T.x, c, T.y * s), fmaf(T.y, c, -T.x * s));
}
