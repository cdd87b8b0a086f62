// Shared parameter block and staging helpers of the fused NCO -> FIR -> decimate ->
// FM kernels (direct-form f32 and f16-split MFMA variants).  Internal.
#pragma once
#include "dd_common.h"
#include "dd_decimw.h"
#include <vector>

// ---- handles (host side) ------------------------------------------------------
// filters.filter object: taps + carried history (filters.py:21-75)
struct dd_fir {
    int K;
    std::vector<double> taps;
    float* taps_rev;        // device: reversed taps, zero padded (direct-form kernels)
    float2* tail[2];        // device: K-1 past inputs (complex64), ping-pong
    float2* tail_const[2];  // device: constant histories (all zeros / all ones) for launch-free resets
    const float2* tail_override;   // non-null: the next launch reads this history instead of tail[parity]
    int parity;
    void* mfma;             // f16-limb Toeplitz operand for the MFMA path (lazy)
    int mfma_tried;
    // float64 real path (audio rate)
    double* taps_dev;
    double* hist[2];
    int hpar;
    int hist_mode;
    int last_kernel;        // DD_KERNEL_* of the last fused launch through this filter
    long long launches;     // fused kernel launches through this filter (dd_fir_launch_count)
    DDDecimWTaps dw_taps;   // k_chain_decim_w's padded taps (M = 0 mod 4), lazy
    char* multi;            // chunk-list launches: seam flags, per-chunk parameter blocks, prefix tables, seam state (grow-only)
    size_t multi_bytes;
    // chunk-list launches: hand-overs that timed out (dd_seam_wait), counted on the device, mirrored into a pinned word
    // behind every launch and looked at by the next chunk-list call on this filter and by dd_stream_sync
    unsigned int* seam_err;         // device
    unsigned int* seam_err_host;    // pinned
    hipEvent_t seam_ev;             // recorded behind the mirror copy
    int seam_pending;               // a mirror copy has been enqueued and not looked at yet
    int state_invalid;              // a chunk-list launch through this filter timed out: the carried history and last FM sample it
                                    // committed are not to be used -- every fused launch reports DD_ERR_TIMEOUT until dd_fir_reset
};

// demod_fm object: carried last sample (demod_fm.py:43-49)
struct dd_fm {
    float2* last;           // device: [2] ping-pong
    int parity;
    int has_last;           // host mirror of "self.__last is not None"
};

// core fused launch (dd_chain.hip); state is read from / written to fir and fm
struct DDFusedArgs {
    const void* in;
    void* out;
    int64_t n;
    int nco;                // apply offsetFreq
    uint64_t cyc;
    int64_t start_index;    // absolute index of in[0] for the NCO phase
    int M;                  // decimation
    int off;                // chunk-relative index of the first kept sample
    int u8;                 // input is interleaved u8 I,Q
    int commit;             // carry the state forward (storeState)
    int force_direct;
    int tight;              // DD_CHAIN_TIGHT: no running-sum kernel
};
int dd_fused_launch(dd_fir* fir, dd_fm* fm, const DDFusedArgs& a, int64_t* n_out, hipStream_t s);
int64_t dd_fused_out_count(const dd_fm* fm, int64_t n, int M, int off);

struct DDChainParams {
    const void* in;            // complex64 (float2) or interleaved u8
    void* out;                 // float radians (FM) or float2 (FIR only)
    const float2* tail_in;     // K-1 post-NCO samples that precede the chunk (oldest first)
    float2* tail_out;          // receives the new tail
    const float2* lasty_in;    // FIR output that precedes the chunk's first kept sample
    float2* lasty_out;         // receives the chunk's last FIR output
    const float* taps_rev;     // reversed taps, zero padded (see dd_chain.hip)
    const float2* nco_tbl;     // 4096-entry phasor table
    uint64_t cyc;              // frac(f/fs) * 2^64
    int64_t abs0;              // absolute index of in[0]
    int64_t L;                 // input samples in this chunk
    int64_t Ld;                // kept (decimated) samples in this chunk
    int K;                     // taps
    int M;                     // decimation
    int off;                   // chunk-relative index of the first kept sample
    int s;                     // FM shift: 1 on the first call (no previous sample), else 0
    int flags;                 // DD_CHAIN_NCO | DD_CHAIN_FM | DD_CHAIN_U8_INPUT
    int T;                     // FIR outputs computed per block
    int nblocks;
    int skip_lo, skip_hi;      // tiles [skip_lo, skip_hi) belong to another launch (persistent interior kernel); equal = none
    // several chunks in ONE launch (dd_chain_process_chunks): the carried state of chunk c-1 reaches chunk c through
    // device memory inside the launch.  seam_wait: polled (then acquired) before this chunk's tiles read tail_in /
    // lasty_in; seam_post: set, behind an agent-scope release, by the tile that has written tail_out / lasty_out.
    unsigned int* seam_wait;
    unsigned int* seam_post;
    unsigned int* seam_err;    // counts waits that gave up (the launch's outputs are then invalid: reported as DD_ERR_TIMEOUT)
    int seam_spin_log2;        // bound of the wait, in polls of one lane (s_sleep 8 between polls)
};

// XCD-aware tile order: hardware deals consecutive workgroups round-robin over the
// 8 XCDs; give each XCD a contiguous run of tiles so neighbouring tiles (which
// share their (K-1)-sample halo) hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int dd_xcd_tile(int bid, int nblocks) {
    const int nx = 8;
    const int q = nblocks / nx, r = nblocks % nx;
    const int x = bid % nx, i = bid / nx;
    // XCD x owns q (+1 if x < r) tiles, starting at x*q + min(x, r)
    return x * q + (x < r ? x : r) + i;
}

// first FIR-output index computed by tile b
__device__ __forceinline__ int64_t dd_tile_pfirst(const DDChainParams& P, int b) {
    return (P.flags & DD_CHAIN_FM) ? ((int64_t)P.s - 1 + (int64_t)b * (P.T - 1)) : (int64_t)b * P.T;
}

// load one staged sample: chunk-relative index n (may be negative -> tail / zero)
__device__ __forceinline__ float2 dd_load_sample(const DDChainParams& P, int64_t n, float2 phasor) {
    if (n < 0) {
        const int64_t i = n + (P.K - 1);
        return (i >= 0) ? P.tail_in[i] : make_float2(0.f, 0.f);
    }
    if (n >= P.L) return make_float2(0.f, 0.f);
    float2 v;
    if (P.flags & DD_CHAIN_U8_INPUT) {
        const uchar2 u = reinterpret_cast<const uchar2*>(P.in)[n];
        v = make_float2((float)u.x - 127.5f, (float)u.y - 127.5f);
    } else {
        v = reinterpret_cast<const float2*>(P.in)[n];
    }
    if (P.flags & DD_CHAIN_NCO) v = dd_cmul(v, phasor);
    return v;
}

__device__ __forceinline__ float dd_fm_angle(float2 cur, float2 prv) {
    const float re = fmaf(cur.x, prv.x, cur.y * prv.y);
    const float im = fmaf(cur.y, prv.x, -cur.x * prv.y);
    return atan2f(im, re);
}
