// Decimating chain, one wave per block of 2048 samples, no barrier anywhere (round 5):
//
//     offsetFreq (comm.py:63-78) -> FIR (comm.py:80-92, filters.py:64-70) -> bwLim [::M] (comm.py:118-125) -> demod_fm (demod_fm.py:40-49)
//
// Why.  k_chain_decim_p (dd_chain.hip) stages a 48 KB tile per 256-thread workgroup between two barriers; three such workgroups fit a CU and
// their load / stage / tap-loop phases overlap only as far as chance has it: the raw-u8 flavour, which reads a quarter of the bytes, is barely
// faster than the complex64 one (0.082-0.094 against 0.098-0.100 ms for 2^26 samples) -- the kernel is bound by its own phases, at 0.68 of the HBM
// peak.  Here a WAVE owns its LDS image and walks rows on its own, the next row's samples in flight while it works on this one (k_chain_cos1k's
// scheme, dd_cosfir.hip).
//
// Rows live on the ABSOLUTE sample grid: row R is the block of W = 2048 samples [R W, (R + 1) W) (absolute indices: the NCO's own count,
// comm.py:75-76), sixteen 16-byte loads per lane whatever M is.  Kept samples sit at absolute indices phi + G M (phi = (abs0 + off) mod M,
// constant along a stream, comm.py:123-125); the outputs whose newest sample falls into a block belong to its row, lane by lane (64 per pass:
// one pass for M >= 32).  The wave keeps the HP = K - 1 (rounded up to even) samples before the block from the row before (copied down inside
// LDS after the tap loop).  A sample after the NCO is a pure function of its absolute index and of the stream's constants (row phasor x phasor
// of its group of 64 x phasor inside the group, each an exact table look-up), whatever the load layout and wherever the recording lies -- and so
// is the carried history (the new tail is recomputed with the same arithmetic): a chunk list over one recording, which continues both grids
// from chunk to chunk, IS one long chunk -- dd_chain_process_chunks makes ONE launch of this kernel, no hand-over inside it, and the outputs
// equal the chunk loop's bit for bit; raw u8 input gives the bits of the same samples as complex64.
//
// A lane runs the K taps over its output's window (LDS reads 16 bytes wide, conflict free for M = 2 mod 4, two-way for M = 4 mod 8; M = 0 mod 8
// through a padded image, DWMap; the taps are wave uniform and come through the scalar cache in pairs; one packed multiply-add handles re and im).  Rows that reach outside the chunk (stream start: the carried history;
// chunk end) take guarded sample-by-sample loads with the same phasors; rows come in runs dealt to the waves in turn (one moving window over
// the stream), a run starts from the last K + M samples of the row before it (halo, and the FIR output the discriminator needs).
#include "dd_chain_kernels.h"
#include "dd_decimw.h"
#include "dd_atan.h"
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4f_a8 __attribute__((ext_vector_type(4), aligned(8)));        // 16-byte load on a sample boundary (complex64) ...
typedef uint32_t v4u_a2 __attribute__((ext_vector_type(4), aligned(2)));     // ... (raw u8): the row grid follows the absolute sample index, not the address

#ifndef DW_W
#define DW_W 2048            // samples per row
#endif
#ifndef DW_WAVES_PER_SIMD
#define DW_WAVES_PER_SIMD 2
#endif
#define DW_NL (DW_W / 128)   // 16-byte loads per lane and row, complex64
#define DW_NL8 (DW_W / 512)  // the same, raw u8
#ifndef DW_TRIP
#define DW_TRIP 16            // taps per trip of the tap loop
#endif
#define DW_PAD DW_TRIP       // zeros behind the block: the tap loop runs in whole trips
#ifndef DW_MAX_M
#define DW_MAX_M 64
#endif
#define DW_NG (DW_W / 64)    // group phasors e^{-j w 64 g}

struct DDDecimWArgs {
    const void* in;
    void* out;
    const float2* tail_in;     // K-1 samples after the NCO that precede the chunk
    float2* tail_out;
    const float2* lasty_in;    // FIR output before the chunk's first kept sample
    float2* lasty_out;
    const float* taps;         // g[j - e], j = 0 .. K16-1 (zeros outside g[0 .. K-1])
    const float2* nco_tbl;
    uint64_t cyc;
    int64_t abs0, L, Ld;
    int64_t R0;                // absolute index of row 0 of this launch
    int nrows, nwaves, run_rows;
    int K, K16, M, HP;
    int e;                     // the window starts one sample early (on an even LDS sample) when 1; the taps start with a zero then
    int phi;                   // (abs0 + off) mod M
    int off;
    int cq, cr;                // (W - 1) / M + 1, (W - 1) % M
    uint32_t minv;             // PAD: 2^32 / M + 1
    int img;                   // samples of the LDS image (halo, block, gaps, zeros behind)
    int s;                     // 1: stream start, no angle for output 0
};

// a * w as one packed multiply and one packed multiply-add with the operand selects and sign modifiers spelt out (left to the compiler the
// swizzled, negated copy of `a` costs two or three more instructions): (a.x w.x, a.y w.x), then + (-a.y w.y, a.x w.y).  The SAME two
// roundings wherever a sample is rotated (rows, guarded rows, tail).
__device__ __forceinline__ v2f dw_cmul(v2f a, v2f w) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ v2f dw_v2(float2 a) { return (v2f){a.x, a.y}; }

// dd_phasor's arithmetic on a table entry fetched by the caller
__device__ __forceinline__ v2f dw_phasor_from(float2 T, uint32_t lo) {
    const float theta = (float)lo * (6.283185307179586f * 5.684341886080802e-14f);
    const float t2 = theta * theta;
    const float c = fmaf(-0.5f, t2, 1.0f);
    const float s = theta * fmaf(-0.16666667f, t2, 1.0f);
    return (v2f){fmaf(T.x, c, T.y * s), fmaf(T.y, c, -T.x * s)};
}
// wave-uniform phase: the table entry comes through the scalar cache (a vector load would queue behind the row's sample loads)
__device__ __forceinline__ v2f dw_phasor_u(uint64_t phase64, const float2* tbl) {
    const uint32_t k = __builtin_amdgcn_readfirstlane((uint32_t)(phase64 >> (64 - DD_NCO_TBITS)));
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(phase64 >> (64 - DD_NCO_TBITS - 32)));
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) float2* dw_const_f2;
    const float2 T = ((dw_const_f2)tbl)[k];
#else
    (void)k;
    const float2 T = make_float2(1.f, 0.f);
#endif
    return dw_phasor_from(T, lo);
}
__device__ __forceinline__ v2f dw_phasor_v(uint64_t phase64, const float2* tbl) {
    return dw_phasor_from(tbl[(uint32_t)(phase64 >> (64 - DD_NCO_TBITS))], (uint32_t)(phase64 >> (64 - DD_NCO_TBITS - 32)));
}

// The NCO factor of the block's sample r (comm.py:77) is (row phasor x w[r & 63]) x G[r >> 6] -- whatever the load layout, so that raw u8
// input (source.py:117-118) and the same samples as complex64 give the same bits.  Per lane: w of its own samples; per row: their products
// with the row phasor (DWRowPh).
struct DWPh {
    v2f w[8];        // complex64: w[0], w[1] = e^{-j w ((2 l) & 63)}, the same + 1; u8: w[k] = e^{-j w (((8 l) & 63) + k)}
};
template <bool U8, bool NCO>
__device__ __forceinline__ void dw_row_ph(v2f prow, const DWPh& ph, DWPh& pw) {
#pragma unroll
    for (int k = 0; k < 8; ++k) pw.w[k] = (NCO && k < (U8 ? 8 : 2)) ? dw_cmul(prow, ph.w[k]) : (v2f){1.f, 0.f};
}

// one sample of the chunk by its chunk-relative index: history (already rotated), raw sample times `ph`, or zero
template <bool U8, bool NCO>
__device__ __forceinline__ v2f dw_sample(const DDDecimWArgs& A, int64_t n, v2f ph) {
    if (n < 0) {
        const int64_t ti = n + (A.K - 1);
        return ti >= 0 ? dw_v2(A.tail_in[ti]) : (v2f){0.f, 0.f};
    }
    if (n >= A.L) return (v2f){0.f, 0.f};
    v2f x;
    if (U8) {
        const uchar2 u = reinterpret_cast<const uchar2*>(A.in)[n];
        x = (v2f){(float)u.x - 127.5f, (float)u.y - 127.5f};
    } else {
        x = dw_v2(reinterpret_cast<const float2*>(A.in)[n]);
    }
    return NCO ? dw_cmul(x, ph) : x;
}

// ---- where a staged sample sits in LDS.  Plain: sample r of [halo | block] at position r.  PAD (M = 0 mod 8): the lanes' windows start M
// samples = a multiple of 16 banks apart -- every 16-byte read of the tap loop a bank conflict (M = 32: all lanes on the same banks).  There the
// image carries two samples of gap after every M, counted from the first window's start: lane i's window starts (M + 2) i further on (= 2 mod 4:
// conflict free), every window meets the gaps at the same places, and the taps carry zeros there (padded taps, built on the host).  The layout follows the row's own
// phase: the halo moves down from the row before's layout into this row's (dw_halo_*).
struct DWMap {
    int c0m;           // (first window's start) mod M, minus M
    uint32_t minv;     // 2^32 / M + 1: x / M = umulhi(x, minv) for the x met here
};
template <bool PAD>
__device__ __forceinline__ int dw_pos(const DWMap& mp, int r) {
    if (!PAD) return r;
    return r + 2 * (int)__umulhi((uint32_t)(r - mp.c0m), mp.minv);
}

// ---- complex64 rows: load j, lane l = samples 128 j + 2 l, + 1 of the block
// (J0, NJ: loads J0 .. J0 + NJ - 1 of the row -- all sixteen, or the last four: what a run needs of the row before it)
template <int J0, int NJ>
__device__ __forceinline__ void dw_issue(const DDDecimWArgs& A, int64_t Brel, int lane, v4f_a8 (&x)[NJ]) {
    const v4f_a8* p = reinterpret_cast<const v4f_a8*>(reinterpret_cast<const float2*>(A.in) + Brel + 2 * lane);
#pragma unroll
    for (int j = 0; j < NJ; ++j) x[j] = __builtin_nontemporal_load(p + 64 * (J0 + j));
}
template <bool NCO, bool PAD, int J0, int NJ>
__device__ __forceinline__ void dw_stage(const DDDecimWArgs& A, float2* buf, const float2* gl, int lane, const DWPh& pw, const v4f_a8 (&x)[NJ], const DWMap& mp) {
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = J0 + jj;
        v2f x0 = (v2f){x[jj].x, x[jj].y}, x1 = (v2f){x[jj].z, x[jj].w};
        if (NCO) {
            const v2f g = dw_v2(gl[2 * j + (lane >> 5)]);
            x0 = dw_cmul(x0, dw_cmul(pw.w[0], g));
            x1 = dw_cmul(x1, dw_cmul(pw.w[1], g));
        }
        const int r = A.HP + 128 * j + 2 * lane;
        *reinterpret_cast<v4f*>(buf + dw_pos<PAD>(mp, r)) = (v4f){x0.x, x0.y, x1.x, x1.y};
    }
}
// the same through guarded sample-by-sample loads, loads jlo .. 15
template <bool NCO, bool PAD>
__device__ __forceinline__ void dw_stage_guarded(const DDDecimWArgs& A, float2* buf, const float2* gl, int lane, int jlo, int64_t Brel, const DWPh& pw, const DWMap& mp) {
    for (int j = jlo; j < DW_NL; ++j) {
        v2f pj = (v2f){1.f, 0.f}, pj1 = (v2f){1.f, 0.f};
        if (NCO) {
            const v2f g = dw_v2(gl[2 * j + (lane >> 5)]);
            pj = dw_cmul(pw.w[0], g);
            pj1 = dw_cmul(pw.w[1], g);
        }
        const int64_t n = Brel + 128 * j + 2 * lane;
        const v2f x0 = dw_sample<false, NCO>(A, n, pj), x1 = dw_sample<false, NCO>(A, n + 1, pj1);
        *reinterpret_cast<v4f*>(buf + dw_pos<PAD>(mp, A.HP + 128 * j + 2 * lane)) = (v4f){x0.x, x0.y, x1.x, x1.y};
    }
}

// ---- raw u8 rows (source.py:117-118): load j, lane l = samples 512 j + 8 l .. + 7 of the block
template <int J0, int NJ>
__device__ __forceinline__ void dw_issue8(const DDDecimWArgs& A, int64_t Brel, int lane, v4u_a2 (&x)[NJ]) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(A.in) + 2 * (Brel + 8 * lane);
#pragma unroll
    for (int j = 0; j < NJ; ++j) x[j] = __builtin_nontemporal_load(reinterpret_cast<const v4u_a2*>(p + 1024 * (J0 + j)));
}
template <bool NCO, bool PAD, int J0, int NJ>
__device__ __forceinline__ void dw_stage8(const DDDecimWArgs& A, float2* buf, const float2* gl, int lane, const DWPh& pw, const v4u_a2 (&x)[NJ], const DWMap& mp) {
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = J0 + jj;
        v2f g = (v2f){1.f, 0.f};
        if (NCO) g = dw_v2(gl[8 * j + (lane >> 3)]);
        const uint32_t d[4] = {x[jj].x, x[jj].y, x[jj].z, x[jj].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v2f xa = (v2f){(float)(d[k] & 0xff) - 127.5f, (float)((d[k] >> 8) & 0xff) - 127.5f};
            v2f xb = (v2f){(float)((d[k] >> 16) & 0xff) - 127.5f, (float)(d[k] >> 24) - 127.5f};
            if (NCO) {
                xa = dw_cmul(xa, dw_cmul(pw.w[2 * k], g));
                xb = dw_cmul(xb, dw_cmul(pw.w[2 * k + 1], g));
            }
            const int r = A.HP + 512 * j + 8 * lane + 2 * k;
            *reinterpret_cast<v4f*>(buf + dw_pos<PAD>(mp, r)) = (v4f){xa.x, xa.y, xb.x, xb.y};
        }
    }
}
template <bool NCO, bool PAD>
__device__ __forceinline__ void dw_stage8_guarded(const DDDecimWArgs& A, float2* buf, const float2* gl, int lane, int jlo, int64_t Brel, const DWPh& pw, const DWMap& mp) {
    for (int j = jlo; j < DW_NL8; ++j) {
        v2f g = (v2f){1.f, 0.f};
        if (NCO) g = dw_v2(gl[8 * j + (lane >> 3)]);
        const int64_t n = Brel + 512 * j + 8 * lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v2f pa = g, pb = g;
            if (NCO) {
                pa = dw_cmul(pw.w[2 * k], g);
                pb = dw_cmul(pw.w[2 * k + 1], g);
            }
            const v2f xa = dw_sample<true, NCO>(A, n + 2 * k, pa), xb = dw_sample<true, NCO>(A, n + 2 * k + 1, pb);
            *reinterpret_cast<v4f*>(buf + dw_pos<PAD>(mp, A.HP + 512 * j + 8 * lane + 2 * k)) = (v4f){xa.x, xa.y, xb.x, xb.y};
        }
    }
}

// the K taps over the window that starts at LDS sample `ws` (even): sixteen taps per trip, four partial sums.  The next trip's samples (eight
// 16-byte LDS reads) and taps (one scalar load) are requested before this trip's multiply-adds: with two waves per SIMD nothing else hides
// their latency (the first version waited for both every eight taps: 0.089 ms for 2^26 raw u8 samples, which move a quarter of the bytes).
__device__ __forceinline__ void dw_taps_load(const v4f* __restrict__ w4, int j, v4f (&x)[DW_TRIP / 2]) {
#pragma unroll
    for (int u = 0; u < DW_TRIP / 2; ++u) x[u] = w4[j / 2 + u];
}
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) v2f* dw_const_f2p;
#else
typedef const v2f* dw_const_f2p;
#endif
__device__ __forceinline__ void dw_taps_coef(dw_const_f2p G, int j, v2f (&c)[DW_TRIP / 2]) {
#pragma unroll
    for (int u = 0; u < DW_TRIP / 2; ++u) c[u] = G[j / 2 + u];
}
// acc += (c.x, c.x) * x resp. (c.y, c.y) * x, the tap pair c in a scalar register pair (left to the compiler the odd tap of a pair is first
// copied into a pair of its own)
__device__ __forceinline__ void dw_mac_lo(v2f& acc, v2f c, v2f x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(c), "v"(x));
}
__device__ __forceinline__ void dw_mac_hi(v2f& acc, v2f c, v2f x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(c), "v"(x));
}
template <bool PAD>
__device__ __forceinline__ v2f dw_taps(const DDDecimWArgs& A, const float2* buf, int ws) {
    const v4f* __restrict__ w4 = reinterpret_cast<const v4f*>(buf + ws);
    const dw_const_f2p G = (dw_const_f2p)A.taps;
    v2f a0 = (v2f){0.f, 0.f}, a1 = (v2f){0.f, 0.f}, a2 = (v2f){0.f, 0.f}, a3 = (v2f){0.f, 0.f};
    v4f xa[DW_TRIP / 2], xb[DW_TRIP / 2];
    v2f ca[DW_TRIP / 2], cb[DW_TRIP / 2];
    dw_taps_load(w4, 0, xa);
    dw_taps_coef(G, 0, ca);
    auto mac = [&](const v4f (&x)[DW_TRIP / 2], const v2f (&c)[DW_TRIP / 2]) {
#pragma unroll
        for (int u = 0; u < DW_TRIP / 2; u += 2) {
            dw_mac_lo(a0, c[u], (v2f){x[u].x, x[u].y});
            dw_mac_hi(a1, c[u], (v2f){x[u].z, x[u].w});
            dw_mac_lo(PAD ? a0 : a2, c[u + 1], (v2f){x[u + 1].x, x[u + 1].y});
            dw_mac_hi(PAD ? a1 : a3, c[u + 1], (v2f){x[u + 1].z, x[u + 1].w});
        }
    };
    // (two trips per turn: the two register sets alternate, nothing is copied)
    for (int j = 0; j < A.K16; j += 2 * DW_TRIP) {
        if (j + DW_TRIP < A.K16) { dw_taps_load(w4, j + DW_TRIP, xb); dw_taps_coef(G, j + DW_TRIP, cb); }
        mac(xa, ca);
        if (j + DW_TRIP < A.K16) {
            if (j + 2 * DW_TRIP < A.K16) { dw_taps_load(w4, j + 2 * DW_TRIP, xa); dw_taps_coef(G, j + 2 * DW_TRIP, ca); }
            mac(xb, cb);
        }
    }
    // the four partial sums by TRUE tap index mod 4, whatever the window's alignment (e = 1: the loop's index runs one ahead): the same
    // additions in the same order for an output wherever its stream's phase puts it -- a chain without NCO counts every chunk from zero
    // (PAD: two partial sums, even and odd taps -- the gaps of the padded image move with e against the taps, but they are two samples wide)
    if (PAD) return A.e ? a1 + a0 : a0 + a1;
    return A.e ? (a1 + a2) + (a3 + a0) : (a0 + a1) + (a2 + a3);
}

// the HP samples before the next row's block: the end of this row's image moves to the front (HP <= 256: two 16-byte pieces per lane at
// most; read before the discriminator, written after it, so that the LDS round trip hides behind it).  PAD: out of this row's layout into the
// next row's.
template <bool PAD>
__device__ __forceinline__ void dw_halo_read(const DDDecimWArgs& A, const float2* buf, int lane, v4f (&h)[2], const DWMap& mp) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (lane + 64 * u < A.HP / 2) h[u] = *reinterpret_cast<const v4f*>(buf + dw_pos<PAD>(mp, DW_W + 2 * (lane + 64 * u)));
}
template <bool PAD>
__device__ __forceinline__ void dw_halo_write(const DDDecimWArgs& A, float2* buf, int lane, const v4f (&h)[2], const DWMap& mp) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (lane + 64 * u < A.HP / 2) *reinterpret_cast<v4f*>(buf + dw_pos<PAD>(mp, 2 * (lane + 64 * u))) = h[u];
}

__device__ __forceinline__ float dw_shr1(float v, float first) {       // wave_shr:1; lane 0 keeps `first`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ v2f dw_lane(v2f v, int l) {                  // (l: wave uniform)
    return (v2f){__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), l)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), l))};
}

#ifdef DW_TRACE
// tools/debug/decimw_trace.py: cycles per phase of an interior row (every stamp drains the wave's counters), summed per wave
#define DW_NPH 6
__device__ unsigned long long g_dw_trace[4096 * (DW_NPH + 2)];
#define DW_T(i) do { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); const unsigned t_ = (unsigned)__builtin_readcyclecounter(); tr[i] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DW_T(i) do { } while (0)
#endif

// where the outputs of a row sit: r0 = block offset of its first kept sample, cnt of them, p0 = chunk-relative index of the first
struct DWRow {
    int r0, cnt;
    int64_t p0;
};
__device__ __forceinline__ void dw_row_next(const DDDecimWArgs& A, DWRow& r) {
    r.p0 += r.cnt;
    r.r0 -= DW_W % A.M;
    if (r.r0 < 0) r.r0 += A.M;
    r.cnt = A.cq - (r.r0 > A.cr ? 1 : 0);                      // (W - 1 - r0) / M + 1
}

// a staged row: the tap loop over its outputs, 64 per pass; the outputs leave; the halo moves down.  ycarry: the FIR output before the
// row's first one on entry, the row's last one on exit.  emit false: the row before a run (only ycarry and the halo matter)
// the LDS layout of a PAD row: gaps counted from its first window's start
__device__ __forceinline__ DWMap dw_row_map(const DDDecimWArgs& A, const DWRow& r) {
    const int ws0 = A.HP - A.K + 1 + r.r0 - A.e;
    const int k = (int)__umulhi((uint32_t)ws0, A.minv);
    return DWMap{ws0 - k * A.M - A.M, A.minv};
}

template <bool FM, bool PAD>
__device__ __forceinline__ void dw_row_outputs(const DDDecimWArgs& A, float2* buf, int lane, const DWRow& r, const DWMap& mp, bool emit, v2f& ycarry, v2f ylast_in
#ifdef DW_TRACE
                                               , unsigned* tr = nullptr, unsigned tprev = 0
#endif
                                               ) {
#ifdef DW_TRACE
    unsigned trd[DW_NPH];
    if (!tr) tr = trd;
#endif
    v4f hl[2];
    DWMap mpn = mp;                                           // the NEXT row's layout: where the halo goes
    if (PAD) {
        DWRow rn = r;
        dw_row_next(A, rn);
        mpn = dw_row_map(A, rn);
    }
    if (!FM && !emit) {                                       // (complex64 output: no output depends on the one before it)
        dw_halo_read<PAD>(A, buf, lane, hl, mp);
        dw_halo_write<PAD>(A, buf, lane, hl, mpn);
        return;
    }
    const int ws0 = A.HP - A.K + 1 + r.r0 - A.e;
    // PAD: the first window starts behind floor(ws0 / M) + 1 gaps, every further one M + 2 samples on
    const int wsp = PAD ? ws0 + 2 * ((int)__umulhi((uint32_t)ws0, A.minv) + 1) : ws0;
    const int wstep = PAD ? A.M + 2 : A.M;
    const int ng = (r.cnt + 63) >> 6;
    for (int t = emit ? 0 : ng - 1; t < ng; ++t) {
        const int i = 64 * t + lane;
        const int ic = i < r.cnt ? i : r.cnt - 1;
        const v2f y = dw_taps<PAD>(A, buf, wsp + ic * wstep);
        if (t == ng - 1) dw_halo_read<PAD>(A, buf, lane, hl, mp);
        DW_T(2);
        if (FM) {
            v2f yp = (v2f){dw_shr1(y.x, ycarry.x), dw_shr1(y.y, ycarry.y)};
            const int last = t == ng - 1 ? (r.cnt - 1) & 63 : 63;
            ycarry = dw_lane(y, last);
            const int64_t p = r.p0 + i;
            if (emit && i < r.cnt && p >= 0 && p < A.Ld) {
                if (p == 0 && A.s == 0) yp = ylast_in;
                if (p >= A.s) {
                    // angle(y[p] conj(y[p-1])) (demod_fm.py:40-49), the M = 1 kernels' arctangent (dd_atan.h)
                    const float re = fmaf(y.x, yp.x, y.y * yp.y), im = fmaf(y.y, yp.x, -y.x * yp.y);
                    reinterpret_cast<float*>(A.out)[p - A.s] = dd_atan2_poly(im, re);
                }
                if (p == A.Ld - 1 && A.lasty_out) *A.lasty_out = make_float2(y.x, y.y);
            }
        } else {
            const int64_t p = r.p0 + i;
            if (emit && i < r.cnt && p >= 0 && p < A.Ld) {
                reinterpret_cast<float2*>(A.out)[p] = make_float2(y.x, y.y);
                if (p == A.Ld - 1 && A.lasty_out) *A.lasty_out = make_float2(y.x, y.y);
            }
        }
        DW_T(3);
    }
    dw_halo_write<PAD>(A, buf, lane, hl, mpn);
    DW_T(4);
}

template <bool U8, bool NCO, bool FM, bool PAD>
__global__ void __launch_bounds__(64, DW_WAVES_PER_SIMD) k_chain_decim_w(const DDDecimWArgs A) {
    extern __shared__ __attribute__((aligned(16))) char dw_smem[];
    float2* const buf = reinterpret_cast<float2*>(dw_smem);
    float2* const gl = buf + A.img;
    const int lane = threadIdx.x;
    const int gw = blockIdx.x;
    const int M = A.M;
    if (PAD) {
        // (the gaps are never written: they start as zeros -- whatever a later row's layout leaves in a gap is finite and meets a zero tap)
        for (int i = lane; i < A.img / 2; i += 64) reinterpret_cast<v4f*>(buf)[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    } else if (lane < DW_PAD) {
        buf[A.HP + DW_W + lane] = make_float2(0.f, 0.f);
    }
    DWPh ph;
#pragma unroll
    for (int k = 0; k < 8; ++k) ph.w[k] = (v2f){1.f, 0.f};
    if (NCO) {
#pragma unroll
        for (int k = 0; k < (U8 ? 8 : 2); ++k) ph.w[k] = dw_phasor_v((uint64_t)(((U8 ? 8 * lane : 2 * lane) & 63) + k) * A.cyc, A.nco_tbl);
        if (lane < DW_NG) {
            const v2f g = dw_phasor_v((uint64_t)(64 * lane) * A.cyc, A.nco_tbl);
            gl[lane] = make_float2(g.x, g.y);
        }
    }
    // the new carried history (the chunk's last K-1 samples after the NCO, older ones from the old history): the value a row gives a sample,
    // recomputed -- its row, its group of 64 and its place in the group follow from its absolute index
    if (gw == 0 && A.tail_out) {
        for (int i = lane; i < A.K - 1; i += 64) {
            const int64_t n = A.L - (A.K - 1) + i;
            v2f p = (v2f){1.f, 0.f};
            if (NCO && n >= 0) {
                const int64_t na = A.abs0 + n;
                int64_t R = na / DW_W;
                if (na - R * DW_W < 0) --R;
                const int r = (int)(na - R * DW_W);
                const v2f prow = dw_phasor_v((uint64_t)(R * DW_W) * A.cyc, A.nco_tbl);
                const v2f pw = dw_cmul(prow, dw_phasor_v((uint64_t)(r & 63) * A.cyc, A.nco_tbl));
                p = dw_cmul(pw, dw_phasor_v((uint64_t)(64 * (r >> 6)) * A.cyc, A.nco_tbl));
            }
            const v2f v = dw_sample<U8, NCO>(A, n, p);
            A.tail_out[i] = make_float2(v.x, v.y);
        }
    }
    // the FIR output before the chunk (demod_fm.py:47-49), fetched once: a load inside a row would wait for the row's prefetch
    v2f ylast_in = (v2f){0.f, 0.f};
    if (FM && A.s == 0) ylast_in = dw_v2(*A.lasty_in);
    const int RR = A.run_rows;
    const int nruns = (A.nrows + RR - 1) / RR;
    // the part of the row before a run that the run needs: the window of its last output (whose FIR output is y[p-1] of the run's first
    // one) and everything after it
    const int rmin = DW_W - M - A.K;
    const int jlo = (rmin > 0 ? rmin : 0) >> (U8 ? 9 : 7);
    static_assert(DW_W - 512 >= 0 && DW_NL >= 4 && DW_NL8 >= 1, "the last 512 samples of a row hold K + M <= 320");
    v4f_a8 x[DW_NL];
    v4u_a2 x8[DW_NL8];
#ifdef DW_TRACE
    unsigned tr[DW_NPH];
#pragma unroll
    for (int i = 0; i < DW_NPH; ++i) tr[i] = 0;
    unsigned trows = 0;
    const unsigned tstart = (unsigned)__builtin_readcyclecounter();
#endif
    for (int run = gw; run < nruns; run += A.nwaves) {
        const int q0 = run * RR, q1 = q0 + RR < A.nrows ? q0 + RR : A.nrows;
        auto brel = [&](int q) { return (A.R0 + q) * (int64_t)DW_W - A.abs0; };
        auto inside = [&](int q) { const int64_t b = brel(q); return b >= 0 && b + DW_W <= A.L; };
        auto row_phasor = [&](int q) { return NCO ? dw_phasor_u((uint64_t)((A.R0 + q) * (int64_t)DW_W) * A.cyc, A.nco_tbl) : (v2f){1.f, 0.f}; };
        // rows [q0, f0) and [f1, q1) reach outside the chunk, [f0, f1) lie inside it
        int f0 = q0;
        while (f0 < q1 && !inside(f0)) ++f0;
        int f1 = f0;
        while (f1 < q1 && inside(f1)) ++f1;
        // the row before the run: its last K + M samples (<= 320) are all the run needs -- the last four loads of the row (u8: the last one)
        // when it lies inside the chunk, requested ahead of the run's first row
        const bool pin = inside(q0 - 1);
        v4f_a8 xp[4];
        v4u_a2 xp8[1];
        if (pin) {
            if constexpr (U8) dw_issue8<DW_NL8 - 1, 1>(A, brel(q0 - 1), lane, xp8); else dw_issue<DW_NL - 4, 4>(A, brel(q0 - 1), lane, xp);
        }
        if (f1 > f0) {
            if constexpr (U8) dw_issue8<0, DW_NL8>(A, brel(f0), lane, x8); else dw_issue<0, DW_NL>(A, brel(f0), lane, x);
        }
        // the row before the run
        DWRow r;
        DWMap mp = DWMap{0, A.minv};
        {
            const int64_t B = (A.R0 + q0 - 1) * (int64_t)DW_W;
            int64_t m = ((int64_t)A.phi - B) % M;
            if (m < 0) m += M;
            r.r0 = (int)m;
            r.cnt = (DW_W - 1 - r.r0) / M + 1;
            r.p0 = (B + r.r0 - A.abs0 - A.off) / M;                  // (exact: B + r0 is a kept sample's absolute index)
            if (PAD) mp = dw_row_map(A, r);
            DWPh pw;
            dw_row_ph<U8, NCO>(row_phasor(q0 - 1), ph, pw);
            if (pin) {
                if constexpr (U8) dw_stage8<NCO, PAD, DW_NL8 - 1, 1>(A, buf, gl, lane, pw, xp8, mp);
                else dw_stage<NCO, PAD, DW_NL - 4, 4>(A, buf, gl, lane, pw, xp, mp);
            } else {
                if constexpr (U8) dw_stage8_guarded<NCO, PAD>(A, buf, gl, lane, jlo, brel(q0 - 1), pw, mp);
                else dw_stage_guarded<NCO, PAD>(A, buf, gl, lane, jlo, brel(q0 - 1), pw, mp);
            }
        }
        v2f ycarry = (v2f){0.f, 0.f};
        dw_row_outputs<FM, PAD>(A, buf, lane, r, mp, false, ycarry, ylast_in);
        for (int q = q0; q < q1; ++q) {
#ifdef DW_TRACE
            unsigned tprev = (unsigned)__builtin_readcyclecounter();
#endif
            const bool fast = q >= f0 && q < f1;
            dw_row_next(A, r);
            if (PAD) mp = dw_row_map(A, r);
            DWPh pw;
            dw_row_ph<U8, NCO>(row_phasor(q), ph, pw);
            DW_T(5);
            if (fast) {
                if constexpr (U8) dw_stage8<NCO, PAD, 0, DW_NL8>(A, buf, gl, lane, pw, x8, mp);
                else dw_stage<NCO, PAD, 0, DW_NL>(A, buf, gl, lane, pw, x, mp);
                DW_T(0);
                if (q + 1 < f1) {
                    // the next row's samples fly during this row's tap loop
                    if constexpr (U8) dw_issue8<0, DW_NL8>(A, brel(q + 1), lane, x8); else dw_issue<0, DW_NL>(A, brel(q + 1), lane, x);
                }
            } else {
                if constexpr (U8) dw_stage8_guarded<NCO, PAD>(A, buf, gl, lane, 0, brel(q), pw, mp);
                else dw_stage_guarded<NCO, PAD>(A, buf, gl, lane, 0, brel(q), pw, mp);
            }
#ifdef DW_TRACE
            if (fast) {
                { __builtin_amdgcn_sched_barrier(0); const unsigned t_ = (unsigned)__builtin_readcyclecounter(); tr[1] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); }
                ++trows;
                dw_row_outputs<FM, PAD>(A, buf, lane, r, mp, true, ycarry, ylast_in, tr, tprev);
            } else
#endif
            dw_row_outputs<FM, PAD>(A, buf, lane, r, mp, true, ycarry, ylast_in);
        }
    }
#ifdef DW_TRACE
    if (gw < 4096 && lane == 0) {
#pragma unroll
        for (int i = 0; i < DW_NPH; ++i) g_dw_trace[gw * (DW_NPH + 2) + i] = tr[i];
        g_dw_trace[gw * (DW_NPH + 2) + DW_NPH] = trows;
        g_dw_trace[gw * (DW_NPH + 2) + DW_NPH + 1] = (unsigned)__builtin_readcyclecounter() - tstart;
    }
#endif
}

#ifdef DW_TRACE
extern "C" int dd_debug_decimw_trace(unsigned long long* out, int nwaves) {
    DD_HIP_CHECK(hipDeviceSynchronize());
    DD_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dw_trace), sizeof(unsigned long long) * (size_t)nwaves * (DW_NPH + 2)));
    return DD_OK;
}
#endif

// ============================================================================ host side
int dd_decimw_supported(int K, int M, int flags, const void* in) {
    if (M < 8 || M > DW_MAX_M || (M & 1) || K < 2 || K > 256) return 0;
    const uintptr_t a = reinterpret_cast<uintptr_t>(in);
    return (a & ((flags & DD_CHAIN_U8_INPUT) ? 1 : 7)) == 0 ? 1 : 0;
}

struct DWPlan {
    int64_t R0;
    int nrows, phi, HP, e, K16, wpc, run_rows, nwaves, nruns, pad, img;
    size_t lds;
};
static int64_t dw_floordiv(int64_t a, int64_t b) { int64_t q = a / b; if (a - q * b < 0) --q; return q; }
// taps as the tap loop meets them: e leading zeros, and in a PAD image two zeros after every M
static int dw_padded_taps_len(int K, int M, int e, int pad) {
    const int Ke = K + e;
    const int raw = pad ? Ke + 2 * ((Ke - 1) / M) : Ke;
    return (raw + DW_TRIP - 1) & ~(DW_TRIP - 1);
}
static void decimw_plan(int64_t abs0, int64_t Ld, int K, int M, int off, int ncu, DWPlan& pl) {
    const int64_t first = abs0 + off;                          // absolute index of the chunk's first kept sample
    int64_t phi = first % M;
    if (phi < 0) phi += M;
    pl.phi = (int)phi;
    pl.R0 = dw_floordiv(first, DW_W);
    const int64_t Rl = Ld > 0 ? dw_floordiv(first + (Ld - 1) * M, DW_W) : pl.R0 - 1;
    pl.nrows = (int)(Rl - pl.R0 + 1);
    pl.HP = K & ~1;                                            // K - 1 rounded up to even
    // a window starts at LDS sample HP - K + 1 + (offset of its kept sample in the block): the parity of that offset is phi's (W and M are
    // even), so the parity of the start is the launch's -- one sample earlier, behind a zero tap, where it is odd
    pl.e = (int)((pl.HP - K + 1 + phi) & 1);
    // lane stride M samples = M / 2 sixteen-byte bank groups of sixteen: conflict free for odd M / 2, two-way for M = 4 mod 8 (cheaper than the
    // padded image's longer tap loop: M = 12 0.158 against 0.183 ms, M = 20 0.128 / 0.132), four-way and worse for M = 0 mod 8: the padded image
    pl.pad = (M % 8) == 0 ? 1 : 0;
    pl.K16 = dw_padded_taps_len(K, M, pl.e, pl.pad);
    const int span = pl.HP + DW_W;
    pl.img = pl.pad ? (span + 2 * (span / M + 4) + 2 * DW_TRIP + 40) & ~1 : span + DW_PAD;
    pl.lds = sizeof(float2) * (size_t)(pl.img + DW_NG);
    int wpc = (int)((160 * 1024) / pl.lds);
    pl.wpc = wpc > 4 * DW_WAVES_PER_SIMD ? 4 * DW_WAVES_PER_SIMD : (wpc < 1 ? 1 : wpc);
    // runs of about 8 rows dealt to the waves in turn, every wave the same number of them where the chunk is long enough
    const int slots = ncu * pl.wpc;
    const int per_wave = (pl.nrows + slots - 1) / slots;
    const int nr = (per_wave + 7) / 8;
    pl.run_rows = nr > 0 ? (per_wave + nr - 1) / nr : 1;
    if (pl.run_rows < 1) pl.run_rows = 1;
    pl.nruns = (pl.nrows + pl.run_rows - 1) / pl.run_rows;
    pl.nwaves = pl.nruns < slots ? pl.nruns : slots;
}
extern "C" int dd_debug_decimw_plan(int64_t abs0, int64_t Ld, int K, int M, int off, int ncu, int64_t* out) {
    DD_REQUIRE(out && Ld >= 0 && ncu >= 1 && off >= 0 && off < M, "arguments");
    if (!dd_decimw_supported(K, M, 0, nullptr)) {
        dd_set_error("k_chain_decim_w takes even M in [8, 64] and 2 .. 256 taps");
        return DD_ERR_UNSUPPORTED;
    }
    DWPlan pl;
    decimw_plan(abs0, Ld, K, M, off, ncu, pl);
    out[0] = pl.R0; out[1] = pl.nrows; out[2] = pl.phi; out[3] = pl.HP; out[4] = pl.e; out[5] = pl.K16; out[6] = pl.wpc; out[7] = pl.run_rows;
    return DD_OK;
}

static const void* decimw_kernel(bool u8, bool nco, bool fm, bool pad) {
    static const void* const k[16] = {
        (const void*)k_chain_decim_w<false, false, false, false>, (const void*)k_chain_decim_w<true, false, false, false>,
        (const void*)k_chain_decim_w<false, true, false, false>,  (const void*)k_chain_decim_w<true, true, false, false>,
        (const void*)k_chain_decim_w<false, false, true, false>,  (const void*)k_chain_decim_w<true, false, true, false>,
        (const void*)k_chain_decim_w<false, true, true, false>,   (const void*)k_chain_decim_w<true, true, true, false>,
        (const void*)k_chain_decim_w<false, false, false, true>,  (const void*)k_chain_decim_w<true, false, false, true>,
        (const void*)k_chain_decim_w<false, true, false, true>,   (const void*)k_chain_decim_w<true, true, false, true>,
        (const void*)k_chain_decim_w<false, false, true, true>,   (const void*)k_chain_decim_w<true, false, true, true>,
        (const void*)k_chain_decim_w<false, true, true, true>,    (const void*)k_chain_decim_w<true, true, true, true>};
    return k[(u8 ? 1 : 0) | (nco ? 2 : 0) | (fm ? 4 : 0) | (pad ? 8 : 0)];
}

int dd_decimw_launch(const DDChainParams& P, const float* taps_g0, const double* taps_host, DDDecimWTaps* cache, hipStream_t stream) {
    if (P.Ld < 1 && !P.tail_out) return DD_OK;                 // (no kept sample: one wave, for the new history alone)
    const bool u8 = (P.flags & DD_CHAIN_U8_INPUT) != 0, nco = (P.flags & DD_CHAIN_NCO) != 0, fm = (P.flags & DD_CHAIN_FM) != 0;
    DWPlan pl;
    decimw_plan(P.abs0, P.Ld, P.K, P.M, P.off, dd_cu_count(), pl);
    static const char* run_env = getenv("DD_DECIMW_RUN");              // tools: rows per run
    if (run_env && atoi(run_env) > 0) {
        pl.run_rows = atoi(run_env);
        pl.nruns = (pl.nrows + pl.run_rows - 1) / pl.run_rows;
        const int slots = dd_cu_count() * pl.wpc;
        pl.nwaves = pl.nruns < slots ? pl.nruns : slots;
    }
    const float* taps = taps_g0 - pl.e;
    if (pl.pad) {
        // the padded taps of (M, e): a small device buffer kept with the filter, rewritten in stream order when the key changes
        const int key = (P.M << 1) | pl.e;
        const int cap = DD_DECIMW_TAPS_CAP;
        static_assert(256 + 1 + 2 * 32 + 2 * DW_TRIP <= DD_DECIMW_TAPS_CAP, "padded taps");
        if (!cache->dev) DD_HIP_CHECK(hipMalloc((void**)&cache->dev, sizeof(float) * cap));
        if (cache->key != key) {
            // (an earlier copy out of this array -- another key -- may still be queued on another stream: one caller thread and one stream per
            //  filter is the contract; on the same stream the runtime has staged pageable memory by the time the call returns)
            float* const t = cache->host;
            for (int j = 0; j < cap; ++j) t[j] = 0.f;
            for (int j = 0; j < P.K; ++j) {                    // g[j] = h[K-1-j] at logical window sample j + e
                const int w = j + pl.e;
                t[w + 2 * (w / P.M)] = (float)taps_host[P.K - 1 - j];
            }
            DD_HIP_CHECK(hipMemcpyAsync(cache->dev, t, sizeof(float) * cap, hipMemcpyHostToDevice, stream));
            cache->key = key;
        }
        taps = cache->dev;
    }
    DDDecimWArgs A;
    A.in = P.in; A.out = P.out;
    A.tail_in = P.tail_in; A.tail_out = P.tail_out; A.lasty_in = P.lasty_in; A.lasty_out = P.lasty_out;
    A.taps = taps;
    A.nco_tbl = P.nco_tbl;
    A.cyc = P.cyc; A.abs0 = P.abs0; A.L = P.L; A.Ld = P.Ld;
    A.R0 = pl.R0;
    A.nrows = pl.nrows; A.nwaves = pl.nwaves; A.run_rows = pl.run_rows;
    A.K = P.K; A.K16 = pl.K16; A.M = P.M; A.HP = pl.HP;
    A.e = pl.e; A.phi = pl.phi; A.off = P.off; A.s = P.s;
    A.cq = (DW_W - 1) / P.M + 1; A.cr = (DW_W - 1) % P.M;
    A.minv = (uint32_t)(0x100000000ull / (uint64_t)P.M) + 1u;
    A.img = pl.img;
    void* kargs[1] = {&A};
    DD_HIP_CHECK(hipLaunchKernel(decimw_kernel(u8, nco, fm, pl.pad != 0), dim3(pl.nwaves > 0 ? pl.nwaves : 1), dim3(64), kargs, pl.lds, stream));
    return DD_OK;
}
