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
    // block-sum form (k_chain_decim_b): taps = the MFMA operand table, e = the block starts one sample early (on an even LDS sample)
    int NI;                    // partial sums per output, ceil(K / M) <= 8
    int nh, h1lo;              // steps of eight samples per block; first step with a non-zero tap among partial sums 4 .. 7
    int F0, F1;                // rows [F0, F1) of the launch lie inside the chunk (whole 16-byte loads), the others reach past its ends
    int c0, d0, wm;            // (phi - R0 W) mod M; R0 W - (abs0 + off); W mod M: a row's first kept sample and output index without 64-bit divisions
    int small;                 // 1: fewer than 2^19 rows -- that arithmetic fits 32 bits
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
// the two halves of dw_cmul, for code that runs several products side by side (a dependent packed operation issued right behind the one it
// waits for costs a wait state: the staging loops run eight products in step)
__device__ __forceinline__ v2f dw_cmul_a(v2f a, v2f w) {
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    return t;
}
__device__ __forceinline__ v2f dw_cmul_b(v2f a, v2f w, v2f t) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// n products a[i] * w[i] in step (same two roundings per product as dw_cmul)
template <int N>
__device__ __forceinline__ void dw_cmul_n(v2f (&a)[N], const v2f (&w)[N]) {
    v2f t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = dw_cmul_a(a[i], w[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = dw_cmul_b(a[i], w[i], t[i]);
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
#ifdef DW_NO_LOADS      // (diagnostic: the kernel without its sample loads -- wrong outputs, the time of everything else)
#pragma unroll
    for (int j = 0; j < NJ; ++j) x[j] = (v4f_a8){(float)lane, 1.f, 2.f, (float)(J0 + j)};
#else
#pragma unroll
    for (int j = 0; j < NJ; ++j) x[j] = __builtin_nontemporal_load(p + 64 * (J0 + j));
#endif
}
template <bool NCO, bool PAD, int J0, int NJ>
__device__ __forceinline__ void dw_stage(const DDDecimWArgs& A, float2* buf, const float2* gl, int lane, const DWPh& pw, const v4f_a8 (&x)[NJ], const DWMap& mp) {
    // (the group phasors of all the row's loads first, in one batch: read one by one, each LDS read waits behind the 16-byte write in front
    //  of it and the staging of a row is sixteen LDS round trips -- 3400 of a row's 8100 cycles, profiles/r06_decimb_notes.txt)
    v2f g[NJ];
    if (NCO) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) g[jj] = dw_v2(gl[2 * (J0 + jj) + (lane >> 5)]);
    }
    // four loads = eight samples at a time: their phasors (pw.w x g) in step, then the rotations in step, then the four 16-byte writes
#ifndef DW_STAGE_NB
#define DW_STAGE_NB 4
#endif
    constexpr int NB = NJ < DW_STAGE_NB ? NJ : DW_STAGE_NB;
    static_assert(NJ % NB == 0, "loads per staging batch");
#pragma unroll
    for (int j0 = 0; j0 < NJ; j0 += NB) {
        v2f v[2 * NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            v[2 * u] = (v2f){x[j0 + u].x, x[j0 + u].y};
            v[2 * u + 1] = (v2f){x[j0 + u].z, x[j0 + u].w};
        }
        if (NCO) {
            v2f p[2 * NB], gg[2 * NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                p[2 * u] = pw.w[0];
                p[2 * u + 1] = pw.w[1];
                gg[2 * u] = gg[2 * u + 1] = g[j0 + u];
            }
            dw_cmul_n<2 * NB>(p, gg);
            dw_cmul_n<2 * NB>(v, p);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int r = A.HP + 128 * (J0 + j0 + u) + 2 * lane;
            *reinterpret_cast<v4f*>(buf + dw_pos<PAD>(mp, r)) = (v4f){v[2 * u].x, v[2 * u].y, v[2 * u + 1].x, v[2 * u + 1].y};
        }
    }
}
// the same through guarded sample-by-sample loads, loads jlo .. 15
template <bool NCO, bool PAD, bool U8 = false>
__device__ __forceinline__ void dw_stage_guarded(const DDDecimWArgs& A, float2* buf, const float2* gl, int lane, int jlo, int64_t Brel, const DWPh& pw, const DWMap& mp) {
    // (round 6: a version that issues a batch's loads on clamped indices before looking at any of them -- no dependent round trip per position --
    //  changed nothing in a chunk loop's time and made the INTERIOR rows of the two-accumulator-set kernels 2-7 % slower by what it did to the
    //  register allocation of the whole kernel: C4 0.0901 -> 0.0920, raw u8 0.0601 -> 0.0644 ms, same call; profiles/r06_decimb_notes.txt.  Not kept.)
    for (int j = jlo; j < DW_NL; ++j) {
        v2f pj = (v2f){1.f, 0.f}, pj1 = (v2f){1.f, 0.f};
        if (NCO) {
            const v2f g = dw_v2(gl[2 * j + (lane >> 5)]);
            pj = dw_cmul(pw.w[0], g);
            pj1 = dw_cmul(pw.w[1], g);
        }
        const int64_t n = Brel + 128 * j + 2 * lane;
        const v2f x0 = dw_sample<U8, NCO>(A, n, pj), x1 = dw_sample<U8, NCO>(A, n + 1, pj1);
        *reinterpret_cast<v4f*>(buf + dw_pos<PAD>(mp, A.HP + 128 * j + 2 * lane)) = (v4f){x0.x, x0.y, x1.x, x1.y};
    }
}

// ---- raw u8 rows of the block-sum kernel (source.py:117-118), four bytes per lane: load j, lane l = samples 128 j + 2 l, + 1 of the block -- the
// complex64 rows' layout, so the 16-byte LDS writes of a load cover 1 KB in lane order.  (Sixteen bytes per lane, dw_issue8: lane l writes
// its eight samples 64 bytes from lane l + 1's -- every write a four-way bank conflict, 60 % of the LDS cycles of the launch;
// profiles/r06_decimb_notes.txt)
typedef uint32_t u32_a2 __attribute__((aligned(2)));
template <int J0, int NJ>
__device__ __forceinline__ void dw_issue4(const DDDecimWArgs& A, int64_t Brel, int lane, uint32_t (&x)[NJ]) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(A.in) + 2 * (Brel + 2 * lane);
#ifdef DW_NO_LOADS
#pragma unroll
    for (int j = 0; j < NJ; ++j) x[j] = (uint32_t)lane * 0x01010101u + (uint32_t)(J0 + j);
#else
#pragma unroll
    for (int j = 0; j < NJ; ++j) x[j] = __builtin_nontemporal_load(reinterpret_cast<const u32_a2*>(p + 256 * (J0 + j)));
#endif
}
template <bool NCO, bool PAD, int J0, int NJ>
__device__ __forceinline__ void dw_stage4(const DDDecimWArgs& A, float2* buf, const float2* gl, int lane, const DWPh& pw, const uint32_t (&x)[NJ], const DWMap& mp) {
    v2f g[NJ];
    if (NCO) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) g[jj] = dw_v2(gl[2 * (J0 + jj) + (lane >> 5)]);
    }
    constexpr int NB = NJ < DW_STAGE_NB ? NJ : DW_STAGE_NB;
    static_assert(NJ % NB == 0, "loads per staging batch");
#pragma unroll
    for (int j0 = 0; j0 < NJ; j0 += NB) {
        v2f v[2 * NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const uint32_t d = x[j0 + u];
            v[2 * u] = (v2f){(float)(d & 0xff) - 127.5f, (float)((d >> 8) & 0xff) - 127.5f};
            v[2 * u + 1] = (v2f){(float)((d >> 16) & 0xff) - 127.5f, (float)(d >> 24) - 127.5f};
        }
        if (NCO) {
            v2f p[2 * NB], gg[2 * NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                p[2 * u] = pw.w[0];
                p[2 * u + 1] = pw.w[1];
                gg[2 * u] = gg[2 * u + 1] = g[j0 + u];
            }
            dw_cmul_n<2 * NB>(p, gg);
            dw_cmul_n<2 * NB>(v, p);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int r = A.HP + 128 * (J0 + j0 + u) + 2 * lane;
            *reinterpret_cast<v4f*>(buf + dw_pos<PAD>(mp, r)) = (v4f){v[2 * u].x, v[2 * u].y, v[2 * u + 1].x, v[2 * u + 1].y};
        }
    }
}

// ---- raw u8 rows (source.py:117-118): load j, lane l = samples 512 j + 8 l .. + 7 of the block
template <int J0, int NJ>
__device__ __forceinline__ void dw_issue8(const DDDecimWArgs& A, int64_t Brel, int lane, v4u_a2 (&x)[NJ]) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(A.in) + 2 * (Brel + 8 * lane);
#ifdef DW_NO_LOADS
#pragma unroll
    for (int j = 0; j < NJ; ++j) x[j] = (v4u_a2){(uint32_t)lane * 0x01010101u, 0x80807f7fu, 0x7f808180u, (uint32_t)(J0 + j)};
#else
#pragma unroll
    for (int j = 0; j < NJ; ++j) x[j] = __builtin_nontemporal_load(reinterpret_cast<const v4u_a2*>(p + 1024 * (J0 + j)));
#endif
}
template <bool NCO, bool PAD, int J0, int NJ>
__device__ __forceinline__ void dw_stage8(const DDDecimWArgs& A, float2* buf, const float2* gl, int lane, const DWPh& pw, const v4u_a2 (&x)[NJ], const DWMap& mp) {
    v2f gs[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) gs[jj] = NCO ? dw_v2(gl[8 * (J0 + jj) + (lane >> 3)]) : (v2f){1.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = J0 + jj;
        const v2f g = gs[jj];
        const uint32_t d[4] = {x[jj].x, x[jj].y, x[jj].z, x[jj].w};
        v2f v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = (v2f){(float)(d[k] & 0xff) - 127.5f, (float)((d[k] >> 8) & 0xff) - 127.5f};
            v[2 * k + 1] = (v2f){(float)((d[k] >> 16) & 0xff) - 127.5f, (float)(d[k] >> 24) - 127.5f};
        }
        if (NCO) {
            v2f p[8], gg[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                p[k] = pw.w[k];
                gg[k] = g;
            }
            dw_cmul_n<8>(p, gg);
            dw_cmul_n<8>(v, p);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = A.HP + 512 * j + 8 * lane + 2 * k;
            *reinterpret_cast<v4f*>(buf + dw_pos<PAD>(mp, r)) = (v4f){v[2 * k].x, v[2 * k].y, v[2 * k + 1].x, v[2 * k + 1].y};
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
#define DW_NPH 7
#define DW_NTR (DW_NPH + 6)     // + rows, whole kernel, kernel start -> first run, run start -> row before staged, the row before's outputs, runs
__device__ unsigned long long g_dw_trace[4096 * DW_NTR];
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

// the new carried history (the chunk's last K-1 samples after the NCO, older ones from the old history): the value a row gives a sample,
// recomputed -- its row, its group of 64 and its place in the group follow from its absolute index
template <bool U8, bool NCO>
__device__ __forceinline__ void dw_new_tail(const DDDecimWArgs& A, int lane) {
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
    if (gw == 0 && A.tail_out) dw_new_tail<U8, NCO>(A, lane);
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
        for (int i = 0; i < DW_NPH; ++i) g_dw_trace[gw * DW_NTR + i] = tr[i];
        g_dw_trace[gw * DW_NTR + DW_NPH] = trows;
        g_dw_trace[gw * DW_NTR + DW_NPH + 1] = (unsigned)__builtin_readcyclecounter() - tstart;
    }
#endif
}

#ifdef DW_TRACE
extern "C" int dd_debug_decimw_trace(unsigned long long* out, int nwaves) {
    DD_HIP_CHECK(hipDeviceSynchronize());
    DD_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dw_trace), sizeof(unsigned long long) * (size_t)nwaves * DW_NTR));
    return DD_OK;
}
#endif

// ============================================================================ block-sum form (round 6)
// k_chain_decim_w's tap loop reads every staged sample K / M times (2.5-4.4 for the reference's /34 and /50): 80 sixteen-byte LDS reads and one
// scalar tap load per sixteen taps, each trip waiting for both -- a wave alone on its SIMD spends 2400 cycles per row in it where its 160 packed
// multiply-adds need 640 (profiles/r05_decim_taps_ubench.txt).  Here every staged sample is read ONCE: the M samples that end at a kept sample
// form the BLOCK of its lane, and the FIR output is a sum of block sums,
//
//     y[g] = sum_i P_i[g - i],   P_i[b] = sum_j h[M i + (M - 1 - j)] x_b[j]        (i < NI = ceil(K / M), h = 0 beyond K - 1)
//
// -- output g takes its newest M taps from its own block, the next M from the block before it, and so on.  A lane forms the NI sums of ITS block
// (the same samples against NI different tap sets) and the sums travel up the lanes: H = P_{NI-1}; H = P_s + (H of the lane below), s = NI-2 .. 0
// (DPP wave_shr:1; lane 0 takes the values the last lane of the row before left behind, kept in scalar registers; a run of rows starts by
// forming them from the last outputs of the row before it).  The block sums are v_mfma_f32_4x4x1_16b_f32: sixteen 4 x 4 outer products
// D_t[m][n] += A_t[m] B_t[n] per instruction, B = the lanes' samples (lane 4 t + n: its own block's sample j, re or im), A broadcast from ONE of
// the sixteen A blocks (cbsz = 4, abid = j mod 16: lane 4 (j mod 16) + m of tap register j / 16 holds h[M m + (M - 1 - j)]) -- so register m of
// a lane accumulates P_m of its block, four partial sums per instruction, one exact fmaf per element (tools/ubench/mfma4x4_layout.hip: layout,
// broadcast and rounding checked against fmaf bit for bit), and the taps of a whole block sit in at most ten registers for the whole launch:
// no tap traffic at all, LDS reads a fifth of the window form's, and the matrix pipe -- idle in every other phase -- does the multiply-adds
// beside the other wave's vector work.  Partial sums 4 .. 7 (K > 4 M) take a second accumulator set, over the samples whose taps there are
// not zero.  The order of the additions of an output depends on (K, M) only -- blocks are counted from the output, not from the row -- so
// chunked and one-call runs and raw u8 / complex64 input agree bit for bit as before.  Where the block would start on an odd LDS sample
// (phi even) it starts one sample early: the reads then cover [kept - M, kept - 1] (the first one under a zero tap) and one more read fetches
// the kept sample.  LDS image: [M samples of halo | 2048 | zeros]; M = 0 mod 4: two samples of gap after every block (lane stride M + 2
// samples: conflict-free 16-byte reads), the gaps are never read.
struct DWCarry {
    v2f h[7];                  // H_{s+1} of the output before the pass's first one, s = 0 .. 6
};

template <int NG>
struct DWAcc {
    // [group][m]: P_{4 group + m}, re / im -- each as TWO chains, the block's even and odd samples (r, i and r1, i1), added at the end: a
    // chain of M / 2 multiply-adds rounds less than one of M (the whole 2^26-sample C3 run against float64: 1.3e-4 -> see
    // tests/test_gpu_fullsize.py), and four to eight independent chains need no wait state between dependent matrix instructions
    v4f r[NG], i[NG], r1[NG], i1[NG];
};

// the block sums of the lane's block (blk: its first sample, even), eight samples (four 16-byte reads) per step, straight-line code for up to
// nine steps: the reads of the step after next are requested before a step's products (the LDS answers after ~300 cycles when eight waves
// stage and read at once; a step's 16-32 products cover 130-270).  The reads are inline asm with the waits placed by hand: as plain loads the
// compiler sinks each one behind the wave-uniform exit in front of its first use -- read, wait, use, nine times the LDS latency per row.  A
// wait for "at most n reads outstanding" with n = the reads requested AFTER the ones needed is safe whatever else the compiler has in
// flight on the same counter (LDS answers in order: while a needed read is out, so are the n behind it).  No read is left in flight when
// the function returns (its register would be free for reuse), none runs past the block.
template <bool PAD>
struct DWOct {
    v4f x[4];
};
template <int OFF>
__device__ __forceinline__ v4f dw_lds_read16(uint32_t addr) {
    v4f r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <bool PAD, int H>
__device__ __forceinline__ void dw_bo_load(uint32_t addr, int M, DWOct<PAD>& d) {
    if constexpr (PAD) {
        // (the kept sample of a block that starts one sample early, j = M, sits behind the gap: sixteen bytes further on)
        d.x[0] = dw_lds_read16<64 * H>(addr + (8 * H >= M ? 16u : 0u));
        d.x[1] = dw_lds_read16<64 * H + 16>(addr + (8 * H + 2 >= M ? 16u : 0u));
        d.x[2] = dw_lds_read16<64 * H + 32>(addr + (8 * H + 4 >= M ? 16u : 0u));
        d.x[3] = dw_lds_read16<64 * H + 48>(addr + (8 * H + 6 >= M ? 16u : 0u));
    } else {
        d.x[0] = dw_lds_read16<64 * H>(addr);
        d.x[1] = dw_lds_read16<64 * H + 16>(addr);
        d.x[2] = dw_lds_read16<64 * H + 32>(addr);
        d.x[3] = dw_lds_read16<64 * H + 48>(addr);
    }
}
template <int N, bool PAD>
__device__ __forceinline__ void dw_bo_wait(DWOct<PAD>& d) {           // the step's samples have arrived once at most N younger reads are out
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(d.x[0]), "+v"(d.x[1]), "+v"(d.x[2]), "+v"(d.x[3]) : "n"(N));
}
// two samples (one 16-byte read) into accumulator set S: block samples j0, j0 + 1 = A blocks J0, J0 + 1 of the tap register
template <int NG, int S, int J0>
__device__ __forceinline__ void dw_bo_mac2(DWAcc<NG>& a, float t, v4f x) {
    a.r[S] = __builtin_amdgcn_mfma_f32_4x4x1f32(t, x.x, a.r[S], 4, J0, 0);
    a.i[S] = __builtin_amdgcn_mfma_f32_4x4x1f32(t, x.y, a.i[S], 4, J0, 0);
    a.r1[S] = __builtin_amdgcn_mfma_f32_4x4x1f32(t, x.z, a.r1[S], 4, J0 + 1, 0);
    a.i1[S] = __builtin_amdgcn_mfma_f32_4x4x1f32(t, x.w, a.i1[S], 4, J0 + 1, 0);
}
template <int NG, int H, bool PAD>
__device__ __forceinline__ void dw_bo_mac(DWAcc<NG>& a, const DWOct<PAD>& d, float ta, float tb, bool g1) {
    constexpr int J = (8 * H) & 15;
    dw_bo_mac2<NG, 0, J>(a, ta, d.x[0]);
    dw_bo_mac2<NG, 0, J + 2>(a, ta, d.x[1]);
    dw_bo_mac2<NG, 0, J + 4>(a, ta, d.x[2]);
    dw_bo_mac2<NG, 0, J + 6>(a, ta, d.x[3]);
    if constexpr (NG == 2) {
        if (g1) {
            dw_bo_mac2<NG, 1, J>(a, tb, d.x[0]);
            dw_bo_mac2<NG, 1, J + 2>(a, tb, d.x[1]);
            dw_bo_mac2<NG, 1, J + 4>(a, tb, d.x[2]);
            dw_bo_mac2<NG, 1, J + 6>(a, tb, d.x[3]);
        }
    }
}
// the first reads of a block's sums, requested ahead of time (k_chain_decim_b: before it issues the next row's global loads -- the reads
// queue behind the row's sixteen LDS writes and need ~400 cycles, the address arithmetic of the loads hides them)
template <bool PAD, int NG>
struct DWPre {
    DWOct<PAD> b0, b1;
};
template <bool PAD, int NG>
__device__ __forceinline__ void dw_bsums_begin(const DDDecimWArgs& A, const float2* __restrict__ blk, DWPre<PAD, NG>& pre) {
    int M = A.M;
    asm volatile("" : "+s"(M));
    const uint32_t addr = (uint32_t)(uintptr_t)blk;            // (LDS: the low 32 bits of the generic address are the byte offset)
    dw_bo_load<PAD, 0>(addr, M, pre.b0);
    if constexpr (NG == 1) dw_bo_load<PAD, 1>(addr, M, pre.b1);
}
template <bool PAD, int NG>
__device__ __forceinline__ void dw_bsums(const DDDecimWArgs& A, const float2* __restrict__ blk, DWPre<PAD, NG>& pre, const float (&ta)[5], const float (&tb)[5], DWAcc<NG>& acc) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        acc.r[g] = (v4f){0.f, 0.f, 0.f, 0.f};
        acc.i[g] = (v4f){0.f, 0.f, 0.f, 0.f};
        acc.r1[g] = (v4f){0.f, 0.f, 0.f, 0.f};
        acc.i1[g] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
    // (through an empty asm: left visible as launch constants the comparisons are hoisted out of the row loop into scalar registers that spill)
    int nh = A.nh, h1lo = A.h1lo, M = A.M;
    asm volatile("" : "+s"(nh), "+s"(h1lo), "+s"(M));
    const uint32_t addr = (uint32_t)(uintptr_t)blk;
    // every step requests a later one without asking whether the block has it (reads past the block's end -- two steps at most -- stay inside
    // the image and are never used): one wait per step, the same on every path.  Two accumulator sets: a step's 32 products cover the LDS
    // latency, the next step's reads are enough (32 registers); one set: the step after next (48)
    // (two sets with the step after next requested as well: the same times, same call -- 0.0904 / 0.0969 / 0.0649 against 0.0905 / 0.0969 / 0.0646)
    if constexpr (NG == 2) {
        DWOct<PAD>& b0 = pre.b0;
        DWOct<PAD>& b1 = pre.b1;
#define DW_BSTEP(H, CUR, NXT)                                                        \
        dw_bo_load<PAD, (H) + 1>(addr, M, NXT);                                      \
        dw_bo_wait<4, PAD>(CUR);                                                     \
        dw_bo_mac<NG, H, PAD>(acc, CUR, ta[(H) >> 1], tb[(H) >> 1], (H) >= h1lo);    \
        if ((H) + 1 >= nh) break;
        do {
            DW_BSTEP(0, b0, b1)
            DW_BSTEP(1, b1, b0)
            DW_BSTEP(2, b0, b1)
            DW_BSTEP(3, b1, b0)
            DW_BSTEP(4, b0, b1)
            DW_BSTEP(5, b1, b0)
            DW_BSTEP(6, b0, b1)
            DW_BSTEP(7, b1, b0)
            DW_BSTEP(8, b0, b1)
        } while (0);
#undef DW_BSTEP
        // the request past the last step lands in registers nobody reads: held until it has landed
        asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(b0.x[0]), "v"(b0.x[1]), "v"(b0.x[2]), "v"(b0.x[3]), "v"(b1.x[0]), "v"(b1.x[1]), "v"(b1.x[2]), "v"(b1.x[3]));
    } else {
        DWOct<PAD>& b0 = pre.b0;
        DWOct<PAD>& b1 = pre.b1;
        DWOct<PAD> b2;
#define DW_BSTEP(H, CUR, NXT)                                                        \
        dw_bo_load<PAD, (H) + 2>(addr, M, NXT);                                      \
        dw_bo_wait<8, PAD>(CUR);                                                     \
        dw_bo_mac<NG, H, PAD>(acc, CUR, ta[(H) >> 1], tb[(H) >> 1], (H) >= h1lo);    \
        if ((H) + 1 >= nh) break;
        do {
            DW_BSTEP(0, b0, b2)
            DW_BSTEP(1, b1, b0)
            DW_BSTEP(2, b2, b1)
            DW_BSTEP(3, b0, b2)
            DW_BSTEP(4, b1, b0)
            DW_BSTEP(5, b2, b1)
            DW_BSTEP(6, b0, b2)
            DW_BSTEP(7, b1, b0)
            DW_BSTEP(8, b2, b1)
        } while (0);
#undef DW_BSTEP
        asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(b0.x[0]), "v"(b0.x[1]), "v"(b0.x[2]), "v"(b0.x[3]), "v"(b1.x[0]), "v"(b1.x[1]), "v"(b1.x[2]), "v"(b1.x[3]),
                     "v"(b2.x[0]), "v"(b2.x[1]), "v"(b2.x[2]), "v"(b2.x[3]));
    }
}

// the LDS layout of a PAD row of the block-sum form: gaps counted from its first block's start
__device__ __forceinline__ int dw_b_first(const DDDecimWArgs& A, const DWRow& r) { return A.HP + r.r0 - A.M + 1 - A.e; }
__device__ __forceinline__ DWMap dw_b_row_map(const DDDecimWArgs& A, const DWRow& r) {
    const int bs0 = dw_b_first(A, r);
    const int k = (int)__umulhi((uint32_t)bs0, A.minv);
    return DWMap{bs0 - k * A.M - A.M, A.minv};
}

// a staged row of the block-sum form: block sums, the sums travel up the lanes, the outputs leave, the halo moves down.  cy: the partial
// sums the row before left for this row's first outputs (in), this row's for the next (out); ycarry as in dw_row_outputs.  emit false: the
// row before a run -- one pass over its LAST 64 outputs, for cy, ycarry and the halo.
// the block of the lane's output in pass t of a row (emit false: the one pass over the row's last 64 outputs)
template <bool PAD>
__device__ __forceinline__ const float2* dw_b_block(const DDDecimWArgs& A, const float2* buf, int lane, const DWRow& r, bool emit, int t) {
    const int bs0 = dw_b_first(A, r);
    const int bsp = PAD ? bs0 + 2 * ((int)__umulhi((uint32_t)bs0, A.minv) + 1) : bs0;
    const int bstep = PAD ? A.M + 2 : A.M;
    const int i = emit ? 64 * t + lane : r.cnt - 64 + lane;
    const int ic = i < 0 ? 0 : (i < r.cnt ? i : r.cnt - 1);
    return buf + bsp + ic * bstep;
}
// pre0: the first pass's first reads have been requested (dw_bsums_begin on dw_b_block(.., 0))
template <bool FM, bool PAD, int NG>
__device__ __forceinline__ void dw_b_row_outputs(const DDDecimWArgs& A, float2* buf, int lane, const DWRow& r, const DWMap& mp, bool emit, v2f& ycarry,
                                                 DWCarry& cy, v2f ylast_in, const float (&ta)[5], const float (&tb)[5], DWPre<PAD, NG>& pre0
#ifdef DW_TRACE
                                                 , unsigned* tr = nullptr, unsigned tprev = 0
#endif
                                                 ) {
#ifdef DW_TRACE
    unsigned trd[DW_NPH];
    if (!tr) tr = trd;
#endif
    v4f hl[2];
    DWMap mpn = mp;
    if (PAD) {
        DWRow rn = r;
        dw_row_next(A, rn);
        mpn = dw_b_row_map(A, rn);
    }
    const int ng = emit ? (r.cnt + 63) >> 6 : 1;
    for (int t = 0; t < ng; ++t) {
        const int i = emit ? 64 * t + lane : r.cnt - 64 + lane;
        const float2* const blk = dw_b_block<PAD>(A, buf, lane, r, emit, t);
        DWAcc<NG> acc;
        if (t > 0) dw_bsums_begin<PAD, NG>(A, blk, pre0);
        dw_bsums<PAD, NG>(A, blk, pre0, ta, tb, acc);
        DW_T(6);
        if (t == ng - 1) dw_halo_read<PAD>(A, buf, lane, hl, mp);
        const int last = (emit && t == ng - 1) ? (r.cnt - 1) & 63 : 63;
        // y = P_0 + (P_1[lane - 1] + (P_2[lane - 2] + ...)): the sums move up one lane per step
        v2f y = (v2f){0.f, 0.f};
        int NI = A.NI;
        asm volatile("" : "+s"(NI));
#pragma unroll
        for (int s = 4 * NG - 1; s >= 0; --s) {
            if (s < NI) {
                const v2f P = (v2f){acc.r[s >> 2][s & 3] + acc.r1[s >> 2][s & 3], acc.i[s >> 2][s & 3] + acc.i1[s >> 2][s & 3]};
                if (s + 1 < NI) {
                    const v2f cin = cy.h[s];
                    cy.h[s] = dw_lane(y, last);
                    y = (v2f){P.x + dw_shr1(y.x, cin.x), P.y + dw_shr1(y.y, cin.y)};
                } else {
                    y = P;
                }
            }
        }
        DW_T(2);
        const int64_t p = r.p0 + i;
        if (FM) {
            v2f yp = (v2f){dw_shr1(y.x, ycarry.x), dw_shr1(y.y, ycarry.y)};
            ycarry = dw_lane(y, last);
            if (emit && i < r.cnt && p >= 0 && p < A.Ld) {
                if (p == 0 && A.s == 0) yp = ylast_in;
                if (p >= A.s) {
                    const float re = fmaf(y.x, yp.x, y.y * yp.y), im = fmaf(y.y, yp.x, -y.x * yp.y);
                    reinterpret_cast<float*>(A.out)[p - A.s] = dd_atan2_poly(im, re);
                }
                if (p == A.Ld - 1 && A.lasty_out) *A.lasty_out = make_float2(y.x, y.y);
            }
        } else {
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

template <bool U8, bool NCO, bool FM, bool PAD, int NG>
__global__ void __launch_bounds__(64, DW_WAVES_PER_SIMD) k_chain_decim_b(const DDDecimWArgs A) {
    extern __shared__ __attribute__((aligned(16))) char dw_smem[];
    float2* const buf = reinterpret_cast<float2*>(dw_smem);
    float2* const gl = buf + A.img;
    const int lane = threadIdx.x;
    const int gw = blockIdx.x;
    const int M = A.M;
#ifdef DW_TRACE
    const unsigned tstart = (unsigned)__builtin_readcyclecounter();
#endif
    const int RR = A.run_rows;
    const int nruns = (A.nrows + RR - 1) / RR;
    v4f_a8 x[DW_NL];
    uint32_t x8[DW_NL];
    auto brel = [&](int q) { return (A.R0 + q) * (int64_t)DW_W - A.abs0; };
    auto inside = [&](int q) { return q >= A.F0 && q < A.F1; };
    auto row_phasor = [&](int q) { return NCO ? dw_phasor_u((uint64_t)((A.R0 + q) * (int64_t)DW_W) * A.cyc, A.nco_tbl) : (v2f){1.f, 0.f}; };
    // rows [q0, f0) and [f1, q1) of a run reach outside the chunk, [f0, f1) lie inside it; pin: so does the row before the run
    struct Run { int q0, q1, f0, f1; bool pin; };
    auto run_of = [&](int run) {
        Run u;
        u.q0 = run * RR;
        u.q1 = u.q0 + RR < A.nrows ? u.q0 + RR : A.nrows;
        u.f0 = u.q0 > A.F0 ? u.q0 : A.F0;
        if (u.f0 > u.q1) u.f0 = u.q1;
        u.f1 = u.q1 < A.F1 ? u.q1 : A.F1;
        if (u.f1 < u.f0) u.f1 = u.f0;
        u.pin = inside(u.q0 - 1);
        return u;
    };
    v4f_a8 xp[4];
    uint32_t xp8[4];
    // what a run starts from: the end of the row before it (the last four loads of that row, u8: the last one) and its first row inside the
    // chunk.  (Tried: requested while the wave's run BEFORE works on its last row -- 5-10 % slower, two far-apart
    // address streams per wave; profiles/r06_decimb_notes.txt)
    auto issue_start = [&](const Run& u) {
        if (u.pin) {
            if constexpr (U8) dw_issue4<DW_NL - 4, 4>(A, brel(u.q0 - 1), lane, xp8); else dw_issue<DW_NL - 4, 4>(A, brel(u.q0 - 1), lane, xp);
        }
        if (u.f1 > u.f0) {
            if constexpr (U8) dw_issue4<0, DW_NL>(A, brel(u.f0), lane, x8); else dw_issue<0, DW_NL>(A, brel(u.f0), lane, x);
        }
    };
    if (gw < nruns) issue_start(run_of(gw));                   // (first thing: the tables below are built while these fly)
    // (nothing outside the staged samples is ever read with a non-zero tap, and the gaps of a PAD image not at all; the zeros behind the block
    //  are what a block that starts one sample early reads behind the row's last kept sample)
    if (lane < 2 * DW_PAD) buf[A.img - 2 * DW_PAD + lane] = make_float2(0.f, 0.f);
    // the taps of a block as the matrix instruction takes them (host: dd_decimw_launch): register g, lane 4 t + m = the tap of block sample
    // 16 g + t in partial sum m (ta) resp. 4 + m (tb) -- the instruction broadcasts the A block t = j mod 16 it is told to; for the whole launch
    float ta[5], tb[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        ta[g] = A.taps[64 * g + lane];
        tb[g] = NG == 2 ? A.taps[320 + 64 * g + lane] : 0.f;
    }
    DWPh ph;
#pragma unroll
    for (int k = 0; k < 8; ++k) ph.w[k] = (v2f){1.f, 0.f};
    if (NCO) {
#pragma unroll
        for (int k = 0; k < 2; ++k) ph.w[k] = dw_phasor_v((uint64_t)(((2 * lane) & 63) + k) * A.cyc, A.nco_tbl);
        if (lane < DW_NG) {
            const v2f g = dw_phasor_v((uint64_t)(64 * lane) * A.cyc, A.nco_tbl);
            gl[lane] = make_float2(g.x, g.y);
        }
    }
    if (gw == 0 && A.tail_out) dw_new_tail<U8, NCO>(A, lane);
    v2f ylast_in = (v2f){0.f, 0.f};
    if (FM && A.s == 0) ylast_in = dw_v2(*A.lasty_in);
    // the part of the row before a run that the run needs: the blocks of its last NI outputs (NI M < K + M samples back from the last kept one)
    const int rmin = DW_W - M - A.NI * M - 2;
    const int jlo = (rmin > 0 ? rmin : 0) >> 7;
#ifdef DW_TRACE
    unsigned tr[DW_NPH];
#pragma unroll
    for (int i = 0; i < DW_NPH; ++i) tr[i] = 0;
    unsigned trows = 0;
#endif
#ifdef DW_TRACE
    unsigned tx[4] = {0, 0, 0, 0};
    tx[0] = (unsigned)__builtin_readcyclecounter() - tstart;
#endif
    for (int run = gw; run < nruns; run += A.nwaves) {
#ifdef DW_TRACE
        const unsigned trun = (unsigned)__builtin_readcyclecounter();
#endif
        const Run cur = run_of(run);
        const int q0 = cur.q0, q1 = cur.q1, f0 = cur.f0, f1 = cur.f1;
        const bool pin = cur.pin;
        if (run != gw) issue_start(cur);
        DWRow r;
        DWMap mp = DWMap{0, A.minv};
        {
            // the row before the run, q = q0 - 1 >= -1: offset of its first kept sample, outputs before it (a 64-bit remainder and quotient
            // cost ~1000 cycles at every run start; with the launch's row 0 as the origin 32 bits do for 2^19 rows)
            if (A.small) {
                const uint32_t u = ((uint32_t)q0 * (uint32_t)A.wm) % (uint32_t)M;
                int m = A.c0 + A.wm - (int)u;                        // in (-M, 2 M)
                if (m >= M) m -= M;
                if (m < 0) m += M;
                r.r0 = m;
                r.cnt = (DW_W - 1 - r.r0) / M + 1;
                r.p0 = ((q0 - 1) * DW_W + A.d0 + r.r0) / M;          // (exact: a kept sample's distance from the chunk's first one)
            } else {
                const int64_t B = (A.R0 + q0 - 1) * (int64_t)DW_W;
                int64_t m = ((int64_t)A.phi - B) % M;
                if (m < 0) m += M;
                r.r0 = (int)m;
                r.cnt = (DW_W - 1 - r.r0) / M + 1;
                r.p0 = (B + r.r0 - A.abs0 - A.off) / M;
            }
            if (PAD) mp = dw_b_row_map(A, r);
            DWPh pw;
            dw_row_ph<false, NCO>(row_phasor(q0 - 1), ph, pw);
            if (pin) {
                if constexpr (U8) dw_stage4<NCO, PAD, DW_NL - 4, 4>(A, buf, gl, lane, pw, xp8, mp);
                else dw_stage<NCO, PAD, DW_NL - 4, 4>(A, buf, gl, lane, pw, xp, mp);
            } else {
                if constexpr (U8) dw_stage_guarded<NCO, PAD, true>(A, buf, gl, lane, jlo, brel(q0 - 1), pw, mp);
                else dw_stage_guarded<NCO, PAD>(A, buf, gl, lane, jlo, brel(q0 - 1), pw, mp);
            }
        }
        v2f ycarry = (v2f){0.f, 0.f};
        DWCarry cy;
#pragma unroll
        for (int s = 0; s < 7; ++s) cy.h[s] = (v2f){0.f, 0.f};
#ifdef DW_TRACE
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // (lgkmcnt(0) only: the staged samples are in LDS; the loads stay in flight)
        const unsigned tpre = (unsigned)__builtin_readcyclecounter();
        tx[1] += tpre - trun;
#endif
        DWPre<PAD, NG> pre;
        dw_bsums_begin<PAD, NG>(A, dw_b_block<PAD>(A, buf, lane, r, false, 0), pre);
        dw_b_row_outputs<FM, PAD, NG>(A, buf, lane, r, mp, false, ycarry, cy, ylast_in, ta, tb, pre);
#ifdef DW_TRACE
        __builtin_amdgcn_s_waitcnt(0xc07f);
        tx[2] += (unsigned)__builtin_readcyclecounter() - tpre;
        ++tx[3];
#endif
        for (int q = q0; q < q1; ++q) {
#ifdef DW_TRACE
            unsigned tprev = (unsigned)__builtin_readcyclecounter();
#endif
            const bool fast = q >= f0 && q < f1;
            dw_row_next(A, r);
            if (PAD) mp = dw_b_row_map(A, r);
            DWPh pw;
            dw_row_ph<false, NCO>(row_phasor(q), ph, pw);
            DW_T(5);
            if (fast) {
                // (tried: a second set of sample registers, the next row requested BEFORE this one is staged so that the wave always has a
                //  request in flight -- no gain, the launch sits on the memory system's rate, not on the bytes in flight; r06_decimb_notes.txt)
                if constexpr (U8) dw_stage4<NCO, PAD, 0, DW_NL>(A, buf, gl, lane, pw, x8, mp);
                else dw_stage<NCO, PAD, 0, DW_NL>(A, buf, gl, lane, pw, x, mp);
                DW_T(0);
                // (the block sums' first LDS reads ahead of the loads' address arithmetic)
                dw_bsums_begin<PAD, NG>(A, dw_b_block<PAD>(A, buf, lane, r, true, 0), pre);
                if (q + 1 < f1) {
                    // the next row's samples fly during this row's block sums
                    if constexpr (U8) dw_issue4<0, DW_NL>(A, brel(q + 1), lane, x8); else dw_issue<0, DW_NL>(A, brel(q + 1), lane, x);
                }
            } else {
                if constexpr (U8) dw_stage_guarded<NCO, PAD, true>(A, buf, gl, lane, 0, brel(q), pw, mp);
                else dw_stage_guarded<NCO, PAD>(A, buf, gl, lane, 0, brel(q), pw, mp);
                dw_bsums_begin<PAD, NG>(A, dw_b_block<PAD>(A, buf, lane, r, true, 0), pre);
            }
#ifdef DW_TRACE
            if (fast) {
                { __builtin_amdgcn_sched_barrier(0); const unsigned t_ = (unsigned)__builtin_readcyclecounter(); tr[1] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); }
                ++trows;
                dw_b_row_outputs<FM, PAD, NG>(A, buf, lane, r, mp, true, ycarry, cy, ylast_in, ta, tb, pre, tr, tprev);
            } else
#endif
            dw_b_row_outputs<FM, PAD, NG>(A, buf, lane, r, mp, true, ycarry, cy, ylast_in, ta, tb, pre);
        }
    }
#ifdef DW_TRACE
    if (gw < 4096 && lane == 0) {
#pragma unroll
        for (int i = 0; i < DW_NPH; ++i) g_dw_trace[gw * DW_NTR + i] = tr[i];
        g_dw_trace[gw * DW_NTR + DW_NPH] = trows;
        g_dw_trace[gw * DW_NTR + DW_NPH + 1] = (unsigned)__builtin_readcyclecounter() - tstart;
#pragma unroll
        for (int i = 0; i < 4; ++i) g_dw_trace[gw * DW_NTR + DW_NPH + 2 + i] = tx[i];
    }
#endif
}

// ============================================================================ host side
int dd_decimw_supported(int K, int M, int flags, const void* in) {
    if (M < 8 || M > DW_MAX_M || (M & 1) || K < 2 || K > 256) return 0;
    const uintptr_t a = reinterpret_cast<uintptr_t>(in);
    return (a & ((flags & DD_CHAIN_U8_INPUT) ? 1 : 7)) == 0 ? 1 : 0;
}

struct DWPlan {
    int64_t R0;
    int nrows, phi, HP, e, K16, wpc, run_rows, nwaves, nruns, pad, img;
    int bsum, NI, j1lo, nh;    // block-sum form (k_chain_decim_b): partial sums per output, first block sample of the second accumulator set, steps of eight samples per block
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
    // block sums (k_chain_decim_b) where an output needs at most eight of them: K <= 8 M -- every K for M >= 32, the reference's /34 and /50
    pl.NI = (K + M - 1) / M;
#ifdef DW_NO_BSUM
    pl.bsum = 0;
#else
    pl.bsum = pl.NI <= 8 ? 1 : 0;
#endif
    pl.j1lo = 0;
    pl.nh = 0;
    if (pl.bsum) {
        pl.HP = M;                                             // a block reaches at most M samples back from its row's first sample
        // a block starts at LDS sample HP + (offset of its kept sample) - M + 1: on an odd one when phi is even -- one sample earlier then
        pl.e = (int)((phi + 1) & 1);
        // lane stride M samples: conflict-free 16-byte reads for odd M / 2; else two samples of gap between the blocks (never read)
        pl.pad = (M % 4) == 0 ? 1 : 0;
        pl.K16 = 0;
        pl.j1lo = 5 * M + pl.e - K > 0 ? 5 * M + pl.e - K : 0;
        pl.nh = (M + pl.e + 7) >> 3;                           // block samples [0, M - 1 + e] in steps of eight
        const int span = pl.HP + DW_W;
        pl.img = pl.pad ? (span + 2 * (span / M + 4) + 2 * DW_PAD) & ~1 : span + 2 * DW_PAD;      // (a block's reads run up to two steps past it)
    } else {
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
    }
    pl.lds = sizeof(float2) * (size_t)(pl.img + DW_NG);
    int wpc = (int)((160 * 1024) / pl.lds);
    pl.wpc = wpc > 4 * DW_WAVES_PER_SIMD ? 4 * DW_WAVES_PER_SIMD : (wpc < 1 ? 1 : wpc);
    // runs of about 8 rows dealt to the waves in turn, every wave the same number of them where the chunk is long enough
    const int slots = ncu * pl.wpc;
    const int per_wave = (pl.nrows + slots - 1) / slots;
    const int nr = (per_wave + 7) / 8;
    pl.run_rows = nr > 0 ? (per_wave + nr - 1) / nr : 1;
    // (a run pays for the row before it: with one row per run -- chunks of up to 2^22 samples -- every wave stages and sums two rows to emit
    //  one.  Two rows per run on half the waves: the sixteen 2^22-sample launches of the C3 chunk loop 0.209 -> 0.189 ms, same call;
    //  profiles/r06_decimb_notes.txt)
    if (pl.run_rows < 2 && pl.nrows >= 2) pl.run_rows = 2;
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
    out[8] = pl.bsum; out[9] = pl.NI; out[10] = pl.j1lo; out[11] = pl.img;
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

template <int NG>
static const void* decimb_kernel(bool u8, bool nco, bool fm, bool pad) {
    static const void* const k[16] = {
        (const void*)k_chain_decim_b<false, false, false, false, NG>, (const void*)k_chain_decim_b<true, false, false, false, NG>,
        (const void*)k_chain_decim_b<false, true, false, false, NG>,  (const void*)k_chain_decim_b<true, true, false, false, NG>,
        (const void*)k_chain_decim_b<false, false, true, false, NG>,  (const void*)k_chain_decim_b<true, false, true, false, NG>,
        (const void*)k_chain_decim_b<false, true, true, false, NG>,   (const void*)k_chain_decim_b<true, true, true, false, NG>,
        (const void*)k_chain_decim_b<false, false, false, true, NG>,  (const void*)k_chain_decim_b<true, false, false, true, NG>,
        (const void*)k_chain_decim_b<false, true, false, true, NG>,   (const void*)k_chain_decim_b<true, true, false, true, NG>,
        (const void*)k_chain_decim_b<false, false, true, true, NG>,   (const void*)k_chain_decim_b<true, false, true, true, NG>,
        (const void*)k_chain_decim_b<false, true, true, true, NG>,    (const void*)k_chain_decim_b<true, true, true, true, NG>};
    return k[(u8 ? 1 : 0) | (nco ? 2 : 0) | (fm ? 4 : 0) | (pad ? 8 : 0)];
}

int dd_decimw_launch(const DDChainParams& P, const float* taps_g0, const double* taps_host, DDDecimWTaps* cache, hipStream_t stream, int* kernel_id) {
    if (P.Ld < 1 && !P.tail_out) return DD_OK;                 // (no kept sample: one wave, for the new history alone)
    const bool u8 = (P.flags & DD_CHAIN_U8_INPUT) != 0, nco = (P.flags & DD_CHAIN_NCO) != 0, fm = (P.flags & DD_CHAIN_FM) != 0;
    DWPlan pl;
    decimw_plan(P.abs0, P.Ld, P.K, P.M, P.off, dd_cu_count(), pl);
    static const char* run_env = DD_TUNE_ENV("DD_DECIMW_RUN");              // tools: rows per run
    if (run_env && atoi(run_env) > 0) {
        pl.run_rows = atoi(run_env);
        pl.nruns = (pl.nrows + pl.run_rows - 1) / pl.run_rows;
        const int slots = dd_cu_count() * pl.wpc;
        pl.nwaves = pl.nruns < slots ? pl.nruns : slots;
    }
    const float* taps = taps_g0 - pl.e;
    if (pl.bsum) {
        // the taps as the matrix instruction takes them, for (M, e): [set][register g][lane 4 t + m] = h[M (4 set + m) + d], d = M - 1 + e - j the
        // distance of block sample j = 16 g + t from the block's kept sample (0 outside the block and beyond h[K - 1]); kept with the filter
        const int key = 0x10000 | (P.M << 1) | pl.e;
        static_assert(640 <= DD_DECIMW_TAPS_CAP, "block-sum taps");
        if (!cache->dev) DD_HIP_CHECK(hipMalloc((void**)&cache->dev, sizeof(float) * DD_DECIMW_TAPS_CAP));
        if (cache->key != key) {
            float* const t = cache->host;
            for (int j = 0; j < DD_DECIMW_TAPS_CAP; ++j) t[j] = 0.f;
            for (int j = 0; j < 80; ++j)
                for (int i = 0; i < 8; ++i) {
                    const int d = P.M - 1 + pl.e - j;
                    if (d < 0 || d > P.M - 1 || P.M * i + d >= P.K) continue;
                    t[320 * (i >> 2) + 64 * (j >> 4) + 4 * (j & 15) + (i & 3)] = (float)taps_host[P.M * i + d];
                }
            DD_HIP_CHECK(hipMemcpyAsync(cache->dev, t, sizeof(float) * DD_DECIMW_TAPS_CAP, hipMemcpyHostToDevice, stream));
            cache->key = key;
        }
        taps = cache->dev;
    } else if (pl.pad) {
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
    A.NI = pl.NI; A.nh = pl.nh; A.h1lo = pl.j1lo >> 3;
    {
        // rows of the launch that lie inside the chunk: abs0 <= (R0 + q) W and (R0 + q + 1) W <= abs0 + L
        const int64_t lo = dw_floordiv(P.abs0 - pl.R0 * DW_W + DW_W - 1, DW_W), hi = dw_floordiv(P.abs0 + P.L - pl.R0 * DW_W, DW_W);
        const int64_t f0 = lo < -1 ? -1 : (lo > pl.nrows ? pl.nrows : lo), f1 = hi < f0 ? f0 : (hi > pl.nrows ? pl.nrows : hi);
        A.F0 = (int)f0; A.F1 = (int)f1;
        int64_t c0 = ((int64_t)pl.phi - pl.R0 * DW_W) % P.M;
        if (c0 < 0) c0 += P.M;
        A.c0 = (int)c0;
        A.d0 = (int)(pl.R0 * DW_W - (P.abs0 + P.off));
        A.wm = DW_W % P.M;
        A.small = pl.nrows < (1 << 19) ? 1 : 0;
    }
    void* kargs[1] = {&A};
    const void* kern = !pl.bsum ? decimw_kernel(u8, nco, fm, pl.pad != 0)
                                : (pl.NI > 4 ? decimb_kernel<2>(u8, nco, fm, pl.pad != 0) : decimb_kernel<1>(u8, nco, fm, pl.pad != 0));
    DD_HIP_CHECK(hipLaunchKernel(kern, dim3(pl.nwaves > 0 ? pl.nwaves : 1), dim3(64), kargs, pl.lds, stream));
    if (kernel_id) *kernel_id = pl.bsum ? DD_KERNEL_DECIM_BLOCKS : DD_KERNEL_DECIM_WAVE;
    return DD_OK;
}

// dd_code_warmup (dd_runtime.hip): the runtime loads a translation unit's code object when one of its kernels is first named
int dd_code_touch_decimw(void) {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, (const void*)k_chain_decim_b<true, true, true, false, 2>) == hipSuccess ? DD_OK : DD_ERR_HIP;
}
