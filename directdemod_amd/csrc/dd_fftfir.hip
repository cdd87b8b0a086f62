// Fused hot path as an f32 overlap-save FFT convolution held in LDS (general taps up to 256, M == 1, FM output):
//
//     offsetFreq (NCO, commuted into the taps) -> FIR as a 1024-point circular convolution -> demod_fm
//
// Why: the three-limb f16 Toeplitz GEMM of dd_mfma.hip spends 108 MFMAs per 1024
// outputs whatever the limb format (f16 x f16 limbs, exact 8-bit data x i8 tap limbs, Karatsuba forms: all 6 matrix
// instructions per 16 taps; tools/ubench/mfma_i8_vs_f16.hip measures the bare streams: 0.148 ms of f16 MFMAs, 0.159 ms
// of i8 MFMAs per 2^26 samples), and the chip holds only 1.54 GHz under it at the board's 1400 W cap.  A block FFT costs ~44
// packed f32 instructions per output for forward transform + spectrum product + inverse transform (+ 12 for the discriminator),
// independent of the tap count, runs on the vector pipe alone at 2.08 GHz under the same cap, and takes complex taps for
// free -- so the NCO
//     y[p] = sum_k g[k] x[p-k] e^{-j th (p-k)} = e^{-j th p} sum_k (g[k] e^{j th k}) x[p-k] = e^{-j th p} w[p]
// moves into the tap spectrum and the discriminator sees  y[p] conj(y[p-1]) = e^{-j th} w[p] conj(w[p-1]).
//
// Numerics: f32 throughout; the error of a block's outputs is relative to the largest FIR output of THAT block (3e-7 of
// it), not to the individual output: an amplitude step of more than ~60 dB inside one 1024-sample block leaves the quiet
// side with the loud side's rounding noise (tests/test_gpu_parity.py::test_mfma_tile_scaling_paths states the bound;
// samples from an 8-bit source span 48 dB).  The f16-limb MFMA kernel keeps its errors within the 255-tap window.
//
// History of this file: a first version ran 4096-point blocks on 4-wave workgroups (16^3, five barriers per
// block, 224 registers, two waves per SIMD: 0.232 ms, vector pipe 47 % busy); k_chain_fft1k below replaced it.
#include "dd_chain_kernels.h"
#include "dd_fftfir.h"
#include "dd_atan.h"
#include <stdlib.h>
#include <stddef.h>
#include <complex>
#include <mutex>

typedef float v2f __attribute__((ext_vector_type(2)));

// ---- complex arithmetic on packed pairs (x = re, y = im): v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 ----
// register that holds output k of ff_bfly16
#define FF_P(k) (4 * ((k) & 3) + ((k) >> 2))

// A complex product a * w with a loop-invariant factor w held in a register pair, written as its two instructions with their
// operand selects and sign modifiers spelt out.  Left to the compiler (a * w.xx + a.yx * (-w.y, w.y)), the splat and the signed
// swizzles of all the factors are hoisted out of the block loop as registers of their own -- 6 registers per factor.
// On gfx950 a packed-f32 result may not be read by the very next VALU instruction (one wait state: the compiler pads with
// s_nop where it cannot find an independent instruction, and it does not reorder asm statements to find one), so the two
// halves of a product are separate statements and the callers keep a dependent pair at least one instruction apart: all
// first halves of a group, then all second halves.
__device__ __forceinline__ v2f ff_mul_lo(v2f a, v2f w) {        // (a.x w.x, a.y w.x)
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    return t;
}
__device__ __forceinline__ v2f ff_fma_hi(v2f a, v2f w, v2f t) {  // t + (-a.y w.y, a.x w.y): completes a * w
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ v2f ff_fma_hic(v2f a, v2f w, v2f t) { // t + (a.y w.y, -a.x w.y): completes a * conj(w)
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// the same with a wave-uniform factor in a scalar register pair
__device__ __forceinline__ v2f ff_mul_lo_s(v2f a, v2f w) {
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
    return t;
}
template <bool CONJ>
__device__ __forceinline__ v2f ff_fma_hi_s(v2f a, v2f w, v2f t) {
    v2f r;
    if (CONJ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}
__device__ __forceinline__ v2f ff_addmj(v2f a, v2f b) {         // a - j b = (a.x + b.y, a.y - b.x)
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f ff_addpj(v2f a, v2f b) {         // a + j b = (a.x - b.y, a.y + b.x)
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a[idx(k)] *= w[k] (or conj(w[k])) for k = 1..15, idx = FF_P (outputs of a butterfly) or the identity
template <bool CONJ, bool PERM>
__device__ __forceinline__ void ff_twiddle15(v2f (&a)[16], const v2f (&w)[16]) {
    v2f t[16];
#pragma unroll
    for (int k = 1; k < 16; ++k) t[k] = ff_mul_lo(a[PERM ? FF_P(k) : k], w[k]);
#pragma unroll
    for (int k = 1; k < 16; ++k) a[PERM ? FF_P(k) : k] = CONJ ? ff_fma_hic(a[PERM ? FF_P(k) : k], w[k], t[k]) : ff_fma_hi(a[PERM ? FF_P(k) : k], w[k], t[k]);
}

// c + (a.x w.x, a.y w.x): first half of c + a * w (second half: ff_fma_hi / ff_fma_hic)
template <bool SC>
__device__ __forceinline__ v2f ff_fma_lo(v2f a, v2f w, v2f c) {
    v2f r;
    if (SC) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"(w), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    return r;
}
template <bool SC>
__device__ __forceinline__ v2f ff_mul_lo_t(v2f a, v2f w) { return SC ? ff_mul_lo_s(a, w) : ff_mul_lo(a, w); }
template <bool CONJ, bool SC>
__device__ __forceinline__ v2f ff_fma_hi_t(v2f a, v2f w, v2f t) {
    return SC ? ff_fma_hi_s<CONJ>(a, w, t) : (CONJ ? ff_fma_hic(a, w, t) : ff_fma_hi(a, w, t));
}
__device__ __forceinline__ v2f ff_2u_minus_s(v2f u, v2f s, v2f two) {          // 2 u - s   (two = (2, 2) in a scalar register pair)
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(u), "s"(two), "v"(s));
    return r;
}

// One stage of four 4-point DFTs (forward W4 = -j, inverse +j) over a[B + S i], i = 0..3, for the four bases B of the
// stage (S = 4: B = 0..3; S = 1: B = 0, 4, 8, 12), whose inputs carry factors: bit e of MASK set = a[e] is to be
// multiplied by w[e] (conj(w[e]) if CONJ) first; bit e of ROT set (only inputs i = 2) = by -j (forward) / +j (inverse).
// The factors are folded into the first additions: x + w y as two packed FMAs, x - w y = 2 x - (x + w y) as a third --
// three instructions where a product and two additions take four (inputs 1 and 0 are multiplied out: two instructions
// each).  Written step by step across the four butterflies so that no instruction reads the result of the one before it
// (a packed result may not be read by the next instruction: the compiler would pad with s_nop).
template <bool INV, bool CONJ, int S, unsigned MASK, unsigned ROT, bool SC>
__device__ __forceinline__ void ff_stage(v2f (&a)[16], const v2f (&w)[16]) {
    const v2f two = {2.0f, 2.0f};
    v2f t0[4], t1[4], u0[4], u1[4], ta[4], tb[4], s0[4], s1[4], s2[4], d[4];
#define FF_E(g, i) ((S == 4 ? (g) : 4 * (g)) + S * (i))
#define FF_M(g, i) ((MASK >> FF_E(g, i)) & 1u)
#define FF_R(g) ((ROT >> FF_E(g, 2)) & 1u)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (FF_M(g, 0)) t0[g] = ff_mul_lo_t<SC>(a[FF_E(g, 0)], w[FF_E(g, 0)]);
        if (FF_M(g, 1)) t1[g] = ff_mul_lo_t<SC>(a[FF_E(g, 1)], w[FF_E(g, 1)]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        u0[g] = FF_M(g, 0) ? ff_fma_hi_t<CONJ, SC>(a[FF_E(g, 0)], w[FF_E(g, 0)], t0[g]) : a[FF_E(g, 0)];
        u1[g] = FF_M(g, 1) ? ff_fma_hi_t<CONJ, SC>(a[FF_E(g, 1)], w[FF_E(g, 1)], t1[g]) : a[FF_E(g, 1)];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (FF_M(g, 2)) ta[g] = ff_fma_lo<SC>(a[FF_E(g, 2)], w[FF_E(g, 2)], u0[g]);
        if (FF_M(g, 3)) tb[g] = ff_fma_lo<SC>(a[FF_E(g, 3)], w[FF_E(g, 3)], u1[g]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (FF_M(g, 2)) s0[g] = ff_fma_hi_t<CONJ, SC>(a[FF_E(g, 2)], w[FF_E(g, 2)], ta[g]);
        else if (FF_R(g)) s0[g] = INV ? ff_addpj(u0[g], a[FF_E(g, 2)]) : ff_addmj(u0[g], a[FF_E(g, 2)]);
        else s0[g] = u0[g] + a[FF_E(g, 2)];
        s2[g] = FF_M(g, 3) ? ff_fma_hi_t<CONJ, SC>(a[FF_E(g, 3)], w[FF_E(g, 3)], tb[g]) : u1[g] + a[FF_E(g, 3)];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (FF_M(g, 2)) s1[g] = ff_2u_minus_s(u0[g], s0[g], two);
        else if (FF_R(g)) s1[g] = INV ? ff_addmj(u0[g], a[FF_E(g, 2)]) : ff_addpj(u0[g], a[FF_E(g, 2)]);
        else s1[g] = u0[g] - a[FF_E(g, 2)];
        d[g] = FF_M(g, 3) ? ff_2u_minus_s(u1[g], s2[g], two) : u1[g] - a[FF_E(g, 3)];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        a[FF_E(g, 0)] = s0[g] + s2[g];
        a[FF_E(g, 2)] = s0[g] - s2[g];
        a[FF_E(g, 1)] = INV ? ff_addpj(s1[g], d[g]) : ff_addmj(s1[g], d[g]);
        a[FF_E(g, 3)] = INV ? ff_addmj(s1[g], d[g]) : ff_addpj(s1[g], d[g]);
    }
#undef FF_E
#undef FF_M
#undef FF_R
}

// 16-point DFT in registers, 4 x 4: a[4p + q] in, output k = c + 4d in a[4c + d] = a[FF_P(k)].  PRE: input a[k] carries the
// factor w[k] (conj(w[k]) for the inverse), k = 1..15 -- the twiddles of the transposed (inverse) graph, folded into the first stage.
template <bool INV, bool PRE>
__device__ __forceinline__ void ff_bfly16(v2f (&a)[16], const v2f (&w)[16]) {
    ff_stage<INV, INV, 4, PRE ? 0xFFFEu : 0u, 0u, false>(a, w);
    // a[4c + q] = u[q][c];  u[q][c] *= W16^{q c} (conjugated for the inverse), constants in scalar register pairs; W16^4 = -j by
    // swapped additions
    v2f cw[16];
    cw[5] = (v2f){0.92387953251128674f, -0.38268343236508977f};
    cw[6] = (v2f){0.70710678118654752f, -0.70710678118654752f};
    cw[7] = (v2f){0.38268343236508977f, -0.92387953251128674f};
    cw[9] = cw[6];
    cw[11] = (v2f){-0.70710678118654752f, -0.70710678118654752f};
    cw[13] = cw[7];
    cw[14] = cw[11];
    cw[15] = (v2f){-0.92387953251128674f, 0.38268343236508977f};
    ff_stage<INV, INV, 1, 0xEAE0u, 0x0400u, true>(a, cw);
}
template <bool INV>
__device__ __forceinline__ void ff_bfly16(v2f (&a)[16]) {
    v2f none[16];
    ff_bfly16<INV, false>(a, none);
}

// (the arctangents live in dd_atan.h: shared with dd_cosfir.hip)
__device__ __forceinline__ float ff_atan2(float y, float x) { return dd_atan2_poly(y, x); }
__device__ __forceinline__ float ff_atan_small(float y, float x) { return dd_atan_small(y, x); }

// ============================================================================
// k_chain_fft1k: the same convolution with ONE WAVE PER BLOCK -- 1024-point blocks (16 x 16 x 4), 768 outputs each.
//
// What the 4096-point kernel above measured: the vector pipe 47 % busy.  One wave issues a VALU
// instruction every 4.8 cycles at best (a packed one occupies the SIMD for ~2.9), so a SIMD needs two or more waves in
// an arithmetic phase at once; with 224 registers and 72 KB of LDS per 4-wave workgroup only two waves share a SIMD,
// the four waves of a workgroup meet at five barriers per block and reach their load / store / exchange phases
// together (a CU's single address unit takes ~16 cycles per 64-lane dwordx2 instruction: 16 back-to-back loads from 8
// waves cost each wave 1000-1500 cycles).  Here a wave owns its block: both exchanges are intra-wave (no barrier at all,
// only lgkmcnt waits), every wave runs at its own phase, 12 waves per CU (3 per SIMD, 168 registers, 10.5 KB of LDS
// each).  1024 points cost 10 butterfly levels instead of 12, which pays for the 25 % overlap: fewer vector
// instructions per output than the 4096-point form, for 25 % more LDS bytes and load instructions per output.
//
//   n = 64 n2 + 4 n1 + n0,  k = k0 + 16 k1 + 256 k2,  W = e^{-2 pi j / 1024}:
//   W^{nk} = W16^{n2 k0} . W^{(4 n1 + n0) k0} . W16^{n1 k1} . W64^{n0 k1} . W4^{n0 k2}
//   forward: B1 (16, over n2) | T1 = W^{lane k0} | X1 | B2 (16, over n1) | T2 = W64^{n0 k1} | X2 | B3 (4 x radix 4, over n0)
//   lane roles: pass 1 (n1, n0) = lane; pass 2 (k0, n0), lane = 4 k0 + n0; pass 3 (k0, j), lane = 4 k0 + j, registers
//   4 c + n0 for k1 = 4 c + j.  X1 element (k0, n1, n0) at 68 k0 + 4 n1 + n0, X2 element (k0, k1, n0) at 84 k0 + 5 k1 + n0
//   (both conflict free for the contiguous ds_write_b64 and the strided ds_read_b64, tools/debug/lds_layout.py); the two
//   images share one buffer: a wave's LDS operations execute in order and it has read an image completely before it
//   writes the next.
#define F1_N 1024
#define F1_ADV 768                  // outputs per block: rows 4..15 of [16][64]
#define F1_S1 68
#define F1_S2 84
#define F1_WAVE_BYTES (16 * F1_S2 * 8)          // 10752
#define F1_WAVES 4
#define F1_HP_OFF (F1_WAVES * F1_WAVE_BYTES)     // the tap spectrum as the lanes multiply it: one copy per workgroup, a lane's sixteen
#define F1_HP_STRIDE 18                          // values contiguous -- eight 16-byte reads per block instead of sixteen 8-byte ones; lane
#define F1_LDS_BYTES (F1_HP_OFF + 64 * F1_HP_STRIDE * 8)   // stride 18 elements = 36 banks: the 16 lanes of a ds_read_b128 group fall on distinct banks
#ifndef F1_RUN_BLOCKS
#define F1_RUN_BLOCKS 8            // blocks per contiguous run of a wave (DDFft1kMap)
#endif

// Which interior blocks a wave takes (blocks 1 .. nblk-2; NI of them, NW waves).  Wave w owns B_w = b or b + 1 blocks (the
// balanced split floor(NI (w+1) / NW) - floor(NI w / NW)) and walks them in K rounds: in round k every wave takes a short
// contiguous run (r0[k] blocks for a wave that owns b, r1[k] for one that owns b + 1) and the runs of a round tile one
// window of the stream in wave order -- so at any time the device works on ONE moving window of ~NW * R * 9 KB instead of NW
// streams 175 KB apart.  tools/ubench/stream_2to1.hip (profiles/r04_stream_2to1.txt): the same 8 B in / 4 B out traffic
// moves at 5.0 TB/s with one contiguous run per wave (K = 1, the round-3 kernel) and at 5.5-6.0 as a moving window
// (6.2 / 7.2 TB/s with non-temporal loads and stores).  Inside a run the 256-sample overlap stays in registers (six
// loads per block); the first block of a run loads all eight row pairs.
#define F1_MAXK 32
struct DDFft1kMap {
    int K, b;
    int r0[F1_MAXK], r1[F1_MAXK];     // run lengths in round k
    int wstart[F1_MAXK];              // interior blocks before round k's window
};

struct DDFft1kTabs {
    const float2* tw1;     // [64][16]   W1024^{lane k}
    const float2* tw2;     // [4][16]    W64^{n0 k}
    const float2* hp;      // [16][64]   H[k0 + 16 (4 c + j) + 256 k2] / 1024 at [4 c + k2][4 k0 + j]
    float2 crot;           // e^{-j theta}
    float theta_sub;       // != 0: |theta| is small -- the discriminator subtracts it from the angle instead of rotating every
                           // product by crot: 12 packed adds instead of 24 packed multiply-adds per block (groups of outputs
                           // that need the full-range arctangent are rotated after all)
    float2 rowph[6];       // complex64 output (no demod_fm): e^{-j theta 128 i}, the NCO factor from one row pair of a block to the next
    int base;              // block b covers FIR outputs [768 b + base, 768 b + base + 768), base in [s - 15, s]: chosen on the host so
                           // that every interior store instruction writes whole 64-byte lines of `out` (a stream START drops one
                           // angle, demod_fm.py:43-49, which put every 512-byte store 4 bytes ahead of a line boundary: PMC WRITE_SIZE
                           // 1.12 x the output; and a chunk loop's `out` pointer need not be aligned at all)
};

// A lane moves TWO consecutive samples per memory instruction (one 16-byte load, one 8-byte store of two angles): the
// CU's address unit takes as long for a 4- or 8-byte-per-lane instruction as for a 16-byte one, and the memory
// instructions were what the first version of this kernel waited for (ablation: without its 12 dwordx2 loads per
// block 0.192 ms instead of 0.229; issued but never consumed 0.224 -- their cost, not their latency).  Loaded that
// way, lane l = 32 h + i holds samples 64 (2 r + h) + 2 i + e of row pair r (e = 0, 1): both columns 2i, 2i+1 at every
// second row.  One v_permlane32_swap per register pair exchanges the upper half of the first register with the lower
// half of the second, after which lane l holds column t = 2 i + h at ALL rows -- the layout the first pass needs; the
// same swap after the last pass turns it back into two consecutive outputs per lane, so the left-hand neighbour of the
// second is in the lane's own registers.
__device__ __forceinline__ void f1_swap(v2f& b, v2f& a) {
    const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(b.x), __float_as_uint(a.x), false, false);
    const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(b.y), __float_as_uint(a.y), false, false);
    b = (v2f){__uint_as_float(x[0]), __uint_as_float(y[0])};
    a = (v2f){__uint_as_float(x[1]), __uint_as_float(y[1])};
}

// row pairs [r0, r0 + n) of the block that starts at sample n0: x[2 r], x[2 r + 1] = samples n0 + 128 r + 2 lane + {0, 1}
// (not yet swapped).  CLAMP: pair indices past `lim` (in pairs, relative to n0) read pair `lim` instead.
template <bool U8, bool CLAMP = false, int NX = 16>
__device__ __forceinline__ void f1_load_pairs(const void* in, int64_t n0, int lane, v2f (&x)[NX], int r0, int n, unsigned lim = 0) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
        unsigned m = (unsigned)lane + 64u * (r0 + i);
        if (CLAMP) m = m < lim ? m : lim;
        if (U8) {
#ifndef FF_NO_NT_LOAD
            const unsigned uu = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(reinterpret_cast<const uchar2*>(in) + n0) + m);
            const uchar4 u = make_uchar4(uu & 255u, (uu >> 8) & 255u, (uu >> 16) & 255u, uu >> 24);
#else
            const uchar4 u = reinterpret_cast<const uchar4*>(reinterpret_cast<const uchar2*>(in) + n0)[m];
#endif
            x[2 * (r0 + i)] = (v2f){(float)u.x - 127.5f, (float)u.y - 127.5f};
            x[2 * (r0 + i) + 1] = (v2f){(float)u.z - 127.5f, (float)u.w - 127.5f};
        } else {
            // non-temporal (streamed once): with the moving-window block map the memory side of this kernel (-DFF_NO_COMPUTE) runs
            // 0.1553 ms instead of 0.1622 (profiles/r04_fft_map_sweep.txt); -DFF_NO_NT_LOAD / -DFF_NO_NT_STORE: plain accesses
#ifndef FF_NO_NT_LOAD
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            const v4f_ v = __builtin_nontemporal_load(reinterpret_cast<const v4f_*>(reinterpret_cast<const float2*>(in) + n0) + m);
#else
            const float4 v = reinterpret_cast<const float4*>(reinterpret_cast<const float2*>(in) + n0)[m];
#endif
            x[2 * (r0 + i)] = (v2f){v.x, v.y};
            x[2 * (r0 + i) + 1] = (v2f){v.z, v.w};
        }
    }
}

#ifdef FF_TRACE
// tools/debug/fft_trace.py: cycles per phase of f1_block (s_memtime stamps; every stamp drains the wave's LDS / scalar counter),
// summed per wave over its interior blocks
#define FF_NPH 14
__device__ unsigned long long g_ff_trace[4096 * (FF_NPH + 2)];
#define FF_T(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned t_ = (unsigned)__builtin_readcyclecounter(); tr[i] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FF_T(i) do { } while (0)
#endif

struct F1Edge {
    int prev_m;            // block position (256 + output index) of the FIR output before the chunk's first one, which comes from the
                           // carried state (255 = just before the block's first output), or -1
    float2 prev;           //   ... un-rotated
    int last_m;            // block position (256 + output index) of the chunk's last FIR output, or -1
    float2 last_rot;       // e^{-j theta (abs0 + L - 1)}
    float2* lasty_out;
};
__device__ __forceinline__ float2 f1_cmulf(v2f a, float2 w) {
    return make_float2(fmaf(a.x, w.x, -a.y * w.y), fmaf(a.x, w.y, a.y * w.x));
}

// sample n of the chunk as this kernel's un-rotated frame sees it: inside the chunk the input itself; before it the
// carried history (K-1 samples AFTER the NCO, filters.py:45,69 / comm.py:77), rotated back; zeros before that; behind
// the chunk's end zeros (only outputs that are not stored depend on those)
template <bool U8>
__device__ __forceinline__ v2f f1_edge_sample(const DDChainParams& P, int64_t n) {
    if (n < 0) {
        const int64_t ti = n + (P.K - 1);
        if (ti < 0) return (v2f){0.f, 0.f};
        const float2 t = P.tail_in[ti];
        if (!(P.flags & DD_CHAIN_NCO)) return (v2f){t.x, t.y};
        const float2 w = dd_phasor((uint64_t)(P.abs0 + n) * P.cyc, P.nco_tbl);     // e^{-j theta (abs0 + n)}
        return (v2f){fmaf(t.x, w.x, t.y * w.y), fmaf(t.y, w.x, -t.x * w.y)};       // t * conj(w)
    }
    // behind the chunk's end: zeros (only outputs that are not stored depend on those samples; a copy of the last sample, as
    // rounds 3's kernel padded, put a run of up to 1023 equal samples -- a DC line 137 x the sample -- into a short chunk's
    // block, and the block's rounding noise is that of its LARGEST output: 2.2e-6 of max|y| on the 3-chunk golden vector)
    if (n >= P.L) return (v2f){0.f, 0.f};
    if (U8) {
        const uchar2 u = reinterpret_cast<const uchar2*>(P.in)[n];
        return (v2f){(float)u.x - 127.5f, (float)u.y - 127.5f};
    }
    const float2 v = reinterpret_cast<const float2*>(P.in)[n];
    return (v2f){v.x, v.y};
}

// angles of row pairs 2 + 2 g, 3 + 2 g (group g of three) from zz (two per lane and row pair: zz[2 r], zz[2 r + 1]):
// angles | two of the next block's loads | two 8-byte stores.  FAST: every product of the group has |im| < tan(pi/8) re
// (wave-uniform, decided per group of 256 outputs: an angle beyond 22.5 degrees -- an amplitude null of a noise-like input,
// SURVEY 8(d) input A -- sends 256 outputs through the full-range form, not the block's 768).
template <bool U8, bool PARTIAL, bool LOADNEXT, bool FAST>
__device__ __forceinline__ void f1_tail_group(const int g, const v2f (&zz)[16], v2f (&a)[16], const void* in, const int64_t n0_next, const int lane, float* const ob, const int lim_lo,
                                              const int limit, const float theta_sub, const v2f crot) {
    float ang[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v2f z = zz[4 + 4 * g + i];
#ifdef FF_NO_DISC
        ang[i] = z.x;
#else
        if (FAST) {
            // theta_sub: the NCO's per-sample rotation taken off the angle (|theta| <= 0.25, so the result stays inside (-pi, pi))
            ang[i] = ff_atan_small(z.y, z.x) - theta_sub;          // (the compiler pairs these into packed instructions)
        } else {
            // full range: the rotation is applied to the product instead (nothing to wrap), and a zero product -- digital
            // silence -- gives exactly 0 like np.angle(0) (demod_fm.py:40-49): ff_atan2's mx == 0 guard
            if (theta_sub != 0.f) {                                // (wave-uniform)
                const v2f t = ff_mul_lo(z, crot);
                z = ff_fma_hi(z, crot, t);
            }
            ang[i] = ff_atan2(z.y, z.x);
        }
#endif
    }
#ifdef FF_LOAD_NOWAIT
    if (LOADNEXT) {                        // same loads, never consumed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const float2*>(in) + n0_next) + lane + 64 * (2 + 2 * g + i);
            float4 junk;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(junk) : "v"(q) : "memory");
        }
    }
#elif !defined(FF_NO_LOAD) && defined(FF_LOADS_IN_TAIL)
    if (LOADNEXT) f1_load_pairs<U8>(in, n0_next, lane, a, 2 + 2 * g, 2);
#endif
#ifdef FF_NO_STORE
    if (ang[0] + ang[1] + ang[2] + ang[3] == 1234.5f) ob[0] = ang[0];
#else
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int o = 128 * (2 * g + i);                    // output of this lane's first sample of the row pair, relative to ob
        if (!PARTIAL) {
#ifndef FF_NO_NT_STORE
            __builtin_nontemporal_store((v2f){ang[2 * i], ang[2 * i + 1]}, reinterpret_cast<v2f*>(ob + o));
#else
            *reinterpret_cast<float2*>(ob + o) = make_float2(ang[2 * i], ang[2 * i + 1]);
#endif
        } else {
            if (2 * lane + o >= lim_lo && 2 * lane + o < limit) ob[o] = ang[2 * i];
            if (2 * lane + o + 1 >= lim_lo && 2 * lane + o + 1 < limit) ob[o + 1] = ang[2 * i + 1];
        }
    }
#endif
}

// (Round 4 also tried the exchanges' sixteen reads as explicit ds_read_b64 -- the compiler merges them into eight ds_read2_b64, which
// the LDS serves at half the rate -- through inline asm with the wait as part of the sequence: 0.1956 against 0.1962 ms, nothing.
// The exchanges cost latency, not LDS throughput; profiles/r04_fft1k_overlap.txt.)
// one block.  On entry a[0..3] hold rows 0..3 of column t (the overlap kept from the previous block, or swapped by the
// caller) and a[4..15] row pairs 2..7 as loaded; on exit, when LOADNEXT, the same for the next block.  out_row4 points
// at the block's first output (row 4, column 0).  PARTIAL: outputs at or beyond `limit` (relative to it) are not stored.
// complex64 output flavour (commSignal.filter without a demodulator, comm.py:80-92): y[p] = e^{-j theta (abs0 + p)} w[p] -- the NCO
// factor the FM flavour never needs (it cancels in y[p] conj(y[p-1]) up to the constant rotation) is applied to every output as
// (block's first output) x (row pair) x (lane): pb = e^{-j theta (abs0 + p0)} from the exact phase table, once per block
struct F1Cx {
    v2f pb;                // e^{-j theta (abs0 + p0)}
    v2f lp0, lp1;          // e^{-j theta (2 lane)}, e^{-j theta (2 lane + 1)}
    const float2* rowph;   // [6] e^{-j theta 128 i} (kernel argument)
};
__device__ __forceinline__ v2f f1_cmul(v2f a, v2f b) { return (v2f){fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)}; }

template <bool U8, bool PARTIAL, bool LOADNEXT, bool CX = false>
__device__ __forceinline__ void f1_block(v2f (&a)[16], v2f (&keep)[4], v2f (&jl)[4], v2f* const Xp, const v2f (&tw1)[16], const v2f (&tw3)[16], const v2f* const hp,
                                         const v2f crot, const float theta_sub, const int lane, const void* in, const int64_t n0_next, float* const out_row4, const int lim_lo, const int limit,
                                         const bool jump = false, const F1Edge* edge = nullptr, const F1Cx* cx = nullptr
#ifdef FF_TRACE
                                         , unsigned* tr = nullptr
#endif
                                         ) {
    const int hi = lane >> 2, lo = lane & 3;
    v2f* const X = Xp;
#ifdef FF_TRACE
    unsigned tdummy[FF_NPH];
    if (!tr) tr = tdummy;
    unsigned tprev = (unsigned)__builtin_readcyclecounter();
#endif
#pragma unroll
    for (int r = 2; r < 8; ++r) f1_swap(a[2 * r], a[2 * r + 1]);
#pragma unroll
    for (int r = 0; r < 4; ++r) keep[r] = a[12 + r];          // the overlap the next block starts with
#ifdef FF_NO_COMPUTE
    v2f zz[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) zz[r] = a[r];
#else
    // column t = 2 (lane & 31) + (lane >> 5) sits at position (t >> 1) + 34 (t & 1) of an X1 row: contiguous per half wave
    const int x1w = (lane & 31) + 34 * (lane >> 5);                        // + 68 k0
    const int x1r = hi * F1_S1 + (lo >> 1) + 34 * (lo & 1);                // + 2 n1     (column 4 n1 + lo)
    const int x2w = hi * F1_S2 + lo;         // + 5 k1
    const int x2r = hi * F1_S2 + 5 * lo;     // + 20 c + n0      (k1 = 4 c + j)
    // ---- forward pass 1 (over n2), T1, X1
    ff_bfly16<false>(a);
    FF_T(0);
    ff_twiddle15<false, true>(a, tw1);
    FF_T(1);
#ifndef FF_NO_LDS
#pragma unroll
    for (int k = 0; k < 16; ++k) X[x1w + F1_S1 * k] = a[FF_P(k)];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = X[x1r + 2 * k];
#endif
    // ---- forward pass 2 (over n1), X2
    FF_T(2);
    ff_bfly16<false>(a);
    FF_T(3);
    FF_T(4);
#ifndef FF_NO_LDS
#pragma unroll
    for (int k = 0; k < 16; ++k) X[x2w + 5 * k] = a[FF_P(k)];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < 4; ++n) a[4 * c + n] = X[x2r + 20 * c + n];
#endif
    // ---- T2 folded into forward pass 3 (radix 4 over n0); the spectrum product folded into inverse pass 3; T2*
    FF_T(5);
    ff_stage<false, false, 1, 0xEEEEu, 0u, false>(a, tw3);
    {
        v2f h[16];
#pragma unroll
        for (int k = 0; k < 16; k += 2) {                      // (this lane's row of the spectrum image in LDS)
            typedef float v4f_h __attribute__((ext_vector_type(4)));
            const v4f_h hh = *reinterpret_cast<const v4f_h*>(hp + k);
            h[k] = (v2f){hh.x, hh.y};
            h[k + 1] = (v2f){hh.z, hh.w};
        }
        ff_stage<true, false, 1, 0xFFFFu, 0u, false>(a, h);
    }
    {
        v2f z[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k & 3) z[k] = ff_mul_lo(a[k], tw3[k]);
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k & 3) a[k] = ff_fma_hic(a[k], tw3[k], z[k]);
    }
    FF_T(6);
#ifndef FF_NO_LDS
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < 4; ++n) X[x2r + 20 * c + n] = a[4 * c + n];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = X[x2w + 5 * k];
#endif
    // ---- inverse pass 2
    FF_T(7);
    ff_bfly16<true>(a);
    FF_T(8);
#ifndef FF_NO_LDS
#pragma unroll
    for (int k = 0; k < 16; ++k) X[x1r + 2 * k] = a[FF_P(k)];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = X[x1w + F1_S1 * k];
#endif
    // ---- T1* folded into inverse pass 1: a[FF_P(r)] = w[n0 + 64 r + t]
    FF_T(9);
    ff_bfly16<true, true>(a, tw1);
    FF_T(10);
    // ---- back to two consecutive outputs per lane: B[r] = w[128 r + 2 lane], A[r] = the one after it (row pairs 1..7;
    // of pair 1 only lane 63's second value is used, as the left-hand neighbour of the block's first output)
    v2f zz[16];
    {
        v2f B[8], A[8];
#pragma unroll
        for (int r = 1; r < 8; ++r) { B[r] = a[FF_P(2 * r)]; A[r] = a[FF_P(2 * r + 1)]; f1_swap(B[r], A[r]); }
        if (CX) {
            // ---- complex64 output: NCO factor, 16-byte stores of two outputs per lane; the next block's loads in between
#ifndef FF_NO_LOAD
            if (LOADNEXT && jump) f1_load_pairs<U8, false, 4>(in, n0_next, lane, jl, 0, 2);
#ifndef FF_LOADS_IN_TAIL
            // (B and A hold this block's results: a[] is free for the next block's samples, requested ahead of every store)
            if (LOADNEXT) f1_load_pairs<U8>(in, n0_next, lane, a, 2, 6);
#endif
#endif
            float2* const oc = reinterpret_cast<float2*>(out_row4) + 2 * lane;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                v2f y[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = 2 + 2 * g + i;
                    const float2 rw = cx->rowph[r - 2];
                    const v2f rp = f1_cmul(cx->pb, (v2f){rw.x, rw.y});
                    y[2 * i] = f1_cmul(B[r], f1_cmul(rp, cx->lp0));
                    y[2 * i + 1] = f1_cmul(A[r], f1_cmul(rp, cx->lp1));
                }
#if !defined(FF_NO_LOAD) && defined(FF_LOADS_IN_TAIL)
                if (LOADNEXT) f1_load_pairs<U8>(in, n0_next, lane, a, 2 + 2 * g, 2);
#endif
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int o = 128 * (2 * g + i);
                    if (!PARTIAL) {
                        typedef float v4f_ __attribute__((ext_vector_type(4)));
#ifndef FF_NO_NT_STORE
                        __builtin_nontemporal_store((v4f_){y[2 * i].x, y[2 * i].y, y[2 * i + 1].x, y[2 * i + 1].y}, reinterpret_cast<v4f_*>(oc + o));
#else
                        *reinterpret_cast<v4f_*>(oc + o) = (v4f_){y[2 * i].x, y[2 * i].y, y[2 * i + 1].x, y[2 * i + 1].y};
#endif
                    } else {
                        if (2 * lane + o >= lim_lo && 2 * lane + o < limit) oc[o] = make_float2(y[2 * i].x, y[2 * i].y);
                        if (2 * lane + o + 1 >= lim_lo && 2 * lane + o + 1 < limit) oc[o + 1] = make_float2(y[2 * i + 1].x, y[2 * i + 1].y);
                    }
                }
            }
            return;
        }
        if (PARTIAL && edge) {
            // chunk edges (cold path).  First block of a chunk that continues a stream: the FIR output before the chunk's
            // first one is the carried state (demod_fm.py:47-49), brought into this kernel's un-rotated frame.  Last block:
            // the chunk's last FIR output, rotated, is the next chunk's carried state.
            if (edge->prev_m >= 0) {
#pragma unroll
                for (int r = 1; r < 8; ++r) {
                    if (128 * r + 2 * lane == edge->prev_m) B[r] = (v2f){edge->prev.x, edge->prev.y};
                    if (128 * r + 2 * lane + 1 == edge->prev_m) A[r] = (v2f){edge->prev.x, edge->prev.y};
                }
            }
            if (edge->last_m >= 0) {
#pragma unroll
                for (int r = 2; r < 8; ++r) {
                    if (128 * r + 2 * lane == edge->last_m) *edge->lasty_out = f1_cmulf(B[r], edge->last_rot);
                    if (128 * r + 2 * lane + 1 == edge->last_m) *edge->lasty_out = f1_cmulf(A[r], edge->last_rot);
                }
            }
        }
        // discriminator: the second value's neighbour is the first; the first's is the second of the lane to the left
        // (wave_shr:1), for lane 0 that of lane 63 in the row pair above (wave_ror:1 of that register as the DPP's `old`
        // operand, which a lane without a source lane keeps)
        v2f pb[8];
#pragma unroll
        for (int r = 2; r < 8; ++r) {
            // (every lane of a wave_ror has a source lane: bound_ctrl set, so that no `old` value has to be materialised)
            const int ox = __builtin_amdgcn_update_dpp(0, __float_as_int(A[r - 1].x), 0x13C, 0xf, 0xf, true);
            const int oy = __builtin_amdgcn_update_dpp(0, __float_as_int(A[r - 1].y), 0x13C, 0xf, 0xf, true);
            pb[r].x = __int_as_float(__builtin_amdgcn_update_dpp(ox, __float_as_int(A[r].x), 0x138, 0xf, 0xf, false));
            pb[r].y = __int_as_float(__builtin_amdgcn_update_dpp(oy, __float_as_int(A[r].y), 0x138, 0xf, 0xf, false));
        }
        v2f t[16];
#pragma unroll
        for (int r = 2; r < 8; ++r) { t[2 * r] = ff_mul_lo(B[r], pb[r]); t[2 * r + 1] = ff_mul_lo(A[r], B[r]); }
#pragma unroll
        for (int r = 2; r < 8; ++r) { zz[2 * r] = ff_fma_hic(B[r], pb[r], t[2 * r]); zz[2 * r + 1] = ff_fma_hic(A[r], B[r], t[2 * r + 1]); }   // w[m] conj(w[m-1])
        if (theta_sub == 0.f) {                        // (wave-uniform)
#pragma unroll
            for (int r = 4; r < 16; ++r) t[r] = ff_mul_lo(zz[r], crot);
#pragma unroll
            for (int r = 4; r < 16; ++r) zz[r] = ff_fma_hi(zz[r], crot, t[r]);         // the NCO's rotation per sample
        }
    }
#endif
    // a[] is dead from here: the next block's samples fly during the angles and stores.  Its first four rows are this
    // block's last four (the 256-sample overlap), kept in `keep`: six loads per block
    // (jump, wave-uniform: the next block is not this one's successor -- its first two row pairs are loaded as well, into
    // registers of their own: the caller swaps them into a[0..3] at the top of the next block, where the wave waits for that
    // block's samples anyway; otherwise the caller copies `keep`)
#ifndef FF_NO_LOAD
    if (LOADNEXT && jump) f1_load_pairs<U8, false, 4>(in, n0_next, lane, jl, 0, 2);
#ifndef FF_LOADS_IN_TAIL
    // ALL of the next block's loads go out here, ahead of every store of this block.  The memory counter (vmcnt) of gfx9 counts
    // loads and stores alike and retires them in issue order, so a wave that waits for a load also waits for every store it
    // issued before that load.  Round 3 interleaved "angles | two loads | two stores" three times: the wait for the last loads
    // at the top of the next block then included four stores issued a few hundred cycles earlier -- a store round trip exposed
    // per block and wave.  (Found when the arithmetic-only build, forced onto the bench input's angle path, measured 0.129 ms
    // and the memory-only build 0.141 ms against 0.200 ms for the kernel: profiles/r04_fft1k_overlap.txt.)
    if (LOADNEXT) f1_load_pairs<U8>(in, n0_next, lane, a, 2, 6);
#endif
#endif
    // wave-uniform fast path per group of two row pairs: every |angle| of its 256 outputs below 22.5 degrees, |im| < tan(pi/8) re
    // (strictly: a product of exactly zero -- 1024 samples of digital silence -- must take the full-range form, whose
    // result for it is 0; the small-angle form would divide 0 by 0)
    bool fast[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        float worst = -1.0f;
#pragma unroll
        for (int r = 4 + 4 * g; r < 8 + 4 * g; ++r) worst = fmaxf(worst, fmaf(-0.41421356f, zz[r].x, fabsf(zz[r].y)));
        fast[g] = __builtin_amdgcn_ballot_w64(worst >= 0.f) == 0;
#ifdef FF_FORCE_FAST
        fast[g] = true;       // (ablation builds without loads run on whatever the registers hold: keep them on the path the bench input takes)
#endif
    }
    FF_T(11);
    float* const ob = out_row4 + 2 * lane;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        if (fast[g]) f1_tail_group<U8, PARTIAL, LOADNEXT, true>(g, zz, a, in, n0_next, lane, ob, lim_lo, limit, theta_sub, crot);
        else f1_tail_group<U8, PARTIAL, LOADNEXT, false>(g, zz, a, in, n0_next, lane, ob, lim_lo, limit, theta_sub, crot);
    }
    FF_T(12);
}

// one edge block (the chunk's first and / or last): samples fetched one by one through f1_edge_sample, stores
// predicated on [s, L), carried state read and written.  Cold: two blocks per chunk.
// The kernel's arguments as they lie in the kernarg segment.  The edge block reads them THERE: structs handed by value to a
// function that is not inlined are first copied to scratch memory -- by every wave, at kernel entry, whether it makes the call
// or not: 9 KB per wave x 3072 waves = 28 MB of stores per launch, the "10 % more bytes written than the output holds" of
// round 3's WRITE_SIZE reading (profiles/r04_inputA_and_write_size_before.txt: 1.098 x on an aligned continuing chunk).
struct F1KernArgs {
    DDChainParams P;
    DDFft1kTabs T;
    DDFft1kMap M;
    int nblk, nwaves;
};
typedef const __attribute__((address_space(4))) F1KernArgs* F1KernArgsPtr;

// (nothing ties F1KernArgs to k_chain_fft1k's parameter list but these checks: the layout rules of the kernarg segment are the C
//  struct rules for these members, and the edge block compares the by-value block count it is handed with the one it reads there)
static_assert(offsetof(F1KernArgs, T) % alignof(DDFft1kTabs) == 0 && offsetof(F1KernArgs, M) % alignof(DDFft1kMap) == 0 &&
              offsetof(F1KernArgs, nwaves) == offsetof(F1KernArgs, nblk) + sizeof(int), "F1KernArgs must mirror k_chain_fft1k's parameter list");
template <bool U8, bool CX>
__device__ __noinline__ void f1_edge_block(F1KernArgsPtr ka, int q, v2f* const X, const v2f* const hp, const int lane, const int nblk_by_value) {
#if defined(__HIP_DEVICE_COMPILE__)
    const F1KernArgs* const kg = (const F1KernArgs*)ka;        // (address-space cast: device pass only)
#else
    const F1KernArgs* const kg = nullptr;
#endif
    const DDChainParams P = kg->P;
    const DDFft1kTabs T = kg->T;
    const int nblk = kg->nblk;
    if (nblk != nblk_by_value) __builtin_trap();               // the struct above no longer matches the kernel's parameters
    // (its own copy of the twiddles: arrays handed to a function that is not inlined would live in scratch memory for
    // the whole kernel)
    const int tcol = 2 * (lane & 31) + (lane >> 5);
    v2f tw1[16], tw3[16];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const float2 u = T.tw1[tcol * 16 + k];
        tw1[k] = (v2f){u.x, u.y};
        if (k & 3) {                                               // pass-3 layout: register 4 c + n0 of lane j holds k1 = 4 c + j
            const float2 w = T.tw2[(k & 3) * 16 + (k & 12) + (lane & 3)];
            tw3[k] = (v2f){w.x, w.y};
        }
    }
    const v2f crot = {T.crot.x, T.crot.y};
    const int64_t p0 = (int64_t)F1_ADV * q + T.base;
    v2f a[16], keep[4], jl[4];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        a[2 * r] = f1_edge_sample<U8>(P, p0 - 256 + 128 * r + 2 * lane);
        a[2 * r + 1] = f1_edge_sample<U8>(P, p0 - 256 + 128 * r + 2 * lane + 1);
    }
    f1_swap(a[0], a[1]);
    f1_swap(a[2], a[3]);
    F1Edge e;
    e.prev_m = -1;
    e.prev = make_float2(0.f, 0.f);
    e.last_m = -1;
    e.last_rot = make_float2(1.f, 0.f);
    e.lasty_out = P.lasty_out;
    const bool nco = (P.flags & DD_CHAIN_NCO) != 0;
    if (!CX && q == 0 && P.s == 0) {
        // y[-1] = e^{-j theta (abs0 - 1)} w[-1]  ->  w[-1] = y[-1] e^{+j theta (abs0 - 1)}
        const float2 ly = *P.lasty_in;
        const float2 w = nco ? dd_phasor((uint64_t)(P.abs0 - 1) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
        e.prev = make_float2(fmaf(ly.x, w.x, ly.y * w.y), fmaf(ly.y, w.x, -ly.x * w.y));
        e.prev_m = (int)(255 - p0);                    // (p0 = base <= 0 here)
    }
    if (q == nblk - 1 && P.lasty_out) {
        e.last_m = (int)(256 + (P.L - 1 - p0));
        e.last_rot = nco ? dd_phasor((uint64_t)(P.abs0 + P.L - 1) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
    }
    const int64_t lo64 = (int64_t)P.s - p0, hi64 = P.L - p0;
    const int lim_lo = lo64 > 0 ? (int)lo64 : 0, lim_hi = hi64 < F1_ADV ? (int)hi64 : F1_ADV;
    if (CX) {
        F1Cx cx;
        const float2 pbf = nco ? dd_phasor((uint64_t)(P.abs0 + p0) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
        const float2 l0 = nco ? dd_phasor((uint64_t)(2 * lane) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
        const float2 l1 = nco ? dd_phasor((uint64_t)(2 * lane + 1) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
        cx.pb = (v2f){pbf.x, pbf.y}; cx.lp0 = (v2f){l0.x, l0.y}; cx.lp1 = (v2f){l1.x, l1.y};
        cx.rowph = kg->T.rowph;
        f1_block<U8, true, false, true>(a, keep, jl, X, tw1, tw3, hp, crot, T.theta_sub, lane, P.in, 0, reinterpret_cast<float*>(reinterpret_cast<float2*>(P.out) + p0), lim_lo, lim_hi, false, &e, &cx);
    } else {
        f1_block<U8, true, false>(a, keep, jl, X, tw1, tw3, hp, crot, T.theta_sub, lane, P.in, 0, reinterpret_cast<float*>(P.out) + (p0 - P.s), lim_lo, lim_hi, false, &e);
    }
    if (q == nblk - 1 && P.tail_out) {
        // the new carried history: the chunk's last K-1 samples after the NCO (older ones from the old history)
        for (int i = lane; i < P.K - 1; i += 64) {
            const int64_t n = P.L - (P.K - 1) + i;
            float2 v;
            if (n < 0) {
                const int64_t ti = n + (P.K - 1);
                v = P.tail_in[ti];                                  // (ti >= 0: i >= 0 and L >= 1)
            } else {
                float2 x;
                if (U8) {
                    const uchar2 u = reinterpret_cast<const uchar2*>(P.in)[n];
                    x = make_float2((float)u.x - 127.5f, (float)u.y - 127.5f);
                } else {
                    x = reinterpret_cast<const float2*>(P.in)[n];
                }
                v = nco ? dd_cmul(x, dd_phasor((uint64_t)(P.abs0 + n) * P.cyc, P.nco_tbl)) : x;
            }
            P.tail_out[i] = v;
        }
    }
}

// The whole chunk in one launch: FIR outputs [0, L) -> FM angles out[p - s] for p >= s, carried state read (history,
// last FIR output) and written.  Block q covers outputs [768 q + base, 768 q + base + 768), base <= s (DDFft1kTabs); every wave
// takes a contiguous run of the nblk blocks; block 0 and block nblk-1 are edge blocks.
template <bool U8, bool CX>
__global__ void __launch_bounds__(64 * F1_WAVES, 3) k_chain_fft1k(const DDChainParams P, const DDFft1kTabs T, const DDFft1kMap M, int nblk, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (wave-uniform by construction: the
    v2f* const X = reinterpret_cast<v2f*>(smem + wave * F1_WAVE_BYTES);                             //  block map below is scalar code)
    const int gw = blockIdx.x * F1_WAVES + wave;
    v2f* const HP = reinterpret_cast<v2f*>(smem + F1_HP_OFF);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float2 h = T.hp[threadIdx.x + 256 * i];
        const int e = threadIdx.x + 256 * i;                   // T.hp is [16][64]: value k of lane l at 64 k + l
        HP[(e & 63) * F1_HP_STRIDE + (e >> 6)] = (v2f){h.x, h.y};
    }
    __syncthreads();                           // the only barrier of the kernel, before any wave may leave
    const v2f* const hp = HP + lane * F1_HP_STRIDE;
    // (the edge blocks are calls: made while no table is live in registers, or everything live is spilled around them)
    F1KernArgsPtr ka = (F1KernArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    if (gw == 0) f1_edge_block<U8, CX>(ka, 0, X, hp, lane, nblk);
    // this wave's interior blocks: run k = [1 + start_k, 1 + start_k + len_k)
    const int ni = nblk - 2;
    const int w0 = ni > 0 ? (int)(((int64_t)ni * gw) / nwaves) : 0, w1 = ni > 0 ? (int)(((int64_t)ni * (gw + 1)) / nwaves) : 0;
    const bool big = (w1 - w0) > M.b;
    const int cbig = w0 - M.b * gw;                 // waves before this one that own b + 1 blocks
#define F1_RUN(k, st, ln) do { const int r0_ = M.r0[k], r1_ = M.r1[k]; st = 1 + M.wstart[k] + gw * r0_ + cbig * (r1_ - r0_); ln = big ? r1_ : r0_; } while (0)
    int k = 0, q = 0, rem = 0;
    for (; k < M.K; ++k) {
        F1_RUN(k, q, rem);
        if (rem > 0) break;
    }
    if (w1 > w0 && k < M.K) {
        const int tcol = 2 * (lane & 31) + (lane >> 5);           // the column this lane transforms in passes 1 and 6
        v2f tw1[16], tw3[16];
#pragma unroll
        for (int i = 1; i < 16; ++i) {
            const float2 u = T.tw1[tcol * 16 + i];
            tw1[i] = (v2f){u.x, u.y};
            if (i & 3) {                                               // pass-3 layout: register 4 c + n0 of lane j holds k1 = 4 c + j
                const float2 w = T.tw2[(i & 3) * 16 + (i & 12) + (lane & 3)];
                tw3[i] = (v2f){w.x, w.y};
            }
        }
        const v2f crot = {T.crot.x, T.crot.y};
        float* const outp = reinterpret_cast<float*>(P.out);
        F1Cx cx;
        if (CX) {
            const bool nco = (P.flags & DD_CHAIN_NCO) != 0;
            const float2 l0 = nco ? dd_phasor((uint64_t)(2 * lane) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
            const float2 l1 = nco ? dd_phasor((uint64_t)(2 * lane + 1) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
            cx.lp0 = (v2f){l0.x, l0.y}; cx.lp1 = (v2f){l1.x, l1.y};
            cx.pb = (v2f){1.f, 0.f};
            cx.rowph = T.rowph;
        }
        v2f a[16], keep[4], jl[4];
        f1_load_pairs<U8>(P.in, (int64_t)F1_ADV * q + T.base - 256, lane, a, 0, 8);
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
        f1_swap(a[0], a[1]);
        f1_swap(a[2], a[3]);
#ifdef FF_TRACE
        unsigned tr[FF_NPH];
#pragma unroll
        for (int i = 0; i < FF_NPH; ++i) tr[i] = 0;
        const unsigned tloop = (unsigned)__builtin_readcyclecounter();
#endif
        for (;;) {
            int qn = q + 1;
            bool jump = false, done = false;
            if (--rem <= 0) {                      // (wave-uniform) the run ends with this block: where is the next one?
                int st = 0, ln = 0;
                for (++k; k < M.K; ++k) {
                    F1_RUN(k, st, ln);
                    if (ln > 0) break;
                }
                if (k < M.K) { qn = st; rem = ln; jump = true; }
                else { qn = q; done = true; }        // (the last one re-reads itself: no branch in the block)
            }
            const int64_t p0 = (int64_t)F1_ADV * q + T.base;
            const int64_t n0_next = (int64_t)F1_ADV * qn + T.base - 256;
            if (CX) {
                const float2 pbf = (P.flags & DD_CHAIN_NCO) ? dd_phasor((uint64_t)(P.abs0 + p0) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
                cx.pb = (v2f){pbf.x, pbf.y};
                f1_block<U8, false, true, true>(a, keep, jl, X, tw1, tw3, hp, crot, T.theta_sub, lane, P.in, n0_next,
                                                reinterpret_cast<float*>(reinterpret_cast<float2*>(P.out) + p0), 0, F1_ADV, jump, nullptr, &cx);
            } else {
#ifdef FF_TRACE
                f1_block<U8, false, true>(a, keep, jl, X, tw1, tw3, hp, crot, T.theta_sub, lane, P.in, n0_next, outp + (p0 - P.s), 0, F1_ADV, jump, nullptr, nullptr, tr);
#else
                f1_block<U8, false, true>(a, keep, jl, X, tw1, tw3, hp, crot, T.theta_sub, lane, P.in, n0_next, outp + (p0 - P.s), 0, F1_ADV, jump);
#endif
            }
            if (done) break;
            if (jump) {
                f1_swap(jl[0], jl[1]);
                f1_swap(jl[2], jl[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = jl[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = keep[r];
            }
            q = qn;
        }
#ifdef FF_TRACE
        tr[13] = (unsigned)__builtin_readcyclecounter() - tloop;
        if (gw < 4096 && lane == 0) {
#pragma unroll
            for (int i = 0; i < FF_NPH; ++i) g_ff_trace[gw * (FF_NPH + 2) + i] = tr[i];
            g_ff_trace[gw * (FF_NPH + 2) + FF_NPH] = (unsigned long long)(w1 - w0);
        }
#endif
    }
#undef F1_RUN
    if (gw == nwaves - 1 && nblk > 1) f1_edge_block<U8, CX>(ka, nblk - 1, X, hp, lane, nblk);
}

// ============================================================================ host side
#define F1_HP_SLOTS 4
struct DDFftState {
    int K;
    std::vector<double> taps;
    float2* tw1;        // [64][16]  W1024^{t k}
    float2* tw2;        // [4][16]   W64^{n0 k}
    // the tap spectrum for the frequency in `cyc`, as pass 3 multiplies it ([16][64]): a ring of device tables, each with a
    // pinned staging copy and an event, so that a caller that retunes from chunk to chunk (a Doppler-tracking loop, cf.
    // decode_funcube.py:228) never stalls the stream: the new spectrum is computed on the host, copied asynchronously in
    // stream order into the NEXT slot, and a slot is only rewritten once everything that used it has finished
    float2* hp[F1_HP_SLOTS];
    float2* hp_host[F1_HP_SLOTS];
    hipEvent_t hp_ev[F1_HP_SLOTS];
    int hp_ev_set[F1_HP_SLOTS];
    int cur;
    uint64_t cyc;
    int have_h;
    int nco;
};

static void fft_pow2(std::vector<std::complex<double>>& v) {
    // iterative radix-2, double precision (host, once per (taps, frequency))
    const int n = (int)v.size();
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(v[i], v[j]);
    }
    for (int len = 2; len <= n; len <<= 1) {
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const double ang = -2.0 * M_PI * (double)k / (double)len;
                const std::complex<double> w(cos(ang), sin(ang));
                const std::complex<double> u = v[i + k], x = v[i + k + len / 2] * w;
                v[i + k] = u + x;
                v[i + k + len / 2] = u - x;
            }
    }
}

int dd_fft1k_supported(int K, int M, int flags) {
    (void)flags;                          // FM angles or complex64 output, complex64 or raw u8 input
    return (M == 1 && K >= 2 && K <= 256) ? 1 : 0;
}

int dd_fft_create(void** st, const double* taps, int K) {
    if (K < 2 || K > 256) return DD_ERR_UNSUPPORTED;
    DDFftState* s = new DDFftState();
    s->K = K;
    s->taps.assign(taps, taps + K);
    s->tw1 = s->tw2 = nullptr;
    for (int i = 0; i < F1_HP_SLOTS; ++i) { s->hp[i] = nullptr; s->hp_host[i] = nullptr; s->hp_ev[i] = nullptr; s->hp_ev_set[i] = 0; }
    s->cur = 0;
    s->cyc = 0;
    s->have_h = 0;
    s->nco = 0;
    std::vector<float2> u1(64 * 16), u2(4 * 16);
    for (int t = 0; t < 64; ++t)
        for (int k = 0; k < 16; ++k) {
            const double ang = -2.0 * M_PI * (double)((t * k) % F1_N) / (double)F1_N;
            u1[t * 16 + k] = make_float2((float)cos(ang), (float)sin(ang));
        }
    for (int n0 = 0; n0 < 4; ++n0)
        for (int k = 0; k < 16; ++k) {
            const double ang = -2.0 * M_PI * (double)((n0 * k) % 64) / 64.0;
            u2[n0 * 16 + k] = make_float2((float)cos(ang), (float)sin(ang));
        }
    hipError_t e = hipMalloc((void**)&s->tw1, u1.size() * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc((void**)&s->tw2, u2.size() * sizeof(float2));
    for (int i = 0; i < F1_HP_SLOTS && e == hipSuccess; ++i) {
        e = hipMalloc((void**)&s->hp[i], 64 * 16 * sizeof(float2));
        if (e == hipSuccess) e = hipHostMalloc((void**)&s->hp_host[i], 64 * 16 * sizeof(float2), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->hp_ev[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMemcpy(s->tw1, u1.data(), u1.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->tw2, u2.data(), u2.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        dd_fft_destroy(s);
        dd_set_error("dd_fft_create: %s", hipGetErrorString(e));
        return DD_ERR_HIP;
    }
    *st = s;
    return DD_OK;
}

void dd_fft_destroy(void* stv) {
    DDFftState* s = reinterpret_cast<DDFftState*>(stv);
    if (!s) return;
    if (s->tw1) (void)hipFree(s->tw1);
    if (s->tw2) (void)hipFree(s->tw2);
    for (int i = 0; i < F1_HP_SLOTS; ++i) {
        if (s->hp[i]) (void)hipFree(s->hp[i]);
        if (s->hp_host[i]) (void)hipHostFree(s->hp_host[i]);
        if (s->hp_ev[i]) (void)hipEventDestroy(s->hp_ev[i]);
    }
    delete s;
}

// tap spectrum for the NCO frequency of this launch (frac(f/fs) = cyc / 2^64), in the order pass 3 multiplies it
static int fft_prepare(DDFftState* s, bool nco, uint64_t cyc, hipStream_t stream) {
    if (s->have_h && s->cyc == cyc && s->nco == (int)nco) return DD_OK;
    const long double frac = nco ? (long double)cyc / 18446744073709551616.0L : 0.0L;      // [0, 1)
    std::vector<std::complex<double>> g(F1_N, std::complex<double>(0.0, 0.0));
    for (int k = 0; k < s->K; ++k) {
        // e^{+j 2 pi frac k}, argument reduced exactly before the call
        long double ph = frac * (long double)k;
        ph -= floorl(ph);
        const long double a = 2.0L * 3.14159265358979323846264338327950288L * ph;
        g[k] = std::complex<double>((double)(s->taps[k] * cosl(a)), (double)(s->taps[k] * sinl(a)));
    }
    fft_pow2(g);
    // the slot after the current one; whatever used it last (>= F1_HP_SLOTS - 1 retunes ago) is waited for -- normally long done
    const int nxt = s->have_h ? (s->cur + 1) % F1_HP_SLOTS : s->cur;
    if (s->have_h) {
        DD_HIP_CHECK(hipEventRecord(s->hp_ev[s->cur], stream));        // every launch so far that read the current table
        s->hp_ev_set[s->cur] = 1;
    }
    if (s->hp_ev_set[nxt]) DD_HIP_CHECK(hipEventSynchronize(s->hp_ev[nxt]));
    float2* const hp = s->hp_host[nxt];
    for (int k0 = 0; k0 < 16; ++k0)
        for (int j = 0; j < 4; ++j)
            for (int c = 0; c < 4; ++c)
                for (int k2 = 0; k2 < 4; ++k2) {
                    const std::complex<double> h = g[k0 + 16 * (4 * c + j) + 256 * k2] / (double)F1_N;
                    hp[(4 * c + k2) * 64 + 4 * k0 + j] = make_float2((float)h.real(), (float)h.imag());
                }
    DD_HIP_CHECK(hipMemcpyAsync(s->hp[nxt], hp, 64 * 16 * sizeof(float2), hipMemcpyHostToDevice, stream));
    s->cur = nxt;
    s->cyc = cyc;
    s->nco = (int)nco;
    s->have_h = 1;
    return DD_OK;
}

#ifdef FF_TRACE
extern "C" int dd_debug_fft_trace(unsigned long long* out, int nwaves) {
    DD_HIP_CHECK(hipDeviceSynchronize());
    DD_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ff_trace), sizeof(unsigned long long) * (size_t)nwaves * (FF_NPH + 2)));
    return DD_OK;
}
#endif

// Host arithmetic of a launch -- where the block grid sits (DDFft1kTabs::base), how many blocks and waves there are and which
// blocks a wave takes in which round (DDFft1kMap) -- as a function of its own, so that the CPU test suite can check it without a
// GPU (dd_debug_fft1k_plan: every interior block exactly once, waves balanced, runs contiguous).
struct DDFft1kPlan { int base, nblk, grid, nwaves; DDFft1kMap map; };
static void fft1k_plan(int64_t L, int s, int out_align_elems, int ncu, int per_cu, int rounds, DDFft1kPlan* pl) {
    pl->base = s - out_align_elems;
    while (pl->base > L - 1) pl->base -= 16;                     // (a chunk of one sample: the last block must hold output L - 1)
    const int nb = (int)((L - pl->base + F1_ADV - 1) / F1_ADV);
    int grid = ncu * (per_cu > 0 ? per_cu : 3);
    if (grid * F1_WAVES > nb) grid = (nb + F1_WAVES - 1) / F1_WAVES;
    const int nw = grid * F1_WAVES;
    // rounds of the block -> wave map: runs of ~F1_RUN_BLOCKS blocks (rounds > 0 forces a count; 1 = round 3's map)
    DDFft1kMap& M = pl->map;
    const int ni = nb > 2 ? nb - 2 : 0;
    const int b = ni / nw;                                       // a wave owns b or b + 1 interior blocks
    int K = rounds > 0 ? rounds : (b + 1 + F1_RUN_BLOCKS - 1) / F1_RUN_BLOCKS;
    if (K > b + 1) K = b + 1;
    if (K > F1_MAXK) K = F1_MAXK;
    if (K < 1) K = 1;
    const int64_t nbig = (int64_t)ni - (int64_t)b * nw;          // waves that own b + 1
    int64_t ws = 0;
    for (int k = 0; k < K; ++k) {
        M.r0[k] = (int)(((int64_t)b * (k + 1)) / K - ((int64_t)b * k) / K);
        M.r1[k] = (int)(((int64_t)(b + 1) * (k + 1)) / K - ((int64_t)(b + 1) * k) / K);
        M.wstart[k] = (int)ws;
        ws += (int64_t)(nw - nbig) * M.r0[k] + nbig * M.r1[k];
    }
    for (int k = K; k < F1_MAXK; ++k) M.r0[k] = M.r1[k] = M.wstart[k] = 0;
    M.K = K;
    M.b = b;
    pl->nblk = nb;
    pl->grid = grid;
    pl->nwaves = nw;
}
// diagnostic (no GPU needed): the plan for a chunk of L samples whose first `s` FIR outputs have no angle, whose `out` pointer sits
// `out_align_elems` elements behind a 64-byte line, on ncu compute units; out[0..6] = base, nblk, grid, nwaves, K, b, F1_MAXK, then
// r0[32], r1[32], wstart[32]
extern "C" int dd_debug_fft1k_plan(int64_t L, int s, int out_align_elems, int ncu, int rounds, int* out) {
    DD_REQUIRE(L >= 1 && (s == 0 || s == 1) && out_align_elems >= 0 && out_align_elems < 16 && ncu >= 1 && out, "arguments");
    DDFft1kPlan pl;
    fft1k_plan(L, s, out_align_elems, ncu, 3, rounds, &pl);
    out[0] = pl.base; out[1] = pl.nblk; out[2] = pl.grid; out[3] = pl.nwaves; out[4] = pl.map.K; out[5] = pl.map.b; out[6] = F1_MAXK;
    for (int k = 0; k < F1_MAXK; ++k) { out[7 + k] = pl.map.r0[k]; out[7 + F1_MAXK + k] = pl.map.r1[k]; out[7 + 2 * F1_MAXK + k] = pl.map.wstart[k]; }
    return DD_OK;
}

// the whole chunk through k_chain_fft1k (one launch, carried state included)
int dd_fft1k_launch(void* stv, const DDChainParams& P, hipStream_t stream) {
    DDFftState* s = reinterpret_cast<DDFftState*>(stv);
    const bool nco = (P.flags & DD_CHAIN_NCO) != 0;
    int rc = fft_prepare(s, nco, P.cyc, stream);
    if (rc != DD_OK) return rc;
    static DDOncePerDevice attr1;
    if (attr1.need()) {
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_fft1k<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS_BYTES));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_fft1k<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS_BYTES));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_fft1k<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS_BYTES));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_fft1k<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS_BYTES));
        attr1.mark();
    }
    DDFft1kTabs T1;
    T1.tw1 = s->tw1; T1.tw2 = s->tw2; T1.hp = s->hp[s->cur];
    {
        const long double frac = nco ? (long double)P.cyc / 18446744073709551616.0L : 0.0L;
        const long double a = 2.0L * 3.14159265358979323846264338327950288L * frac;
        // theta in (-pi, pi]; small -> subtracted from the angle, else applied as a rotation of every product
        const long double th = a > 3.14159265358979323846264338327950288L ? a - 2.0L * 3.14159265358979323846264338327950288L : a;
        static const char* rot_env = DD_TUNE_ENV("DD_FFT_ROTATE");          // tools: force the rotation form
        if (fabsl(th) <= 0.25L && th != 0.0L && !(rot_env && atoi(rot_env))) {
            T1.theta_sub = (float)th;
            T1.crot = make_float2((float)cosl(a), (float)-sinl(a));     // (the full-range angle path rotates by it instead)
        } else {
            T1.theta_sub = 0.f;
            T1.crot = make_float2((float)cosl(a), (float)-sinl(a));
        }
    }
    // the block grid is laid so that block b's first angle, out[768 b + base - s], starts a 64-byte line of `out`
    // (DD_FFT_FRAME=0, tools: the grid starts at output 0 whatever the alignment, as before round 4)
    static const char* frame_env = DD_TUNE_ENV("DD_FFT_FRAME");
    const bool cxout = !(P.flags & DD_CHAIN_FM);
    {
        const long double frac = nco ? (long double)P.cyc / 18446744073709551616.0L : 0.0L;
        for (int i = 0; i < 6; ++i) {
            long double ph = frac * (long double)(128 * i);
            ph -= floorl(ph);
            const long double ang = 2.0L * 3.14159265358979323846264338327950288L * ph;
            T1.rowph[i] = make_float2((float)cosl(ang), (float)-sinl(ang));
        }
    }
    // float32 angles: 16 per 64-byte line; complex64 outputs: 8 per line (and no demod_fm shift)
    const int a16 = cxout ? (int)((reinterpret_cast<uintptr_t>(P.out) >> 3) & 7) : (int)((reinterpret_cast<uintptr_t>(P.out) >> 2) & 15);
    static const char* wg_env1 = DD_TUNE_ENV("DD_FFT_WGS_PER_CU");           // tools: occupancy experiments
    static const char* rounds_env = DD_TUNE_ENV("DD_FFT_ROUNDS");            // tools: a fixed number of rounds (1 = round 3's map)
    DDFft1kPlan pl;
    fft1k_plan(P.L, P.s, (frame_env && atoi(frame_env) == 0) ? P.s : a16, dd_cu_count(), wg_env1 ? atoi(wg_env1) : 3, rounds_env ? atoi(rounds_env) : 0, &pl);
    T1.base = pl.base;
    const int nb1 = pl.nblk, grid1 = pl.grid, nw1 = pl.nwaves;
    const DDFft1kMap& M1 = pl.map;
    const bool u8 = (P.flags & DD_CHAIN_U8_INPUT) != 0;
    if (u8 && cxout) hipLaunchKernelGGL((k_chain_fft1k<true, true>), dim3(grid1), dim3(64 * F1_WAVES), F1_LDS_BYTES, stream, P, T1, M1, nb1, nw1);
    else if (u8) hipLaunchKernelGGL((k_chain_fft1k<true, false>), dim3(grid1), dim3(64 * F1_WAVES), F1_LDS_BYTES, stream, P, T1, M1, nb1, nw1);
    else if (cxout) hipLaunchKernelGGL((k_chain_fft1k<false, true>), dim3(grid1), dim3(64 * F1_WAVES), F1_LDS_BYTES, stream, P, T1, M1, nb1, nw1);
    else hipLaunchKernelGGL((k_chain_fft1k<false, false>), dim3(grid1), dim3(64 * F1_WAVES), F1_LDS_BYTES, stream, P, T1, M1, nb1, nw1);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// dd_code_warmup (dd_runtime.hip): the runtime loads a translation unit's code object when one of its kernels is first named
int dd_code_touch_fftfir(void) {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, (const void*)k_chain_fft1k<false, false>) == hipSuccess ? DD_OK : DD_ERR_HIP;
}
