// Fused hot path as an f32 overlap-save FFT convolution held in LDS (general taps, M == 1):
//
//     offsetFreq (NCO, commuted into the taps) -> FIR as a 4096-point circular convolution -> demod_fm
//
// Why: the three-limb f16 Toeplitz GEMM of dd_mfma.hip spends 108 MFMAs per
// 1024 outputs whatever the limb format (f16 x f16 limbs, exact 8-bit data x i8 tap limbs, Karatsuba forms: all 6
// matrix instructions per 16 taps), and the chip holds only 1.57 GHz under it.  A 4096-point block FFT costs ~80
// packed f32 instructions per block point for forward transform + spectrum product + inverse transform, independent
// of the tap count up to 256, runs on the vector pipe alone, and takes complex taps for free -- so the NCO
//     y[p] = sum_k g[k] x[p-k] e^{-j th (p-k)} = e^{-j th p} sum_k (g[k] e^{j th k}) x[p-k] = e^{-j th p} w[p]
// moves into the tap spectrum and the discriminator sees  y[p] conj(y[p-1]) = e^{-j th} w[p] conj(w[p-1]).
//
// Block geometry: N = 4096 = 16^3, 256 threads, 16 points per thread and pass.  Block q of a run covers FIR outputs
// p0 .. p0+3839 (p0 = p_a + 3840 q); it reads the 4096 inputs n0 = p0-256 .. p0+3839, so block position m = t + 256 r
// (thread t, register r) holds w[n0 + m]; rows r = 1..15 are the block's 3840 outputs, row 0 is the 256-sample
// overlap (positions >= K-1 of a circular convolution are linear; K <= 256) of which only m = 255 is used, as the
// left-hand neighbour of output 0.
//
// Index algebra (n = 256 n2 + 16 n1 + n0, k = k0 + 16 k1 + 256 k2, W = e^{-2 pi j / 4096}):
//     W^{nk} = W16^{n2 k0} . W^{(16 n1 + n0) k0} . W16^{n1 k1} . W256^{n0 k1} . W16^{n0 k2}
//   forward (DIF): B1 over n2 | T1 = W^{t k0} | X1 | B2 over n1 | T2 = W256^{n0 k1} | X2 | B3 over n0
//   inverse (DIT, the transposed graph): B3^H | X2^T | T2* | B2^H | X1^T | T1* | B1^H
// so each thread keeps ONE set of 15 + 15 twiddles for both directions, the spectrum is multiplied in the permuted
// order in which pass 3 leaves it (thread (k0,k1), register k2: its 16 H values are loop-invariant registers), the
// inverse exchanges write back to the very addresses the thread read in the forward exchange (no buffer hazards), and
// the result arrives in natural order in the layout the input was loaded in.  Five workgroup barriers per block.
//
// LDS: X1 [16][272] and X2 [16][289] complex64 (row strides chosen so that both the 16-lane-contiguous ds_write_b64
// and the strided ds_read_b64 of every exchange are bank-conflict free), 72 KB per workgroup, two workgroups per CU.
#include "dd_chain_kernels.h"
#include "dd_fftfir.h"
#include <stdlib.h>
#include <complex>
#include <mutex>

typedef float v2f __attribute__((ext_vector_type(2)));

#define FF_N 4096
#define FF_THREADS 256
#define FF_ADV 3840                 // outputs per block = rows 1..15
#define FF_S1 272                   // X1 row stride (complex): 2*272 mod 64 = 32
#define FF_S2 289                   // X2 row stride (complex): 2*289 mod 64 = 2, element (k0, k1, n0) at k0*289 + k1*17 + n0
#define FF_X1_BYTES (16 * FF_S1 * 8)
#define FF_X2_BYTES (16 * FF_S2 * 8)
#define FF_EDGE_OFF (FF_X1_BYTES + FF_X2_BYTES)
#define FF_LDS_BYTES (FF_EDGE_OFF + 4 * 16 * 8)

struct DDFftTabs {
    const float2* tw1;     // [256][16]  W4096^{t k}
    const float2* tw2;     // [16][16]   W256^{n0 k}
    const float2* hp;      // [256][16]  H[k0 + 16 k1 + 256 k2] / 4096 at [16 k0 + k1][k2]
    float2 crot;           // e^{-j theta}: the NCO's per-sample rotation as the discriminator sees it
    float2* dbg;           // diagnostic instantiation only: receives the 16 registers of every thread after stage dbg_stage
    int dbg_stage;
};

// ---- complex arithmetic on packed pairs (x = re, y = im): v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 ----
__device__ __forceinline__ v2f ff_cmul(v2f a, v2f w) {          // a * w
    const v2f t = a * w.xx;
    return __builtin_elementwise_fma(a.yx, (v2f){-w.y, w.y}, t);
}
__device__ __forceinline__ v2f ff_cmulc(v2f a, v2f w) {         // a * conj(w)
    const v2f t = a * w.xx;
    return __builtin_elementwise_fma(a.yx, (v2f){w.y, -w.y}, t);
}
template <bool INV>
__device__ __forceinline__ v2f ff_tw(v2f a, v2f w) { return INV ? ff_cmulc(a, w) : ff_cmul(a, w); }
// register that holds output k of ff_bfly16
#define FF_P(k) (4 * ((k) & 3) + ((k) >> 2))

// the same with a loop-invariant factor held in a register pair: written as the two instructions with their operand
// selects and sign modifiers spelt out.  Left to the compiler, the splat (w.x, w.x) and the signed swizzles (-w.y, w.y),
// (w.y, -w.y) of all 46 factors are hoisted out of the block loop as registers of their own -- 6 registers per factor.
// On gfx950 a packed-f32 result may not be read by the very next VALU instruction (one wait state; the compiler pads
// with s_nop where it cannot find an independent instruction, and it does not look inside asm statements), so the two
// halves of a product are issued as separate statements and the callers keep a dependent pair at least one
// instruction apart: all first halves of a group, then all second halves.
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

// 4-point DFT, forward W4 = -j, inverse +j; ROT2: input x2 still carries a factor -j (forward) / +j (inverse)
template <bool INV, bool ROT2>
__device__ __forceinline__ void ff_r4(v2f& x0, v2f& x1, v2f& x2, v2f& x3) {
    v2f s0, s1;
    if (ROT2) {
        s0 = INV ? ff_addpj(x0, x2) : ff_addmj(x0, x2);
        s1 = INV ? ff_addmj(x0, x2) : ff_addpj(x0, x2);
    } else {
        s0 = x0 + x2;
        s1 = x0 - x2;
    }
    const v2f s2 = x1 + x3, d = x1 - x3;
    x0 = s0 + s2;
    x2 = s0 - s2;
    x1 = INV ? ff_addpj(s1, d) : ff_addmj(s1, d);
    x3 = INV ? ff_addmj(s1, d) : ff_addpj(s1, d);
}

// 16-point DFT in registers, 4 x 4: a[4p + q] in, output k = c + 4d in a[4c + d] = a[FF_P(k)]
template <bool INV>
__device__ __forceinline__ void ff_bfly16(v2f (&a)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) ff_r4<INV, false>(a[q], a[4 + q], a[8 + q], a[12 + q]);
    // a[4c + q] = u[q][c];  u[q][c] *= W16^{q c} (conjugated for the inverse); constants in scalar register pairs
    const v2f w1 = {0.92387953251128674f, -0.38268343236508977f};
    const v2f w2 = {0.70710678118654752f, -0.70710678118654752f};
    const v2f w3 = {0.38268343236508977f, -0.92387953251128674f};
    const v2f w6 = {-0.70710678118654752f, -0.70710678118654752f};
    const v2f w9 = {-0.92387953251128674f, 0.38268343236508977f};
    const v2f t5 = ff_mul_lo_s(a[5], w1), t6 = ff_mul_lo_s(a[6], w2), t7 = ff_mul_lo_s(a[7], w3), t9 = ff_mul_lo_s(a[9], w2);
    const v2f t11 = ff_mul_lo_s(a[11], w6), t13 = ff_mul_lo_s(a[13], w3), t14 = ff_mul_lo_s(a[14], w6), t15 = ff_mul_lo_s(a[15], w9);
    a[5] = ff_fma_hi_s<INV>(a[5], w1, t5);
    a[6] = ff_fma_hi_s<INV>(a[6], w2, t6);
    a[7] = ff_fma_hi_s<INV>(a[7], w3, t7);
    a[9] = ff_fma_hi_s<INV>(a[9], w2, t9);
    a[11] = ff_fma_hi_s<INV>(a[11], w6, t11);
    a[13] = ff_fma_hi_s<INV>(a[13], w3, t13);
    a[14] = ff_fma_hi_s<INV>(a[14], w6, t14);
    a[15] = ff_fma_hi_s<INV>(a[15], w9, t15);
    // u[2][2] *= W16^4 = -j: folded into the additions of the second stage (ROT2)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c == 2) ff_r4<INV, true>(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
        else ff_r4<INV, false>(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
    }
}

__device__ __forceinline__ float ff_atan2(float y, float x) {
    // odd degree-15 minimax polynomial on [0,1] + octant fix-up (same as dd_mfma.hip's discriminator)
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float z = t * t;
    float p = -4.054567120e-03f;
    p = fmaf(p, z, 2.186295773e-02f);
    p = fmaf(p, z, -5.591232695e-02f);
    p = fmaf(p, z, 9.642197381e-02f);
    p = fmaf(p, z, -1.390862959e-01f);
    p = fmaf(p, z, 1.994656567e-01f);
    p = fmaf(p, z, -3.332986079e-01f);
    p = fmaf(p, z, 9.999993356e-01f);
    float r = p * t;
    r = (mx == 0.f) ? 0.f : r;
    r = (ay > ax) ? 1.5707963267948966f - r : r;
    r = (x < 0.f) ? 3.141592653589793f - r : r;
    return copysignf(r, y);
}

// atan(y/x) for x > 0, |y| <= tan(pi/8) x (minimax fit, 2.3e-8 rad evaluated in f32), no octant logic
__device__ __forceinline__ float ff_atan_small(float y, float x) {
    const float t = y * __builtin_amdgcn_rcpf(x);
    const float z = t * t;
    float p = fmaf(7.902598251e-02f, z, -1.382445378e-01f);
    p = fmaf(p, z, 1.997187931e-01f);
    p = fmaf(p, z, -3.333275667e-01f);
    return fmaf(t, z * p, t);
}

__device__ __forceinline__ float ff_lane_left(float v) {       // wave_shr:1 (lane 0 keeps its own value; patched by the caller)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}

template <bool U8>
__device__ __forceinline__ void ff_load_block(const void* in, int64_t n0, int64_t nmax, int t, v2f (&x)[16]) {
    // x[r] = sample n0 + t + 256 r (n0 is workgroup-uniform: scalar base, 32-bit lane offsets).  Only the run's last
    // block may reach past the chunk's end -- behind the run's last output -- and clamps its indices.
    const unsigned lim = (n0 + FF_N - 1 <= nmax) ? (unsigned)(FF_N - 1) : (unsigned)(nmax - n0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned m = (unsigned)t + 256u * r;
        m = m < lim ? m : lim;
        if (U8) {
            const uchar2 u = (reinterpret_cast<const uchar2*>(in) + n0)[m];
            x[r] = (v2f){(float)u.x - 127.5f, (float)u.y - 127.5f};
        } else {
            const float2 v = (reinterpret_cast<const float2*>(in) + n0)[m];
            x[r] = (v2f){v.x, v.y};
        }
    }
}

// FIR outputs [p_a, p_b) of the chunk (interior: every input the blocks read up to the run's last output lies inside the
// chunk) -> FM angles out[p - s].  Persistent: workgroup g takes a contiguous run of the nblk blocks.
// timing ablations (tools/mkvariant.sh N dd_fftfir -DFF_NO_xxx; results are wrong by construction, never shipped):
//   FF_NO_STORE  no output stores        FF_NO_LOAD   no input loads (the first block's samples are reused)
//   FF_NO_BARRIER no workgroup barriers  FF_NO_LDS    no exchanges at all (registers pass straight through)
//   FF_NO_DISC   no discriminator (one component of the FIR output is stored)
#ifdef FF_NO_BARRIER
#define FF_SYNC() asm volatile("" ::: "memory")
#else
#define FF_SYNC() __syncthreads()
#endif
#define FF_STAMP(i)                                                                                   \
    if (DBG == 2) {                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        const unsigned long long tn = __builtin_readcyclecounter();                                   \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        acc_t[i] += tn - tp;                                                                          \
        tp = tn;                                                                                      \
    }
#define FF_DUMP(st)                                                                                   \
    if (DBG == 1 && T.dbg_stage == (st)) {                                                                 \
        for (int k = 0; k < 16; ++k) T.dbg[t * 16 + k] = make_float2(a[k].x, a[k].y);                 \
        return;                                                                                       \
    }

template <bool U8, int DBG = 0>
__global__ void __launch_bounds__(FF_THREADS, 2) k_chain_fft(const DDChainParams P, const DDFftTabs T, int64_t p_a, int64_t p_b, int nblk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f* const X1 = reinterpret_cast<v2f*>(smem);
    v2f* const X2 = reinterpret_cast<v2f*>(smem + FF_X1_BYTES);
    v2f* const EDGE = reinterpret_cast<v2f*>(smem + FF_EDGE_OFF);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int hi = t >> 4, lo = t & 15;

    v2f tw1[16], tw2[16], hp[16];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const float2 a = T.tw1[t * 16 + k], b = T.tw2[lo * 16 + k];
        tw1[k] = (v2f){a.x, a.y};
        tw2[k] = (v2f){b.x, b.y};
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float2 h = T.hp[t * 16 + k];
        hp[k] = (v2f){h.x, h.y};
    }
    const v2f crot = {T.crot.x, T.crot.y};

    const int G = gridDim.x, g = blockIdx.x;
    const int q_begin = (int)(((int64_t)nblk * g) / G), q_end = (int)(((int64_t)nblk * (g + 1)) / G);
    if (q_begin >= q_end) return;
    const int64_t nmax = P.L - 1;
    float* const outp = reinterpret_cast<float*>(P.out);

    v2f a[16], nx[16];
    ff_load_block<U8>(P.in, p_a + (int64_t)FF_ADV * q_begin - 256, nmax, t, nx);

    // exchange addresses (complex units)
    const int x1w = t;                          // + 272 k0      thread (n1, n0) writes element k0
    const int x1r = hi * FF_S1 + lo;            // + 16 n1       thread (k0, n0) reads element n1
    const int x2w = hi * FF_S2 + lo;            // + 17 k1       thread (k0, n0) writes element k1
    const int x2r = hi * FF_S2 + lo * 17;       // + n0          thread (k0, k1) reads element n0

    // every load issued so far (tables, first block) is complete before the loop: the compiler's wait-count analysis merges
    // the loop header's state with the pre-loop state on every iteration, so a table register first used inside the loop
    // would otherwise get a vmcnt wait there that, in the steady state, waits for the block loads issued just before it
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
    unsigned long long acc_t[24], tp = 0;
    if (DBG == 2) {
#pragma unroll
        for (int i = 0; i < 24; ++i) acc_t[i] = 0;
        tp = __builtin_readcyclecounter();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = nx[r];
    for (int q = q_begin; q < q_end; ++q) {
        const int64_t p0 = p_a + (int64_t)FF_ADV * q;
#ifndef FF_NO_LOAD
        if (q + 1 < q_end) ff_load_block<U8>(P.in, p0 + FF_ADV - 256, nmax, t, nx);
#endif

        FF_DUMP(0)
        // ---- forward pass 1 (over n2), T1, X1
        ff_bfly16<false>(a);
        FF_DUMP(1)
        ff_twiddle15<false, true>(a, tw1);
        FF_DUMP(2)
        FF_STAMP(0)
#ifndef FF_NO_LDS
#pragma unroll
        for (int k = 0; k < 16; ++k) X1[x1w + FF_S1 * k] = a[FF_P(k)];
#endif
        FF_STAMP(1)
        FF_SYNC();
        FF_STAMP(2)
#ifndef FF_NO_LDS
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = X1[x1r + 16 * k];
#endif
        FF_DUMP(3)
        FF_STAMP(3)
        // ---- forward pass 2 (over n1), T2, X2
        ff_bfly16<false>(a);
        FF_DUMP(4)
        ff_twiddle15<false, true>(a, tw2);
        FF_DUMP(5)
        FF_STAMP(4)
#ifndef FF_NO_LDS
#pragma unroll
        for (int k = 0; k < 16; ++k) X2[x2w + 17 * k] = a[FF_P(k)];
#endif
        FF_STAMP(5)
        FF_SYNC();
        FF_STAMP(6)
#ifndef FF_NO_LDS
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = X2[x2r + k];
#endif
        FF_DUMP(6)
        FF_STAMP(7)
        // ---- forward pass 3 (over n0), spectrum product, inverse pass 3
        ff_bfly16<false>(a);
        FF_DUMP(7)
        {
            v2f z[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) z[k] = ff_mul_lo(a[FF_P(k)], hp[k]);
#pragma unroll
            for (int k = 0; k < 16; ++k) z[k] = ff_fma_hi(a[FF_P(k)], hp[k], z[k]);
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = z[k];
        }
        FF_DUMP(15)
        ff_bfly16<true>(a);
        FF_DUMP(8)
        FF_STAMP(8)
#ifndef FF_NO_LDS
#pragma unroll
        for (int k = 0; k < 16; ++k) X2[x2r + k] = a[FF_P(k)];
#endif
        FF_STAMP(9)
        FF_SYNC();
        FF_STAMP(10)
#ifndef FF_NO_LDS
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = X2[x2w + 17 * k];
#endif
        FF_DUMP(9)
        FF_STAMP(11)
        // ---- T2*, inverse pass 2
        ff_twiddle15<true, false>(a, tw2);
        FF_DUMP(10)
        ff_bfly16<true>(a);
        FF_DUMP(11)
        FF_STAMP(12)
#ifndef FF_NO_LDS
#pragma unroll
        for (int k = 0; k < 16; ++k) X1[x1r + 16 * k] = a[FF_P(k)];
#endif
        FF_STAMP(13)
        FF_SYNC();
        FF_STAMP(14)
#ifndef FF_NO_LDS
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = X1[x1w + FF_S1 * k];
#endif
        FF_DUMP(12)
        FF_STAMP(15)
        // ---- T1*, inverse pass 1: a[FF_P(r)] = w[n0 + t + 256 r]
        ff_twiddle15<true, false>(a, tw1);
        FF_DUMP(13)
        ff_bfly16<true>(a);
        FF_DUMP(14)

        // ---- discriminator: rows 1..15, left-hand neighbour = lane - 1 (wave_shr:1); lane 0 takes the previous wave's
        // lane 63 (wave 0: row r-1 of wave 3) from LDS -- the value read there is the DPP's `old` operand, which a lane
        // without a source lane keeps
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < 16; ++r) EDGE[wave * 16 + r] = a[FF_P(r)];
        }
        FF_STAMP(16)
        FF_SYNC();
        FF_STAMP(17)
        {
            const int eidx = wave > 0 ? (wave - 1) * 16 : 3 * 16 - 1;
            v2f zz[16];
#pragma unroll
            for (int r = 1; r < 16; ++r) {
                const v2f e = EDGE[eidx + r];
                const v2f cur = a[FF_P(r)];
                v2f prv;
                prv.x = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(e.x), __float_as_int(cur.x), 0x138, 0xf, 0xf, false));
                prv.y = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(e.y), __float_as_int(cur.y), 0x138, 0xf, 0xf, false));
                zz[r] = prv;
            }
#pragma unroll
            for (int r = 1; r < 16; ++r) { const v2f prv = zz[r]; zz[r] = ff_mul_lo(a[FF_P(r)], prv); a[FF_P(r)] = ff_fma_hic(a[FF_P(r)], prv, zz[r]); }
            // a[FF_P(r)] = w[m] conj(w[m-1]); the NCO's rotation per sample
#pragma unroll
            for (int r = 1; r < 16; ++r) zz[r] = ff_mul_lo(a[FF_P(r)], crot);
#pragma unroll
            for (int r = 1; r < 16; ++r) zz[r] = ff_fma_hi(a[FF_P(r)], crot, zz[r]);
            FF_STAMP(18)
            bool small = true;
#pragma unroll
            for (int r = 1; r < 16; ++r) small = small && (fabsf(zz[r].y) <= 0.41421356f * zz[r].x);
            float ang[16];
#ifdef FF_NO_DISC
#pragma unroll
            for (int r = 1; r < 16; ++r) ang[r] = zz[r].x;
            if (false) {
#else
            if (__builtin_amdgcn_ballot_w64(!small) == 0) {
#endif        // wave-uniform: every |angle| of the wave's 960 outputs below 22.5 degrees
#pragma unroll
                for (int r = 1; r < 16; ++r) ang[r] = ff_atan_small(zz[r].y, zz[r].x);
            } else {
#ifndef FF_NO_DISC
#pragma unroll
                for (int r = 1; r < 16; ++r) ang[r] = ff_atan2(zz[r].y, zz[r].x);
#endif
            }
            FF_STAMP(19)
            // the next block's samples move into the working registers BEFORE this block's stores are issued: the wait for
            // those loads (issued a whole block ago) then cannot include the stores -- vmcnt counts in order, and behind
            // the branches of the store section the compiler has to assume that no younger operation covers the loads
            // (written as volatile moves that memory operations may not cross: as plain assignments the copies are placed
            // on the loop's back edge, behind the stores)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_mov_b64 %0, %1" : "=v"(a[r]) : "v"(nx[r]) : "memory");
            FF_STAMP(20)
            float* const ob = outp + (p0 - P.s) + t;
#ifdef FF_NO_STORE
            float sacc = 0.f;
#pragma unroll
            for (int r = 1; r < 16; ++r) sacc += ang[r];
            if (sacc == 1234.5f) ob[0] = sacc;
#else
            if (p0 + FF_ADV <= p_b) {
#pragma unroll
                for (int r = 1; r < 16; ++r) ob[256 * (r - 1)] = ang[r];
            } else {
#pragma unroll
                for (int r = 1; r < 16; ++r)
                    if (p0 + t + 256 * (r - 1) < p_b) ob[256 * (r - 1)] = ang[r];
            }
#endif
        }
        FF_STAMP(21)
    }
    if (DBG == 2 && lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(T.dbg) + ((size_t)blockIdx.x * 4 + wave) * 24;
#pragma unroll
        for (int i = 0; i < 22; ++i) o[i] = acc_t[i];
        o[23] = (unsigned long long)(q_end - q_begin);
    }
}

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
#define F1_HP_OFF (F1_WAVES * F1_WAVE_BYTES)     // the tap spectrum as the lanes multiply it, [16][64] complex64: one copy per workgroup
#define F1_LDS_BYTES (F1_HP_OFF + 16 * 64 * 8)

struct DDFft1kTabs {
    const float2* tw1;     // [64][16]   W1024^{lane k}
    const float2* tw2;     // [4][16]    W64^{n0 k}
    const float2* hp;      // [16][64]   H[k0 + 16 (4 c + j) + 256 k2] / 1024 at [4 c + k2][4 k0 + j]
    float2 crot;
    int stagger;           // s_sleep units (64 cycles) by which the second / third third of the grid start later
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
template <bool U8, bool CLAMP = false>
__device__ __forceinline__ void f1_load_pairs(const void* in, int64_t n0, int lane, v2f (&x)[16], int r0, int n, unsigned lim = 0) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
        unsigned m = (unsigned)lane + 64u * (r0 + i);
        if (CLAMP) m = m < lim ? m : lim;
        if (U8) {
            const uchar4 u = reinterpret_cast<const uchar4*>(reinterpret_cast<const uchar2*>(in) + n0)[m];
            x[2 * (r0 + i)] = (v2f){(float)u.x - 127.5f, (float)u.y - 127.5f};
            x[2 * (r0 + i) + 1] = (v2f){(float)u.z - 127.5f, (float)u.w - 127.5f};
        } else {
            const float4 v = reinterpret_cast<const float4*>(reinterpret_cast<const float2*>(in) + n0)[m];
            x[2 * (r0 + i)] = (v2f){v.x, v.y};
            x[2 * (r0 + i) + 1] = (v2f){v.z, v.w};
        }
    }
}

// angles of row pairs 2..7 from zz (two per lane and row pair: zz[2 r], zz[2 r + 1]), three groups of two row pairs:
// angles | two of the next block's loads | two 8-byte stores
template <bool U8, bool PARTIAL, bool LOADNEXT, bool FAST>
__device__ __forceinline__ void f1_tail(const v2f (&zz)[16], v2f (&a)[16], const void* in, const int64_t n0_next, const int lane, float* const ob, const int limit) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        float ang[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const v2f z = zz[4 + 4 * g + i];
#ifdef FF_NO_DISC
            ang[i] = z.x;
#else
            ang[i] = FAST ? ff_atan_small(z.y, z.x) : ff_atan2(z.y, z.x);
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
#elif !defined(FF_NO_LOAD)
        if (LOADNEXT) f1_load_pairs<U8>(in, n0_next, lane, a, 2 + 2 * g, 2);
#endif
#ifdef FF_NO_STORE
        if (ang[0] + ang[1] + ang[2] + ang[3] == 1234.5f) ob[0] = ang[0];
#else
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int o = 128 * (2 * g + i);                    // output of this lane's first sample of the row pair, relative to ob
            if (!PARTIAL) {
                *reinterpret_cast<float2*>(ob + o) = make_float2(ang[2 * i], ang[2 * i + 1]);
            } else {
                if (2 * lane + o < limit) ob[o] = ang[2 * i];
                if (2 * lane + o + 1 < limit) ob[o + 1] = ang[2 * i + 1];
            }
        }
#endif
    }
}

// one block.  On entry a[0..3] hold rows 0..3 of column t (the overlap kept from the previous block, or swapped by the
// caller) and a[4..15] row pairs 2..7 as loaded; on exit, when LOADNEXT, the same for the next block.  out_row4 points
// at the block's first output (row 4, column 0).  PARTIAL: outputs at or beyond `limit` (relative to it) are not stored.
template <bool U8, bool PARTIAL, bool LOADNEXT>
__device__ __forceinline__ void f1_block(v2f (&a)[16], v2f (&keep)[4], v2f* const X, const v2f (&tw1)[16], const v2f (&tw2)[16], const v2f* const hp,
                                         const v2f crot, const int lane, const void* in, const int64_t n0_next, float* const out_row4, const int limit) {
    const int hi = lane >> 2, lo = lane & 3;
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
    ff_twiddle15<false, true>(a, tw1);
#ifndef FF_NO_LDS
#pragma unroll
    for (int k = 0; k < 16; ++k) X[x1w + F1_S1 * k] = a[FF_P(k)];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = X[x1r + 2 * k];
#endif
    // ---- forward pass 2 (over n1), T2, X2
    ff_bfly16<false>(a);
    ff_twiddle15<false, true>(a, tw2);
#ifndef FF_NO_LDS
#pragma unroll
    for (int k = 0; k < 16; ++k) X[x2w + 5 * k] = a[FF_P(k)];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < 4; ++n) a[4 * c + n] = X[x2r + 20 * c + n];
#endif
    // ---- forward pass 3 (radix 4 over n0), spectrum product, inverse pass 3
#pragma unroll
    for (int c = 0; c < 4; ++c) ff_r4<false, false>(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
    {
        v2f z[16], h[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = hp[64 * k];        // (this lane's column of the spectrum image in LDS)
#pragma unroll
        for (int k = 0; k < 16; ++k) z[k] = ff_mul_lo(a[k], h[k]);
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = ff_fma_hi(a[k], h[k], z[k]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) ff_r4<true, false>(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
#ifndef FF_NO_LDS
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < 4; ++n) X[x2r + 20 * c + n] = a[4 * c + n];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = X[x2w + 5 * k];
#endif
    // ---- T2*, inverse pass 2
    ff_twiddle15<true, false>(a, tw2);
    ff_bfly16<true>(a);
#ifndef FF_NO_LDS
#pragma unroll
    for (int k = 0; k < 16; ++k) X[x1r + 2 * k] = a[FF_P(k)];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = X[x1w + F1_S1 * k];
#endif
    // ---- T1*, inverse pass 1: a[FF_P(r)] = w[n0 + 64 r + t]
    ff_twiddle15<true, false>(a, tw1);
    ff_bfly16<true>(a);
    // ---- back to two consecutive outputs per lane: B[r] = w[128 r + 2 lane], A[r] = the one after it (row pairs 1..7;
    // of pair 1 only lane 63's second value is used, as the left-hand neighbour of the block's first output)
    v2f zz[16];
    {
        v2f B[8], A[8];
#pragma unroll
        for (int r = 1; r < 8; ++r) { B[r] = a[FF_P(2 * r)]; A[r] = a[FF_P(2 * r + 1)]; f1_swap(B[r], A[r]); }
        // discriminator: the second value's neighbour is the first; the first's is the second of the lane to the left
        // (wave_shr:1), for lane 0 that of lane 63 in the row pair above (wave_ror:1 of that register as the DPP's `old`
        // operand, which a lane without a source lane keeps)
        v2f pb[8];
#pragma unroll
        for (int r = 2; r < 8; ++r) {
            const int ox = __builtin_amdgcn_update_dpp(0, __float_as_int(A[r - 1].x), 0x13C, 0xf, 0xf, false);
            const int oy = __builtin_amdgcn_update_dpp(0, __float_as_int(A[r - 1].y), 0x13C, 0xf, 0xf, false);
            pb[r].x = __int_as_float(__builtin_amdgcn_update_dpp(ox, __float_as_int(A[r].x), 0x138, 0xf, 0xf, false));
            pb[r].y = __int_as_float(__builtin_amdgcn_update_dpp(oy, __float_as_int(A[r].y), 0x138, 0xf, 0xf, false));
        }
        v2f t[16];
#pragma unroll
        for (int r = 2; r < 8; ++r) { t[2 * r] = ff_mul_lo(B[r], pb[r]); t[2 * r + 1] = ff_mul_lo(A[r], B[r]); }
#pragma unroll
        for (int r = 2; r < 8; ++r) { zz[2 * r] = ff_fma_hic(B[r], pb[r], t[2 * r]); zz[2 * r + 1] = ff_fma_hic(A[r], B[r], t[2 * r + 1]); }   // w[m] conj(w[m-1])
#pragma unroll
        for (int r = 4; r < 16; ++r) t[r] = ff_mul_lo(zz[r], crot);
#pragma unroll
        for (int r = 4; r < 16; ++r) zz[r] = ff_fma_hi(zz[r], crot, t[r]);             // the NCO's rotation per sample
    }
#endif
    // a[] is dead from here: the next block's samples fly during the angles and stores.  Its first four rows are this
    // block's last four (the 256-sample overlap), kept in `keep`: six loads per block
#ifndef FF_NO_LOAD
    if (LOADNEXT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = keep[r];
    }
#endif
    // wave-uniform fast path: every |angle| of the block's 768 outputs below 22.5 degrees (|im| <= tan(pi/8) re)
    float worst = -1.0f;
#pragma unroll
    for (int r = 4; r < 16; ++r) worst = fmaxf(worst, fmaf(-0.41421356f, zz[r].x, fabsf(zz[r].y)));
    const bool fast = __builtin_amdgcn_ballot_w64(worst > 0.f) == 0;
    float* const ob = out_row4 + 2 * lane;
    if (fast) f1_tail<U8, PARTIAL, LOADNEXT, true>(zz, a, in, n0_next, lane, ob, limit);
    else f1_tail<U8, PARTIAL, LOADNEXT, false>(zz, a, in, n0_next, lane, ob, limit);
}

// FIR outputs [p_a, p_b) of the chunk -> FM angles out[p - s]; every wave takes a contiguous run of the nblk blocks
template <bool U8>
__global__ void __launch_bounds__(64 * F1_WAVES, 3) k_chain_fft1k(const DDChainParams P, const DDFft1kTabs T, int64_t p_a, int64_t p_b, int nblk, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2f* const X = reinterpret_cast<v2f*>(smem + wave * F1_WAVE_BYTES);
    const int gw = blockIdx.x * F1_WAVES + wave;
    const int q_begin = (int)(((int64_t)nblk * gw) / nwaves), q_end = (int)(((int64_t)nblk * (gw + 1)) / nwaves);
    v2f* const HP = reinterpret_cast<v2f*>(smem + F1_HP_OFF);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float2 h = T.hp[threadIdx.x + 256 * i];
        HP[threadIdx.x + 256 * i] = (v2f){h.x, h.y};
    }
    __syncthreads();                           // the only barrier of the kernel, before any wave may leave
    if (q_begin >= q_end) return;
    const v2f* const hp = HP + lane;
    const int tcol = 2 * (lane & 31) + (lane >> 5);           // the column this lane transforms in passes 1 and 6
    v2f tw1[16], tw2[16];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const float2 u = T.tw1[tcol * 16 + k], w = T.tw2[(lane & 3) * 16 + k];
        tw1[k] = (v2f){u.x, u.y};
        tw2[k] = (v2f){w.x, w.y};
    }
    const v2f crot = {T.crot.x, T.crot.y};
    float* const outp = reinterpret_cast<float*>(P.out);
    v2f a[16], keep[4];
    // the run's last block may be partial; it is also the only one whose 1024 samples may reach past the chunk's end (behind
    // the run's last output), so it is loaded on its own, with clamped indices, instead of being prefetched
    const bool last_partial = (q_end == nblk) && (p_a + (int64_t)F1_ADV * nblk > p_b);
    const int q_full_end = last_partial ? q_end - 1 : q_end;
    if (q_begin < q_full_end) {
        f1_load_pairs<U8>(P.in, p_a + (int64_t)F1_ADV * q_begin - 256, lane, a, 0, 8);
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): see k_chain_fft
        f1_swap(a[0], a[1]);
        f1_swap(a[2], a[3]);
    }
    for (int q = q_begin; q < q_full_end; ++q) {
        const int64_t p0 = p_a + (int64_t)F1_ADV * q;
        const int64_t n0_next = (q + 1 < q_full_end) ? p0 + F1_ADV - 256 : p0 - 256;     // (the last one re-reads itself: no branch in the block)
        f1_block<U8, false, true>(a, keep, X, tw1, tw2, hp, crot, lane, P.in, n0_next, outp + (p0 - P.s), 0);
    }
    if (last_partial) {
        const int64_t p0 = p_a + (int64_t)F1_ADV * (q_end - 1);
        const int64_t room = (P.L - 2 - (p0 - 256)) >> 1;                                 // last whole pair of the chunk, relative to the block
        f1_load_pairs<U8, true>(P.in, p0 - 256, lane, a, 0, 8, (unsigned)(room < F1_N / 2 - 1 ? room : F1_N / 2 - 1));
        f1_swap(a[0], a[1]);
        f1_swap(a[2], a[3]);
        f1_block<U8, true, false>(a, keep, X, tw1, tw2, hp, crot, lane, P.in, 0, outp + (p0 - P.s), (int)(p_b - p0));
    }
}

// ============================================================================ host side
struct DDFftState {
    int K;
    std::vector<double> taps;
    float2* tw1;
    float2* tw2;
    float2* hp;
    float2* tw1k;       // tables of k_chain_fft1k
    float2* tw2k;
    float2* hp1;
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

int dd_fft_supported(int K, int M, int flags) {
    return (M == 1 && K >= 1 && K <= 256 && (flags & DD_CHAIN_FM)) ? 1 : 0;
}

int dd_fft_create(void** st, const double* taps, int K) {
    if (K < 1 || K > 256) return DD_ERR_UNSUPPORTED;
    DDFftState* s = new DDFftState();
    s->K = K;
    s->taps.assign(taps, taps + K);
    s->tw1 = s->tw2 = s->hp = s->tw1k = s->tw2k = s->hp1 = nullptr;
    s->cyc = 0;
    s->have_h = 0;
    s->nco = 0;
    std::vector<float2> t1(256 * 16), t2(16 * 16);
    for (int t = 0; t < 256; ++t)
        for (int k = 0; k < 16; ++k) {
            const double ang = -2.0 * M_PI * (double)((t * k) % FF_N) / (double)FF_N;
            t1[t * 16 + k] = make_float2((float)cos(ang), (float)sin(ang));
        }
    for (int n0 = 0; n0 < 16; ++n0)
        for (int k = 0; k < 16; ++k) {
            const double ang = -2.0 * M_PI * (double)((n0 * k) % 256) / 256.0;
            t2[n0 * 16 + k] = make_float2((float)cos(ang), (float)sin(ang));
        }
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
    hipError_t e = hipMalloc((void**)&s->tw1, t1.size() * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc((void**)&s->tw1k, u1.size() * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc((void**)&s->tw2k, u2.size() * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc((void**)&s->hp1, 64 * 16 * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(s->tw1k, u1.data(), u1.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->tw2k, u2.data(), u2.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&s->tw2, t2.size() * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc((void**)&s->hp, 256 * 16 * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(s->tw1, t1.data(), t1.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->tw2, t2.data(), t2.size() * sizeof(float2), hipMemcpyHostToDevice);
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
    if (s->hp) (void)hipFree(s->hp);
    if (s->tw1k) (void)hipFree(s->tw1k);
    if (s->tw2k) (void)hipFree(s->tw2k);
    if (s->hp1) (void)hipFree(s->hp1);
    delete s;
}

// tap spectrum for the NCO frequency of this launch (frac(f/fs) = cyc / 2^64), permuted for pass 3
static int fft_prepare(DDFftState* s, bool nco, uint64_t cyc, hipStream_t stream) {
    if (s->have_h && s->cyc == cyc && s->nco == (int)nco) return DD_OK;
    const long double frac = nco ? (long double)cyc / 18446744073709551616.0L : 0.0L;      // [0, 1)
    std::vector<std::complex<double>> g(FF_N, std::complex<double>(0.0, 0.0));
    for (int k = 0; k < s->K; ++k) {
        // e^{+j 2 pi frac k}, argument reduced exactly before the call
        long double ph = frac * (long double)k;
        ph -= floorl(ph);
        const long double a = 2.0L * 3.14159265358979323846264338327950288L * ph;
        g[k] = std::complex<double>((double)(s->taps[k] * cosl(a)), (double)(s->taps[k] * sinl(a)));
    }
    std::vector<std::complex<double>> g1(g.begin(), g.begin() + F1_N);     // the same taps, zero padded to 1024
    fft_pow2(g);
    fft_pow2(g1);
    std::vector<float2> hp1(64 * 16);
    for (int k0 = 0; k0 < 16; ++k0)
        for (int j = 0; j < 4; ++j)
            for (int c = 0; c < 4; ++c)
                for (int k2 = 0; k2 < 4; ++k2) {
                    const std::complex<double> h = g1[k0 + 16 * (4 * c + j) + 256 * k2] / (double)F1_N;
                    hp1[(4 * c + k2) * 64 + 4 * k0 + j] = make_float2((float)h.real(), (float)h.imag());
                }
    std::vector<float2> hp(256 * 16);
    for (int k0 = 0; k0 < 16; ++k0)
        for (int k1 = 0; k1 < 16; ++k1)
            for (int k2 = 0; k2 < 16; ++k2) {
                const std::complex<double> h = g[k0 + 16 * k1 + 256 * k2] / (double)FF_N;
                hp[(16 * k0 + k1) * 16 + k2] = make_float2((float)h.real(), (float)h.imag());
            }
    // the table may still be read by an earlier launch on this stream: stream-ordered copy from a staging vector that
    // lives until the copy has been consumed
    DD_HIP_CHECK(hipStreamSynchronize(stream));
    DD_HIP_CHECK(hipMemcpy(s->hp, hp.data(), hp.size() * sizeof(float2), hipMemcpyHostToDevice));
    DD_HIP_CHECK(hipMemcpy(s->hp1, hp1.data(), hp1.size() * sizeof(float2), hipMemcpyHostToDevice));
    s->cyc = cyc;
    s->nco = (int)nco;
    s->have_h = 1;
    return DD_OK;
}

int dd_fft_launch(void* stv, const DDChainParams& P, int64_t p_a, int64_t p_b, hipStream_t stream) {
    DDFftState* s = reinterpret_cast<DDFftState*>(stv);
    if (p_b <= p_a) return DD_OK;
    const bool nco = (P.flags & DD_CHAIN_NCO) != 0;
    int rc = fft_prepare(s, nco, P.cyc, stream);
    if (rc != DD_OK) return rc;
    static DDOncePerDevice attr_set;
    if (attr_set.need()) {
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_fft<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_BYTES));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_fft<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_BYTES));
        attr_set.mark();
    }
    DDFftTabs T;
    T.tw1 = s->tw1;
    T.tw2 = s->tw2;
    T.hp = s->hp;
    T.dbg = nullptr;
    T.dbg_stage = 0;
    {
        const long double frac = nco ? (long double)P.cyc / 18446744073709551616.0L : 0.0L;
        const long double a = 2.0L * 3.14159265358979323846264338327950288L * frac;
        T.crot = make_float2((float)cosl(a), (float)-sinl(a));
    }
    const char* kern_env = getenv("DD_MFMA_KERNEL");
    if (kern_env && strcmp(kern_env, "fft1k") == 0) {
        static DDOncePerDevice attr1;
        if (attr1.need()) {
            DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_fft1k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS_BYTES));
            DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_fft1k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS_BYTES));
            attr1.mark();
        }
        DDFft1kTabs T1;
        T1.tw1 = s->tw1k; T1.tw2 = s->tw2k; T1.hp = s->hp1; T1.crot = T.crot;
        { const char* e = getenv("DD_FFT_STAGGER"); T1.stagger = e ? atoi(e) : 0; }
        const int nb1 = (int)((p_b - p_a + F1_ADV - 1) / F1_ADV);
        static const char* wg_env1 = getenv("DD_FFT_WGS_PER_CU");
        const int per_cu1 = wg_env1 ? atoi(wg_env1) : 3;
        int grid1 = dd_cu_count() * (per_cu1 > 0 ? per_cu1 : 3);
        if (grid1 * F1_WAVES > nb1) grid1 = (nb1 + F1_WAVES - 1) / F1_WAVES;
        if (P.flags & DD_CHAIN_U8_INPUT) hipLaunchKernelGGL((k_chain_fft1k<true>), dim3(grid1), dim3(64 * F1_WAVES), F1_LDS_BYTES, stream, P, T1, p_a, p_b, nb1, grid1 * F1_WAVES);
        else hipLaunchKernelGGL((k_chain_fft1k<false>), dim3(grid1), dim3(64 * F1_WAVES), F1_LDS_BYTES, stream, P, T1, p_a, p_b, nb1, grid1 * F1_WAVES);
        DD_LAUNCH_CHECK();
        return DD_OK;
    }
    const int64_t nblk64 = (p_b - p_a + FF_ADV - 1) / FF_ADV;
    const int nblk = (int)nblk64;
    static const char* wg_env = getenv("DD_FFT_WGS_PER_CU");
    const int per_cu = wg_env ? atoi(wg_env) : 2;
    int grid = dd_cu_count() * (per_cu > 0 ? per_cu : 2);
    if (grid > nblk) grid = nblk;
    static const char* st_env = getenv("DD_FFT_STAMPS");
    static int st_count = 0;
    if (st_env && !(P.flags & DD_CHAIN_U8_INPUT) && st_count++ == atoi(st_env)) {
        // diagnostic: per-wave cycle sums of the 19 segments of a block, averaged over waves and blocks
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_fft<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_BYTES));
        unsigned long long* buf = nullptr;
        const size_t nw = (size_t)grid * 4 * 24;
        DD_HIP_CHECK(hipMalloc((void**)&buf, nw * 8));
        T.dbg = reinterpret_cast<float2*>(buf);
        hipLaunchKernelGGL((k_chain_fft<false, 2>), dim3(grid), dim3(FF_THREADS), FF_LDS_BYTES, stream, P, T, p_a, p_b, nblk);
        std::vector<unsigned long long> hb(nw);
        DD_HIP_CHECK(hipMemcpy(hb.data(), buf, nw * 8, hipMemcpyDeviceToHost));
        (void)hipFree(buf);
        const char* nm[22] = {"load wait + pass 1", "X1 writes", "barrier", "X1 reads", "pass 2", "X2 writes", "barrier", "X2 reads", "pass 3 + H + inverse 3",
                              "X2' writes", "barrier", "X2' reads", "inverse 2", "X1' writes", "barrier", "X1' reads", "inverse 1 + edge", "barrier", "edge reads, z = w conj(w') c", "angles", "next block into registers", "stores"};
        double tot = 0;
        for (int i = 0; i < 22; ++i) {
            double sum = 0, nb = 0;
            for (size_t w = 0; w < (size_t)grid * 4; ++w) { sum += (double)hb[w * 24 + i]; nb += (double)hb[w * 24 + 23]; }
            fprintf(stderr, "[fft stamps] %-26s %8.0f cycles per block\n", nm[i], sum / nb);
            tot += sum / nb;
        }
        fprintf(stderr, "[fft stamps] total %.0f cycles per block and wave, %d workgroups, %d blocks\n", tot, grid, nblk);
        return DD_OK;
    }
    if (P.flags & DD_CHAIN_U8_INPUT) hipLaunchKernelGGL((k_chain_fft<true>), dim3(grid), dim3(FF_THREADS), FF_LDS_BYTES, stream, P, T, p_a, p_b, nblk);
    else hipLaunchKernelGGL((k_chain_fft<false>), dim3(grid), dim3(FF_THREADS), FF_LDS_BYTES, stream, P, T, p_a, p_b, nblk);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// diagnostic: one block (4096 complex64 samples at `in`, device) through the kernel up to `stage`; out receives the
// 256 x 16 register image after that stage (tests/tools compare it with a NumPy model of the same data flow)
extern "C" int dd_debug_fft_block(const float* in_c64, const double* taps, int ntaps, uint64_t cycles_q64, int nco, int stage, float* out_c64, void* stream) {
    void* st = nullptr;
    int rc = dd_fft_create(&st, taps, ntaps);
    if (rc != DD_OK) return rc;
    DDFftState* s = reinterpret_cast<DDFftState*>(st);
    hipStream_t hs = dd_stream(stream);
    rc = fft_prepare(s, nco != 0, cycles_q64, hs);
    if (rc != DD_OK) { dd_fft_destroy(st); return rc; }
    DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_fft<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_BYTES));
    DDChainParams P;
    memset(&P, 0, sizeof(P));
    float* scratch = nullptr;
    DD_HIP_CHECK(hipMalloc((void**)&scratch, FF_N * sizeof(float)));
    P.in = in_c64;
    P.out = scratch;
    P.L = FF_N;
    P.Ld = FF_N;
    P.K = ntaps;
    P.M = 1;
    P.flags = DD_CHAIN_FM | (nco ? DD_CHAIN_NCO : 0);
    P.cyc = cycles_q64;
    DDFftTabs T;
    T.tw1 = s->tw1; T.tw2 = s->tw2; T.hp = s->hp;
    T.crot = make_float2(1.f, 0.f);
    T.dbg = reinterpret_cast<float2*>(out_c64);
    T.dbg_stage = stage;
    hipLaunchKernelGGL((k_chain_fft<false, 1>), dim3(1), dim3(FF_THREADS), FF_LDS_BYTES, hs, P, T, (int64_t)256, (int64_t)(256 + FF_ADV), 1);
    hipError_t e = hipStreamSynchronize(hs);
    (void)hipFree(scratch);
    dd_fft_destroy(st);
    if (e != hipSuccess) { dd_set_error("dd_debug_fft_block: %s", hipGetErrorString(e)); return DD_ERR_HIP; }
    return DD_OK;
}
