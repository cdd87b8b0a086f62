// Fused M == 1 hot path for cosine-series windows of 255 taps (filters.hamming, filters.py:199; any b[k] = a0 + a1 cos(2 pi k / 254)):
//
//     offsetFreq (comm.py:63-78) -> FIR as three RUNNING SUMS (comm.py:80-92, filters.py:64-70) -> demod_fm (demod_fm.py:40-49)
//
// Why: k_chain_fft1k runs at the board's power cap -- 0.22-0.23 J of switching energy per 2^26-sample
// launch, half of it for the two 1024-point transforms per 768 outputs.  For a cosine-series window the convolution needs no
// transform: with xt[n] = x[n] e^{-j w n} (the NCO, applied to the samples) and phi = 2 pi / 254,
//     R[n] = sum_{k<255} xt[n-k]      C[n] = sum_{k<255} cos(phi k) xt[n-k]      S[n] = sum_{k<255} sin(phi k) xt[n-k]
//     (C, S)[n] = Rot_phi((C, S)[n-1] - (xt[n-255], 0)) + (xt[n], 0)            R[n] = R[n-1] + xt[n] - xt[n-255]
//     y[n] = a0 R[n] + a1 C[n]
// (tools/ubench/cosfir_arith.hip measured this arithmetic alone before the kernel existed: 0.093 ms and 0.082 J per 2^26 samples against
// 0.123 ms and 0.129 J for the FFT kernel's; profiles/r05_cosfir_ubench.txt).  The kernel: 0.141-0.148 ms, 0.154-0.162 J per launch.
//
// Layout.  A wave walks runs of ROWS of 1024 samples; lane L owns samples 16 L .. 16 L + 15 of the row, so the recurrence runs
// serially inside a lane and the 64 lanes are tied together by a two-pass scan:
//   pass A   lane totals T_L = sum_i A^{15-i} (xt_i, 0): plain sums with constant weights (phi is fixed by K = 255)
//   scan     pre_l = sum_{l' <= l} A^{16 (l - l')} T_l' inside each DPP row of 16 lanes (Kogge-Stone through DPP moves, weights = rotations
//            by 16 phi 2^k)
//   window   V_L = pre_l + A^{16 (l + 1)} tot' - A^256 pre'_l - A^255 (xt[16 (L - 15)], 0): the 255-sample window that ends with lane L, from
//            the prefix 16 lanes back (one ds_bpermute per component; lanes L < 16: the previous row's, kept in a register) and its row's total
//   pass B   the recurrence from the true state V_{L-1} at the lane's first sample, subtracting xt[n-255] as it leaves: 8 packed
//            instructions per sample; y up to the positive factor a0, which the discriminator does not see
// Every quantity is rebuilt from at most the last two rows: no error is carried along the stream (the float32 model
// tools/sim/cosfir_sim.py: FIR error 4.7e-7 of max|y|, FM median 1.2e-8 rad against the float64 definition).
// xt[n-255] is lane L-16's sample i+1: the row's samples after the NCO go through LDS once (two row images per wave -- this row and
// the one before it, whose last 16 groups are what lanes 0..15 look back at -- of 64 groups of 16 samples, 144 bytes apart: conflict
// free for the 16-byte row-major stores and for the lanes' own 16-byte reads), which also turns the coalesced load layout (lane l
// holds samples 128 j + 2 l, +1) into the lane-contiguous one.  No barrier: a wave owns its LDS image.  Raw u8 input (source.py:117-118)
// arrives lane-contiguous already (32 bytes per lane and row).  The lane's 16 angles go back through the dead part of the row image and
// leave as four 1 KB row-major stores, issued by the NEXT row behind its loads.  Complex64 output (CX: the FIR output itself, commSignal.filter
// alone) takes the same road through the image of the row before: eight 1 KB stores.
//
// Edges and runs.  The row grid is laid by the alignment of `out` (row q covers samples [base + 1024 q, +1024), base in [s - 15, s], so
// that every lane's 16 angles are one 64-byte line).  Rows come in runs of 8 dealt to the waves in turn (one moving window over the
// stream; short chunks: one run per wave); a run starts from the last 256 samples of the row before it (c1_prime_light).  Rows that
// touch the stream start, the chunk end or the carried state (history after the NCO, last FIR output: DDChainParams) load sample by
// sample with exact phases and predicate their stores.  One launch per chunk.
#include "dd_chain_kernels.h"
#include "dd_cosfir.h"
#include "dd_cosfit.h"
#include "dd_atan.h"
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define C1_K 255
// rotations by phi = 2 pi / 254, by 256 phi (= 2 phi) and by 16 phi 2^k: fixed by K, so literals -- as kernel arguments they were fourteen of the
// wave-uniform values the compiler keeps in scalar registers (it holds more than the 102 there are and spills the rest through v_readlane)
#define C1_C1 9.996940573e-01f
#define C1_S1 2.473442728e-02f
#define C1_C2 9.987764162e-01f
#define C1_S2 4.945371992e-02f
#define C1_WC0 9.226918151e-01f
#define C1_WS0 3.855383436e-01f
#define C1_WC1 7.027203712e-01f
#define C1_WS1 7.114661481e-01f
#define C1_WC2 -1.236815966e-02f
#define C1_WS2 9.999235114e-01f
#define C1_WC3 -9.996940573e-01f
#define C1_WS3 -2.473442728e-02f
#define C1_ROW 1024
#define C1_GROUP_BYTES 144                     // 16 samples of 8 bytes + 16 bytes of padding
#define C1_BUF_BYTES (64 * C1_GROUP_BYTES)     // one row after the NCO: 9216
#define C1_WAVE_BYTES (2 * C1_BUF_BYTES)       // this row and the one before it (whose last 16 groups are the samples 255 back of lanes 0..15)
#define C1_WAVES 4
#define C1_LDS_BYTES (C1_WAVES * C1_WAVE_BYTES)

struct DDCos1kLane { float b15c, b15s; };   // rotation by 16 phi ((lane & 15) + 1)

// the kernel's only argument
struct DDCos1kArgs {
    const void* in;            // complex64 or interleaved u8
    void* out;                 // angles (float), or the FIR output itself (float2: CX)
    const float2* tail_in;     // 254 samples after the NCO that precede the chunk
    float2* tail_out;
    const float2* lasty_in;    // FIR output before the chunk's first sample
    float2* lasty_out;
    const float2* nco_tbl;     // 4096-entry phasor table
    const DDCos1kLane* lane_tab;
    uint64_t cyc;              // frac(f / fs) 2^64
    int64_t abs0;              // absolute index of in[0]
    int64_t L;
    int s;                     // 1: stream start, no angle for sample 0
    int base;                  // first sample of row 0 (chunk-relative)
    int nrows, nwaves;
    int run_rows;              // 0: wave w takes ONE run, rows [nrows w / nwaves, nrows (w + 1) / nwaves); R > 0: runs of R rows dealt to the
                               // waves in turn (run j = rows [jR, jR + R) goes to wave j mod nwaves: the device walks one moving window)
    float a0, a1;              // y = a0 R + a1 C
    float2 q1, q2, q4;         // e^{-j w 128}, e^{-j w 256}, e^{-j w 512}: row-major load layout (complex64 input)
    float2 e1, e2, e3, e4, e8; // e^{-j w i}: lane-contiguous layout (u8 input; e1 also pairs the complex64 samples)
};

__device__ __forceinline__ v2f c1_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f c1_fma(float a, v2f b, v2f c) { return __builtin_elementwise_fma((v2f){a, a}, b, c); }
// A complex product a * w as its two packed instructions with operand selects and sign modifiers spelt out (left to the compiler the
// swizzled, negated copy of an operand is built with two more instructions).  A packed-f32 result may not be read by the very next
// VALU instruction on gfx950: callers issue the first halves of a group of products, then the second halves.
__device__ __forceinline__ v2f c1_mul_lo(v2f a, v2f w) {         // (a.x w.x, a.y w.x)
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    return t;
}
__device__ __forceinline__ v2f c1_fma_hi(v2f a, v2f w, v2f t) {  // t + (-a.y w.y, a.x w.y): completes a * w
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ v2f c1_fma_hic(v2f a, v2f w, v2f t) { // t + (a.y w.y, -a.x w.y): completes a * conj(w)
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ v2f c1_cmul(v2f a, v2f w) { return c1_fma_hi(a, w, c1_mul_lo(a, w)); }
__device__ __forceinline__ v2f c1_v2(float2 a) { return (v2f){a.x, a.y}; }

template <int CTRL>
__device__ __forceinline__ float c1_dpp0(float v) {                         // lanes without a source lane read 0 (bound_ctrl)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// wave_shr:1; lane 0 keeps `first`
__device__ __forceinline__ float c1_shr1(float v, float first) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ v2f c1_shr1(v2f v, v2f first) { return (v2f){c1_shr1(v.x, first.x), c1_shr1(v.y, first.y)}; }
// lane 15 of the lane's own DPP row, in every lane of it (row_newbcast:15)
__device__ __forceinline__ float c1_rowlast(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x15F, 0xF, 0xF, false)); }
__device__ __forceinline__ v2f c1_rowlast(v2f v) { return (v2f){c1_rowlast(v.x), c1_rowlast(v.y)}; }
__device__ __forceinline__ float c1_lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ v2f c1_lane63(v2f v) { return (v2f){c1_lane63(v.x), c1_lane63(v.y)}; }
__device__ __forceinline__ float c1_bperm(int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }
__device__ __forceinline__ v2f c1_bperm(int addr, v2f v) { return (v2f){c1_bperm(addr, v.x), c1_bperm(addr, v.y)}; }

struct C1St { v2f C, S, R; };

#define C1_ROW_SHR(n) (0x110 + (n))
// r += (r of the lane n back in the DPP row): one instruction (the DPP operand of a VOP2 add); lanes without a source lane keep r
template <int CTRL>
__device__ __forceinline__ float c1_add_dpp(float r) {
    static_assert(CTRL == 0x111 || CTRL == 0x112 || CTRL == 0x114 || CTRL == 0x118, "dpp control");
    if (CTRL == 0x111) asm("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r));
    if (CTRL == 0x112) asm("v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r));
    if (CTRL == 0x114) asm("v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r));
    if (CTRL == 0x118) asm("v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r));
    return r;
}
// one Kogge-Stone step of the weighted prefix inside a DPP row of 16 lanes: t += Rot(t of the lane n back)
template <int CTRL>
__device__ __forceinline__ void c1_scan_step(C1St& t, float wc, float ws) {
    const v2f Cs = (v2f){c1_dpp0<CTRL>(t.C.x), c1_dpp0<CTRL>(t.C.y)};
    const v2f Ss = (v2f){c1_dpp0<CTRL>(t.S.x), c1_dpp0<CTRL>(t.S.y)};
    t.R = (v2f){c1_add_dpp<CTRL>(t.R.x), c1_add_dpp<CTRL>(t.R.y)};
    t.C = c1_fma(wc, Cs, c1_fma(-ws, Ss, t.C));
    t.S = c1_fma(ws, Cs, c1_fma(wc, Ss, t.S));
}

// what a wave carries from one row to the next
struct C1Carry {
    C1St W;          // the previous row's prefixes of lanes 48..63 (inside their DPP row) as lanes L < 16 need them (lane L holds lane 48 + L's)
    C1St V63;        // windowed state at the previous row's last sample (wave-uniform): enters lane 0
    v2f y63;         // the previous row's last FIR output (wave-uniform): y[n-1] of lane 0's first sample
    v2f prow;        // e^{-j w (abs0 + S)} of the NEXT row times the lane's own factor, looked up a row ahead (the table read is a
                     // dependent global load: taken at the top of a row it stalls the wave for a memory round trip per row)
    int64_t pend_S;  // >= INT64_MIN + 1: the row whose angles wait in the OTHER buffer's first 4 KB for their stores (C1_NO_PEND = none)
    int cur;         // byte offset of this row's LDS buffer in the wave's image (0 or C1_BUF_BYTES; the other one holds the row before)
};

#define C1_NO_PEND ((int64_t)-4611686018427387904LL)
// the angles of row S, left in the image at `img` by the row before: four 1 KB row-major stores
__device__ __forceinline__ void c1_flush_angles(const DDCos1kArgs& A, const char* img, int64_t S, int lane) {
    float* const o = reinterpret_cast<float*>(A.out) + (S - A.s) + 4 * lane;
    const int G = lane >> 2, sw = (G >> 1) & 3;          // (group 16 g + G: the 16 g part does not reach the swizzle bits)
    v4f v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const v4f*>(img + 64 * (16 * g + G) + 16 * ((lane & 3) ^ sw));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#ifdef C1_ABL_NO_STORE
        if (v[g].x + v[g].y + v[g].z + v[g].w == 1234.5f) o[256 * g] = v[g].x;
#else
#ifdef C1_NT_STORE
        __builtin_nontemporal_store(v[g], reinterpret_cast<v4f*>(o + 256 * g));
#else
        *reinterpret_cast<v4f*>(o + 256 * g) = v[g];            // (plain: 0.1514 ms against 0.1633 with the non-temporal hint, same call)
#endif
#endif
    }
}

// CX: the 1024 FIR outputs of row S, left in the image at `img` (the padded layout of the sample image) by the row before: eight 16-byte
// row-major reads per lane, then eight 1 KB stores
__device__ __forceinline__ void c1_cx_read(const char* img, int lane, v4f (&v)[8]) {
    const char* const rd = img + (lane >> 3) * C1_GROUP_BYTES + 16 * (lane & 7);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const v4f*>(rd + 8 * j * C1_GROUP_BYTES);
}
__device__ __forceinline__ void c1_cx_store(const DDCos1kArgs& A, int64_t S, int lane, const v4f (&v)[8]) {
    float2* const o = reinterpret_cast<float2*>(A.out) + S + 2 * lane;
#pragma unroll
    // (non-temporal: with 8 bytes written per 8 read the hint pays -- 0.195 against 0.210-0.216 ms in the same calls; the angles' stores, 4 per 8, lose by it)
#ifdef C1_CX_PLAIN_STORE
    for (int j = 0; j < 8; ++j) *reinterpret_cast<v4f*>(o + 128 * j) = v[j];
#else
    for (int j = 0; j < 8; ++j) __builtin_nontemporal_store(v[j], reinterpret_cast<v4f*>(o + 128 * j));
#endif
}

// Ablation switches for timing experiments (tools/mkvariant.sh N dd_cosfir -DC1_ABL_...; the outputs of such a build are wrong):
//   C1_ABL_NO_LOAD   no global loads          C1_ABL_NO_STORE  no global stores         C1_ABL_NO_PHASOR  no per-row phase table look-up
//   C1_ABL_NO_LDS    no LDS traffic           C1_ABL_NO_FM     no discriminator         C1_ABL_NO_SCAN    no scan / window stage
//   C1_ABL_MEMONLY   the memory side alone: loads, both LDS transpositions, stores; an "angle" is re + im of the sample
//   C1_CX_IMMEDIATE  (results stay right) complex64 output stored by the row that made it instead of the next one
//   C1_CX_PLAIN_STORE (results stay right) complex64 output stored without the non-temporal hint
#ifdef C1_TRACE
// tools/debug/cos_trace.py: cycles per phase of a row (s_memtime stamps; every stamp drains the wave's LDS / scalar counter), summed
// per wave over its interior rows
#define C1_NPH 8
__device__ unsigned long long g_c1_trace[4096 * (C1_NPH + 2)];
#define C1_T(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned t_ = (unsigned)__builtin_readcyclecounter(); tr[i] += t_ - tprev; tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define C1_T(i) do { } while (0)
#endif

template <bool U8>
__device__ __forceinline__ void c1_issue_loads(const DDCos1kArgs& A, int64_t S, int lane, v4f (&xin)[8]) {
#ifdef C1_ABL_NO_LOAD
    return;
#endif
    if (U8) {
        const v4f* p = reinterpret_cast<const v4f*>(reinterpret_cast<const unsigned char*>(A.in) + 2 * (S + 16 * lane));
        xin[0] = __builtin_nontemporal_load(p);
        xin[1] = __builtin_nontemporal_load(p + 1);
    } else {
        const v4f* p = reinterpret_cast<const v4f*>(reinterpret_cast<const float2*>(A.in) + S + 2 * lane);
#pragma unroll
#ifdef C1_PLAIN_LOAD
        for (int j = 0; j < 8; ++j) xin[j] = p[64 * j];
#else
        for (int j = 0; j < 8; ++j) xin[j] = __builtin_nontemporal_load(p + 64 * j);
#endif
    }
}

template <bool NCO>
__device__ __forceinline__ v2f c1_row_phasor(const DDCos1kArgs& A, int64_t S, v2f ql) {
    if (!NCO) return (v2f){1.f, 0.f};
#ifdef C1_ABL_NO_PHASOR
    return c1_cmul((v2f){0.6f, 0.8f}, ql);
#else
    // (wave-uniform: the table entry comes through the SCALAR cache -- a vector load here would sit in vmcnt behind the next row's
    //  eight sample loads, and waiting for it would wait for all of them: a memory round trip exposed per row)
    const uint64_t phase64 = (uint64_t)(A.abs0 + S) * A.cyc;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(phase64 >> (64 - DD_NCO_TBITS - 32)));
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) float2* c1_const_f2;
    const uint32_t k = __builtin_amdgcn_readfirstlane((uint32_t)(phase64 >> (64 - DD_NCO_TBITS)));
    const float2 Tk = ((c1_const_f2)A.nco_tbl)[k];
#else
    const float2 Tk = make_float2(1.f, 0.f);
#endif
    const float theta = (float)lo * (6.283185307179586f * 5.684341886080802e-14f);   // 2 pi 2^-44 (dd_phasor's residual)
    const float t2 = theta * theta;
    const float cc = fmaf(-0.5f, t2, 1.0f), ss = theta * fmaf(-0.16666667f, t2, 1.0f);
    const v2f pr = (v2f){fmaf(Tk.x, cc, Tk.y * ss), fmaf(Tk.y, cc, -Tk.x * ss)};
    return c1_cmul(pr, ql);
#endif
}

// One row.  EDGE: sample-by-sample loads (history, chunk end), predicated stores, the carried FIR output.  emit: store the row's outputs.
// CX: the output is the FIR output itself (commSignal.filter alone, comm.py:80-92) instead of the discriminator's angles.
template <bool U8, bool NCO, bool EDGE, bool CX>
__device__ __forceinline__ void c1_row(const DDCos1kArgs& A, const DDCos1kLane& lt, const int lane, char* const lds,
                                       const int64_t S, const bool emit, const bool prefetch_next,
                                       v4f (&xin)[8], v4f (&xnext)[8], const v2f ql, C1Carry& cr
#ifdef C1_TRACE
                                       , unsigned* tr = nullptr
#endif
                                       ) {
#ifdef C1_TRACE
    unsigned tprev = (unsigned)__builtin_readcyclecounter();
    unsigned trdummy[C1_NPH];
    if (!tr) tr = trdummy;
#endif
    const float c = C1_C1, s = C1_S1;
    char* const cur = lds + cr.cur;
    char* const own = cur + lane * C1_GROUP_BYTES;                              // this lane's 16 samples of the row
    // CX: the outputs of the row before wait in THIS row's buffer (dead when they were put there); they are read before the samples
    // overwrite them -- a wave's LDS accesses keep their order -- and stored behind this row's prefetch
    v4f pv[8];
    const int64_t cx_pend = CX ? cr.pend_S : C1_NO_PEND;
    if (CX && cx_pend != C1_NO_PEND) { c1_cx_read(cur, lane, pv); cr.pend_S = C1_NO_PEND; }
    // lane - 16's group: this row's for lanes 16.., the previous row's lanes 48.. for lanes 0..15
    const char* const old = lane < 16 ? lds + (C1_BUF_BYTES - cr.cur) + (48 + lane) * C1_GROUP_BYTES : cur + (lane - 16) * C1_GROUP_BYTES;
    v2f xt[16];
    if (EDGE) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t n = S + 16 * lane + i;
            v2f v = (v2f){0.f, 0.f};
            if (n < 0) {
                const int64_t ti = n + (C1_K - 1);
                if (ti >= 0) v = c1_v2(A.tail_in[ti]);                           // (after the NCO already)
            } else if (n < A.L) {
                float2 x;
                if (U8) {
                    const uchar2 u = reinterpret_cast<const uchar2*>(A.in)[n];
                    x = make_float2((float)u.x - 127.5f, (float)u.y - 127.5f);
                } else {
                    x = reinterpret_cast<const float2*>(A.in)[n];
                }
                if (NCO) x = dd_cmul(x, dd_phasor((uint64_t)(A.abs0 + n) * A.cyc, A.nco_tbl));
                v = c1_v2(x);
            }
            xt[i] = v;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) *reinterpret_cast<v4f*>(own + 16 * t) = (v4f){xt[2 * t].x, xt[2 * t].y, xt[2 * t + 1].x, xt[2 * t + 1].y};
    } else {
        // (xin: this row's samples, requested by the caller or by the row before; cr.prow: this row's phase factor)
        const v2f prow = cr.prow;
        if (U8) {
            // 32 bytes = the lane's own 16 samples: (x - 127.5) e^{-j w (S + 16 L + i)}
            unsigned u[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) { u[t] = __float_as_uint(xin[0][t]); u[4 + t] = __float_as_uint(xin[1][t]); }
            v2f p4[4];
            if (NCO) {
                p4[0] = prow;
                p4[1] = c1_cmul(prow, c1_v2(A.e4));
                p4[2] = c1_cmul(prow, c1_v2(A.e8));
                p4[3] = c1_cmul(p4[2], c1_v2(A.e4));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned word = u[i >> 1];
                const float re = (float)((word >> (16 * (i & 1))) & 0xFF) - 127.5f;
                const float im = (float)((word >> (16 * (i & 1) + 8)) & 0xFF) - 127.5f;
                v2f x = (v2f){re, im};
                if (NCO) {
                    const v2f pb = p4[i >> 2];
                    const v2f pi = (i & 3) == 0 ? pb : c1_cmul(pb, c1_v2((i & 3) == 1 ? A.e1 : ((i & 3) == 2 ? A.e2 : A.e3)));
                    x = c1_cmul(x, pi);
                }
                xt[i] = x;
            }
#ifndef C1_ABL_NO_LDS
#pragma unroll
            for (int t = 0; t < 8; ++t) *reinterpret_cast<v4f*>(own + 16 * t) = (v4f){xt[2 * t].x, xt[2 * t].y, xt[2 * t + 1].x, xt[2 * t + 1].y};
#endif
        } else {
            // row-major registers: xin[j] = samples 128 j + 2 l, + 1  ->  NCO  ->  LDS, group 8 j + (l >> 3), slot l & 7
            char* const wr = cur + (lane >> 3) * C1_GROUP_BYTES + 16 * (lane & 7);
            v2f pj[8];
            if (NCO) {
                pj[0] = prow;
                pj[1] = c1_cmul(prow, c1_v2(A.q1));
                pj[2] = c1_cmul(prow, c1_v2(A.q2));
                pj[3] = c1_cmul(pj[2], c1_v2(A.q1));
                pj[4] = c1_cmul(prow, c1_v2(A.q4));
                pj[5] = c1_cmul(pj[4], c1_v2(A.q1));
                pj[6] = c1_cmul(pj[4], c1_v2(A.q2));
                pj[7] = c1_cmul(pj[6], c1_v2(A.q1));
            }
            v2f pj1[8], t0[8], t1[8];
            if (NCO) {
                const v2f e1 = c1_v2(A.e1);
#pragma unroll
                for (int j = 0; j < 8; ++j) t0[j] = c1_mul_lo(pj[j], e1);
#pragma unroll
                for (int j = 0; j < 8; ++j) pj1[j] = c1_fma_hi(pj[j], e1, t0[j]);
#pragma unroll
                for (int j = 0; j < 8; ++j) { t0[j] = c1_mul_lo((v2f){xin[j].x, xin[j].y}, pj[j]); t1[j] = c1_mul_lo((v2f){xin[j].z, xin[j].w}, pj1[j]); }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v2f x0 = (v2f){xin[j].x, xin[j].y}, x1 = (v2f){xin[j].z, xin[j].w};
                if (NCO) {
                    x0 = c1_fma_hi(x0, pj[j], t0[j]);
                    x1 = c1_fma_hi(x1, pj1[j], t1[j]);
                }
#ifdef C1_ABL_NO_LDS
                xt[2 * j] = x0; xt[2 * j + 1] = x1;
#else
                *reinterpret_cast<v4f*>(wr + 8 * j * C1_GROUP_BYTES) = (v4f){x0.x, x0.y, x1.x, x1.y};
#endif
            }
        }
    }
    C1_T(0);
    // the next row's samples fly during this row's arithmetic (requested ahead of every store of this row: vmcnt retires in order),
    // into the OTHER register set: the caller alternates the two, so that nothing is copied -- or waited for -- at the loop's back edge
    if (prefetch_next) { c1_issue_loads<U8>(A, S + C1_ROW, lane, xnext); cr.prow = c1_row_phasor<NCO>(A, S + C1_ROW, ql); }
    // the angles of the row before leave now: their LDS round trip and their stores overlap this row's arithmetic
    if (CX) { if (cx_pend != C1_NO_PEND) c1_cx_store(A, cx_pend, lane, pv); }
    else if (cr.pend_S != C1_NO_PEND) { c1_flush_angles(A, lds + (C1_BUF_BYTES - cr.cur), cr.pend_S, lane); cr.pend_S = C1_NO_PEND; }
#ifndef C1_ABL_NO_LDS
    if (!EDGE && !U8) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const v4f v = *reinterpret_cast<const v4f*>(own + 16 * t);
            xt[2 * t] = (v2f){v.x, v.y};
            xt[2 * t + 1] = (v2f){v.z, v.w};
        }
    }
#endif
#ifdef C1_ABL_MEMONLY
    if (!EDGE) {
        cr.cur = C1_BUF_BYTES - cr.cur;
        if (!emit) return;
        {
            const int sw = (lane >> 1) & 3;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                *reinterpret_cast<v4f*>(cur + 64 * lane + 16 * (t ^ sw)) = (v4f){xt[4 * t].x + xt[4 * t].y, xt[4 * t + 1].x + xt[4 * t + 1].y, xt[4 * t + 2].x + xt[4 * t + 2].y, xt[4 * t + 3].x + xt[4 * t + 3].y};
        }
        cr.pend_S = S;
        return;
    }
#endif
    // d[i] = xt[n - 255] = sample i + 1 of the group 16 lanes back
    v2f d[16];
#ifdef C1_ABL_NO_LDS
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = xt[(i + 5) & 15] * 0.5f;
#else
    d[0] = *reinterpret_cast<const v2f*>(old + 8);
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        const v4f v = *reinterpret_cast<const v4f*>(old + 16 * t);
        d[2 * t - 1] = (v2f){v.x, v.y};
        d[2 * t] = (v2f){v.z, v.w};
    }
    // (the first sample of the NEXT group: lane 15's is the first sample of this row, in this row's buffer)
    d[15] = *reinterpret_cast<const v2f*>(lane == 15 ? cur : old + C1_GROUP_BYTES);
#endif
    C1_T(1);
    // ---- pass A: lane totals of the un-windowed recurrence
    // T = sum_i A^{15-i} (xt[i], 0) = (sum cos((15-i) phi) xt[i], sum sin((15-i) phi) xt[i]): plain sums with constant weights (phi is
    // fixed by K = 255), two partial sums each so that consecutive multiply-adds do not wait for one another
    C1St t;
    {
        static constexpr float CW[16] = {1.000000000e+00f, 9.996940573e-01f, 9.987764162e-01f, 9.972476384e-01f, 9.951086592e-01f, 9.923607874e-01f,
                                         9.890057045e-01f, 9.850454633e-01f, 9.804824871e-01f, 9.753195679e-01f, 9.695598648e-01f, 9.632069021e-01f,
                                         9.562645670e-01f, 9.487371075e-01f, 9.406291296e-01f, 9.319455943e-01f};
        static constexpr float SW[16] = {0.000000000e+00f, 2.473442728e-02f, 4.945371992e-02f, 7.414275255e-02f, 9.878641831e-02f, 1.233696381e-01f,
                                         1.478773698e-01f, 1.722946174e-01f, 1.966064405e-01f, 2.207979630e-01f, 2.448543824e-01f, 2.687609789e-01f,
                                         2.925031245e-01f, 3.160662917e-01f, 3.394360625e-01f, 3.625981373e-01f};
        v2f C0 = xt[15], C1 = CW[1] * xt[14], S0 = SW[2] * xt[13], S1 = SW[1] * xt[14], R0 = xt[15] + xt[13], R1 = xt[14];
        C0 = c1_fma(CW[2], xt[13], C0);
#pragma unroll
        for (int i = 12; i >= 0; --i) {
            const int m = 15 - i;
            if (i & 1) { C0 = c1_fma(CW[m], xt[i], C0); S0 = c1_fma(SW[m], xt[i], S0); R0 += xt[i]; }
            else { C1 = c1_fma(CW[m], xt[i], C1); S1 = c1_fma(SW[m], xt[i], S1); R1 += xt[i]; }
        }
        t.C = C0 + C1; t.S = S0 + S1; t.R = R0 + R1;
    }
    C1_T(2);
    // ---- inclusive weighted prefix inside each DPP row of 16 lanes
#ifndef C1_ABL_NO_SCAN
    c1_scan_step<C1_ROW_SHR(1)>(t, C1_WC0, C1_WS0);
    c1_scan_step<C1_ROW_SHR(2)>(t, C1_WC1, C1_WS1);
    c1_scan_step<C1_ROW_SHR(4)>(t, C1_WC2, C1_WS2);
    c1_scan_step<C1_ROW_SHR(8)>(t, C1_WC3, C1_WS3);
#endif
    // ---- window.  t is the prefix INSIDE the lane's DPP row of 16 lanes (l = L & 15); the 256 samples that end with lane L are lanes
    // 16 rho .. L of its own row and lanes l + 1 .. 15 of the row before, whose sum is A^{16 (l + 1)} (tot' - A^{16 (15 - l)} pre'_l):
    //     V_L = pre_l + A^{16 (l + 1)} tot' - A^256 pre'_l - A^255 (xt[16 (L - 15)], 0)
    // pre' = the prefix 16 lanes back (lanes L < 16: the previous wave row's lanes 48.., carried), tot' = its lane 15 (row_newbcast).
    // No sum here spans more than 32 lanes, and the two wave-wide scan steps (row_bcast 15 / 31) are not needed at all.
    const int back16 = ((lane - 16) & 63) << 2;
    C1St Wn;                                                 // lane L: this row's pre of lane L - 16 (L >= 16), of lane 48 + L (L < 16: the next row's W)
    Wn.C = c1_bperm(back16, t.C);
    Wn.S = c1_bperm(back16, t.S);
    Wn.R = c1_bperm(back16, t.R);
    const bool low = lane < 16;
    C1St W;
    W.C = low ? cr.W.C : Wn.C;
    W.S = low ? cr.W.S : Wn.S;
    W.R = low ? cr.W.R : Wn.R;
    C1St tot;
    tot.C = c1_rowlast(W.C); tot.S = c1_rowlast(W.S); tot.R = c1_rowlast(W.R);
    const v2f e = d[15];
    C1St V;
    V.C = c1_fma(-C1_C2, W.C, c1_fma(C1_S2, W.S, c1_fma(-c, e, t.C)));
    V.S = c1_fma(-C1_S2, W.C, c1_fma(-C1_C2, W.S, c1_fma(-s, e, t.S)));
    V.R = t.R - W.R - e;
#ifndef C1_ABL_NO_SCAN
    V.C = c1_fma(lt.b15c, tot.C, c1_fma(-lt.b15s, tot.S, V.C));
    V.S = c1_fma(lt.b15s, tot.C, c1_fma(lt.b15c, tot.S, V.S));
    V.R += tot.R;
#endif
    // state at the lane's first sample = V of the lane before (lane 0: the previous row's last)
    C1St u;
    u.C = c1_shr1(V.C, cr.V63.C);
    u.S = c1_shr1(V.S, cr.V63.S);
    u.R = c1_shr1(V.R, cr.V63.R);
    cr.W = Wn;
    cr.V63.C = c1_lane63(V.C); cr.V63.S = c1_lane63(V.S); cr.V63.R = c1_lane63(V.R);
    C1_T(3);
    // ---- pass B: the windowed recurrence.  y / a0 = R + (a1 / a0) C (filters.py:199): the discriminator does not see the positive
    // factor a0 (the edge rows, which hand y to the next chunk, put it back), so an output is one multiply-add
    v2f y[16];
    const float kap = A.a1 / A.a0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const v2f D = u.C - d[i];                                  // the sample 255 back leaves the window (its weight there: cos(254 phi) = 1, sin = 0)
        const v2f Cn = c1_fma(c, D, c1_fma(-s, u.S, xt[i]));
        u.S = c1_fma(s, D, c * u.S);
        u.C = Cn;
        u.R += xt[i] - d[i];
        y[i] = c1_fma(kap, u.C, u.R);
    }
    C1_T(4);
    cr.cur = C1_BUF_BYTES - cr.cur;                          // this row's buffer is the next row's "row before"
    if (EDGE) {
        // the FIR output before the chunk's first sample is carried state (demod_fm.py:47-49); the chunk's last one becomes it
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t n = S + 16 * lane + i;
            if (!CX && A.s == 0 && n == -1) y[i] = c1_v2(*A.lasty_in);
            if (n == A.L - 1 && A.lasty_out) *A.lasty_out = make_float2(A.a0 * y[i].x, A.a0 * y[i].y);
        }
    }
    if (CX) {
        if (!emit) return;
        const float a0 = A.a0;
        if (EDGE) {
            float2* const o = reinterpret_cast<float2*>(A.out);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t n = S + 16 * lane + i;
                if (n >= 0 && n < A.L) o[n] = make_float2(a0 * y[i].x, a0 * y[i].y);
            }
            return;
        }
        // The lane holds 16 consecutive outputs (128 bytes); they cross the wave through the image of the row BEFORE (dead: this row has
        // taken its samples 255 back from it; it is the NEXT row's buffer) in the padded layout of the sample image, and leave as eight
        // 1 KB row-major stores issued by the next row (or by the kernel after the wave's last interior row).
        char* const img = lds + cr.cur;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            *reinterpret_cast<v4f*>(img + lane * C1_GROUP_BYTES + 16 * t) = (v4f){a0 * y[2 * t].x, a0 * y[2 * t].y, a0 * y[2 * t + 1].x, a0 * y[2 * t + 1].y};
#ifdef C1_CX_IMMEDIATE
        { v4f v[8]; c1_cx_read(img, lane, v); c1_cx_store(A, S, lane, v); }
#else
        cr.pend_S = S;
#endif
        return;
    }
    const v2f yl = c1_shr1(y[15], cr.y63);
    cr.y63 = c1_lane63(y[15]);
    if (!emit) return;
    // ---- demod_fm: angle(y[n] conj(y[n-1])).  Stage by stage over the lane's 16 samples, so that the products, the reciprocals and the
    // polynomials of different samples fill one another's latencies; the small-angle form where all 256 outputs of a group of four
    // samples per lane allow it -- decided for the four groups at once, the usual case being "all four".
    float ang[16];
    {
        v2f z[16], tz[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) tz[i] = c1_mul_lo(y[i], i ? y[i - 1] : yl);
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = c1_fma_hic(y[i], i ? y[i - 1] : yl, tz[i]);          // (re, im) of y[n] conj(y[n-1])
#ifdef C1_ABL_NO_FM
#pragma unroll
        for (int i = 0; i < 16; ++i) ang[i] = z[i].x + z[i].y;
#else
        // small-angle form (|angle| <= 22.5 degrees: re > 0 and |im / re| <= tan(pi / 8)) for the whole row when all 1024 outputs allow it
        // -- the quotients are formed first and the test is made on them; a product of exactly zero (digital silence) fails `re > 0` and
        // takes the full-range form, which returns np.angle(0) = 0
        float r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = __builtin_amdgcn_rcpf(z[i].x);
        v2f tt[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) tt[k] = (v2f){z[2 * k].y * r[2 * k], z[2 * k + 1].y * r[2 * k + 1]};
        float tmax = fmaxf(fabsf(tt[0].x), fabsf(tt[0].y)), remin = fminf(z[0].x, z[1].x);
#pragma unroll
        for (int k = 1; k < 8; ++k) { tmax = fmaxf(tmax, fmaxf(fabsf(tt[k].x), fabsf(tt[k].y))); remin = fminf(remin, fminf(z[2 * k].x, z[2 * k + 1].x)); }
        if (__builtin_amdgcn_ballot_w64(!(tmax <= 0.41421356f) || !(remin > 0.f)) == 0) {
            v2f zz[8], pp[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) zz[k] = tt[k] * tt[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) pp[k] = c1_fma(7.902598251e-02f, zz[k], (v2f){-1.382445378e-01f, -1.382445378e-01f});
#pragma unroll
            for (int k = 0; k < 8; ++k) pp[k] = c1_fma(pp[k], zz[k], (v2f){1.997187931e-01f, 1.997187931e-01f});
#pragma unroll
            for (int k = 0; k < 8; ++k) pp[k] = c1_fma(pp[k], zz[k], (v2f){-3.333275667e-01f, -3.333275667e-01f});
#pragma unroll
            for (int k = 0; k < 8; ++k) pp[k] = zz[k] * pp[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const v2f a2 = c1_fma(tt[k], pp[k], tt[k]); ang[2 * k] = a2.x; ang[2 * k + 1] = a2.y; }
        } else {
            // rare: per group of four samples (256 outputs), as k_chain_fft1k decides it
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float worst = -1.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) worst = fmaxf(worst, fmaf(-0.41421356f, z[4 * g + k].x, fabsf(z[4 * g + k].y)));
                if (__builtin_amdgcn_ballot_w64(worst >= 0.f) == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) ang[4 * g + k] = dd_atan_small(z[4 * g + k].y, z[4 * g + k].x);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) ang[4 * g + k] = dd_atan2_poly(z[4 * g + k].y, z[4 * g + k].x);
                }
            }
        }
#endif
    }
    C1_T(5);
    if (EDGE) {
        float* const o = reinterpret_cast<float*>(A.out) + (S - A.s) + 16 * lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t n = S + 16 * lane + i;
            if (n >= A.s && n < A.L) o[i] = ang[i];
        }
        return;
    }
    // The lane holds 16 consecutive angles (one 64-byte line); a store instruction made of 64 such lines costs four times a
    // coalesced one in the memory pipeline (measured: the kernel at 0.42 ms with them, 0.14 ms without any store).  So the angles
    // cross the wave through LDS -- the first 4 KB of the row's sample image, which nothing reads any more (the next row's lanes
    // 0..15 look at its LAST 16 groups) -- and leave as four 1 KB row-major stores.
    // Image: angle m of the row at dword 16 (m >> 4) + 4 (((m >> 2) & 3) ^ ((m >> 5) & 3)) + (m & 3): unpadded, the 16-byte chunks of
    // a lane's line swizzled by its lane pair, which keeps the eight lanes of a ds_write_b128 group and the sixteen of a
    // ds_read_b128 group on distinct banks.
    char* const img = cur;
    {
        const int sw = (lane >> 1) & 3;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            *reinterpret_cast<v4f*>(img + 64 * lane + 16 * (t ^ sw)) = (v4f){ang[4 * t], ang[4 * t + 1], ang[4 * t + 2], ang[4 * t + 3]};
    }
    cr.pend_S = S;                                           // (stored by the next row, or by the kernel after the wave's last interior row)
    C1_T(6);
}

// The row BEFORE a run, cheaply.  What a run needs from it depends on its last 256 samples only: their image in LDS (the samples 255 back of
// the run's first 16 lanes), the prefixes of lanes 48..63 (as lanes 0..15 will use them: only the sum of the lanes behind matters, so the prefix
// may start at lane 48), the row total, the windowed state and the FIR output at its last sample -- and that output is a0 R + a1 C of the
// state itself: no second pass, no discriminator.  A quarter of a row's loads and a third of its instructions; lanes 0..47 work on zeros.
// (Rows that touch the stream start or the carried state go through c1_row<EDGE> instead.)
template <bool U8>
__device__ __forceinline__ void c1_prime_issue(const DDCos1kArgs& A, int64_t S, int lane, v4f (&xp)[2]) {
    xp[0] = xp[1] = (v4f){0.f, 0.f, 0.f, 0.f};
#ifdef C1_ABL_NO_LOAD
    return;
#endif
    if (U8) {
        if (lane >= 48) {
            const v4f* p = reinterpret_cast<const v4f*>(reinterpret_cast<const unsigned char*>(A.in) + 2 * (S + 16 * lane));
            xp[0] = __builtin_nontemporal_load(p);
            xp[1] = __builtin_nontemporal_load(p + 1);
        }
    } else {
        const v4f* p = reinterpret_cast<const v4f*>(reinterpret_cast<const float2*>(A.in) + S + 2 * lane);
        xp[0] = __builtin_nontemporal_load(p + 64 * 6);
        xp[1] = __builtin_nontemporal_load(p + 64 * 7);
    }
}
template <bool U8, bool NCO>
__device__ __forceinline__ void c1_prime_light(const DDCos1kArgs& A, const DDCos1kLane& lt, const int lane, char* const lds, const int64_t S,
                                               const v4f (&xp)[2], const v2f ql, C1Carry& cr) {
    const float c = C1_C1, s = C1_S1;
    char* const cur = lds + cr.cur;
    char* const own = cur + lane * C1_GROUP_BYTES;
    const bool act = lane >= 48;
    const v2f prow = c1_row_phasor<NCO>(A, S, ql);
    v2f xt[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xt[i] = (v2f){0.f, 0.f};
    if (U8) {
        if (act) {
            unsigned u[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) { u[t] = __float_as_uint(xp[0][t]); u[4 + t] = __float_as_uint(xp[1][t]); }
            v2f p4[4];
            if (NCO) {
                p4[0] = prow;
                p4[1] = c1_cmul(prow, c1_v2(A.e4));
                p4[2] = c1_cmul(prow, c1_v2(A.e8));
                p4[3] = c1_cmul(p4[2], c1_v2(A.e4));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned word = u[i >> 1];
                v2f x = (v2f){(float)((word >> (16 * (i & 1))) & 0xFF) - 127.5f, (float)((word >> (16 * (i & 1) + 8)) & 0xFF) - 127.5f};
                if (NCO) {
                    const v2f pb = p4[i >> 2];
                    x = c1_cmul(x, (i & 3) == 0 ? pb : c1_cmul(pb, c1_v2((i & 3) == 1 ? A.e1 : ((i & 3) == 2 ? A.e2 : A.e3))));
                }
                xt[i] = x;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) *reinterpret_cast<v4f*>(own + 16 * t) = (v4f){xt[2 * t].x, xt[2 * t].y, xt[2 * t + 1].x, xt[2 * t + 1].y};
        }
    } else {
        char* const wr = cur + (lane >> 3) * C1_GROUP_BYTES + 16 * (lane & 7);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 6 + jj;
            v2f x0 = (v2f){xp[jj].x, xp[jj].y}, x1 = (v2f){xp[jj].z, xp[jj].w};
            if (NCO) {
                v2f pj = c1_cmul(c1_cmul(prow, c1_v2(A.q4)), c1_v2(A.q2));
                if (jj) pj = c1_cmul(pj, c1_v2(A.q1));
                x0 = c1_cmul(x0, pj);
                x1 = c1_cmul(x1, c1_cmul(pj, c1_v2(A.e1)));
            }
            *reinterpret_cast<v4f*>(wr + 8 * j * C1_GROUP_BYTES) = (v4f){x0.x, x0.y, x1.x, x1.y};
        }
        if (act) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const v4f v = *reinterpret_cast<const v4f*>(own + 16 * t);
                xt[2 * t] = (v2f){v.x, v.y};
                xt[2 * t + 1] = (v2f){v.z, v.w};
            }
        }
    }
    const v2f e = *reinterpret_cast<const v2f*>(cur + 48 * C1_GROUP_BYTES);          // the first of the 256 samples: it is NOT in the 255-sample window
    C1St t;
    {
        static constexpr float CW[16] = {1.000000000e+00f, 9.996940573e-01f, 9.987764162e-01f, 9.972476384e-01f, 9.951086592e-01f, 9.923607874e-01f,
                                         9.890057045e-01f, 9.850454633e-01f, 9.804824871e-01f, 9.753195679e-01f, 9.695598648e-01f, 9.632069021e-01f,
                                         9.562645670e-01f, 9.487371075e-01f, 9.406291296e-01f, 9.319455943e-01f};
        static constexpr float SW[16] = {0.000000000e+00f, 2.473442728e-02f, 4.945371992e-02f, 7.414275255e-02f, 9.878641831e-02f, 1.233696381e-01f,
                                         1.478773698e-01f, 1.722946174e-01f, 1.966064405e-01f, 2.207979630e-01f, 2.448543824e-01f, 2.687609789e-01f,
                                         2.925031245e-01f, 3.160662917e-01f, 3.394360625e-01f, 3.625981373e-01f};
        v2f C0 = xt[15], C1 = CW[1] * xt[14], S0 = SW[2] * xt[13], S1 = SW[1] * xt[14], R0 = xt[15] + xt[13], R1 = xt[14];
        C0 = c1_fma(CW[2], xt[13], C0);
#pragma unroll
        for (int i = 12; i >= 0; --i) {
            const int m = 15 - i;
            if (i & 1) { C0 = c1_fma(CW[m], xt[i], C0); S0 = c1_fma(SW[m], xt[i], S0); R0 += xt[i]; }
            else { C1 = c1_fma(CW[m], xt[i], C1); S1 = c1_fma(SW[m], xt[i], S1); R1 += xt[i]; }
        }
        t.C = C0 + C1; t.S = S0 + S1; t.R = R0 + R1;
    }
    // lanes 48..63 are one DPP row: four steps give their prefixes (the lanes before them hold zeros)
    c1_scan_step<C1_ROW_SHR(1)>(t, C1_WC0, C1_WS0);
    c1_scan_step<C1_ROW_SHR(2)>(t, C1_WC1, C1_WS1);
    c1_scan_step<C1_ROW_SHR(4)>(t, C1_WC2, C1_WS2);
    c1_scan_step<C1_ROW_SHR(8)>(t, C1_WC3, C1_WS3);
    const int back16 = ((lane - 16) & 63) << 2;
    cr.W.C = c1_bperm(back16, t.C);
    cr.W.S = c1_bperm(back16, t.S);
    cr.W.R = c1_bperm(back16, t.R);
    const C1St P63 = {c1_lane63(t.C), c1_lane63(t.S), c1_lane63(t.R)};
    cr.V63.C = c1_fma(-c, e, P63.C);
    cr.V63.S = c1_fma(-s, e, P63.S);
    cr.V63.R = P63.R - e;
    cr.y63 = c1_fma(A.a1 / A.a0, cr.V63.C, cr.V63.R);
    cr.cur = C1_BUF_BYTES - cr.cur;
}

// The whole chunk in one launch.  Row q covers samples [base + 1024 q, base + 1024 (q + 1)); wave gw takes rows
// [nrows gw / nwaves, nrows (gw + 1) / nwaves), after running the row before them without stores.
template <bool U8, bool NCO, bool CX>
__global__ void __launch_bounds__(64 * C1_WAVES, 2) k_chain_cos1k(const DDCos1kArgs A) {
    extern __shared__ __attribute__((aligned(16))) char c1_smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * C1_WAVES + wave;
    const int nrows = A.nrows, nwaves = A.nwaves;
    char* const lds = c1_smem + wave * C1_WAVE_BYTES;
    const DDCos1kLane lt = A.lane_tab[lane];
    v2f ql = (v2f){1.f, 0.f};
    if (NCO) ql = c1_v2(dd_phasor((uint64_t)(U8 ? 16 * lane : 2 * lane) * A.cyc, A.nco_tbl));
    const int R = A.run_rows;
    const int nruns = R > 0 ? (nrows + R - 1) / R : nwaves;
    for (int run = gw; run < nruns; run += nwaves) {
    const int q0 = R > 0 ? run * R : (int)(((int64_t)nrows * gw) / nwaves);
    const int q1 = R > 0 ? (q0 + R < nrows ? q0 + R : nrows) : (int)(((int64_t)nrows * (gw + 1)) / nwaves);
    if (q1 <= q0) continue;
    // the "row before" of the wave's first row starts as zeros where it is read (the hardware does not clear LDS; its old
    // contents may be NaN patterns): the last 16 groups of the second buffer, plus the word after it that lane 15 of a row never reads
    if (lane < 16) {
#pragma unroll
        for (int t = 0; t < 9; ++t) *reinterpret_cast<v4f*>(lds + C1_BUF_BYTES + (48 + lane) * C1_GROUP_BYTES + 16 * t) = (v4f){0.f, 0.f, 0.f, 0.f};
    }
    C1Carry cr;
    cr.W = cr.V63 = (C1St){(v2f){0.f, 0.f}, (v2f){0.f, 0.f}, (v2f){0.f, 0.f}};
    cr.y63 = (v2f){0.f, 0.f};
    cr.prow = (v2f){1.f, 0.f};
    cr.cur = 0;
    cr.pend_S = C1_NO_PEND;
#ifdef C1_TRACE
    unsigned tr[C1_NPH];
#pragma unroll
    for (int i = 0; i < C1_NPH; ++i) tr[i] = 0;
    const unsigned tloop = (unsigned)__builtin_readcyclecounter();
#endif
    // a row is an edge row when it holds samples before the first output, the carried state or the chunk's end.  Edge rows sit at
    // the two ends of a wave's range only: [q0 - 1, f0) edge, [f0, f1) interior, [f1, q1) edge.
    auto edge = [&](int q) { const int64_t lo = (int64_t)A.base + (int64_t)C1_ROW * q; return lo < A.s || lo + C1_ROW > A.L || q == nrows - 1; };
#ifdef C1_ABL_MEMONLY
    int f0 = edge(q0 - 1) ? q0 - 1 : q0;                      // (the memory-only build carries no state: no row before the run)
    const int qfirst = f0;
#else
    int f0 = q0 - 1;
    const int qfirst = q0 - 1;
#endif
    while (f0 < q1 && edge(f0)) ++f0;
    int f1 = f0;
    while (f1 < q1 && !edge(f1)) ++f1;
    v4f xa[8], xb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xa[j] = xb[j] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int q = qfirst; q < f0; ++q)
        c1_row<U8, NCO, true, CX>(A, lt, lane, lds, (int64_t)A.base + (int64_t)C1_ROW * q, q >= q0, false, xa, xb, ql, cr);
    if (f1 > f0) {
        const int64_t S0 = (int64_t)A.base + (int64_t)C1_ROW * f0;
#if !defined(C1_ABL_MEMONLY) && !defined(C1_FULL_PRIME)
        if (f0 == q0 - 1 && f1 > q0) {
            // the row before the run is an interior row: its last 256 samples are all the run needs (c1_prime_light); the run's first row
            // is requested right behind them
            v4f xp[2];
            c1_prime_issue<U8>(A, S0, lane, xp);
            ++f0;
            c1_issue_loads<U8>(A, S0 + C1_ROW, lane, xa);
            c1_prime_light<U8, NCO>(A, lt, lane, lds, S0, xp, ql, cr);
            cr.prow = c1_row_phasor<NCO>(A, S0 + C1_ROW, ql);
        } else
#endif
        {
        c1_issue_loads<U8>(A, S0, lane, xa);
        cr.prow = c1_row_phasor<NCO>(A, S0, ql);
        }
        // two rows per trip: the sample registers alternate (xa: even rows of the run, xb: odd ones)
        for (int q = f0; q < f1; q += 2) {
            const int64_t S = (int64_t)A.base + (int64_t)C1_ROW * q;
#ifdef C1_TRACE
            c1_row<U8, NCO, false, CX>(A, lt, lane, lds, S, q >= q0, q + 1 < f1, xa, xb, ql, cr, q >= q0 ? tr : nullptr);
            if (q + 1 < f1) c1_row<U8, NCO, false, CX>(A, lt, lane, lds, S + C1_ROW, true, q + 2 < f1, xb, xa, ql, cr, tr);
#else
            c1_row<U8, NCO, false, CX>(A, lt, lane, lds, S, q >= q0, q + 1 < f1, xa, xb, ql, cr);
            if (q + 1 < f1) c1_row<U8, NCO, false, CX>(A, lt, lane, lds, S + C1_ROW, true, q + 2 < f1, xb, xa, ql, cr);
#endif
        }
    }
    if (cr.pend_S != C1_NO_PEND) {
        if (CX) { v4f pv[8]; c1_cx_read(lds + cr.cur, lane, pv); c1_cx_store(A, cr.pend_S, lane, pv); }
        else c1_flush_angles(A, lds + (C1_BUF_BYTES - cr.cur), cr.pend_S, lane);
        cr.pend_S = C1_NO_PEND;
    }
    for (int q = f1; q < q1; ++q)
        c1_row<U8, NCO, true, CX>(A, lt, lane, lds, (int64_t)A.base + (int64_t)C1_ROW * q, q >= q0, false, xa, xb, ql, cr);
#ifdef C1_TRACE
    tr[7] = (unsigned)__builtin_readcyclecounter() - tloop;
    if (gw < 4096 && lane == 0) {
#pragma unroll
        for (int i = 0; i < C1_NPH; ++i) g_c1_trace[gw * (C1_NPH + 2) + i] = tr[i];
        g_c1_trace[gw * (C1_NPH + 2) + C1_NPH] = (unsigned long long)(q1 - q0);
    }
#endif
    if (q1 == nrows && A.tail_out) {
        // the new carried history: the chunk's last K-1 samples after the NCO (older ones from the old history)
        for (int i = lane; i < C1_K - 1; i += 64) {
            const int64_t n = A.L - (C1_K - 1) + i;
            float2 v;
            if (n < 0) {
                v = A.tail_in[n + (C1_K - 1)];
            } else {
                float2 x;
                if (U8) {
                    const uchar2 u = reinterpret_cast<const uchar2*>(A.in)[n];
                    x = make_float2((float)u.x - 127.5f, (float)u.y - 127.5f);
                } else {
                    x = reinterpret_cast<const float2*>(A.in)[n];
                }
                v = NCO ? dd_cmul(x, dd_phasor((uint64_t)(A.abs0 + n) * A.cyc, A.nco_tbl)) : x;
            }
            A.tail_out[i] = v;
        }
    }
    }   // runs
}

// ============================================================================ host side
struct DDCos1kState {
    double a0, a1;
    DDCos1kLane* lane_tab;
};

int dd_cos1k_supported(const double* taps, int K, int M, int flags) {
    (void)flags;                                   // FM angles or complex64 output, complex64 or raw u8 input
    if (M != 1 || K != C1_K) return 0;
    DDCosFit f;
    if (!dd_cos_fit_cached(taps, K, &f)) return 0;
    if (f.Q != 1) return 0;                        // a0 + a1 cos(2 pi k / (K-1)): Hamming, Hann and their relatives
    // the kernel works with y / a0 (the discriminator does not see a positive factor) and forms a1 / a0: a (near) pure cosine has no a0
    // to divide by -- such taps take the transform kernel
    return (isfinite(f.a[0]) && isfinite(f.a[1]) && fabs(f.a[0]) >= 1e-3 * fabs(f.a[1]) && f.a[0] != 0.0) ? 1 : 0;
}

int dd_cos1k_create(void** st, const double* taps, int K) {
    if (K != C1_K) return DD_ERR_UNSUPPORTED;
    DDCosFit f;
    if (!dd_cos_fit_cached(taps, K, &f) || f.Q != 1 || !dd_cos1k_supported(taps, K, 1, 0)) return DD_ERR_UNSUPPORTED;
    DDCos1kState* s = new DDCos1kState();
    s->a0 = f.a[0];
    s->a1 = f.a[1];
    s->lane_tab = nullptr;
    DDCos1kLane h[64];
    const long double phi16 = 16.0L * 2.0L * 3.14159265358979323846264338327950288L / (long double)(C1_K - 1);
    for (int l = 0; l < 64; ++l) {
        h[l].b15c = (float)cosl(phi16 * ((l & 15) + 1)); h[l].b15s = (float)sinl(phi16 * ((l & 15) + 1));
    }
    hipError_t e = hipMalloc((void**)&s->lane_tab, sizeof(h));
    if (e == hipSuccess) e = hipMemcpy(s->lane_tab, h, sizeof(h), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (s->lane_tab) (void)hipFree(s->lane_tab);
        delete s;
        dd_set_error("dd_cos1k_create: %s", hipGetErrorString(e));
        return DD_ERR_HIP;
    }
    *st = s;
    return DD_OK;
}

void dd_cos1k_destroy(void* stv) {
    DDCos1kState* s = reinterpret_cast<DDCos1kState*>(stv);
    if (!s) return;
    if (s->lane_tab) (void)hipFree(s->lane_tab);
    delete s;
}

#ifdef C1_TRACE
extern "C" int dd_debug_cos1k_trace(unsigned long long* out, int nwaves) {
    DD_HIP_CHECK(hipDeviceSynchronize());
    DD_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c1_trace), sizeof(unsigned long long) * (size_t)nwaves * (C1_NPH + 2)));
    return DD_OK;
}
#endif

// where the row grid sits and how many rows and waves a chunk takes (host arithmetic, also reachable without a GPU: dd_debug_cos1k_plan)
static void cos1k_plan(int64_t L, int s, int out_align_elems, int ncu, int wg_per_cu, int* base, int* nrows, int* grid, int* nwaves) {
    int b = s - out_align_elems;                                   // out[b - s] starts a 64-byte line
    while (b > L - 1) b -= 16;                                     // (a chunk of one sample: the last row must hold sample L - 1)
    const int64_t nr = (L - b + C1_ROW - 1) / C1_ROW;
    int g = ncu * (wg_per_cu > 0 ? wg_per_cu : 2);
    if ((int64_t)g * C1_WAVES > nr) g = (int)((nr + C1_WAVES - 1) / C1_WAVES);
    *base = b; *nrows = (int)nr; *grid = g; *nwaves = g * C1_WAVES;
}
extern "C" int dd_debug_cos1k_plan(int64_t L, int s, int out_align_elems, int ncu, int* out) {
    DD_REQUIRE(L >= 1 && (s == 0 || s == 1) && out_align_elems >= 0 && out_align_elems < 16 && ncu >= 1 && out, "arguments");
    cos1k_plan(L, s, out_align_elems, ncu, 2, &out[0], &out[1], &out[2], &out[3]);
    return DD_OK;
}

static const void* cos1k_kernel(bool u8, bool nco, bool cx) {
    static const void* const k[8] = {
        (const void*)k_chain_cos1k<false, false, false>, (const void*)k_chain_cos1k<true, false, false>,
        (const void*)k_chain_cos1k<false, true, false>,  (const void*)k_chain_cos1k<true, true, false>,
        (const void*)k_chain_cos1k<false, false, true>,  (const void*)k_chain_cos1k<true, false, true>,
        (const void*)k_chain_cos1k<false, true, true>,   (const void*)k_chain_cos1k<true, true, true>};
    return k[(u8 ? 1 : 0) | (nco ? 2 : 0) | (cx ? 4 : 0)];
}

int dd_cos1k_launch(void* stv, const DDChainParams& P, hipStream_t stream) {
    DDCos1kState* s = reinterpret_cast<DDCos1kState*>(stv);
    if (P.L < 1) return DD_OK;
    static DDOncePerDevice attr;
    if (attr.need()) {
        for (int i = 0; i < 8; ++i) DD_HIP_CHECK(hipFuncSetAttribute(cos1k_kernel(i & 1, i & 2, i & 4), hipFuncAttributeMaxDynamicSharedMemorySize, C1_LDS_BYTES));
        attr.mark();
    }
    const bool nco = (P.flags & DD_CHAIN_NCO) != 0;
    const long double PI2 = 2.0L * 3.14159265358979323846264338327950288L;
    const long double frac = nco ? (long double)P.cyc / 18446744073709551616.0L : 0.0L;
    DDCos1kArgs A;
    A.in = P.in; A.out = P.out;
    A.tail_in = P.tail_in; A.tail_out = P.tail_out; A.lasty_in = P.lasty_in; A.lasty_out = P.lasty_out;
    A.nco_tbl = P.nco_tbl; A.lane_tab = s->lane_tab;
    A.cyc = P.cyc; A.abs0 = P.abs0; A.L = P.L; A.s = P.s;
    A.a0 = (float)s->a0; A.a1 = (float)s->a1;
    auto ph = [&](int m) { long double p = frac * (long double)m; p -= floorl(p); const long double a = PI2 * p; return make_float2((float)cosl(a), (float)-sinl(a)); };
    A.q1 = ph(128); A.q2 = ph(256); A.q4 = ph(512);
    A.e1 = ph(1); A.e2 = ph(2); A.e3 = ph(3); A.e4 = ph(4); A.e8 = ph(8);
    const bool cx = !(P.flags & DD_CHAIN_FM);
    // rows start where `out` starts a line: 16 angles = 64 bytes, 16 complex outputs = 128
    const int a16 = (int)((reinterpret_cast<uintptr_t>(P.out) >> (cx ? 3 : 2)) & 15);
    static const char* wg_env = DD_TUNE_ENV("DD_COS_WGS_PER_CU");            // tools: occupancy experiments
    int grid;
    cos1k_plan(P.L, P.s, a16, dd_cu_count(), wg_env ? atoi(wg_env) : 2, &A.base, &A.nrows, &grid, &A.nwaves);
    static const char* run_env = DD_TUNE_ENV("DD_COS_RUN");                 // tools: rows per run of the moving-window map (0 = one run per wave)
    A.run_rows = run_env ? atoi(run_env) : -1;
    static const char* grid_env = DD_TUNE_ENV("DD_COS_GRID");               // tools: a fixed number of workgroups
    if (grid_env && atoi(grid_env) > 0 && (int64_t)atoi(grid_env) * C1_WAVES <= A.nrows) { grid = atoi(grid_env); A.nwaves = grid * C1_WAVES; }
    // runs of 8 rows dealt to the waves in turn once every wave gets at least two of them: the device then walks one moving window of
    // nwaves x 64 KB instead of nwaves streams far apart (memory side alone 5.64 -> 6.05 TB/s, the kernel 0.1455 -> 0.1417 ms in one call,
    // profiles/r05_cos1k_memory_only.txt; a run start costs c1_prime_light: 2 KB read twice and a third of a row's instructions)
    if (A.run_rows < 0) A.run_rows = A.nrows >= 16 * A.nwaves ? 8 : 0;
    const bool u8 = (P.flags & DD_CHAIN_U8_INPUT) != 0;
    const dim3 g(grid), b(64 * C1_WAVES);
    void* kargs[1] = {&A};
    DD_HIP_CHECK(hipLaunchKernel(cos1k_kernel(u8, nco, cx), g, b, kargs, C1_LDS_BYTES, stream));
    return DD_OK;
}

// dd_code_warmup (dd_runtime.hip): the runtime loads a translation unit's code object when one of its kernels is first named
int dd_code_touch_cosfir(void) {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, (const void*)k_chain_cos1k<false, true, false>) == hipSuccess ? DD_OK : DD_ERR_HIP;
}
