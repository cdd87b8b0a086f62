// Fused hot path on the matrix cores (general real taps, M == 1):
//
//     offsetFreq (NCO) -> FIR as a Toeplitz GEMM on v_mfma_f32_32x32x16_f16 -> demod_fm
//
// Why: in direct form the 255-tap complex FIR costs 1020 flop per 12 algorithmic
// bytes -- compute bound at ~23 % of the HBM roofline on the f32 VALU *or* the f32
// MFMA (same 157 TF peak).  The f16 matrix pipe is 16x faster, and
// float32-grade accuracy is kept by splitting both operands into two f16 limbs
// (x = xh + xl, g = gh + gl, 11+11 significant bits each) and accumulating
//     gh*xh + gh*xl + gl*xh        (the dropped gl*xl term is < 2^-22 relative)
// in the MFMA's f32 accumulator: 3/16 of the f32 cost.
//
// GEMM shape per wave (one "strip" of 1024 consecutive outputs):
//     D[i][j] = y[32 i + j] = sum_m A[i][m] * B[m][j],   i, j in [0, 32)
//     A[i][m] = s[32 i + m]          signal window of segment i   (LDS, f16 limbs)
//     B[m][j] = g2[m - j]            Toeplitz band of the reversed taps (constant fragments)
// K-dimension = 31 + (HALO+1) padded to 16*NKS.  With A = signal, the 32 lanes of
// one accumulator register hold 32 CONSECUTIVE outputs.
//
// LDS image: four f16 planes (re_hi, re_lo, im_hi, im_lo) of the NCO-rotated,
// power-of-two-scaled tile; every 32 samples are followed by 16 B of padding so the
// 64-byte-strided ds_read_b128 of the A fragments is bank-conflict free
// (dword index 20 i + 4 h, distinct for the 16 lanes of every b128 lane group).
//
// Two kernels:
//   k_chain_mfma_ws    interior tiles: persistent, wave-specialised (4 matrix + 12 vector
//                      waves per CU), see the block comment above it;
//   k_chain_mfma_edge  stream start/end, unaligned or u8 input, partial tiles: one tile
//                      per 4-wave workgroup, fully predicated, same MFMA core.
#include "dd_chain_kernels.h"
#include "dd_fftfir.h"
#include "dd_cosfir.h"
#include <stdlib.h>
#include <atomic>

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define MF_WAVES 4
#define MF_THREADS (MF_WAVES * 64)
#define MF_STRIP 1024
#define MF_T (MF_WAVES * MF_STRIP)
#define MF_ADV (MF_T - 32)

struct DDMfmaTaps {
    const v8h* frag;     // [limb][ks][lane] B fragments
    float inv_tapscale;  // 1 / (power-of-two scale applied to the taps)
    unsigned long long* stamps;   // diagnostic build only (env DD_STAMPS): per-wave segment cycle sums
};

__device__ __forceinline__ float dd_pow2_scale_for(float m) {
    // power of two s with m*s in [2^13, 2^14)  (s = 1 for m == 0 / denormal)
    const uint32_t eb = (__float_as_uint(m) >> 23) & 0xff;
    int se = 267 - (int)eb;
    se = se > 254 ? 254 : (se < 1 ? 1 : se);
    return eb == 0 ? 1.0f : __uint_as_float((uint32_t)se << 23);
}

// atan2 for the discriminator: odd degree-15 minimax polynomial on [0,1] (max error
// 4e-8 rad in exact arithmetic, 1.5e-7 rad evaluated in f32) + octant fix-up.
__device__ __forceinline__ float dd_fast_atan2(float y, float x) {
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
    r = (mx == 0.f) ? 0.f : r;                          // atan2(0,0) = 0 like np.angle
    r = (ay > ax) ? 1.5707963267948966f - r : r;
    r = (x < 0.f) ? 3.141592653589793f - r : r;
    return copysignf(r, y);
}

// atan(y/x) for x > 0, |y| <= tan(pi/8) x: t + t z (c0 + c1 z + c2 z^2 + c3 z^3), z = t^2
// (minimax fit on [0, tan(pi/8)]; 2.3e-8 rad evaluated in f32), no octant logic
__device__ __forceinline__ float dd_atan_small(float y, float x) {
    const float t = y * __builtin_amdgcn_rcpf(x);
    const float z = t * t;
    float p = fmaf(7.902598251e-02f, z, -1.382445378e-01f);
    p = fmaf(p, z, 1.997187931e-01f);
    p = fmaf(p, z, -3.333275667e-01f);
    return fmaf(t, z * p, t);
}

__device__ __forceinline__ float dd_fm_angle_fast(float cx, float cy, float px, float py) {
    const float re = fmaf(cx, px, cy * py);
    const float im = fmaf(cy, px, -cx * py);
    return dd_fast_atan2(im, re);
}

// value of the lane one to the left (DPP wave_shr:1); lanes 0/32 are patched by the caller
__device__ __forceinline__ float dd_lane_left(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dd_readlane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// bytes of LDS used by one tile's staging (planes + phasors + reduction + strip hand-over)
#define MF_LDS_TILE_BYTES(NKS) ((4 * MfmaGeom<NKS>::PLANE + 8 * MfmaGeom<NKS>::NGRP + 4 * MF_WAVES + 8 * MF_WAVES + 15) & ~15)

template <int NKS>
struct MfmaGeom {
    static constexpr int HALO = 16 * NKS - 32;
    static constexpr int SPAN = MF_T + HALO;                     // staged samples (multiple of 32)
    static constexpr int PLANE = SPAN * 2 + (SPAN / 32) * 16;    // bytes per f16 plane incl. padding
    static constexpr int NIT = (SPAN / 2 + MF_THREADS - 1) / MF_THREADS;
    static constexpr int NGRP = (SPAN + 63) / 64;                // 64-sample NCO phasor groups (SPAN need not be a multiple of 64)
};

// samples n0, n0+1 of the chunk as the FIR sees them: stream edges, carried history (already
// NCO-rotated), u8 ingest.  Every load is unconditional on a clamped index and the value is
// selected afterwards: a predicated load makes hipcc branch and drain vmcnt per element
// (measured: one edge tile took 15 us that way).
__device__ __forceinline__ float4 dd_edge_fetch2(const DDChainParams& P, int64_t n0) {
    const bool u8 = (P.flags & DD_CHAIN_U8_INPUT) != 0;
    const int K1 = P.K - 1;
    float v[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int64_t n = n0 + k;
        const int64_t nc = n < 0 ? 0 : (n >= P.L ? P.L - 1 : n);
        float2 x;
        if (u8) {
            const uchar2 u = reinterpret_cast<const uchar2*>(P.in)[nc];
            x = make_float2((float)u.x - 127.5f, (float)u.y - 127.5f);
        } else {
            x = reinterpret_cast<const float2*>(P.in)[nc];
        }
        const int64_t ti = n + K1;                       // index into the carried history
        const int64_t tc = ti < 0 ? 0 : (ti >= K1 ? (K1 > 0 ? K1 - 1 : 0) : ti);
        const float2 t = P.tail_in[tc];
        const bool in_chunk = n >= 0 && n < P.L;
        const bool in_tail = n < 0 && ti >= 0;
        v[2 * k] = in_chunk ? x.x : (in_tail ? t.x : 0.f);
        v[2 * k + 1] = in_chunk ? x.y : (in_tail ? t.y : 0.f);
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

// edge tiles: issue the tile's global loads (two consecutive samples per lane per step)
template <int NKS>
__device__ __forceinline__ void dd_tile_load(const DDChainParams& P, int b, float4 (&raw)[MfmaGeom<NKS>::NIT]) {
    using G = MfmaGeom<NKS>;
    const int tid = threadIdx.x;
    const int64_t ns = (int64_t)b * MF_ADV - 32 - G::HALO;
#pragma unroll
    for (int it = 0; it < G::NIT; ++it) raw[it] = dd_edge_fetch2(P, ns + 2 * (tid + MF_THREADS * it));
}

// phasor of the first sample of this thread's 64-sample group of tile b (threads
// < NGRP); issued together with the tile's loads so its table fetch is off the
// critical path
template <int NKS>
__device__ __forceinline__ float2 dd_tile_w2(const DDChainParams& P, int b) {
    using G = MfmaGeom<NKS>;
    static_assert(G::NGRP <= MF_THREADS, "one group phasor per thread");
    const int64_t ns = (int64_t)b * MF_ADV - 32 - G::HALO;
    const int g = threadIdx.x < G::NGRP ? threadIdx.x : G::NGRP - 1;
    if (!(P.flags & DD_CHAIN_NCO)) return make_float2(1.f, 0.f);
    return dd_phasor((uint64_t)(P.abs0 + ns + (int64_t)g * 64) * P.cyc, P.nco_tbl);
}

// rotate, scale, split into f16 limbs, write the LDS planes.  Returns the tile's
// power-of-two scale.  Contains one barrier (max reduction; it also fences the
// previous tile's LDS reads).
template <int NKS>
__device__ __forceinline__ float dd_tile_stage(const DDChainParams& P, int b, const float4 (&raw)[MfmaGeom<NKS>::NIT],
                                               char* smem, float2 w1a, float2 w1b, float2 w2mine) {
    using G = MfmaGeom<NKS>;
    char* planes = smem;
    float2* w2 = reinterpret_cast<float2*>(smem + 4 * G::PLANE);
    float* red = reinterpret_cast<float*>(w2 + G::NGRP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t ns = (int64_t)b * MF_ADV - 32 - G::HALO;
    const bool nco = (P.flags & DD_CHAIN_NCO) != 0;
    const int K = P.K;

    float m = 0.f;
#pragma unroll
    for (int it = 0; it < G::NIT; ++it) {
        m = fmaxf(m, fmaxf(fmaxf(fabsf(raw[it].x), fabsf(raw[it].y)), fmaxf(fabsf(raw[it].z), fabsf(raw[it].w))));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();                    // previous tile: all LDS reads (planes, w2, red, wlast) are done
    if (lane == 0) red[wave] = m;
    if (tid < G::NGRP) w2[tid] = w2mine;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = dd_pow2_scale_for(m);
    const float inv_scale = 1.0f / scale;

    const int64_t tail_first = P.L - (K - 1);       // first sample of the new history
    const bool tail_writer = (b == P.nblocks - 1) && P.tail_out != nullptr;
#pragma unroll
    for (int it = 0; it < G::NIT; ++it) {
        const int e = 2 * (tid + MF_THREADS * it);
        if (e >= G::SPAN) continue;
        const int64_t n = ns + e;
        float2 pa = make_float2(scale, 0.f), pb = make_float2(scale, 0.f);
        if (nco) {
            const float2 g = w2[e >> 6];
            const float2 gs = make_float2(g.x * scale, g.y * scale);
            if (n >= 0) pa = dd_cmul(gs, w1a);                 // history samples (n < 0) are already rotated
            if (n + 1 >= 0) pb = dd_cmul(gs, w1b);
        }
        const float2 xa = dd_cmul(make_float2(raw[it].x, raw[it].y), pa);
        const float2 xb = dd_cmul(make_float2(raw[it].z, raw[it].w), pb);
        if (tail_writer) {
            if (n >= tail_first && n < P.L) P.tail_out[n - tail_first] = make_float2(xa.x * inv_scale, xa.y * inv_scale);
            if (n + 1 >= tail_first && n + 1 < P.L) P.tail_out[n + 1 - tail_first] = make_float2(xb.x * inv_scale, xb.y * inv_scale);
        }
        v2h rh, rl, ih, il;
        rh.x = (_Float16)xa.x; rh.y = (_Float16)xb.x;
        ih.x = (_Float16)xa.y; ih.y = (_Float16)xb.y;
        rl.x = (_Float16)(xa.x - (float)rh.x); rl.y = (_Float16)(xb.x - (float)rh.y);
        il.x = (_Float16)(xa.y - (float)ih.x); il.y = (_Float16)(xb.y - (float)ih.y);
        const int off = 2 * e + 16 * (e >> 5);
        *reinterpret_cast<v2h*>(planes + off) = rh;
        *reinterpret_cast<v2h*>(planes + G::PLANE + off) = rl;
        *reinterpret_cast<v2h*>(planes + 2 * G::PLANE + off) = ih;
        *reinterpret_cast<v2h*>(planes + 3 * G::PLANE + off) = il;
    }
    return scale;
}

// epilogue.  lane (j = lane & 31, h = lane >> 5), register r holds output
//   p = P0 + 1024*wave + 32*row + j,  row = (r & 3) + 8 (r >> 2) + 4 h
template <int NKS>
__device__ __forceinline__ void dd_tile_epilogue(const DDChainParams& P, int b, v16f& cre, v16f& cim, float unscale, char* smem) {
    using G = MfmaGeom<NKS>;
    float2* wlast = reinterpret_cast<float2*>(smem + 4 * G::PLANE + sizeof(float2) * G::NGRP + sizeof(float) * MF_WAVES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t P0 = (int64_t)b * MF_ADV - 32;
    const int64_t pw = P0 + (int64_t)wave * MF_STRIP;
    const int64_t p_lo = P0 + 32;                          // first output this tile owns
    const bool fm = (P.flags & DD_CHAIN_FM) != 0;

    if (!fm) {
        float2* out = reinterpret_cast<float2*>(P.out);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int64_t p = pw + 32 * row + j;
            if (p >= p_lo && p < P.Ld)
                out[p] = make_float2(cre[r] * unscale, cim[r] * unscale);
        }
        return;
    }

    if (P.s == 0 && b == 0 && wave == 0 && lane == 31) {   // p == -1: sample carried from the previous chunk
        const float2 ly = *P.lasty_in;                     // (any positive scale: only its angle matters)
        cre[0] = ly.x;
        cim[0] = ly.y;
    }
    if (lane == 63) wlast[wave] = make_float2(cre[15], cim[15]);
    __syncthreads();
    const float2 prev_strip = (wave > 0) ? wlast[wave - 1] : make_float2(0.f, 0.f);

    // column-0 neighbours: row-1 is register r-1 of lane 31/63, or sits across the
    // 4-row split of the accumulator layout (register r+3 / r-1 of the other half)
    float* out = reinterpret_cast<float*>(P.out) + (pw - P.s) + j + 128 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float pre = dd_lane_left(cre[r]);
        float pim = dd_lane_left(cim[r]);
        float a_re, a_im, c_re, c_im;                      // for lane 0 and lane 32
        if ((r & 3) != 0) {
            a_re = dd_readlane(cre[r - 1], 31); a_im = dd_readlane(cim[r - 1], 31);
            c_re = dd_readlane(cre[r - 1], 63); c_im = dd_readlane(cim[r - 1], 63);
        } else {
            if (r > 0) { a_re = dd_readlane(cre[r - 1], 63); a_im = dd_readlane(cim[r - 1], 63); }
            else { a_re = prev_strip.x; a_im = prev_strip.y; }
            c_re = dd_readlane(cre[r + 3], 31); c_im = dd_readlane(cim[r + 3], 31);
        }
        pre = (lane == 0) ? a_re : pre;  pim = (lane == 0) ? a_im : pim;
        pre = (lane == 32) ? c_re : pre; pim = (lane == 32) ? c_im : pim;
        const int rowbase = (r & 3) + 8 * (r >> 2);        // row = rowbase + 4h
        const float ang = dd_fm_angle_fast(cre[r], cim[r], pre, pim);
        const int64_t p = pw + 32 * (rowbase + 4 * h) + j;
        if (p >= p_lo && p >= P.s && p < P.Ld) out[32 * rowbase] = ang;
        if (p == P.Ld - 1) *P.lasty_out = make_float2(cre[r] * unscale, cim[r] * unscale);
    }
}

// Edge tiles (stream start/end, unaligned or u8 input, partial tiles): one tile per 4-wave
// workgroup, fully predicated, tap fragments held in registers (stand-alone edge kernel).
template <int NKS>
__device__ __forceinline__ void dd_edge_tile(const DDChainParams& P, const DDMfmaTaps& taps, int b, char* smem) {
    using G = MfmaGeom<NKS>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4 raw[G::NIT];
    dd_tile_load<NKS>(P, b, raw);
    float2 w1a = make_float2(1.f, 0.f), w1b = make_float2(1.f, 0.f);
    if (P.flags & DD_CHAIN_NCO) {
        w1a = dd_phasor((uint64_t)((2 * tid) & 63) * P.cyc, P.nco_tbl);
        w1b = dd_phasor((uint64_t)(((2 * tid) & 63) + 1) * P.cyc, P.nco_tbl);
    }
    const float scale = dd_tile_stage<NKS>(P, b, raw, smem, w1a, w1b, dd_tile_w2<NKS>(P, b));
    v8h bh[NKS], bl[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        bh[ks] = taps.frag[ks * 64 + lane];
        bl[ks] = taps.frag[(NKS + ks) * 64 + lane];
    }
    __syncthreads();
    const int i = lane & 31, h = lane >> 5;
    const int sb = wave * MF_STRIP;
    const char* abase = smem + (2 * sb + (sb >> 1)) + 80 * i + 16 * h;
    v16f cre, cim;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cre[r] = 0.f; cim[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int off = 32 * ks + 16 * (ks >> 1);
        const v8h arh = *reinterpret_cast<const v8h*>(abase + off);
        const v8h arl = *reinterpret_cast<const v8h*>(abase + G::PLANE + off);
        const v8h aih = *reinterpret_cast<const v8h*>(abase + 2 * G::PLANE + off);
        const v8h ail = *reinterpret_cast<const v8h*>(abase + 3 * G::PLANE + off);
        const v8h th = bh[ks], tl = bl[ks];
        cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, th, cre, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, th, cim, 0, 0, 0);
        cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(arl, th, cre, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(ail, th, cim, 0, 0, 0);
        cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, tl, cre, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, tl, cim, 0, 0, 0);
    }
    dd_tile_epilogue<NKS>(P, b, cre, cim, taps.inv_tapscale / scale, smem);
}

// The same edge tile inside the register budget of the 16-wave ws kernel (a kernel that
// spills there runs its persistent loop ~12 % slower -- measured -- even though the spills
// sit in this path only): the tile is fetched twice, once for its peak and once to stage
// it, three fetches in flight, instead of being held in 36 registers.
template <int NKS>
__device__ __forceinline__ void dd_edge_tile_lean(const DDChainParams& P, const DDMfmaTaps& taps, int b, char* smem, const v8h* lds_taps) {
    using G = MfmaGeom<NKS>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t ns = (int64_t)b * MF_ADV - 32 - G::HALO;
    char* planes = smem;
    float* red = reinterpret_cast<float*>(smem + 4 * G::PLANE + 8 * G::NGRP);   // same LDS layout as dd_tile_stage
    const bool nco = (P.flags & DD_CHAIN_NCO) != 0;
    float m = 0.f;
#pragma unroll 3
    for (int it = 0; it < G::NIT; ++it) {
        const float4 v = dd_edge_fetch2(P, ns + 2 * (tid + MF_THREADS * it));
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = dd_pow2_scale_for(m);
    const float inv_scale = 1.0f / scale;
    const int64_t tail_first = P.L - (P.K - 1);     // first sample of the new history
    const bool tail_writer = (b == P.nblocks - 1) && P.tail_out != nullptr;
#pragma unroll 3
    for (int it = 0; it < G::NIT; ++it) {
        const int e = 2 * (tid + MF_THREADS * it);
        if (e >= G::SPAN) continue;
        const int64_t n = ns + e;
        const float4 v = dd_edge_fetch2(P, n);
        float2 pa = make_float2(scale, 0.f), pb = make_float2(scale, 0.f);
        if (nco) {                                  // history samples (n < 0) are already rotated
            if (n >= 0) { const float2 w = dd_phasor((uint64_t)(P.abs0 + n) * P.cyc, P.nco_tbl); pa = make_float2(w.x * scale, w.y * scale); }
            if (n + 1 >= 0) { const float2 w = dd_phasor((uint64_t)(P.abs0 + n + 1) * P.cyc, P.nco_tbl); pb = make_float2(w.x * scale, w.y * scale); }
        }
        const float2 xa = dd_cmul(make_float2(v.x, v.y), pa);
        const float2 xb = dd_cmul(make_float2(v.z, v.w), pb);
        if (tail_writer) {
            if (n >= tail_first && n < P.L) P.tail_out[n - tail_first] = make_float2(xa.x * inv_scale, xa.y * inv_scale);
            if (n + 1 >= tail_first && n + 1 < P.L) P.tail_out[n + 1 - tail_first] = make_float2(xb.x * inv_scale, xb.y * inv_scale);
        }
        v2h rh, rl, ih, il;
        rh.x = (_Float16)xa.x; rh.y = (_Float16)xb.x;
        ih.x = (_Float16)xa.y; ih.y = (_Float16)xb.y;
        rl.x = (_Float16)(xa.x - (float)rh.x); rl.y = (_Float16)(xb.x - (float)rh.y);
        il.x = (_Float16)(xa.y - (float)ih.x); il.y = (_Float16)(xb.y - (float)ih.y);
        const int off = 2 * e + 16 * (e >> 5);
        *reinterpret_cast<v2h*>(planes + off) = rh;
        *reinterpret_cast<v2h*>(planes + G::PLANE + off) = rl;
        *reinterpret_cast<v2h*>(planes + 2 * G::PLANE + off) = ih;
        *reinterpret_cast<v2h*>(planes + 3 * G::PLANE + off) = il;
    }
    __syncthreads();
    const int i = lane & 31, h = lane >> 5;
    const int sb = wave * MF_STRIP;
    const char* abase = smem + (2 * sb + (sb >> 1)) + 80 * i + 16 * h;
    v16f cre, cim;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cre[r] = 0.f; cim[r] = 0.f; }
#pragma unroll 2
    for (int ks = 0; ks < NKS; ++ks) {
        const int off = 32 * ks + 16 * (ks >> 1);
        const v8h arh = *reinterpret_cast<const v8h*>(abase + off);
        const v8h arl = *reinterpret_cast<const v8h*>(abase + G::PLANE + off);
        const v8h aih = *reinterpret_cast<const v8h*>(abase + 2 * G::PLANE + off);
        const v8h ail = *reinterpret_cast<const v8h*>(abase + 3 * G::PLANE + off);
        const v8h th = lds_taps[ks * 64 + lane];
        const v8h tl = lds_taps[(NKS + ks) * 64 + lane];
        cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, th, cre, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, th, cim, 0, 0, 0);
        cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(arl, th, cre, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(ail, th, cim, 0, 0, 0);
        cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, tl, cre, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, tl, cim, 0, 0, 0);
    }
    dd_tile_epilogue<NKS>(P, b, cre, cim, taps.inv_tapscale / scale, smem);
}

template <int NKS>
__global__ void __launch_bounds__(MF_THREADS, 2) k_chain_mfma_edge(const DDChainParams P, const DDMfmaTaps taps, int t_first, int t_last) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // edge tiles are [0, t_first) and [t_last, nblocks)
    const int b = (int)blockIdx.x < t_first ? (int)blockIdx.x : t_last + ((int)blockIdx.x - t_first);
    dd_edge_tile<NKS>(P, taps, b, smem);
}

// ---------------------------------------------------------------------------------
// Interior tiles: wave-specialised persistent kernel (one 16-wave workgroup per CU).
//
//   waves 0..3   "matrix" waves, one per SIMD: the 108 MFMAs of one 1024-output strip
//                per tile (A = signal limbs and B = tap limbs, both read from LDS),
//                then the strip's FIR outputs go to an LDS y-buffer in output order;
//   waves 4..15  "vector" waves, three per SIMD: NCO rotation + f16 limb split of the
//                next tile into the other LDS plane buffer, and the discriminator
//                epilogue of the previous tile read back from the y-buffer: 4
//                consecutive outputs per lane, so y[n-1] is in the lane's own
//                registers and every store is a 16-byte-per-lane contiguous line.
//
// Why: the vector side is latency/issue bound when only one or two waves per SIMD
// run it (measured: ~10 cycles per VALU instruction, 2.6x the matrix time).  Three
// vector waves per SIMD hide that latency by thread-level parallelism while the
// matrix wave keeps the MFMA pipe busy; 128 VGPRs per wave make 4 waves/SIMD fit.
//
// Pipeline over the workgroup's tiles, ONE workgroup barrier per phase p:
//   vector: issue loads of tile p+2 | epilogue of tile p-2 from the y-buffer, signal "y read" |
//           convert tile p -> planes[p&1] | peak of tile p+1 |  barrier
//   matrix: own epilogue unit of tile p-2, signal "y read" | 108 MFMAs of tile p-1 from
//           planes[(p-1)&1] | wait until all 16 waves have read y, write y of tile p-1 |  barrier
// ---------------------------------------------------------------------------------
#define WS_THREADS 1024
#define WS_MWAVES 4
#define WS_VTHREADS (WS_THREADS - 64 * WS_MWAVES)     // 768
#define WS_VWAVES (WS_VTHREADS / 64)                   // 12

template <int NKS>
struct WsGeom {
    using G = MfmaGeom<NKS>;
    static constexpr int NQ = G::SPAN / 2;                        // sample pairs per tile
    static constexpr int NIT = (NQ + WS_VTHREADS - 1) / WS_VTHREADS;   // 3 steps for 12 vector waves: 94 % lane use
    static constexpr int PLANES_BYTES = 4 * G::PLANE;             // one plane buffer
    static constexpr int YBUF_OFF = 2 * PLANES_BYTES;
    // y-buffer: planar (real plane, imaginary plane), each with 16 B of slack in front (y[-1] of lane 0).
    // Planar so that a matrix wave stores its accumulators as they sit in registers (ds_write2_b32 pairs):
    // the interleaved float2 form needed 24 v_mov per strip to pair re with im first.
    static constexpr int YPLANE_F = 4 + MF_T;
    static constexpr int YBUF_BYTES = 2 * YPLANE_F * 4;
    static constexpr int TAPS_OFF = YBUF_OFF + YBUF_BYTES;
    static constexpr int TAPS_BYTES = 2 * NKS * 64 * 16;
    static constexpr int RED_OFF = TAPS_OFF + TAPS_BYTES;         // [2][WS_VWAVES] float
    static constexpr int SCALE_OFF = RED_OFF + 2 * WS_VWAVES * 4; // [4] float
    static constexpr int YDONE_OFF = SCALE_OFF + 16;              // int: waves that finished reading the y-buffer
    static constexpr int NONUNIT_OFF = YDONE_OFF + 16;            // [4] int: "some wave saw a tile part outside the unit range", per tile slot
    static constexpr int TILEW_OFF = NONUNIT_OFF + 16;            // [4] float2: NCO phasor of a tile's first sample (complex-output flavour)
    static constexpr int LDS_BYTES = (TILEW_OFF + 32 + 15) & ~15;
};

template <int NKS, bool U8>
__device__ __forceinline__ void dd_ws_load(const DDChainParams& P, int b, int vt, float4 (&raw)[WsGeom<NKS>::NIT]) {
    using G = MfmaGeom<NKS>;
    using W = WsGeom<NKS>;
    const int64_t ns = (int64_t)b * MF_ADV - 32 - G::HALO;
    if (U8) {
        // raw u8 I,Q pairs (source.py:117-118): two samples = one dword per lane, widened here
        const char* base = reinterpret_cast<const char*>(P.in) + 2 * ns;                            // wave-uniform
#pragma unroll
        for (int it = 0; it < W::NIT; ++it) {
            int q = vt + WS_VTHREADS * it;
            if (WS_VTHREADS * (it + 1) > W::NQ) q = q < W::NQ ? q : W::NQ - 1;
            const uint32_t d = *reinterpret_cast<const uint32_t*>(base + 4u * (unsigned)q);
            raw[it] = make_float4((float)(d & 0xff) - 127.5f, (float)((d >> 8) & 0xff) - 127.5f,
                                  (float)((d >> 16) & 0xff) - 127.5f, (float)(d >> 24) - 127.5f);
        }
        return;
    }
    const char* base = reinterpret_cast<const char*>(reinterpret_cast<const float2*>(P.in) + ns);   // wave-uniform
#pragma unroll
    for (int it = 0; it < W::NIT; ++it) {
        int q = vt + WS_VTHREADS * it;
        if (WS_VTHREADS * (it + 1) > W::NQ) q = q < W::NQ ? q : W::NQ - 1;   // partial last step: re-read, write masked
        raw[it] = *reinterpret_cast<const float4*>(base + 16u * (unsigned)q);   // two consecutive samples
    }
}

// NCO with TILE-RELATIVE phase: x[n] e^{-j w (n0+k)} = e^{-j w n0} (x[n] e^{-j w k}).  The
// lane's phasors W[k] (k = its fixed positions inside a tile) are loop invariant registers;
// the per-tile factor e^{-j w n0} is common to every output of the tile, so it cancels in
// the discriminator's y[n] conj(y[n-1]) and is applied in the epilogue only for complex
// output.  Plain v_fma/v_mul on purpose: packed f32 ops beside the matrix waves' MFMAs
// cost ~5x a plain op (measured: the vector phases ran 2.3x slower with v_pk_*).
template <int NKS, bool UNIT_SCALE>
__device__ __forceinline__ void dd_ws_convert(const float4 (&raw)[WsGeom<NKS>::NIT], char* planes,
                                              const float2 (&wk)[WsGeom<NKS>::NIT][2], float scale, int vt) {
    using G = MfmaGeom<NKS>;
    using W = WsGeom<NKS>;
#pragma unroll
    for (int it = 0; it < W::NIT; ++it) {
        const int q = vt + WS_VTHREADS * it;
        if (WS_VTHREADS * (it + 1) > W::NQ && q >= W::NQ) continue;
        const int e = 2 * q;
        float2 pa = wk[it][0], pb = wk[it][1];
        if (!UNIT_SCALE) {
            pa = make_float2(pa.x * scale, pa.y * scale);
            pb = make_float2(pb.x * scale, pb.y * scale);
        }
        const float2 xa = dd_cmul(make_float2(raw[it].x, raw[it].y), pa);
        const float2 xb = dd_cmul(make_float2(raw[it].z, raw[it].w), pb);
        uint32_t rh, rl, ih, il;
        // (hi = RNE(x), lo = RNE(x - hi); left to the compiler's selection on purpose: a
        // hand-written 3.5-instruction form built on v_fma_mix with an f16 SOURCE operand made
        // the matrix waves' MFMAs on the same SIMD run at 50 cycles each instead of 35 -- like
        // packed f32 ops, that form does not coexist with the matrix pipe)
#ifndef DD_WS_MIX_SPLIT
        {   // the 12-instruction plain split of k_chain_mfma_ab (dd_ab_split2); round 2 A/B: FM 0.2361 -> 0.2356 ms, complex output
            // 0.2866 -> 0.2767 ms (the compiler's form below, with six v_fma_mix* per sample pair, stays selectable: -DDD_WS_MIX_SPLIT)
            float t0, t1, t2, t3;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(rh) : "v"(xa.x), "v"(xb.x));
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ih) : "v"(xa.y), "v"(xb.y));
            asm("v_cvt_f32_f16_e32 %0, %1" : "=v"(t0) : "v"(rh));
            asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(t1) : "v"(rh));
            asm("v_cvt_f32_f16_e32 %0, %1" : "=v"(t2) : "v"(ih));
            asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(t3) : "v"(ih));
            t0 = xa.x - t0; t1 = xb.x - t1; t2 = xa.y - t2; t3 = xb.y - t3;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(rl) : "v"(t0), "v"(t1));
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(il) : "v"(t2), "v"(t3));
        }
#else
        {
            v2h rh_, rl_, ih_, il_;
            rh_.x = (_Float16)xa.x; rh_.y = (_Float16)xb.x;
            ih_.x = (_Float16)xa.y; ih_.y = (_Float16)xb.y;
            rl_.x = (_Float16)fmaf((float)rh_.x, -1.0f, xa.x); rl_.y = (_Float16)fmaf((float)rh_.y, -1.0f, xb.x);
            il_.x = (_Float16)fmaf((float)ih_.x, -1.0f, xa.y); il_.y = (_Float16)fmaf((float)ih_.y, -1.0f, xb.y);
            rh = __builtin_bit_cast(uint32_t, rh_); rl = __builtin_bit_cast(uint32_t, rl_);
            ih = __builtin_bit_cast(uint32_t, ih_); il = __builtin_bit_cast(uint32_t, il_);
        }
#endif
        const int off = 2 * e + 16 * (e >> 5);
        *reinterpret_cast<uint32_t*>(planes + off) = rh;
        *reinterpret_cast<uint32_t*>(planes + G::PLANE + off) = rl;
        *reinterpret_cast<uint32_t*>(planes + 2 * G::PLANE + off) = ih;
        *reinterpret_cast<uint32_t*>(planes + 3 * G::PLANE + off) = il;
    }
}

// epilogue of one 256-output unit u of tile b: lane handles outputs o = 256u + 4 lane + {0..3}
template <int NKS, bool CX = false>
__device__ __forceinline__ void dd_ws_epilogue_unit(const DDChainParams& P, int b, int u, int lane, const float2* yb, float unscale,
                                                    const float2* tilew = nullptr) {
    const int o = 256 * u + 4 * lane;
    if (o < 32) return;                                    // the tile's first column belongs to the previous tile
    // planar y-buffer: yb points at real[0]; imaginary plane YPLANE_F floats further
    const float* yre = reinterpret_cast<const float*>(yb);
    const float* yim = yre + WsGeom<NKS>::YPLANE_F;
    const float4 r4 = *reinterpret_cast<const float4*>(yre + o), i4 = *reinterpret_cast<const float4*>(yim + o);
    const float2 ym = make_float2(yre[o - 1], yim[o - 1]);
    const float4 y01 = make_float4(r4.x, i4.x, r4.y, i4.y);
    const float4 y23 = make_float4(r4.z, i4.z, r4.w, i4.w);
    const int64_t p = (int64_t)b * MF_ADV - 32 + o;
    if (P.flags & DD_CHAIN_FM) {
        // z_k = y_k conj(y_{k-1})
        const float re0 = fmaf(y01.x, ym.x, y01.y * ym.y), im0 = fmaf(y01.y, ym.x, -y01.x * ym.y);
        const float re1 = fmaf(y01.z, y01.x, y01.w * y01.y), im1 = fmaf(y01.w, y01.x, -y01.z * y01.y);
        const float re2 = fmaf(y23.x, y01.z, y23.y * y01.w), im2 = fmaf(y23.y, y01.z, -y23.x * y01.w);
        const float re3 = fmaf(y23.z, y23.x, y23.w * y23.y), im3 = fmaf(y23.w, y23.x, -y23.z * y23.y);
        float a0, a1, a2, a3;
        // wave-uniform fast path: every |angle| below 22.5 degrees (an oversampled FM signal
        // always is): atan(t) needs no octant logic.  One ballot decides for the whole wave.  (Strict: a product of
        // exactly zero -- digital silence -- must take the full-range form, which returns 0 for it; the small-angle one
        // would compute 0 * rcp(0) = NaN)
        const unsigned long long big = __builtin_amdgcn_ballot_w64(!(fabsf(im0) < 0.41421354f * re0)) |
                                       __builtin_amdgcn_ballot_w64(!(fabsf(im1) < 0.41421354f * re1)) |
                                       __builtin_amdgcn_ballot_w64(!(fabsf(im2) < 0.41421354f * re2)) |
                                       __builtin_amdgcn_ballot_w64(!(fabsf(im3) < 0.41421354f * re3));
        if (big == 0) {
            a0 = dd_atan_small(im0, re0); a1 = dd_atan_small(im1, re1);
            a2 = dd_atan_small(im2, re2); a3 = dd_atan_small(im3, re3);
        } else {
            a0 = dd_fast_atan2(im0, re0); a1 = dd_fast_atan2(im1, re1);
            a2 = dd_fast_atan2(im2, re2); a3 = dd_fast_atan2(im3, re3);
        }
        float* out = reinterpret_cast<float*>(P.out) + (p - P.s);
        if (P.s == 0) {
            *reinterpret_cast<float4*>(out) = make_float4(a0, a1, a2, a3);
        } else {                                           // first chunk of a stream: outputs shifted by one
            out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
        }
    } else {
        // complex output: undo the power-of-two scales and apply the tile's NCO factor
        float2 t = make_float2(unscale, 0.f);
        if (P.flags & DD_CHAIN_NCO) {
            float2 ph;
            if (CX) {
                // complex-output flavour: the tile's phasor was fetched two phases ago by one lane and left in
                // LDS -- fetched from the table here, its vmcnt wait would drain the vector waves' in-flight tile
                // loads (vmcnt retires in order) and stall a matrix wave for a memory latency per phase
                ph = *tilew;
            } else {
                const int64_t ns = (int64_t)b * MF_ADV - 32 - MfmaGeom<NKS>::HALO;
                ph = dd_phasor((uint64_t)(P.abs0 + ns) * P.cyc, P.nco_tbl);
            }
            t = make_float2(ph.x * unscale, ph.y * unscale);
        }
        const float2 o0 = dd_cmul(make_float2(y01.x, y01.y), t), o1 = dd_cmul(make_float2(y01.z, y01.w), t);
        const float2 o2 = dd_cmul(make_float2(y23.x, y23.y), t), o3 = dd_cmul(make_float2(y23.z, y23.w), t);
        float4* out = reinterpret_cast<float4*>(reinterpret_cast<float2*>(P.out) + p);
        out[0] = make_float4(o0.x, o0.y, o1.x, o1.y);
        out[1] = make_float4(o2.x, o2.y, o3.x, o3.y);
    }
}

// max over the wave in 6 DPP steps (row_shr 1,2,4,8 then row_bcast 15/31); valid in lane 63
__device__ __forceinline__ float dd_wave_max(float m) {
#define DD_DPP_MAX(ctrl, rmask)                                                                              \
    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), ctrl, rmask, 0xf, false)));
    DD_DPP_MAX(0x111, 0xf)      // row_shr:1   (values are >= 0, so the 0 filled into invalid lanes is neutral)
    DD_DPP_MAX(0x112, 0xf)      // row_shr:2
    DD_DPP_MAX(0x114, 0xf)      // row_shr:4
    DD_DPP_MAX(0x118, 0xf)      // row_shr:8   -> lane 15 of each row holds the row max
    DD_DPP_MAX(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
    DD_DPP_MAX(0x143, 0xc)      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave max
#undef DD_DPP_MAX
    return m;
}

// One vector-wave phase p (ONE workgroup barrier per phase):
//   loads of tile p+2 are issued first (two phases of prefetch distance: HBM latency under
//   load is several microseconds); epilogue of tile p-2 out of the y-buffer, then the
//   wave signals "y-buffer read" on an LDS counter (the matrix waves wait on it before
//   they overwrite the buffer -- they get there thousands of cycles later, so the wait is
//   free); conversion of tile p with the scale published during phase p-1; tile max and
//   group phasors of tile p+1 for the next phase; barrier.
template <int NKS, bool U8, bool CX, bool ST>
__device__ __forceinline__ void dd_ws_vphase(const DDChainParams& P, const DDMfmaTaps& taps, char* smem, int t_begin, int n, int p,
                                             float4 (&rcur)[WsGeom<NKS>::NIT], float4 (&rnext)[WsGeom<NKS>::NIT],
                                             float4 (&rld)[WsGeom<NKS>::NIT],
                                             const float2 (&wk)[WsGeom<NKS>::NIT][2], int vt, int vw, int lane,
                                             unsigned long long (&acc_t)[8]) {
    using G = MfmaGeom<NKS>;
    using W = WsGeom<NKS>;
    const bool stamp = ST && taps.stamps != nullptr;        // ST: in-kernel stamps compiled in (tools only)
    unsigned long long tp = stamp ? __builtin_readcyclecounter() : 0;
#define DD_STAMP(i) if (stamp) { const unsigned long long tn = __builtin_readcyclecounter(); acc_t[i] += tn - tp; tp = tn; }
    float* redall = reinterpret_cast<float*>(smem + W::RED_OFF);
    float* scales = reinterpret_cast<float*>(smem + W::SCALE_OFF);
    const float2* yb = reinterpret_cast<const float2*>(smem + W::YBUF_OFF + 16);
    // complex-output flavour: one lane fetches tile p's start phasor BEFORE this phase's tile loads (so that
    // waiting for it does not wait for them) and publishes it in LDS for the epilogues of phase p+2
    float2* tilew = reinterpret_cast<float2*>(smem + W::TILEW_OFF);
    const bool mk_tilew = CX && vt == 0 && p < n && (P.flags & DD_CHAIN_NCO);
    float2 tw = make_float2(1.f, 0.f);
    if (CX && mk_tilew) tw = dd_phasor((uint64_t)(P.abs0 + (int64_t)(t_begin + p) * MF_ADV - 32 - G::HALO) * P.cyc, P.nco_tbl);
    {
        const int bl = t_begin + (p + 2 < n ? p + 2 : n - 1);   // past the end: harmless re-read, never used
        dd_ws_load<NKS, U8>(P, bl, vt, rld);
    }
    DD_STAMP(0)
    if (p >= 2 && p - 2 < n) {              // epilogue of tile p-2 (y-buffer written in phase p-1)
        const float unscale = taps.inv_tapscale / scales[(p - 2) & 3];
        if (CX) dd_ws_epilogue_unit<NKS, true>(P, t_begin + p - 2, vw, lane, yb, unscale, tilew + ((p - 2) & 3));
        else dd_ws_epilogue_unit<NKS>(P, t_begin + p - 2, vw, lane, yb, unscale);      // units 12..15: matrix waves
    }
    if (CX && mk_tilew) tilew[p & 3] = tw;
    if (lane == 0) atomicAdd(reinterpret_cast<int*>(smem + W::YDONE_OFF), 1);   // this wave is done with the y-buffer
    DD_STAMP(1)
    if (p < n) {                                            // convert tile p (max published in phase p-1)
        const float* red = redall + (p & 1) * WS_VWAVES;
        // common case: no wave raised the tile's non-unit flag -> unit scale, no reduction to read
        int* nonunit = reinterpret_cast<int*>(smem + W::NONUNIT_OFF);
        float m = 1.0f;
        if (__builtin_amdgcn_readfirstlane(nonunit[p & 3]) != 0) {
            m = red[0];
#pragma unroll
            for (int k = 1; k < WS_VWAVES; ++k) m = fmaxf(m, red[k]);
        }
        if (vt == 0) nonunit[(p + 2) & 3] = 0;              // re-arm the slot tile p+2's producers raise in phase p+1
        // f16 limbs hold the tile as it is when its peak lies in [0.25, 32768) (hi limb cannot
        // overflow after the rotation, the lo limb's subnormal floor stays below 2^-22 of the peak):
        // the common case (8-bit SDR samples peak at 181) skips the scaling multiplies
        const bool unit = (m >= 0.25f) && (m < 32768.0f);
        const float scale = unit ? 1.0f : dd_pow2_scale_for(m);
        if (vt == 0) scales[p & 3] = scale;
        if (unit) dd_ws_convert<NKS, true>(rcur, smem + (p & 1) * W::PLANES_BYTES, wk, scale, vt);
        else dd_ws_convert<NKS, false>(rcur, smem + (p & 1) * W::PLANES_BYTES, wk, scale, vt);
    }
    DD_STAMP(2)
    if (p + 1 < n) {                                        // tile max + group phasors of tile p+1
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < W::NIT; ++it) {
            m = fmaxf(fmaxf(m, fabsf(rnext[it].x)), fmaxf(fabsf(rnext[it].y), fmaxf(fabsf(rnext[it].z), fabsf(rnext[it].w))));
        }
        // Only the question "does the tile fit the f16 limbs unscaled" needs answering in the
        // common case: two ballots.  A wave whose lanes all sit inside [.., 32768) with at least
        // one >= 0.25 publishes 1.0 (any value of the unit range does); otherwise its true max
        // (DPP reduction, result in lane 63) -- the tile max over the waves' entries then still
        // selects the same scale as the exact max would.
        const bool hi_any = __builtin_amdgcn_ballot_w64(!(m < 32768.0f)) != 0;
        const bool lo_any = __builtin_amdgcn_ballot_w64(m >= 0.25f) != 0;
        if (hi_any || !lo_any) {
            m = dd_wave_max(m);
            if (lane == 63) atomicOr(reinterpret_cast<int*>(smem + W::NONUNIT_OFF) + ((p + 1) & 3), 1);
        } else m = 1.0f;
        if (lane == 63) redall[((p + 1) & 1) * WS_VWAVES + vw] = m;
    }
    DD_STAMP(3)
    __syncthreads();
    DD_STAMP(4)
}

template <int NKS, bool U8, bool CX, bool ST>
__device__ __forceinline__ void dd_ws_vector(const DDChainParams& P, const DDMfmaTaps& taps, char* smem, int t_begin, int t_end, int nph) {
    using W = WsGeom<NKS>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int vt = tid - 64 * WS_MWAVES, vw = vt >> 6;
    const int n = t_end - t_begin;

    // the first two tiles are requested before anything else: their (cold) HBM latency covers the
    // phasor table fetches below and the tap copy of the kernel prologue
    float4 r0[W::NIT], r1[W::NIT], r2[W::NIT];
    dd_ws_load<NKS, U8>(P, t_begin, vt, r0);
    dd_ws_load<NKS, U8>(P, t_begin + (n > 1 ? 1 : 0), vt, r1);
    // tile-relative NCO phasors of this lane's sample positions (loop invariant)
    float2 wk[W::NIT][2];
#pragma unroll
    for (int it = 0; it < W::NIT; ++it) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int pos = 2 * (vt + WS_VTHREADS * it) + k;
            wk[it][k] = (P.flags & DD_CHAIN_NCO) ? dd_phasor((uint64_t)pos * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
        }
    }
    {   // tile max and group phasors of tile 0 (what phase p-1 does for tile p)
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < W::NIT; ++it) {
            m = fmaxf(fmaxf(m, fabsf(r0[it].x)), fmaxf(fabsf(r0[it].y), fmaxf(fabsf(r0[it].z), fabsf(r0[it].w))));
        }
        m = dd_wave_max(m);
        if (lane == 63) reinterpret_cast<float*>(smem + W::RED_OFF)[vw] = m;
    }
    __syncthreads();                                        // prologue barrier (matched in dd_ws_matrix)

    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = 0; p < nph; p += 3) {                      // nph is a multiple of 3: no conditional phases
        dd_ws_vphase<NKS, U8, CX, ST>(P, taps, smem, t_begin, n, p, r0, r1, r2, wk, vt, vw, lane, acc_t);
        dd_ws_vphase<NKS, U8, CX, ST>(P, taps, smem, t_begin, n, p + 1, r1, r2, r0, wk, vt, vw, lane, acc_t);
        dd_ws_vphase<NKS, U8, CX, ST>(P, taps, smem, t_begin, n, p + 2, r2, r0, r1, wk, vt, vw, lane, acc_t);
    }
    if (ST && taps.stamps && lane == 0) {
        for (int q = 0; q < 6; ++q) taps.stamps[((size_t)blockIdx.x * 16 + (tid >> 6)) * 8 + q] = acc_t[q];
        taps.stamps[((size_t)blockIdx.x * 16 + (tid >> 6)) * 8 + 7] = (unsigned long long)nph | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
    }
}

// ------------------------------------------------------------------ matrix waves
template <int NKS, bool CX, bool ST>
__device__ __forceinline__ void dd_ws_matrix(const DDChainParams& P, const DDMfmaTaps& taps, char* smem, int t_begin, int t_end, int nph) {
    using G = MfmaGeom<NKS>;
    using W = WsGeom<NKS>;
    const int tid = threadIdx.x, lane = tid & 63, mw = tid >> 6;
    const int n = t_end - t_begin;
    const int i = lane & 31, h = lane >> 5;
    const int sb = mw * MF_STRIP;
    const int aoff = (2 * sb + (sb >> 1)) + 80 * i + 16 * h;
    const v8h* tb = reinterpret_cast<const v8h*>(smem + W::TAPS_OFF) + lane;
    float* ywr = reinterpret_cast<float*>(smem + W::YBUF_OFF + 16) + sb + 128 * h + i;    // + 32*rowbase(r); imaginary plane YPLANE_F further
    int* ydone = reinterpret_cast<int*>(smem + W::YDONE_OFF);
    __builtin_amdgcn_s_setprio(3);                          // MFMAs issue as soon as the pipe frees up
    __syncthreads();                                        // prologue barrier (tile 0's max is published)

    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool stamp = ST && taps.stamps != nullptr;
    const unsigned long long t_clk0 = stamp ? __builtin_readcyclecounter() : 0;
    const unsigned long long t_rt0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
    for (int p = 0; p < nph; ++p) {
        unsigned long long tp = stamp ? __builtin_readcyclecounter() : 0;
        const bool go = p >= 1 && p <= n;
        if (p >= 2 && p - 2 < n) {          // the matrix waves take the last 4 epilogue units
            const float unscale = taps.inv_tapscale / reinterpret_cast<const float*>(smem + W::SCALE_OFF)[(p - 2) & 3];
            if (CX) dd_ws_epilogue_unit<NKS, true>(P, t_begin + p - 2, WS_VWAVES + mw, lane,
                                                   reinterpret_cast<const float2*>(smem + W::YBUF_OFF + 16), unscale,
                                                   reinterpret_cast<const float2*>(smem + W::TILEW_OFF) + ((p - 2) & 3));
            else
            dd_ws_epilogue_unit<NKS>(P, t_begin + p - 2, WS_VWAVES + mw, lane,
                                     reinterpret_cast<const float2*>(smem + W::YBUF_OFF + 16), unscale);
        }
        if (lane == 0) atomicAdd(ydone, 1);                 // done reading the y-buffer (own epilogue unit)
        const char* abase = smem + ((p - 1) & 1) * W::PLANES_BYTES + aoff;
        v16f cre, cim;
#pragma unroll
        for (int r = 0; r < 16; ++r) { cre[r] = 0.f; cim[r] = 0.f; }
        // software pipeline, distance two k-steps: the six fragments of k-step ks+2 are issued during the
        // six MFMAs of k-step ks, one read per MFMA gap (no read burst between MFMA groups), so a
        // fragment has ~250-380 cycles to arrive; sched_barrier pins the order.
        v8h f[3][6];
#define DD_WS_LOADF(buf, ks)                                                                     \
        {                                                                                        \
            const int off_ = 32 * (ks) + 16 * ((ks) >> 1);                                       \
            f[buf][0] = *reinterpret_cast<const v8h*>(abase + off_);                             \
            f[buf][1] = *reinterpret_cast<const v8h*>(abase + G::PLANE + off_);                  \
            f[buf][2] = *reinterpret_cast<const v8h*>(abase + 2 * G::PLANE + off_);              \
            f[buf][3] = *reinterpret_cast<const v8h*>(abase + 3 * G::PLANE + off_);              \
            f[buf][4] = tb[(ks) * 64];                                                           \
            f[buf][5] = tb[(NKS + (ks)) * 64];                                                   \
        }
        // one k-step: the six MFMAs of buffer `buf`, with one fragment read of the next
        // k-step (buffer nb) issued in each MFMA gap (no read burst between MFMA groups)
#define DD_WS_STEP(buf, nb, ksn, pre)                                                            \
        {                                                                                        \
            const int off_ = 32 * (ksn) + 16 * ((ksn) >> 1);                                     \
            cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[buf][0], f[buf][4], cre, 0, 0, 0);   \
            if (pre) f[nb][0] = *reinterpret_cast<const v8h*>(abase + off_);                     \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[buf][2], f[buf][4], cim, 0, 0, 0);   \
            if (pre) f[nb][4] = tb[(ksn) * 64];                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[buf][1], f[buf][4], cre, 0, 0, 0);   \
            if (pre) f[nb][2] = *reinterpret_cast<const v8h*>(abase + 2 * G::PLANE + off_);      \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[buf][3], f[buf][4], cim, 0, 0, 0);   \
            if (pre) f[nb][1] = *reinterpret_cast<const v8h*>(abase + G::PLANE + off_);          \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[buf][0], f[buf][5], cre, 0, 0, 0);   \
            if (pre) f[nb][3] = *reinterpret_cast<const v8h*>(abase + 3 * G::PLANE + off_);      \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[buf][2], f[buf][5], cim, 0, 0, 0);   \
            if (pre) f[nb][5] = tb[(NKS + (ksn)) * 64];                                          \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }
        DD_STAMP(0)
        int yd_early = 0;
        if (go) {
            DD_WS_LOADF(0, 0)
            DD_WS_LOADF(1, 1)
#pragma unroll
            for (int ks = 0; ks < NKS - 1; ++ks) {
                DD_WS_STEP(ks % 3, (ks + 2) % 3, (ks + 2 < NKS ? ks + 2 : ks), (ks + 2 < NKS))
            }
            yd_early = __hip_atomic_load(ydone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // in flight under the last k-step
            DD_WS_STEP((NKS - 1) % 3, (NKS + 1) % 3, NKS - 1, false)
        }
        DD_STAMP(1)
        // all 16 waves have finished reading the y-buffer of tile p-2 (normally long ago)
        if (yd_early < 16 * (p + 1))
        while (__hip_atomic_load(ydone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 16 * (p + 1)) __builtin_amdgcn_s_sleep(2);
        if (go) {
            // register r of lane (i, h) is output 32 (rowbase(r) + 4h) + i of the strip
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ywr[32 * ((r & 3) + 8 * (r >> 2))] = cre[r];
                ywr[W::YPLANE_F + 32 * ((r & 3) + 8 * (r >> 2))] = cim[r];
            }
        }
        DD_STAMP(2)
        __syncthreads();
        DD_STAMP(3)
    }
    if (stamp && lane == 0) {
        if (mw == 0) {   // in-kernel clock: shader ticks per 100 MHz reference tick over the whole loop
            taps.stamps[((size_t)blockIdx.x * 16 + mw) * 8 + 5] = __builtin_readcyclecounter() - t_clk0;
            taps.stamps[((size_t)blockIdx.x * 16 + mw) * 8 + 6] = __builtin_amdgcn_s_memrealtime() - t_rt0;
            taps.stamps[((size_t)blockIdx.x * 16 + 1) * 8 + 4] = t_rt0;                              // loop start (abs)
            taps.stamps[((size_t)blockIdx.x * 16 + 2) * 8 + 4] = __builtin_amdgcn_s_memrealtime();   // loop end (abs)
        }
        for (int q = 0; q < 4; ++q) taps.stamps[((size_t)blockIdx.x * 16 + mw) * 8 + q] = acc_t[q];
        taps.stamps[((size_t)blockIdx.x * 16 + mw) * 8 + 7] = (unsigned long long)nph | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
    }
}

// U8: the input is raw interleaved uint8 I,Q (2 B/sample) instead of complex64; only the vector waves' loads differ.
// CX: complex-output flavour (launched only without DD_CHAIN_FM): tile phasors handed over through LDS.  CX = false is
// the code as it was before the flavour existed, token for token -- this kernel's speed depends on code generation in
// ways its instruction mix does not explain, so the headline instantiation is kept byte-identical (assembly diff).
template <int NKS, bool U8, bool CX, bool ST = false>
__global__ void __launch_bounds__(WS_THREADS) k_chain_mfma_ws(const DDChainParams P, const DDMfmaTaps taps, int t_first, int t_last, int nwg) {
    using W = WsGeom<NKS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wg = blockIdx.x;
    if (wg >= nwg) {
        // edge tiles ([0, t_first) and [t_last, nblocks)) ride along as trailing workgroups:
        // they are dispatched when the first persistent workgroups retire, i.e. inside the
        // spread of the persistent workgroups' finish times, and cost no launch of their own
        if (threadIdx.x >= MF_THREADS) return;              // 4 waves do the tile (exited waves leave the barrier)
        const int e = wg - nwg;
        const int b = e < t_first ? e : t_last + (e - t_first);
        v8h* tl = reinterpret_cast<v8h*>(smem + W::TAPS_OFF);
        for (int idx = threadIdx.x; idx < 2 * NKS * 64; idx += MF_THREADS) tl[idx] = taps.frag[idx];
        dd_edge_tile_lean<NKS>(P, taps, b, smem, tl);        // (its staging barriers order the tap copy)
        return;
    }
    if (ST && taps.stamps && threadIdx.x == 0) taps.stamps[((size_t)wg * 16) * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    const int nt = t_last - t_first;
    const int t_begin = t_first + (int)(((int64_t)wg * nt) / nwg);
    const int t_end = t_first + (int)(((int64_t)(wg + 1) * nt) / nwg);
    if (t_begin >= t_end) return;
    // tap limb fragments -> LDS once per workgroup (all 1024 threads; 2*NKS*64 entries)
    {
        v8h* tl = reinterpret_cast<v8h*>(smem + W::TAPS_OFF);
        for (int idx = threadIdx.x; idx < 2 * NKS * 64; idx += WS_THREADS) tl[idx] = taps.frag[idx];
        if (threadIdx.x == 0) *reinterpret_cast<int*>(smem + W::YDONE_OFF) = 0;
        if (threadIdx.x < 4) reinterpret_cast<int*>(smem + W::NONUNIT_OFF)[threadIdx.x] = threadIdx.x == 0 ? 1 : 0;   // tile 0: read the true max
    }
    __syncthreads();
    const int nph = ((t_end - t_begin + 2 + 2) / 3) * 3;      // phases, rounded up to the vector loop's unroll of 3
    if (threadIdx.x < 64 * WS_MWAVES) dd_ws_matrix<NKS, CX, ST>(P, taps, smem, t_begin, t_end, nph);
    else dd_ws_vector<NKS, U8, CX, ST>(P, taps, smem, t_begin, t_end, nph);
}

#include "dd_mfma_ab.h"

// ============================================================================
// host side
// ============================================================================
struct DDMfmaState {
    int K;
    int nks;
    v8h* frag;          // device
    float inv_tapscale;
    std::vector<double> taps;
    void* fft;          // overlap-save FFT form of the interior run (dd_fftfir.hip), lazy
    int fft_tried;
    void* cos;          // running-sum form for cosine-series windows of 255 taps (dd_cosfir.hip), lazy
    int cos_tried;
};

static int mfma_nks_for(int K) {
    const int opts[4] = {6, 10, 12, 18};
    for (int i = 0; i < 4; ++i)
        if (K <= 16 * opts[i] - 31) return opts[i];
    return 0;
}

int dd_mfma_supported(int K, int M, int flags) {
    (void)flags;
    return (M == 1 && mfma_nks_for(K) != 0) ? 1 : 0;
}

int dd_mfma_create(void** st, const double* taps, int K) {
    const int nks = mfma_nks_for(K);
    if (!nks) return DD_ERR_UNSUPPORTED;
    const int HALO = 16 * nks - 32;
    // correlation form: y[o] = sum_j' g2[j'] s[o + j'], j' in [0, HALO], element 0 of
    // the window is HALO samples before the output sample
    double mx = 0.0;
    for (int k = 0; k < K; ++k) mx = fmax(mx, fabs(taps[k]));
    int ex = 0;
    if (mx > 0.0) frexp(mx, &ex);                         // mx = f * 2^ex, f in [0.5, 1)
    const double tapscale = ldexp(1.0, -ex);
    std::vector<double> g2(HALO + 1, 0.0);
    for (int k = 0; k < K; ++k) g2[HALO - k] = taps[k] * tapscale;   // tap k multiplies the sample k before the output
    std::vector<_Float16> frag((size_t)2 * nks * 64 * 8);
    for (int ks = 0; ks < nks; ++ks) {
        for (int lane = 0; lane < 64; ++lane) {
            const int j = lane & 31, h = lane >> 5;
            for (int t = 0; t < 8; ++t) {
                const int m = 16 * ks + 8 * h + t;
                const int idx = m - j;
                const double g = (idx >= 0 && idx <= HALO) ? g2[idx] : 0.0;
                const _Float16 gh = (_Float16)g;
                const _Float16 gl = (_Float16)(g - (double)gh);
                frag[((size_t)(0 * nks + ks) * 64 + lane) * 8 + t] = gh;
                frag[((size_t)(1 * nks + ks) * 64 + lane) * 8 + t] = gl;
            }
        }
    }
    DDMfmaState* s = new DDMfmaState();
    s->K = K;
    s->nks = nks;
    s->frag = nullptr;
    s->inv_tapscale = (float)(1.0 / tapscale);
    s->taps.assign(taps, taps + K);
    s->fft = nullptr;
    s->fft_tried = 0;
    s->cos = nullptr;
    s->cos_tried = 0;
    hipError_t e = hipMalloc((void**)&s->frag, frag.size() * sizeof(_Float16));
    if (e == hipSuccess) e = hipMemcpy(s->frag, frag.data(), frag.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (s->frag) (void)hipFree(s->frag);
        delete s;
        dd_set_error("dd_mfma_create: %s", hipGetErrorString(e));
        return DD_ERR_HIP;
    }
    *st = s;
    return DD_OK;
}

void dd_mfma_destroy(void* st) {
    DDMfmaState* s = reinterpret_cast<DDMfmaState*>(st);
    if (!s) return;
    if (s->fft) dd_fft_destroy(s->fft);
    if (s->cos) dd_cos1k_destroy(s->cos);
    (void)hipFree(s->frag);
    delete s;
}

#define DD_STAMP_WGS 1024      // workgroups the DD_STAMPS diagnostic buffer holds

// Which M = 1 kernel runs: by tap class, unless a tool or test has forced one.  The choice is a process-wide word set
// through dd_debug_select_kernel (a debug entry like dd_debug_fill_lds); the environment variable DD_MFMA_KERNEL only
// seeds it, read ONCE when the first chain is launched (no getenv in the launch path).
enum { DD_KSEL_UNREAD = -1, DD_KSEL_AUTO = 0, DD_KSEL_AB = 1, DD_KSEL_WS = 2, DD_KSEL_FFT1K = 3, DD_KSEL_COS1K = 4, DD_KSEL_DECIMP = 5 };
static std::atomic<int> g_kernel_sel{DD_KSEL_UNREAD};
static int kernel_sel_parse(const char* name) {
    if (!name || !*name || strcmp(name, "auto") == 0) return DD_KSEL_AUTO;
    if (strcmp(name, "ab") == 0) return DD_KSEL_AB;
#ifdef DD_WITH_WS
    if (strcmp(name, "ws") == 0) return DD_KSEL_WS;            // round 1's y-buffer kernel: only in a -DDD_WITH_WS build (tools/mkvariant.sh)
#endif
    if (strcmp(name, "fft1k") == 0) return DD_KSEL_FFT1K;
    if (strcmp(name, "cos1k") == 0) return DD_KSEL_COS1K;
    if (strcmp(name, "decimp") == 0) return DD_KSEL_DECIMP;      // M > 1: the tile kernels of rounds 1-4 instead of k_chain_decim_w (M = 1: as "auto")
    return -2;
}
static int kernel_sel() {
    int c = g_kernel_sel.load(std::memory_order_relaxed);
    if (c == DD_KSEL_UNREAD) {
        c = kernel_sel_parse(getenv("DD_MFMA_KERNEL"));
        if (c < 0) c = DD_KSEL_AUTO;
        g_kernel_sel.store(c, std::memory_order_relaxed);
    }
    return c;
}
int dd_kernel_sel_decimp(void) { return kernel_sel() == DD_KSEL_DECIMP ? 1 : 0; }
extern "C" int dd_debug_select_kernel(const char* name) {
    const int c = kernel_sel_parse(name);
    if (c < 0) {
        dd_set_error("dd_debug_select_kernel: unknown kernel '%s' (auto, ab, fft1k, cos1k, decimp; \"ws\" -- k_chain_mfma_ws, round 1 -- is not in the "
                     "product library since round 6: build a variant with -DDD_WITH_WS)", name);
        return DD_ERR_INVALID;
    }
    g_kernel_sel.store(c, std::memory_order_relaxed);
    return DD_OK;
}

template <int NKS>
static int mfma_launch_t(DDMfmaState* st, DDChainParams& P, hipStream_t s, int* kernel_id) {
    using G = MfmaGeom<NKS>;
    const size_t lds = (size_t)MF_LDS_TILE_BYTES(NKS);
    const size_t lds_ws = (size_t)WsGeom<NKS>::LDS_BYTES;
    static DDOncePerDevice attr_set;
    if (attr_set.need()) {
#ifdef DD_WITH_WS
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_ws<NKS, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ws));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_ws<NKS, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ws));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_ws<NKS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ws));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_ws<NKS, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ws));
#endif
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_edge<NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_ab<NKS, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AbGeom<NKS>::LDS_BYTES));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_ab<NKS, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AbGeom<NKS>::LDS_BYTES));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_ab<NKS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AbGeom<NKS>::LDS_BYTES));
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_ab<NKS, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AbGeom<NKS>::LDS_BYTES));
        if (NKS == 18) {                                    // the instantiations with the in-kernel stamps (DD_STAMPS, tools only)
            DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_ab<18, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AbGeom<18>::LDS_BYTES));
#ifdef DD_WITH_WS
            DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_ws<18, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WsGeom<18>::LDS_BYTES));
            DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma_ws<18, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WsGeom<18>::LDS_BYTES));
#endif
        }
        attr_set.mark();
    }
    DDMfmaTaps t;
    t.frag = st->frag;
    t.inv_tapscale = st->inv_tapscale;
    t.stamps = nullptr;
    static unsigned long long* stamp_buf = nullptr;
    // DD_STAMPS=<k>: per-wave stage stamps, printed for the k-th launch (k <= 3 -> the 4th: a cold GPU;
    // a few hundred -> the clock the chip holds under sustained load)
    static const char* stamps_env = DD_TUNE_ENV("DD_STAMPS");
    static const int stamps_at = stamps_env ? (atoi(stamps_env) > 3 ? atoi(stamps_env) : 3) : -1;
    static int launches = 0;
    const bool want_stamps_env = stamps_env != nullptr && launches++ == stamps_at;
    const int ksel = kernel_sel();
    const bool ws_env = ksel == DD_KSEL_WS;
    // only these have an instantiation with the stamps compiled in: 255-tap class, complex64 input; FM output, or k_chain_mfma_ws
    const bool want_stamps = want_stamps_env && NKS == 18 && !(P.flags & DD_CHAIN_U8_INPUT) && ((P.flags & DD_CHAIN_FM) || ws_env);
    if (want_stamps) {
        if (!stamp_buf) DD_HIP_CHECK(hipMalloc((void**)&stamp_buf, DD_STAMP_WGS * 16 * 8 * 8));
        DD_HIP_CHECK(hipMemsetAsync(stamp_buf, 0, DD_STAMP_WGS * 16 * 8 * 8, s));
        t.stamps = stamp_buf;
    }
    // interior tiles: whole span inside the chunk, all outputs emitted, aligned complex64
    int t_first = 1, t_last = 1;
    const bool u8in = (P.flags & DD_CHAIN_U8_INPUT) != 0;
    const bool aligned = (reinterpret_cast<uintptr_t>(P.in) & (u8in ? 3 : 15)) == 0;
    if (aligned && P.nblocks > 2) {
        // tile b: ns = b*ADV - 32 - HALO >= 0 ; ns + SPAN <= L ; b*ADV - 32 + T <= Ld
        int64_t lo = (32 + G::HALO + MF_ADV - 1) / MF_ADV;
        if (lo < 1) lo = 1;
        int64_t hi1 = (P.L - G::SPAN + 32 + G::HALO) / MF_ADV;         // last b with ns + SPAN <= L
        int64_t hi2 = (P.Ld - MF_T + 32) / MF_ADV;                     // last b with P0 + T <= Ld
        int64_t hi = hi1 < hi2 ? hi1 : hi2;
        if (hi > P.nblocks - 2) hi = P.nblocks - 2;
        if (hi >= lo) { t_first = (int)lo; t_last = (int)hi + 1; }
    }
    const int n_int = t_last - t_first;
    // dd_debug_select_kernel (tools and tests switch kernels inside one process): "ab" / "ws" force the MFMA kernels,
    // "fft1k" the overlap-save FFT kernel wherever it applies; "auto" = by tap class
    const bool force_fft1k = ksel == DD_KSEL_FFT1K;
    const bool force_mfma = ksel == DD_KSEL_AB || ksel == DD_KSEL_WS;
    // The FFT kernel's time does not depend on the tap count (0.221 ms per 2^26 samples, 0.207 from raw u8); the MFMA
    // kernel's does: 0.227 ms in the 162..257-tap class, 0.20 below.  FM output only.
    const bool fft_class = NKS == 18;
    // 255 taps of the form a0 + a1 cos(2 pi k / 254) (filters.hamming), FM or complex64 output: the running-sum kernel (dd_cosfir.hip),
    // a third of the overlap-save form's arithmetic.  "cos1k" forces it wherever it applies (it never applies to other taps),
    // "fft1k" / "ab" / "ws" keep it off.
    // (DD_CHAIN_TIGHT: the caller asks for the transform kernel's stop-band error bound)
    if ((ksel == DD_KSEL_AUTO || ksel == DD_KSEL_COS1K) && st->K == 255 && !(P.flags & DD_CHAIN_TIGHT)) {
        if (!st->cos && !st->cos_tried) {
            st->cos_tried = 1;
            if (!dd_cos1k_supported(st->taps.data(), st->K, 1, P.flags) || dd_cos1k_create(&st->cos, st->taps.data(), st->K) != DD_OK) st->cos = nullptr;
        }
        if (st->cos) {
            int rc = dd_cos1k_launch(st->cos, P, s);
            if (rc != DD_OK) return rc;
            if (kernel_id) *kernel_id = DD_KERNEL_COS_RS;
            return DD_OK;
        }
    }
    // (any input alignment: its block grid follows the alignment of `out`, dd_fftfir.hip DDFft1kTabs::base)
    if ((force_fft1k || (fft_class && !force_mfma)) && dd_fft1k_supported(st->K, 1, P.flags)) {
        if (!st->fft && !st->fft_tried) {
            st->fft_tried = 1;
            if (dd_fft_create(&st->fft, st->taps.data(), st->K) != DD_OK) st->fft = nullptr;
        }
        if (st->fft) {
            int rc = dd_fft1k_launch(st->fft, P, s);
            if (rc != DD_OK) return rc;
            if (kernel_id) *kernel_id = DD_KERNEL_FFT_OS;
            return DD_OK;
        }
    }
    if (n_int > 0) {
        // one 16-wave workgroup per CU (LDS bound): persistent workgroups for the interior run plus
        // one workgroup per edge tile (both sides of the run), all resident at once -- the edge
        // tiles cost neither a launch of their own nor a tail after the persistent loop
        const int n_edge = t_first + (P.nblocks - t_last);
        const int ncu = dd_cu_count() < DD_STAMP_WGS ? dd_cu_count() : DD_STAMP_WGS;
        const int cus = n_edge < ncu / 2 ? ncu - n_edge : ncu / 2;
        int grid = (n_int + 3) / 4 < cus ? (n_int + 3) / 4 : cus;
        const bool cx = !(P.flags & DD_CHAIN_FM);
        const dim3 g(grid + n_edge), b(WS_THREADS);
        // the two-matrix-set kernel (dd_mfma_ab.h).  (k_chain_mfma_ws, round 1's y-buffer kernel, was reachable through
        // dd_debug_select_kernel("ws") only: since round 6 it is compiled in -DDD_WITH_WS builds alone -- VERDICT r5 item 8)
#ifdef DD_WITH_WS
        const bool use_ab = !ws_env;
#else
        const bool use_ab = true;
        (void)ws_env; (void)lds_ws;
#endif
        const size_t lds_ab = (size_t)AbGeom<NKS>::LDS_BYTES;
        if (use_ab) {
            if (u8in && cx) hipLaunchKernelGGL((k_chain_mfma_ab<NKS, true, true>), g, b, lds_ab, s, P, t, t_first, t_last, grid);
            else if (u8in) hipLaunchKernelGGL((k_chain_mfma_ab<NKS, true, false>), g, b, lds_ab, s, P, t, t_first, t_last, grid);
            else if (cx) hipLaunchKernelGGL((k_chain_mfma_ab<NKS, false, true>), g, b, lds_ab, s, P, t, t_first, t_last, grid);
            else if (NKS == 18 && t.stamps)                  // DD_STAMPS (tools): the instantiation with the in-kernel stamps compiled in
                hipLaunchKernelGGL((k_chain_mfma_ab<18, false, false, true>), g, b, (size_t)AbGeom<18>::LDS_BYTES, s, P, t, t_first, t_last, grid);
            else hipLaunchKernelGGL((k_chain_mfma_ab<NKS, false, false>), g, b, lds_ab, s, P, t, t_first, t_last, grid);
        }
#ifdef DD_WITH_WS
        else
        if (u8in && cx) hipLaunchKernelGGL((k_chain_mfma_ws<NKS, true, true>), g, b, lds_ws, s, P, t, t_first, t_last, grid);
        else if (u8in) hipLaunchKernelGGL((k_chain_mfma_ws<NKS, true, false>), g, b, lds_ws, s, P, t, t_first, t_last, grid);
        else if (NKS == 18 && t.stamps && cx) hipLaunchKernelGGL((k_chain_mfma_ws<18, false, true, true>), g, b, (size_t)WsGeom<18>::LDS_BYTES, s, P, t, t_first, t_last, grid);
        else if (NKS == 18 && t.stamps) hipLaunchKernelGGL((k_chain_mfma_ws<18, false, false, true>), g, b, (size_t)WsGeom<18>::LDS_BYTES, s, P, t, t_first, t_last, grid);
        else if (cx) hipLaunchKernelGGL((k_chain_mfma_ws<NKS, false, true>), g, b, lds_ws, s, P, t, t_first, t_last, grid);
        else hipLaunchKernelGGL((k_chain_mfma_ws<NKS, false, false>), g, b, lds_ws, s, P, t, t_first, t_last, grid);
#endif
        DD_LAUNCH_CHECK();
        if (kernel_id) *kernel_id = use_ab ? DD_KERNEL_MFMA_AB : DD_KERNEL_MFMA_WS;
        if (want_stamps) {
            static int printed = 0;
            std::vector<unsigned long long> hb(DD_STAMP_WGS * 16 * 8);
            DD_HIP_CHECK(hipMemcpyAsync(hb.data(), stamp_buf, hb.size() * 8, hipMemcpyDeviceToHost, s));
            DD_HIP_CHECK(hipStreamSynchronize(s));
            if (printed++ == 0) {
                const char* vn_ws[6] = {"V:issue loads", "V:epilogue", "V:convert", "V:next tile max", "V:barrier wait", "-"};
                const char* mn_ws[4] = {"M:epilogue unit", "M:108 mfma", "M:y wait + write", "M:barrier wait"};
                const char* vn_ab[6] = {"V:issue loads", "V:convert", "V:next tile range", "V:barrier wait", "-", "-"};
                const char* mn_ab[4] = {"M:108 mfma + publish (per pair of phases)", "M:discriminator", "M:barrier waits", "-"};
                const char** vn = use_ab ? vn_ab : vn_ws;
                const char** mn = use_ab ? mn_ab : mn_ws;
                const int n_m = use_ab ? 8 : 4;              // matrix waves per workgroup
                const double nphd = (double)(hb[7] & 0xffffffffull);
                {   // wave placement: SIMD id (HW_ID bits 5:4) of each of the 16 waves, histogram over workgroups
                    int bad = 0;
                    for (int w = 0; w < grid; ++w) {
                        int cnt[4] = {0, 0, 0, 0};
                        for (int wv = 0; wv < 4; ++wv) cnt[(hb[((size_t)w * 16 + wv) * 8 + 7] >> 36) & 3]++;
                        if (cnt[0] != 1 || cnt[1] != 1 || cnt[2] != 1 || cnt[3] != 1) ++bad;
                    }
                    fprintf(stderr, "[stamps] workgroups whose 4 matrix waves do NOT sit on 4 different SIMDs: %d of %d; wg0 simd ids:", bad, grid);
                    for (int wv = 0; wv < 16; ++wv) fprintf(stderr, " %d", (int)((hb[((size_t)wv) * 8 + 7] >> 36) & 3));
                    fprintf(stderr, "\n");
                }
                fprintf(stderr, "[stamps] in-kernel clock: %.3f GHz (%llu shader ticks / %llu ref ticks @100 MHz), %d phases\n",
                        (double)hb[5] / (double)hb[6] * 0.1, hb[5], hb[6], (int)nphd);
                {
                    unsigned long long e_min = ~0ull, e_max = 0, ls_min = ~0ull, ls_max = 0, le_min = ~0ull, le_max = 0;
                    for (int w = 0; w < grid; ++w) {
                        const unsigned long long e = hb[((size_t)w * 16) * 8 + 4], a = hb[((size_t)w * 16 + 1) * 8 + 4], b2 = hb[((size_t)w * 16 + 2) * 8 + 4];
                        e_min = e < e_min ? e : e_min; e_max = e > e_max ? e : e_max;
                        ls_min = a < ls_min ? a : ls_min; ls_max = a > ls_max ? a : ls_max;
                        le_min = b2 < le_min ? b2 : le_min; le_max = b2 > le_max ? b2 : le_max;
                    }
                    fprintf(stderr, "[stamps] workgroup timeline (us, relative to the first kernel entry): entry %.1f..%.1f, loop start %.1f..%.1f, loop end %.1f..%.1f\n",
                            0.0, (e_max - e_min) * 0.01, (ls_min - e_min) * 0.01, (ls_max - e_min) * 0.01, (le_min - e_min) * 0.01, (le_max - e_min) * 0.01);
                }
                for (int wv = 0; wv < 16; ++wv) {
                    const int nq = use_ab ? (wv < n_m ? 3 : 4) : (wv < 4 ? 4 : 5);
                    fprintf(stderr, "[stamps] wave %2d:", wv);
                    for (int q = 0; q < nq; ++q) {
                        double sum = 0;
                        for (int w = 0; w < grid; ++w) sum += (double)hb[((size_t)w * 16 + wv) * 8 + q];
                        fprintf(stderr, " %s=%.0f", wv < n_m ? mn[q] : vn[q], sum / grid / nphd);
                    }
                    fprintf(stderr, "\n");
                }
            }
        }
    } else {
        hipLaunchKernelGGL(k_chain_mfma_edge<NKS>, dim3(P.nblocks), dim3(MF_THREADS), lds, s, P, t, P.nblocks, P.nblocks);
        DD_LAUNCH_CHECK();
        if (kernel_id) *kernel_id = DD_KERNEL_MFMA_TILES;
    }
    return DD_OK;
}

int dd_mfma_launch(void* stv, const DDChainParams& Pin, hipStream_t s, int* kernel_id) {
    DDMfmaState* st = reinterpret_cast<DDMfmaState*>(stv);
    DDChainParams P = Pin;
    P.T = MF_T;
    P.nblocks = (int)((P.Ld + MF_ADV - 1) / MF_ADV);
    if (P.nblocks < 1) P.nblocks = 1;
    switch (st->nks) {
        case 6: return mfma_launch_t<6>(st, P, s, kernel_id);
        case 10: return mfma_launch_t<10>(st, P, s, kernel_id);
        case 12: return mfma_launch_t<12>(st, P, s, kernel_id);
        case 18: return mfma_launch_t<18>(st, P, s, kernel_id);
    }
    return DD_ERR_UNSUPPORTED;
}

// dd_code_warmup (dd_runtime.hip): the runtime loads a translation unit's code object when one of its kernels is first named
int dd_code_touch_mfma(void) {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, (const void*)k_chain_mfma_edge<16>) == hipSuccess ? DD_OK : DD_ERR_HIP;
}
