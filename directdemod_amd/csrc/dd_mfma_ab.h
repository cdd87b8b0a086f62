// k_chain_mfma_ab -- the FM-output interior kernel with TWO alternating sets of matrix waves
// (included by dd_mfma.hip after k_chain_mfma_ws, whose helpers and LDS plane image it shares).
//
// Where k_chain_mfma_ws loses its time (PMC, profiles/r01_mfma_ws_pmc_sq.txt: matrix pipe 56 % busy): inside a
// phase the matrix wave of a SIMD first runs its own discriminator unit, then the 108 MFMAs, then waits for the
// y-buffer, copies its accumulators into it and meets the barrier -- 1460 of 5110 cycles with the pipe idle -- and
// the FIR outputs take a round trip through LDS (32 KB written and read per tile) on their way to the discriminator.
//
// Here a SIMD hosts two matrix waves, A and B, that alternate tiles:
//
//     phase q      A: 108 MFMAs of tile q-1            B: discriminator of tile q-2, straight from ITS accumulators
//     phase q+1    A: discriminator of tile q-1        B: 108 MFMAs of tile q
//
// so the matrix pipe meets a fresh MFMA stream right after every barrier, and the wave that has just finished a
// tile keeps its 32 accumulator registers and turns them into angles while the other wave computes: no y-buffer,
// no hand-over counter, no LDS traffic for the outputs.  In the accumulator layout a lane holds column j of 16
// rows, so y[n-1] is the same register one lane to the left (DPP row_shr:1); only the first lane of each 16-lane
// row needs a value DPP cannot deliver, which lanes 15/31/47/63 leave in a 2 KB LDS table at the end of the MFMA
// phase (column 0 takes the previous row's column 31; the last output of a strip goes to its right-hand neighbour).  The 8 remaining waves (two per SIMD) do what the 12 vector
// waves of k_chain_mfma_ws do minus the discriminator: tile loads two phases ahead, NCO rotation + f16 limb split
// into the plane buffer of the next tile, and the next tile's range check.  Registers: 128 per wave either way
// (16 waves); LDS: two plane buffers + tap fragments (124 KB).
#pragma once

#define AB_VWAVES 8
#define AB_VTHREADS (64 * AB_VWAVES)
#define AB_REG_ROWS 16        // accumulator registers (of 16) whose outputs the owning matrix wave turns into angles itself (with 8
                              // the other half go through LDS to the vector waves: measured slower, 0.2393 ms against 0.2340)
#define AB_RED_ENTRIES 12     // range-check slots per tile: 8 vector waves, the halo step (entry 8), padding to 16 bytes
#ifndef DD_AB_EPI_PRIO
#define DD_AB_EPI_PRIO 0
#endif
#ifndef DD_AB_VEC_PRIO
#define DD_AB_VEC_PRIO 0
#endif
// The round-2 experiments that were built, parity-tested, measured and found no faster (boundary table
// deferred, first k-step prefetched across the barrier, ratio-form discriminator, grouped fragment reads, the halo step
// on a vector wave) are no longer switches of this header: tools/variants/mfma_ab_switches.patch puts them back.
// timing ablations (tools/mkvariant.sh ... -DDD_AB_NO_xxx; results are wrong by construction, never shipped):
//   DD_AB_NO_EPI      matrix waves skip the discriminator      DD_AB_NO_MFMA   matrix waves skip the MFMAs
//   DD_AB_NO_CONVERT  vector waves skip the rotation / split   DD_AB_NO_LOAD   vector waves skip the tile loads

template <int NKS>
struct AbGeom {
    using G = MfmaGeom<NKS>;
    static constexpr int PLANES_BYTES = 4 * G::PLANE;
    static constexpr int TAPS_OFF = 2 * PLANES_BYTES;
    static constexpr int TAPS_BYTES = 2 * NKS * 64 * 16;
    static constexpr int RED_OFF = TAPS_OFF + TAPS_BYTES;                // [2][AB_RED_ENTRIES] float: 8 vector waves + the halo step
    static constexpr int NONUNIT_OFF = RED_OFF + 2 * AB_RED_ENTRIES * 4; // [4] int
    // left-hand neighbours that DPP row_shr:1 cannot deliver (the first lane of each 16-lane row), per matrix set, float2:
    //   X0[4 waves][16] (+1: the slot the last strip's last output falls into)  for lanes 0   (column 0, rows of half 0)
    //   XA[4][16] for lanes 16 (= lane 15, same register)    X1[4][16] for lanes 32 (column 0, half 1)    XB[4][16] for lanes 48 (= lane 47)
    static constexpr int X0_ENTRIES = 4 * 16 + 1;
    static constexpr int BCOL_OFF = NONUNIT_OFF + 16;
    static constexpr int BCOL_SET_BYTES = (X0_ENTRIES + 3 * 4 * 16 + 1) * 8;
    // second half of every strip's FIR outputs (registers 8..15 = outputs 512..1023), per matrix set, for the vector
    // waves' share of the discriminator: planar [re | im][4 strips][4 floats of slack (the last = output 511) + 512]
    static constexpr int YH_STRIDE = 4 + 512;
    static constexpr int YH_OFF = (BCOL_OFF + 2 * BCOL_SET_BYTES + 15) & ~15;
    static constexpr int YH_SET_BYTES = 2 * 4 * YH_STRIDE * 4;
    static constexpr int YHCNT_OFF = YH_OFF + (AB_REG_ROWS < 16 ? 2 * YH_SET_BYTES : 0);   // [2] int: y-half buffers written (x4 waves), per set
    static constexpr int CONVCNT_OFF = YHCNT_OFF + 8;                    // [2] int: waves that have finished converting a tile into plane buffer 0 / 1
    static constexpr int WKX_OFF = (YHCNT_OFF + 16 + 15) & ~15;          // [2 sets][64 lanes][4] float2: the halo step's tile-relative phasors (kept out of the matrix waves' registers)
    static constexpr int SCALE_OFF = WKX_OFF + 2 * 64 * 32;              // [4] float: the power-of-two scale of tile p (complex output undoes it)
    static constexpr int LDS_BYTES = SCALE_OFF + 16;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(MF_LDS_TILE_BYTES(NKS) <= TAPS_OFF, "the edge tile's image must not reach the tap fragments");
};

// ---- vector side: "quads".  A lane owns FOUR consecutive samples of a step (two 16-byte loads, 32 contiguous bytes):
// each f16 plane then receives 8 bytes per lane and step (ds_write_b64) instead of two 4-byte stores -- the LDS store
// path, not the arithmetic, was what the conversion waited for (ablation: without its ds_write_b32 the conversion
// phase took 1600 cycles instead of 3600).  A tile's span is 1088 quads (255 taps): the 8 vector waves take
// 2 steps x 512 quads = the last 4096 samples; the first 64 quads (the 256-sample halo) are ONE more wave-step, which
// as a third step of one vector wave made that wave the slowest of the workgroup -- it goes to the matrix wave
// (strip 0) that is in its discriminator phase, which has the time.
#define AB_VSTEPS 2

template <int NKS>
struct AbQ {
    using G = MfmaGeom<NKS>;
    static constexpr int NQUAD = G::SPAN / 4;                         // 1088
    static constexpr int XQUADS = NQUAD - AB_VSTEPS * AB_VTHREADS;    // the halo step: 64 quads for 255 taps (fewer for the shorter tap classes)
    static_assert(XQUADS >= 1 && XQUADS <= 64, "the halo must fit one wave-step of quads");
};

struct AbRaw { float4 a, b; };                                         // samples 4q, 4q+1 | 4q+2, 4q+3

// quad q (0 <= q < NQUAD, wave-uniform base) of tile b
template <int NKS, bool U8>
__device__ __forceinline__ AbRaw dd_ab_load_quad(const DDChainParams& P, int b, int q) {
    using G = MfmaGeom<NKS>;
    const int64_t ns = (int64_t)b * MF_ADV - 32 - G::HALO;
    AbRaw r;
    if (U8) {
        const char* base = reinterpret_cast<const char*>(P.in) + 2 * ns;                            // wave-uniform
        const uint2 d = *reinterpret_cast<const uint2*>(base + 8u * (unsigned)q);                  // 4 u8 pairs (source.py:117-118)
        r.a = make_float4((float)(d.x & 0xff) - 127.5f, (float)((d.x >> 8) & 0xff) - 127.5f,
                          (float)((d.x >> 16) & 0xff) - 127.5f, (float)(d.x >> 24) - 127.5f);
        r.b = make_float4((float)(d.y & 0xff) - 127.5f, (float)((d.y >> 8) & 0xff) - 127.5f,
                          (float)((d.y >> 16) & 0xff) - 127.5f, (float)(d.y >> 24) - 127.5f);
        return r;
    }
    const char* base = reinterpret_cast<const char*>(reinterpret_cast<const float2*>(P.in) + ns);   // wave-uniform
#ifndef DD_AB_NO_SADDR
    // told to the compiler in so many words (both halves through readfirstlane, and back into the GLOBAL address space:
    // a pointer rebuilt from an integer is a generic pointer, its loads are flat loads and a flat load's wait is
    // vmcnt(0)): the loads then take the tile's base from scalar registers and a 32-bit lane offset -- no 64-bit
    // vector address arithmetic per tile and step
    typedef const __attribute__((address_space(1))) char* gptr_t;
    const uint64_t bu = reinterpret_cast<uint64_t>(base);
    const gptr_t gb = (gptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bu >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bu));
    typedef float ab_v4f __attribute__((ext_vector_type(4)));
    const ab_v4f va = *(const __attribute__((address_space(1))) ab_v4f*)(gb + 32u * (unsigned)q);
    const ab_v4f vb = *(const __attribute__((address_space(1))) ab_v4f*)(gb + 32u * (unsigned)q + 16u);
    r.a = make_float4(va.x, va.y, va.z, va.w);
    r.b = make_float4(vb.x, vb.y, vb.z, vb.w);
#else
    r.a = *reinterpret_cast<const float4*>(base + 32u * (unsigned)q);
    r.b = *reinterpret_cast<const float4*>(base + 32u * (unsigned)q + 16u);
#endif
    return r;
}

// f16 limb split of two values: hi = RNE_f16(x) packed (v_cvt_pk_f16_f32), unpacked again (v_cvt_f32_f16 on either
// half), lo = RNE_f16(x - hi): 6 plain vector instructions.  Left to the compiler the split of a sample pair is 16-17
// instructions, six of them v_fma_mixlo/mixhi_f16 fused with the rotation -- and beside a busy matrix pipe a mix op
// costs 8.7 cycles of the SIMD's issue against 4.6 for a plain conversion (tools/ubench/valu_beside_mfma.hip).
__device__ __forceinline__ void dd_ab_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    float t0, t1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    asm("v_cvt_f32_f16_e32 %0, %1" : "=v"(t0) : "v"(hi));
    asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(t1) : "v"(hi));
    t0 = x0 - t0;
    t1 = x1 - t1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(t0), "v"(t1));
}

// rotate (tile-relative NCO phase, see dd_ws_convert), split into f16 limbs, write 8 bytes into each of the four planes
template <int NKS, bool UNIT_SCALE>
__device__ __forceinline__ void dd_ab_convert_quad(const AbRaw& raw, char* planes, int q, const float2 (&wk)[4], float scale) {
    using G = MfmaGeom<NKS>;
    float2 w0 = wk[0], w1 = wk[1], w2 = wk[2], w3 = wk[3];
    if (!UNIT_SCALE) {
        w0 = make_float2(w0.x * scale, w0.y * scale); w1 = make_float2(w1.x * scale, w1.y * scale);
        w2 = make_float2(w2.x * scale, w2.y * scale); w3 = make_float2(w3.x * scale, w3.y * scale);
    }
    const float2 x0 = dd_cmul(make_float2(raw.a.x, raw.a.y), w0), x1 = dd_cmul(make_float2(raw.a.z, raw.a.w), w1);
    const float2 x2 = dd_cmul(make_float2(raw.b.x, raw.b.y), w2), x3 = dd_cmul(make_float2(raw.b.z, raw.b.w), w3);
    uint2 rh, rl, ih, il;
    dd_ab_split2(x0.x, x1.x, rh.x, rl.x);
    dd_ab_split2(x2.x, x3.x, rh.y, rl.y);
    dd_ab_split2(x0.y, x1.y, ih.x, il.x);
    dd_ab_split2(x2.y, x3.y, ih.y, il.y);
    const int e = 4 * q;
    const int off = 2 * e + 16 * (e >> 5);                 // 8-byte aligned: a quad never straddles a 32-sample pad
    *reinterpret_cast<uint2*>(planes + off) = rh;
    *reinterpret_cast<uint2*>(planes + G::PLANE + off) = rl;
    *reinterpret_cast<uint2*>(planes + 2 * G::PLANE + off) = ih;
    *reinterpret_cast<uint2*>(planes + 3 * G::PLANE + off) = il;
}

__device__ __forceinline__ float dd_ab_absmax(const AbRaw& r, float m) {
    m = fmaxf(fmaxf(m, fabsf(r.a.x)), fmaxf(fabsf(r.a.y), fmaxf(fabsf(r.a.z), fabsf(r.a.w))));
    return fmaxf(fmaxf(m, fabsf(r.b.x)), fmaxf(fabsf(r.b.y), fmaxf(fabsf(r.b.z), fabsf(r.b.w))));
}

// "does the tile fit the f16 limbs unscaled" -- this wave's share of the answer for the tile in slot `slot`:
// red entry `ent` = 1.0 (any value of the unit range) when all of the wave's samples lie below 32768 and at least one
// reaches 0.25, else the wave's true maximum together with the tile's non-unit flag (see dd_ws_vphase)
__device__ __forceinline__ void dd_ab_publish_range(float m, char* smem, int red_off, int nonunit_off, int slot, int ent, int lane) {
    const bool hi_any = __builtin_amdgcn_ballot_w64(!(m < 32768.0f)) != 0;
    const bool lo_any = __builtin_amdgcn_ballot_w64(m >= 0.25f) != 0;
    if (hi_any || !lo_any) {
        m = dd_wave_max(m);
        if (lane == 63) atomicOr(reinterpret_cast<int*>(smem + nonunit_off) + (slot & 3), 1);
    } else m = 1.0f;
    if (lane == 63) reinterpret_cast<float*>(smem + red_off)[(slot & 1) * AB_RED_ENTRIES + ent] = m;
}

// the power-of-two scale of tile p (1 in the common case: no wave raised the tile's non-unit flag)
__device__ __forceinline__ float dd_ab_tile_scale(const char* smem, int red_off, int flag, int p, bool& unit) {
    const float* red = reinterpret_cast<const float*>(smem + red_off) + (p & 1) * AB_RED_ENTRIES;
    float m = 1.0f;
    if (__builtin_amdgcn_readfirstlane(flag) != 0) {
        m = red[0];
#pragma unroll
        for (int k = 1; k < AB_RED_ENTRIES; ++k) m = fmaxf(m, red[k]);
    }
    unit = (m >= 0.25f) && (m < 32768.0f);                 // the f16 limbs hold the tile unscaled (see dd_ws_vphase)
    return unit ? 1.0f : dd_pow2_scale_for(m);
}

// "this wave's share of tile t is in the plane buffer": one LDS add per wave, issued behind its plane stores (the LDS
// executes a wave's operations in order, so whoever sees the count sees the stores).  The matrix set that computes
// tile t next phase reads the count at the end of its discriminator phase; with all AB_CONV_WAVES shares in, it
// requests its first two k-steps of fragments BEFORE the barrier and its MFMAs start the moment the barrier opens.
#define AB_CONV_WAVES (AB_VWAVES + 1)
__device__ __forceinline__ void dd_ab_conv_done(char* smem, int cnt_off, int t, int lane) {
}

// one 256-output unit of the vector waves' share: outputs 512 + 256 u + 4 lane + {0..3} of strip s, tile b
struct AbUnit { float4 r4, i4; float2 ym; };
template <int NKS>
__device__ __forceinline__ AbUnit dd_ab_unit_read(const float* yh_set, int vw, int lane) {
    using A = AbGeom<NKS>;
    const float* yre = yh_set + (vw >> 1) * A::YH_STRIDE + 4 + 256 * (vw & 1) + 4 * lane;
    const float* yim = yre + 4 * A::YH_STRIDE;
    AbUnit d;
    d.r4 = *reinterpret_cast<const float4*>(yre);
    d.i4 = *reinterpret_cast<const float4*>(yim);
    d.ym = make_float2(yre[-1], yim[-1]);
    return d;
}
__device__ __forceinline__ void dd_ab_unit_store(const DDChainParams& P, int b, int vw, int lane, const AbUnit& d) {
    const float4 r4 = d.r4, i4 = d.i4;
    const float2 ym = d.ym;
    // z_k = y_k conj(y_{k-1})
    const float re0 = fmaf(r4.x, ym.x, i4.x * ym.y), im0 = fmaf(i4.x, ym.x, -r4.x * ym.y);
    const float re1 = fmaf(r4.y, r4.x, i4.y * i4.x), im1 = fmaf(i4.y, r4.x, -r4.y * i4.x);
    const float re2 = fmaf(r4.z, r4.y, i4.z * i4.y), im2 = fmaf(i4.z, r4.y, -r4.z * i4.y);
    const float re3 = fmaf(r4.w, r4.z, i4.w * i4.z), im3 = fmaf(i4.w, r4.z, -r4.w * i4.z);
    const float mn = fminf(fminf(fmaf(0.41421354f, re0, -fabsf(im0)), fmaf(0.41421354f, re1, -fabsf(im1))),
                           fminf(fmaf(0.41421354f, re2, -fabsf(im2)), fmaf(0.41421354f, re3, -fabsf(im3))));
    float a0, a1, a2, a3;
    if (__builtin_amdgcn_ballot_w64(!(mn > 0.f)) == 0) {
        a0 = dd_atan_small(im0, re0); a1 = dd_atan_small(im1, re1);
        a2 = dd_atan_small(im2, re2); a3 = dd_atan_small(im3, re3);
    } else {
        a0 = dd_fast_atan2(im0, re0); a1 = dd_fast_atan2(im1, re1);
        a2 = dd_fast_atan2(im2, re2); a3 = dd_fast_atan2(im3, re3);
    }
    const int64_t p = (int64_t)b * MF_ADV - 32 + (int64_t)(vw >> 1) * MF_STRIP + 512 + 256 * (vw & 1) + 4 * lane;
    float* out = reinterpret_cast<float*>(P.out) + (p - P.s);
    if (P.s == 0) {
        *reinterpret_cast<float4*>(out) = make_float4(a0, a1, a2, a3);
    } else {                                               // first chunk of a stream: outputs shifted by one
        out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
    }
}

#define DD_AB_STAMP(i) if (stamp) { const unsigned long long tn = __builtin_readcyclecounter(); acc_t[i] += tn - tp; tp = tn; }

// one vector-wave phase p: loads of tile p+2 | conversion of tile p | range check of tile p+1 | barrier
template <int NKS, bool U8, bool CX>
__device__ __forceinline__ void dd_ab_vphase(const DDChainParams& P, char* smem, int t_begin, int n, int p,
                                             AbRaw (&rcur)[AB_VSTEPS], AbRaw (&rnext)[AB_VSTEPS], AbRaw (&rld)[AB_VSTEPS],
                                             const float2 (&wk)[AB_VSTEPS][4],
                                             int vt, int vw, int lane, bool stamp, unsigned long long (&acc_t)[8],
                                             AbRaw& xraw, const float2 (&wkx)[4]) {
    using A = AbGeom<NKS>;
    using Q = AbQ<NKS>;
    unsigned long long tp = stamp ? __builtin_readcyclecounter() : 0;
    // the tile's non-unit flag is requested first: its LDS round trip (hundreds of cycles behind the matrix waves'
    // fragment stream) passes under the address arithmetic and the issue of the tile loads
    const int nu_flag = reinterpret_cast<const int*>(smem + A::NONUNIT_OFF)[p & 3];
#ifndef DD_AB_NO_LOAD
    {
        const int bl = t_begin + (p + 2 < n ? p + 2 : n - 1);   // past the end: harmless re-read, never used
#pragma unroll
        for (int st = 0; st < AB_VSTEPS; ++st) rld[st] = dd_ab_load_quad<NKS, U8>(P, bl, Q::XQUADS + vt + AB_VTHREADS * st);
    }
#endif
    // this wave's unit of the discriminator of tile p-2 (its MFMAs ran in phase p-1, by set (p-1) & 1): the outputs
    // are requested from LDS now and turned into angles behind the conversion
    const bool do_unit = AB_REG_ROWS < 16 && p >= 2 && p - 2 < n;
    AbUnit ud;
    ud.r4 = ud.i4 = make_float4(1.f, 0.f, 0.f, 0.f);
    ud.ym = make_float2(1.f, 0.f);
#ifndef DD_AB_NO_EPI
    if (do_unit) ud = dd_ab_unit_read<NKS>(reinterpret_cast<const float*>(smem + A::YH_OFF + ((p - 1) & 1) * A::YH_SET_BYTES), vw, lane);
#endif
    DD_AB_STAMP(0)
#ifndef DD_AB_NO_CONVERT
    if (p < n) {                                            // convert tile p (range published in phase p-1)
        bool unit;
        const float scale = dd_ab_tile_scale(smem, A::RED_OFF, nu_flag, p, unit);
        if (vt == 0) reinterpret_cast<int*>(smem + A::NONUNIT_OFF)[(p + 2) & 3] = 0;   // re-arm the slot tile p+2's producers raise in phase p+1
        if (CX && vt == 0) reinterpret_cast<float*>(smem + A::SCALE_OFF)[p & 3] = scale;
        char* planes = smem + (p & 1) * A::PLANES_BYTES;
#pragma unroll
        for (int st = 0; st < AB_VSTEPS; ++st) {
            if (unit) dd_ab_convert_quad<NKS, true>(rcur[st], planes, Q::XQUADS + vt + AB_VTHREADS * st, wk[st], scale);
            else dd_ab_convert_quad<NKS, false>(rcur[st], planes, Q::XQUADS + vt + AB_VTHREADS * st, wk[st], scale);
        }
        dd_ab_conv_done(smem, A::CONVCNT_OFF, p, lane);
    }
#endif
    DD_AB_STAMP(1)
#ifndef DD_AB_NO_EPI
    if (do_unit) dd_ab_unit_store(P, t_begin + p - 2, vw, lane, ud);
#endif
    if (p + 1 < n) {                                        // does tile p+1 fit the f16 limbs unscaled?
        float m = 0.f;
#pragma unroll
        for (int st = 0; st < AB_VSTEPS; ++st) m = dd_ab_absmax(rnext[st], m);
        dd_ab_publish_range(m, smem, A::RED_OFF, A::NONUNIT_OFF, p + 1, vw, lane);
    }
    DD_AB_STAMP(2)
    __syncthreads();
    DD_AB_STAMP(3)
}

template <int NKS, bool U8, bool CX, bool ST>
__device__ __forceinline__ void dd_ab_vector(const DDChainParams& P, const DDMfmaTaps& taps, char* smem, int t_begin, int t_end, int nph) {
    using A = AbGeom<NKS>;
    using Q = AbQ<NKS>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int vt = tid - 64 * 8, vw = vt >> 6;
    const int n = t_end - t_begin;
    AbRaw r0[AB_VSTEPS], r1[AB_VSTEPS], r2[AB_VSTEPS];
#pragma unroll
    for (int st = 0; st < AB_VSTEPS; ++st) {
        r0[st] = dd_ab_load_quad<NKS, U8>(P, t_begin, Q::XQUADS + vt + AB_VTHREADS * st);
        r1[st] = dd_ab_load_quad<NKS, U8>(P, t_begin + (n > 1 ? 1 : 0), Q::XQUADS + vt + AB_VTHREADS * st);
    }
    float2 wk[AB_VSTEPS][4];                               // tile-relative NCO phasors of this lane's sample positions
#pragma unroll
    for (int st = 0; st < AB_VSTEPS; ++st) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pos = 4 * (Q::XQUADS + vt + AB_VTHREADS * st) + k;
            wk[st][k] = (P.flags & DD_CHAIN_NCO) ? dd_phasor((uint64_t)pos * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
        }
    }
    AbRaw xraw;
    float2 wkx[4];
    xraw.a = xraw.b = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) wkx[k] = make_float2(1.f, 0.f);
    {   // range of tile 0 (what phase p-1 does for tile p); its non-unit flag is preset: the true maximum is read
        float m = 0.f;
#pragma unroll
        for (int st = 0; st < AB_VSTEPS; ++st) m = dd_ab_absmax(r0[st], m);
        m = dd_wave_max(m);
        if (lane == 63) reinterpret_cast<float*>(smem + A::RED_OFF)[vw] = m;
    }
    __syncthreads();                                        // prologue barrier (matched in dd_ab_matrix)
    if (DD_AB_VEC_PRIO) __builtin_amdgcn_s_setprio(DD_AB_VEC_PRIO);
    const bool stamp = ST && taps.stamps != nullptr;       // ST: the in-kernel stamps are compiled in (tools only: ~30 scalar instructions per wave and phase)
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = 0; p < nph; p += 3) {                      // nph is a multiple of 6
        dd_ab_vphase<NKS, U8, CX>(P, smem, t_begin, n, p, r0, r1, r2, wk, vt, vw, lane, stamp, acc_t, xraw, wkx);
        dd_ab_vphase<NKS, U8, CX>(P, smem, t_begin, n, p + 1, r1, r2, r0, wk, vt, vw, lane, stamp, acc_t, xraw, wkx);
        dd_ab_vphase<NKS, U8, CX>(P, smem, t_begin, n, p + 2, r2, r0, r1, wk, vt, vw, lane, stamp, acc_t, xraw, wkx);
    }
    if (stamp && lane == 0) {
        for (int q = 0; q < 4; ++q) taps.stamps[((size_t)blockIdx.x * 16 + (tid >> 6)) * 8 + q] = acc_t[q];
        taps.stamps[((size_t)blockIdx.x * 16 + (tid >> 6)) * 8 + 7] = (unsigned long long)nph | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
    }
}

// ------------------------------------------------------------------ matrix waves
// The discriminator's angle from the RATIO t = im / re (v_rcp_f32 + v_mul: 1.5 ulp) and re, so that im need not stay in
// a register behind the ratio (the straight-line epilogue holds 16 rows of everything it still needs).
//   dd_atan_ratio_small: |t| <= tan 22.5 deg, re > 0 (same polynomial as dd_atan_small)
//   dd_atan_ratio_full : any t including +-inf (re = 0) and NaN (0 / 0 -> 0 like np.angle); re < 0 adds the half turn.
//                        im's sign is sign(t) sign(re).  Same degree-15 polynomial as dd_fast_atan2.
__device__ __forceinline__ float dd_atan_ratio_small(float t) {
    const float z = t * t;
    float p = fmaf(7.902598251e-02f, z, -1.382445378e-01f);
    p = fmaf(p, z, 1.997187931e-01f);
    p = fmaf(p, z, -3.333275667e-01f);
    return fmaf(t, z * p, t);
}
__device__ __forceinline__ float dd_atan_ratio_full(float t, float re) {
    const float at = fabsf(t);
    const bool inv = at > 1.0f;
    const float u = inv ? __builtin_amdgcn_rcpf(at) : at;  // rcp(inf) = 0
    const float z = u * u;
    float p = -4.054567120e-03f;
    p = fmaf(p, z, 2.186295773e-02f);
    p = fmaf(p, z, -5.591232695e-02f);
    p = fmaf(p, z, 9.642197381e-02f);
    p = fmaf(p, z, -1.390862959e-01f);
    p = fmaf(p, z, 1.994656567e-01f);
    p = fmaf(p, z, -3.332986079e-01f);
    p = fmaf(p, z, 9.999993356e-01f);
    float r = p * u;
    r = inv ? 1.5707963267948966f - r : r;
    const bool neg = __float_as_int(re) < 0;                // sign bit (re = -0 counts as negative: t carries rcp(-0) = -inf)
    r = neg ? 3.141592653589793f - r : r;
    r = copysignf(r, neg ? -t : t);                         // sign(im) = sign(t) sign(re)
    return (t != t && re == 0.f) ? 0.f : r;                 // 0 / 0: atan2(0, 0) = 0
}

// discriminator of one strip out of the accumulators.  Lane (j = lane & 31, h = lane >> 5), register r holds output
// 32 (rowbase(r) + 4 h) + j of the strip, rowbase(r) = (r & 3) + 8 (r >> 2).  xrd: this lane's column-0 table (X0 of the
// wave for h = 0, X1 for h = 1), entry r = the output that precedes column 0 of register r's row.
template <int NKS>
__device__ __forceinline__ void dd_ab_epilogue(const DDChainParams& P, int b, int mw, int lane, const v16f& cre, const v16f& cim,
                                               const float2* xrd) {
    const int j = lane & 31, h = lane >> 5;
    const int64_t pw = (int64_t)b * MF_ADV - 32 + (int64_t)mw * MF_STRIP;
    float* out = reinterpret_cast<float*>(P.out) + (pw - P.s) + j + 128 * h;
    // One wave turns 1024 outputs into angles: written as ONE straight-line block over all 16 rows (16-way
    // instruction-level parallelism; a single in-order wave issues a dependent chain at ~12 cycles per
    // instruction, measured, and four rows at a time with a branch per group took 3500 cycles per strip).
    // 1. the left-hand neighbours of the row-leading lanes (0, 16, 32, 48), requested up front
    constexpr int NR = AB_REG_ROWS;
    float2 bv[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) bv[r] = xrd[r];
    // 2. z = y[n] conj(y[n-1]); y[n-1] is the same register one lane to the left: DPP row_shr:1, which leaves the
    //    first lane of each 16-lane row (no source lane) at the old value of the destination = its table entry
    float re[NR], im[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float pre = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(bv[r].x), __float_as_int(cre[r]), 0x111, 0xf, 0xf, false));
        const float pim = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(bv[r].y), __float_as_int(cim[r]), 0x111, 0xf, 0xf, false));
        re[r] = fmaf(cre[r], pre, cim[r] * pim);
        im[r] = fmaf(cim[r], pre, -cre[r] * pim);
    }
    // 3. wave-uniform fast path: every |angle| below 22.5 degrees (an oversampled FM signal always is), i.e.
    //    min over the rows of (tan(22.5 deg) re - |im|) > 0 in every lane (a NaN fails the test, and so does a product of
    //    exactly zero -- digital silence -- for which the small-angle form would compute 0 * rcp(0) = NaN and the
    //    full-range form returns np.angle(0) = 0)
    float mn = fmaf(0.41421354f, re[0], -fabsf(im[0]));
#pragma unroll
    for (int r = 1; r + 1 < NR; r += 2)                      // v_min3_f32: two rows per instruction
        mn = fminf(fminf(mn, fmaf(0.41421354f, re[r], -fabsf(im[r]))), fmaf(0.41421354f, re[r + 1], -fabsf(im[r + 1])));
    if ((NR & 1) == 0) mn = fminf(mn, fmaf(0.41421354f, re[NR - 1], -fabsf(im[NR - 1])));
    const bool all_small = __builtin_amdgcn_ballot_w64(!(mn > 0.f)) == 0;
    float a[NR];
    if (all_small) {
#pragma unroll
        for (int r = 0; r < NR; ++r) a[r] = dd_atan_small(im[r], re[r]);
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r) a[r] = dd_fast_atan2(im[r], re[r]);
    }
    // 4. row r of the lane is output 32 (rowbase(r) + 4 h) + j: 128 contiguous bytes per half wave and row.
    //    The tile's first 32 outputs (strip 0, row 0) belong to the previous tile.
#ifdef DD_AB_NO_STORE
    if (P.K != 12345) {                                    // (ablation: angles computed, one store per strip)
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) sum += a[r];
        out[0] = sum;
        return;
    }
#endif
    if (mw != 0 || h != 0) out[0] = a[0];
#pragma unroll
    for (int r = 1; r < NR; ++r) out[32 * ((r & 3) + 8 * (r >> 2))] = a[r];
}

// Complex-output flavour: the strip leaves straight from the accumulators as complex64, times the tile's start phasor
// (the conversion rotates tile-relative) and the inverse of the power-of-two scales (tile, taps).  Row r of the lane is
// output 32 (rowbase(r) + 4 h) + j: 256 contiguous bytes per half wave and row.  No neighbours, no boundary table.
template <int NKS>
__device__ __forceinline__ void dd_ab_store_cx(const DDChainParams& P, const DDMfmaTaps& taps, const char* smem, int b, int slot, int mw, int lane,
                                               const v16f& cre, const v16f& cim) {
    using A = AbGeom<NKS>;
    const int j = lane & 31, h = lane >> 5;
    const float unscale = taps.inv_tapscale / reinterpret_cast<const float*>(smem + A::SCALE_OFF)[slot & 3];
    float2 t = make_float2(unscale, 0.f);
    if (P.flags & DD_CHAIN_NCO) {
        const int64_t ns = (int64_t)b * MF_ADV - 32 - MfmaGeom<NKS>::HALO;
        const float2 ph = dd_phasor((uint64_t)(P.abs0 + ns) * P.cyc, P.nco_tbl);
        t = make_float2(ph.x * unscale, ph.y * unscale);
    }
    const int64_t pw = (int64_t)b * MF_ADV - 32 + (int64_t)mw * MF_STRIP;
    float2* out = reinterpret_cast<float2*>(P.out) + pw + j + 128 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float2 o = make_float2(fmaf(cre[r], t.x, -cim[r] * t.y), fmaf(cre[r], t.y, cim[r] * t.x));
        if (r != 0 || mw != 0 || h != 0) out[32 * ((r & 3) + 8 * (r >> 2))] = o;   // the tile's first 32 outputs belong to the previous tile
    }
}

// after the MFMAs: registers AB_REG_ROWS..15 (outputs 512..1023 of the strip, plus output 511 in the slack) to the set's
// y-half buffer, in output order: the vector waves run their discriminator in the next phase
__device__ __forceinline__ void dd_ab_write_yhalf(int lane, const v16f& cre, const v16f& cim, float* yre_strip, int plane_floats) {
    const int j = lane & 31, h = lane >> 5;
    float* wre = yre_strip + 4 + 128 * h + j;
    float* wim = wre + plane_floats;
#pragma unroll
    for (int r = AB_REG_ROWS; r < 16; ++r) {
        const int row = (r & 3) + 8 * ((r >> 2) - 2);      // row - 16
        wre[32 * row] = cre[r];
        wim[32 * row] = cim[r];
    }
    if (lane == 63) {                                      // output 511 = register 7, lane 63
        yre_strip[3] = cre[7];
        yre_strip[3 + plane_floats] = cim[7];
    }
}

// after the MFMAs: the last lane of each 16-lane row leaves its 16 outputs where the first lane of the next row (the
// lane DPP row_shr:1 cannot serve) will look for them.  Lanes 15 / 47: same register, tables XA / XB.  Lanes 31 / 63
// (column 31): the next row of the strip starts in another register -- lane 31 of register q is followed by lane 0 of
// register q + 1 ((q & 3) != 3) or lane 32 of register q - 3; lane 63 by lane 32 of register q + 1 or lane 0 of
// register q + 1 (q = 15: lane 0, register 0 of the NEXT strip: the X0 tables of the four waves are contiguous).
// Entry 0 of strip 0's X0 table has no producer (the output before the tile's first one belongs to the previous tile, and
// so do the 32 outputs of that row, which are not stored): the wave writes its own output there, so that the row's
// lane-0 product is |y|^2 (angle 0) and the wave-uniform small-angle test never sees what the LDS held before the launch.
// The one entry another wave reads -- the strip's last output, lane 63 of register 15, for lane 0 / register 0 of the
// next strip -- is written at the end of the MFMA phase (dd_ab_publish_last, in front of the barrier); the 63 entries
// the wave reads back itself wait until the start of its discriminator phase (dd_ab_publish), when the matrix pipe
// is busy with the other set: the ~200 cycles of masked stores no longer sit between the last MFMA and the barrier.
__device__ __forceinline__ void dd_ab_publish_last(int lane, const v16f& cre, const v16f& cim, float2* x0w) {
    if (lane == 63) {
        float* d = reinterpret_cast<float*>(x0w + 16);
        d[0] = cre[15];
        d[1] = cim[15];
    }
}
__device__ __forceinline__ void dd_ab_publish(int lane, int mw, const v16f& cre, const v16f& cim, float2* xtab0, float2* x0w, float2* xaw, float2* x1w, float2* xbw) {
#ifndef DD_AB_NO_X0_SELF          // (-DDD_AB_NO_X0_SELF: the round-2 bug, for checking that tests/test_gpu_determinism.py catches it)
    if (mw == 0 && lane == 0) {
        float* d = reinterpret_cast<float*>(xtab0);          // (strip 0's X0 = the set's table base: a constant address, no register kept for it)
        d[0] = cre[0];
        d[1] = cim[0];
    }
#endif
    if ((lane & 15) == 15) {
        const int g = lane >> 4;
        float2* pa = g == 0 ? xaw : (g == 1 ? x0w + 1 : (g == 2 ? xbw : x1w + 1));     // (q & 3) != 3
        float2* pb = g == 0 ? xaw : (g == 1 ? x1w - 3 : (g == 2 ? xbw : x0w + 1));     // (q & 3) == 3
        // (two 4-byte stores per entry: an 8-byte store wants re and im in adjacent registers, 32 v_mov per strip)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float* d = reinterpret_cast<float*>(((q & 3) != 3 ? pa : pb) + q);
            d[0] = cre[q];
            d[1] = cim[q];
        }
    }
}

// fragments of k-steps 0 and 1 of a strip (what the MFMA loop needs before its first instruction)
#ifndef AB_HEAD_KSTEPS
#define AB_HEAD_KSTEPS 1      // k-steps requested a phase early: 2 would cover the whole start-up, and spills (48 live registers across the barrier)
#endif
template <int NKS, int FIRST, int LAST>
__device__ __forceinline__ void dd_ab_frag_head(const char* abase, const v8h* tb, v8h (&f)[3][6]) {
    using G = MfmaGeom<NKS>;
    if (FIRST <= 0 && 0 < LAST) { DD_WS_LOADF(0, 0) }
    if (FIRST <= 1 && 1 < LAST) { DD_WS_LOADF(1, 1) }
}
template <int NKS>
__device__ __forceinline__ void dd_ab_mfma_strip(const char* abase, const v8h* tb, v16f& cre, v16f& cim, v8h (&f)[3][6], bool have_head) {
    using G = MfmaGeom<NKS>;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cre[r] = 0.f; cim[r] = 0.f; }
    if (!have_head) dd_ab_frag_head<NKS, 0, AB_HEAD_KSTEPS>(abase, tb, f);
    dd_ab_frag_head<NKS, AB_HEAD_KSTEPS, 2>(abase, tb, f);
#pragma unroll
    for (int ks = 0; ks < NKS - 1; ++ks) {
        DD_WS_STEP(ks % 3, (ks + 2) % 3, (ks + 2 < NKS ? ks + 2 : ks), (ks + 2 < NKS))
    }
    DD_WS_STEP((NKS - 1) % 3, (NKS + 1) % 3, NKS - 1, false)
}

// End of a set's discriminator phase: if every converting wave has reported tile t = qnext - 1 (the set's next MFMA
// tile) complete, request the fragments of its first two k-steps now; they land while the wave waits at the barrier.
template <int NKS>
__device__ __forceinline__ bool dd_ab_try_head(const char* smem, int n, int qnext, int aoff, const v8h* tb, v8h (&f)[3][6]) {
    return false;
}

// the halo step (quads 0..63 of a tile) on the matrix wave of strip 0, set SET: conversion in the set's discriminator
// phase, range check of the next one at the end of its MFMA phase (the data were requested a phase earlier)
template <int NKS, bool U8>
__device__ __forceinline__ void dd_ab_halo_convert(const DDChainParams& P, char* smem, int t_begin, int n, int q, int lane,
                                                   AbRaw& xraw, const float4* wkx_lds, int nu_flag) {
    using A = AbGeom<NKS>;
#ifndef DD_AB_NO_CONVERT
    if (q < n) {
        bool unit;
        const float scale = dd_ab_tile_scale(smem, A::RED_OFF, nu_flag, q, unit);
        const float4 w01 = wkx_lds[0], w23 = wkx_lds[1];
        const float2 wkx[4] = {make_float2(w01.x, w01.y), make_float2(w01.z, w01.w), make_float2(w23.x, w23.y), make_float2(w23.z, w23.w)};
        char* planes = smem + (q & 1) * A::PLANES_BYTES;
        if (AbQ<NKS>::XQUADS == 64 || lane < AbQ<NKS>::XQUADS) {
            if (unit) dd_ab_convert_quad<NKS, true>(xraw, planes, lane, wkx, scale);
            else dd_ab_convert_quad<NKS, false>(xraw, planes, lane, wkx, scale);
        }
        dd_ab_conv_done(smem, A::CONVCNT_OFF, q, lane);
    }
#endif
#ifndef DD_AB_NO_LOAD
    // the set's next tile (two phases ahead); lanes past the halo re-read its last quad (never written)
    xraw = dd_ab_load_quad<NKS, U8>(P, t_begin + (q + 2 < n ? q + 2 : n - 1), lane < AbQ<NKS>::XQUADS ? lane : AbQ<NKS>::XQUADS - 1);
#endif
}

template <int NKS, int SET, bool U8, bool CX, bool ST>
__device__ __forceinline__ void dd_ab_matrix(const DDChainParams& P, const DDMfmaTaps& taps, char* smem, int t_begin, int t_end, int nph) {
    using A = AbGeom<NKS>;
    const int tid = threadIdx.x, lane = tid & 63, mw = (tid >> 6) & 3;
    const int n = t_end - t_begin;
    const int i = lane & 31, h = lane >> 5;
    const int sb = mw * MF_STRIP;
    const int aoff = (2 * sb + (sb >> 1)) + 80 * i + 16 * h;
    const v8h* tb = reinterpret_cast<const v8h*>(smem + A::TAPS_OFF) + lane;
    float2* xtab = reinterpret_cast<float2*>(smem + A::BCOL_OFF + SET * A::BCOL_SET_BYTES);
    float2* x0 = xtab + 16 * mw;
    float2* xa = xtab + A::X0_ENTRIES + 16 * mw;
    float2* x1 = xtab + A::X0_ENTRIES + 64 + 16 * mw;
    float2* xb = xtab + A::X0_ENTRIES + 128 + 16 * mw;
    const int lg = lane >> 4;
    const float2* xrd = lg == 0 ? x0 : (lg == 1 ? xa : (lg == 2 ? x1 : xb));
    // halo step: set 1 owns the even tiles' (its discriminator phases are the even ones), set 0 the odd tiles'
    const bool halo = mw == 0;
    AbRaw xraw;
    xraw.a = xraw.b = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* wkx_lds = reinterpret_cast<float4*>(smem + A::WKX_OFF) + (SET * 64 + lane) * 2;     // written and read by this lane only
    if (halo) {
        xraw = dd_ab_load_quad<NKS, U8>(P, t_begin + (SET == 1 ? 0 : (n > 1 ? 1 : 0)), lane < AbQ<NKS>::XQUADS ? lane : AbQ<NKS>::XQUADS - 1);
        float2 w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (P.flags & DD_CHAIN_NCO) ? dd_phasor((uint64_t)(4 * lane + k) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
        wkx_lds[0] = make_float4(w[0].x, w[0].y, w[1].x, w[1].y);
        wkx_lds[1] = make_float4(w[2].x, w[2].y, w[3].x, w[3].y);
        if (SET == 1) {                                     // tile 0: the true maximum (its non-unit flag is preset)
            const float m = dd_wave_max(dd_ab_absmax(xraw, 0.f));
            if (lane == 63) reinterpret_cast<float*>(smem + A::RED_OFF)[AB_VWAVES] = m;
        }
    }
    __syncthreads();                                        // prologue barrier (tile 0's range is published)

    const bool stamp = ST && taps.stamps != nullptr;       // ST: the in-kernel stamps are compiled in (tools only: ~30 scalar instructions per wave and phase)
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_clk0 = stamp ? __builtin_readcyclecounter() : 0;
    const unsigned long long t_rt0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
    v16f cre, cim;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cre[r] = 0.f; cim[r] = 0.f; }
    v8h f[3][6];                                            // fragment ring of the MFMA loop (k-steps 0, 1 requested a phase early)
    bool head = false;
    // Set SET computes tile t in phase t + 1 where ((t + 1) & 1) == SET and runs its discriminator in phase t + 2.  The
    // loop body is [discriminator phase pe | MFMA phase pe + 1]: the fragments requested at the end of the first are
    // consumed at the start of the second, so nothing but the accumulators is live across the loop's back edge.
    // Set 0's phase 0 has no tile to compute: only the range of the halo it converts in phase 1.
    if (SET == 0) {
        unsigned long long tp = stamp ? __builtin_readcyclecounter() : 0;
        if (halo && 1 < n) dd_ab_publish_range(dd_ab_absmax(xraw, 0.f), smem, A::RED_OFF, A::NONUNIT_OFF, 1, AB_VWAVES, lane);
        DD_AB_STAMP(0)
        __syncthreads();
        DD_AB_STAMP(2)
    }
    for (int pe = 1 - SET; pe < nph; pe += 2) {
        unsigned long long tp = stamp ? __builtin_readcyclecounter() : 0;
        const int qm = pe + 1;                              // this set's MFMA phase of the pair (tile pe)
        {   // phase pe: the rest of tile pe - 2's table, halo step of tile pe, discriminator of tile pe - 2
            const bool epi = pe >= 2 && pe - 2 < n;
            const int nu_flag = halo ? reinterpret_cast<const int*>(smem + A::NONUNIT_OFF)[pe & 3] : 0;
            if (halo) dd_ab_halo_convert<NKS, U8>(P, smem, t_begin, n, pe, lane, xraw, wkx_lds, nu_flag);
#ifndef DD_AB_NO_EPI
            if (epi) {
                if (CX) dd_ab_store_cx<NKS>(P, taps, smem, t_begin + pe - 2, pe - 2, mw, lane, cre, cim);
                else dd_ab_epilogue<NKS>(P, t_begin + pe - 2, mw, lane, cre, cim, xrd);
            }
#endif
            head = dd_ab_try_head<NKS>(smem, n, qm, aoff, tb, f);
            DD_AB_STAMP(1)
            __syncthreads();
            DD_AB_STAMP(2)
        }
        if (qm >= nph) break;                               // (set 0: the phase after the last one)
        if (qm <= n) {
            __builtin_amdgcn_s_setprio(3);                  // MFMAs issue as soon as the pipe frees up
            const char* abase = smem + (pe & 1) * A::PLANES_BYTES + aoff;
#ifndef DD_AB_NO_MFMA
            dd_ab_mfma_strip<NKS>(abase, tb, cre, cim, f, head);
#else
            (void)abase; (void)tb;
#endif
            if (!CX) dd_ab_publish(lane, mw, cre, cim, reinterpret_cast<float2*>(smem + A::BCOL_OFF + SET * A::BCOL_SET_BYTES), x0, xa, x1, xb);
            if (AB_REG_ROWS < 16)
                dd_ab_write_yhalf(lane, cre, cim, reinterpret_cast<float*>(smem + A::YH_OFF + SET * A::YH_SET_BYTES) + mw * A::YH_STRIDE, 4 * A::YH_STRIDE);
            __builtin_amdgcn_s_setprio(DD_AB_EPI_PRIO);
        }
        // range of the halo the set converts in its next discriminator phase (tile qm + 1, requested a phase ago)
        if (halo && qm + 1 < n) dd_ab_publish_range(dd_ab_absmax(xraw, 0.f), smem, A::RED_OFF, A::NONUNIT_OFF, qm + 1, AB_VWAVES, lane);
        DD_AB_STAMP(0)
        __syncthreads();
        DD_AB_STAMP(2)
    }
    if (stamp && lane == 0) {
        const size_t w = (size_t)blockIdx.x * 16 + (tid >> 6);
        if ((tid >> 6) == 0) {   // in-kernel clock: shader ticks per 100 MHz reference tick over the whole loop
            taps.stamps[w * 8 + 5] = __builtin_readcyclecounter() - t_clk0;
            taps.stamps[w * 8 + 6] = __builtin_amdgcn_s_memrealtime() - t_rt0;
            taps.stamps[((size_t)blockIdx.x * 16 + 1) * 8 + 4] = t_rt0;                              // loop start (abs)
            taps.stamps[((size_t)blockIdx.x * 16 + 2) * 8 + 4] = __builtin_amdgcn_s_memrealtime();   // loop end (abs)
        }
        for (int q = 0; q < 3; ++q) taps.stamps[w * 8 + q] = acc_t[q];
        taps.stamps[w * 8 + 7] = (unsigned long long)nph | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
    }
}

// U8: raw interleaved uint8 I,Q input.  CX: complex64 output (no DD_CHAIN_FM) instead of angles.
template <int NKS, bool U8, bool CX, bool ST = false>
__global__ void __launch_bounds__(WS_THREADS) k_chain_mfma_ab(const DDChainParams P, const DDMfmaTaps taps, int t_first, int t_last, int nwg) {
    using A = AbGeom<NKS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wg = blockIdx.x;
    if (wg >= nwg) {
        // edge tiles ([0, t_first) and [t_last, nblocks)) ride along as trailing 4-wave workgroups (see k_chain_mfma_ws)
        if (threadIdx.x >= MF_THREADS) return;
        const int e = wg - nwg;
        const int b = e < t_first ? e : t_last + (e - t_first);
        v8h* tl = reinterpret_cast<v8h*>(smem + A::TAPS_OFF);
        for (int idx = threadIdx.x; idx < 2 * NKS * 64; idx += MF_THREADS) tl[idx] = taps.frag[idx];
        dd_edge_tile_lean<NKS>(P, taps, b, smem, tl);
        return;
    }
    if (ST && taps.stamps && threadIdx.x == 0) taps.stamps[((size_t)wg * 16) * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    const int nt = t_last - t_first;
    const int t_begin = t_first + (int)(((int64_t)wg * nt) / nwg);
    const int t_end = t_first + (int)(((int64_t)(wg + 1) * nt) / nwg);
    if (t_begin >= t_end) return;
    {
        v8h* tl = reinterpret_cast<v8h*>(smem + A::TAPS_OFF);
        for (int idx = threadIdx.x; idx < 2 * NKS * 64; idx += WS_THREADS) tl[idx] = taps.frag[idx];
        if (threadIdx.x < 4) reinterpret_cast<int*>(smem + A::NONUNIT_OFF)[threadIdx.x] = threadIdx.x == 0 ? 1 : 0;   // tile 0: read the true max
        if (threadIdx.x < 2 * AB_RED_ENTRIES) reinterpret_cast<float*>(smem + A::RED_OFF)[threadIdx.x] = 0.f;          // (unused entries stay 0)
        if (threadIdx.x < 4) reinterpret_cast<int*>(smem + A::YHCNT_OFF)[threadIdx.x] = 0;                            // (+ the two conversion counters)
    }
    __syncthreads();
    const int nph = ((t_end - t_begin + 2 + 5) / 6) * 6;      // phases: a multiple of the vector loop's 3 and the matrix sets' 2
    if (threadIdx.x < 256) dd_ab_matrix<NKS, 0, U8, CX, ST>(P, taps, smem, t_begin, t_end, nph);
    else if (threadIdx.x < 512) dd_ab_matrix<NKS, 1, U8, CX, ST>(P, taps, smem, t_begin, t_end, nph);
    else dd_ab_vector<NKS, U8, CX, ST>(P, taps, smem, t_begin, t_end, nph);
}
