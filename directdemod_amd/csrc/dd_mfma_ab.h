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
#ifndef DD_AB_EPI_PRIO
#define DD_AB_EPI_PRIO 0
#endif
// timing ablations (tools/mkvariant.sh ... -DDD_AB_NO_xxx; results are wrong by construction, never shipped):
//   DD_AB_NO_EPI      matrix waves skip the discriminator      DD_AB_NO_MFMA   matrix waves skip the MFMAs
//   DD_AB_NO_CONVERT  vector waves skip the rotation / split   DD_AB_NO_LOAD   vector waves skip the tile loads

template <int NKS>
struct AbGeom {
    using G = MfmaGeom<NKS>;
    static constexpr int NQ = G::SPAN / 2;                               // sample pairs per tile
    static constexpr int NIT = (NQ + AB_VTHREADS - 1) / AB_VTHREADS;     // 5 for 255 taps (4.25: the oldest two vector waves take the fifth step)
    static constexpr int PLANES_BYTES = 4 * G::PLANE;
    static constexpr int TAPS_OFF = 2 * PLANES_BYTES;
    static constexpr int TAPS_BYTES = 2 * NKS * 64 * 16;
    static constexpr int RED_OFF = TAPS_OFF + TAPS_BYTES;                // [2][AB_VWAVES] float
    static constexpr int NONUNIT_OFF = RED_OFF + 2 * AB_VWAVES * 4;      // [4] int
    // left-hand neighbours that DPP row_shr:1 cannot deliver (the first lane of each 16-lane row), per matrix set, float2:
    //   X0[4 waves][16] (+1: the slot the last strip's last output falls into)  for lanes 0   (column 0, rows of half 0)
    //   XA[4][16] for lanes 16 (= lane 15, same register)    X1[4][16] for lanes 32 (column 0, half 1)    XB[4][16] for lanes 48 (= lane 47)
    static constexpr int X0_ENTRIES = 4 * 16 + 1;
    static constexpr int BCOL_OFF = NONUNIT_OFF + 16;
    static constexpr int BCOL_SET_BYTES = (X0_ENTRIES + 3 * 4 * 16 + 1) * 8;
    static constexpr int LDS_BYTES = (BCOL_OFF + 2 * BCOL_SET_BYTES + 15) & ~15;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(MF_LDS_TILE_BYTES(NKS) <= TAPS_OFF, "the edge tile's image must not reach the tap fragments");
};

template <int NKS, bool U8>
__device__ __forceinline__ void dd_ab_load(const DDChainParams& P, int b, int vt, float4 (&raw)[AbGeom<NKS>::NIT]) {
    using G = MfmaGeom<NKS>;
    using A = AbGeom<NKS>;
    const int64_t ns = (int64_t)b * MF_ADV - 32 - G::HALO;
    if (U8) {
        const char* base = reinterpret_cast<const char*>(P.in) + 2 * ns;                            // wave-uniform
#pragma unroll
        for (int it = 0; it < A::NIT; ++it) {
            int q = vt + AB_VTHREADS * it;
            if (AB_VTHREADS * (it + 1) > A::NQ) q = q < A::NQ ? q : A::NQ - 1;
            const uint32_t d = *reinterpret_cast<const uint32_t*>(base + 4u * (unsigned)q);
            raw[it] = make_float4((float)(d & 0xff) - 127.5f, (float)((d >> 8) & 0xff) - 127.5f,
                                  (float)((d >> 16) & 0xff) - 127.5f, (float)(d >> 24) - 127.5f);
        }
        return;
    }
    const char* base = reinterpret_cast<const char*>(reinterpret_cast<const float2*>(P.in) + ns);   // wave-uniform
#pragma unroll
    for (int it = 0; it < A::NIT; ++it) {
        int q = vt + AB_VTHREADS * it;
        if (AB_VTHREADS * (it + 1) > A::NQ) q = q < A::NQ ? q : A::NQ - 1;   // partial last step: re-read, write masked
        raw[it] = *reinterpret_cast<const float4*>(base + 16u * (unsigned)q);   // two consecutive samples
    }
}

// rotate (tile-relative NCO phase, see dd_ws_convert), split into f16 limbs, write the four planes
template <int NKS, bool UNIT_SCALE>
__device__ __forceinline__ void dd_ab_convert(const float4 (&raw)[AbGeom<NKS>::NIT], char* planes,
                                              const float2 (&wk)[AbGeom<NKS>::NIT][2], float scale, int vt) {
    using G = MfmaGeom<NKS>;
    using A = AbGeom<NKS>;
#pragma unroll
    for (int it = 0; it < A::NIT; ++it) {
        const int q = vt + AB_VTHREADS * it;
        if (AB_VTHREADS * (it + 1) > A::NQ && q >= A::NQ) continue;
        const int e = 2 * q;
        float2 pa = wk[it][0], pb = wk[it][1];
        if (!UNIT_SCALE) {
            pa = make_float2(pa.x * scale, pa.y * scale);
            pb = make_float2(pb.x * scale, pb.y * scale);
        }
        const float2 xa = dd_cmul(make_float2(raw[it].x, raw[it].y), pa);
        const float2 xb = dd_cmul(make_float2(raw[it].z, raw[it].w), pb);
        // f16 limb split in 12 plain vector instructions per sample pair, written out: hi = RNE_f16(x) packed two at
        // a time (v_cvt_pk_f16_f32), unpacked again (v_cvt_f32_f16 on either half), lo = RNE_f16(x - hi).  Left to the
        // compiler this is 16-17 instructions, six of them v_fma_mixlo/mixhi_f16 fused with the rotation -- and
        // beside a busy matrix pipe a mix op costs 8.7 cycles of the SIMD's issue against 4.6 for a plain
        // conversion (tools/ubench/valu_beside_mfma.hip).
        uint32_t rh, rl, ih, il;
        {
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
        const int off = 2 * e + 16 * (e >> 5);
#ifdef DD_AB_NO_DSWRITE
        if (scale != 12345.f) continue;                    // (ablation: the limbs are computed, never written)
#endif
        *reinterpret_cast<uint32_t*>(planes + off) = rh;
        *reinterpret_cast<uint32_t*>(planes + G::PLANE + off) = rl;
        *reinterpret_cast<uint32_t*>(planes + 2 * G::PLANE + off) = ih;
        *reinterpret_cast<uint32_t*>(planes + 3 * G::PLANE + off) = il;
    }
}

#define DD_AB_STAMP(i) if (stamp) { const unsigned long long tn = __builtin_readcyclecounter(); acc_t[i] += tn - tp; tp = tn; }

// one vector-wave phase p: loads of tile p+2 | conversion of tile p | range check of tile p+1 | barrier
template <int NKS, bool U8>
__device__ __forceinline__ void dd_ab_vphase(const DDChainParams& P, char* smem, int t_begin, int n, int p,
                                             float4 (&rcur)[AbGeom<NKS>::NIT], float4 (&rnext)[AbGeom<NKS>::NIT],
                                             float4 (&rld)[AbGeom<NKS>::NIT], const float2 (&wk)[AbGeom<NKS>::NIT][2],
                                             int vt, int vw, int lane, bool stamp, unsigned long long (&acc_t)[8]) {
    using A = AbGeom<NKS>;
    unsigned long long tp = stamp ? __builtin_readcyclecounter() : 0;
    float* redall = reinterpret_cast<float*>(smem + A::RED_OFF);
    int* nonunit = reinterpret_cast<int*>(smem + A::NONUNIT_OFF);
#ifndef DD_AB_NO_LOAD
    {
        const int bl = t_begin + (p + 2 < n ? p + 2 : n - 1);   // past the end: harmless re-read, never used
        dd_ab_load<NKS, U8>(P, bl, vt, rld);
    }
#endif
    DD_AB_STAMP(0)
#ifndef DD_AB_NO_CONVERT
    if (p < n) {                                            // convert tile p (range published in phase p-1)
        const float* red = redall + (p & 1) * AB_VWAVES;
        float m = 1.0f;
        if (__builtin_amdgcn_readfirstlane(nonunit[p & 3]) != 0) {
            m = red[0];
#pragma unroll
            for (int k = 1; k < AB_VWAVES; ++k) m = fmaxf(m, red[k]);
        }
        if (vt == 0) nonunit[(p + 2) & 3] = 0;              // re-arm the slot tile p+2's producers raise in phase p+1
        const bool unit = (m >= 0.25f) && (m < 32768.0f);   // the f16 limbs hold the tile unscaled (see dd_ws_vphase)
        const float scale = unit ? 1.0f : dd_pow2_scale_for(m);
        if (unit) dd_ab_convert<NKS, true>(rcur, smem + (p & 1) * A::PLANES_BYTES, wk, scale, vt);
        else dd_ab_convert<NKS, false>(rcur, smem + (p & 1) * A::PLANES_BYTES, wk, scale, vt);
    }
#endif
    DD_AB_STAMP(1)
    if (p + 1 < n) {                                        // does tile p+1 fit the f16 limbs unscaled?
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < A::NIT; ++it)
            m = fmaxf(fmaxf(m, fabsf(rnext[it].x)), fmaxf(fabsf(rnext[it].y), fmaxf(fabsf(rnext[it].z), fabsf(rnext[it].w))));
        const bool hi_any = __builtin_amdgcn_ballot_w64(!(m < 32768.0f)) != 0;
        const bool lo_any = __builtin_amdgcn_ballot_w64(m >= 0.25f) != 0;
        if (hi_any || !lo_any) {
            m = dd_wave_max(m);
            if (lane == 63) atomicOr(nonunit + ((p + 1) & 3), 1);
        } else m = 1.0f;
        if (lane == 63) redall[((p + 1) & 1) * AB_VWAVES + vw] = m;
    }
    DD_AB_STAMP(2)
    __syncthreads();
    DD_AB_STAMP(3)
}

template <int NKS, bool U8>
__device__ __forceinline__ void dd_ab_vector(const DDChainParams& P, const DDMfmaTaps& taps, char* smem, int t_begin, int t_end, int nph) {
    using A = AbGeom<NKS>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int vt = tid - 64 * 8, vw = vt >> 6;
    const int n = t_end - t_begin;
    float4 r0[A::NIT], r1[A::NIT], r2[A::NIT];
    dd_ab_load<NKS, U8>(P, t_begin, vt, r0);
    dd_ab_load<NKS, U8>(P, t_begin + (n > 1 ? 1 : 0), vt, r1);
    float2 wk[A::NIT][2];                                  // tile-relative NCO phasors of this lane's sample positions
#pragma unroll
    for (int it = 0; it < A::NIT; ++it) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int pos = 2 * (vt + AB_VTHREADS * it) + k;
            wk[it][k] = (P.flags & DD_CHAIN_NCO) ? dd_phasor((uint64_t)pos * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
        }
    }
    {   // range of tile 0 (what phase p-1 does for tile p); its non-unit flag is preset: the true maximum is read
        float m = 0.f;
#pragma unroll
        for (int it = 0; it < A::NIT; ++it)
            m = fmaxf(fmaxf(m, fabsf(r0[it].x)), fmaxf(fabsf(r0[it].y), fmaxf(fabsf(r0[it].z), fabsf(r0[it].w))));
        m = dd_wave_max(m);
        if (lane == 63) reinterpret_cast<float*>(smem + A::RED_OFF)[vw] = m;
    }
    __syncthreads();                                        // prologue barrier (matched in dd_ab_matrix)
    const bool stamp = taps.stamps != nullptr;
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = 0; p < nph; p += 3) {                      // nph is a multiple of 6
        dd_ab_vphase<NKS, U8>(P, smem, t_begin, n, p, r0, r1, r2, wk, vt, vw, lane, stamp, acc_t);
        dd_ab_vphase<NKS, U8>(P, smem, t_begin, n, p + 1, r1, r2, r0, wk, vt, vw, lane, stamp, acc_t);
        dd_ab_vphase<NKS, U8>(P, smem, t_begin, n, p + 2, r2, r0, r1, wk, vt, vw, lane, stamp, acc_t);
    }
    if (stamp && lane == 0) {
        for (int q = 0; q < 4; ++q) taps.stamps[((size_t)blockIdx.x * 16 + (tid >> 6)) * 8 + q] = acc_t[q];
        taps.stamps[((size_t)blockIdx.x * 16 + (tid >> 6)) * 8 + 7] = (unsigned long long)nph | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
    }
}

// ------------------------------------------------------------------ matrix waves
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
    // 1. the left-hand neighbours of the row-leading lanes (0, 16, 32, 48) for all 16 registers, requested up front
    float2 bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = xrd[r];
    // 2. z = y[n] conj(y[n-1]); y[n-1] is the same register one lane to the left: DPP row_shr:1, which leaves the
    //    first lane of each 16-lane row (no source lane) at the old value of the destination = its table entry
    float re[16], im[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float pre = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(bv[r].x), __float_as_int(cre[r]), 0x111, 0xf, 0xf, false));
        const float pim = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(bv[r].y), __float_as_int(cim[r]), 0x111, 0xf, 0xf, false));
        re[r] = fmaf(cre[r], pre, cim[r] * pim);
        im[r] = fmaf(cim[r], pre, -cre[r] * pim);
    }
    // 3. wave-uniform fast path: every |angle| below 22.5 degrees (an oversampled FM signal always is), i.e.
    //    min over the 16 rows of (tan(22.5 deg) re - |im|) >= 0 in every lane (a NaN fails the test)
    float mn = fmaf(0.41421354f, re[0], -fabsf(im[0]));
#pragma unroll
    for (int r = 1; r < 16; ++r) mn = fminf(mn, fmaf(0.41421354f, re[r], -fabsf(im[r])));
    const bool all_small = __builtin_amdgcn_ballot_w64(!(mn >= 0.f)) == 0;
    float a[16];
    if (all_small) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = dd_atan_small(im[r], re[r]);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = dd_fast_atan2(im[r], re[r]);
    }
    // 4. row r of the lane is output 32 (rowbase(r) + 4 h) + j: 128 contiguous bytes per half wave and row.
    //    The tile's first 32 outputs (strip 0, row 0) belong to the previous tile.
#ifdef DD_AB_NO_STORE
    if (P.K != 12345) {                                    // (ablation: angles computed, one store per strip)
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += a[r];
        out[0] = sum;
        return;
    }
#endif
    if (mw != 0 || h != 0) out[0] = a[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) out[32 * ((r & 3) + 8 * (r >> 2))] = a[r];
}

// after the MFMAs: the last lane of each 16-lane row leaves its 16 outputs where the first lane of the next row (the
// lane DPP row_shr:1 cannot serve) will look for them.  Lanes 15 / 47: same register, tables XA / XB.  Lanes 31 / 63
// (column 31): the next row of the strip starts in another register -- lane 31 of register q is followed by lane 0 of
// register q + 1 ((q & 3) != 3) or lane 32 of register q - 3; lane 63 by lane 32 of register q + 1 or lane 0 of
// register q + 1 (q = 15: lane 0, register 0 of the NEXT strip: the X0 tables of the four waves are contiguous).
__device__ __forceinline__ void dd_ab_publish(int lane, const v16f& cre, const v16f& cim, float2* x0w, float2* xaw, float2* x1w, float2* xbw) {
    if ((lane & 15) == 15) {
        const int g = lane >> 4;
        float2* pa = g == 0 ? xaw : (g == 1 ? x0w + 1 : (g == 2 ? xbw : x1w + 1));     // (q & 3) != 3
        float2* pb = g == 0 ? xaw : (g == 1 ? x1w - 3 : (g == 2 ? xbw : x0w + 1));     // (q & 3) == 3
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float2 v = make_float2(cre[q], cim[q]);
            if ((q & 3) != 3) pa[q] = v;
            else pb[q] = v;
        }
    }
}

template <int NKS>
__device__ __forceinline__ void dd_ab_mfma_strip(const char* abase, const v8h* tb, v16f& cre, v16f& cim) {
    using G = MfmaGeom<NKS>;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cre[r] = 0.f; cim[r] = 0.f; }
    v8h f[3][6];
    DD_WS_LOADF(0, 0)
    DD_WS_LOADF(1, 1)
#pragma unroll
    for (int ks = 0; ks < NKS - 1; ++ks) {
        DD_WS_STEP(ks % 3, (ks + 2) % 3, (ks + 2 < NKS ? ks + 2 : ks), (ks + 2 < NKS))
    }
    DD_WS_STEP((NKS - 1) % 3, (NKS + 1) % 3, NKS - 1, false)
}

template <int NKS, int SET>
__device__ __forceinline__ void dd_ab_matrix(const DDChainParams& P, const DDMfmaTaps& taps, char* smem, int t_begin, int t_end, int nph) {
    using G = MfmaGeom<NKS>;
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
    __syncthreads();                                        // prologue barrier (tile 0's range is published)

    const bool stamp = taps.stamps != nullptr;
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_clk0 = stamp ? __builtin_readcyclecounter() : 0;
    const unsigned long long t_rt0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
    v16f cre, cim;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cre[r] = 0.f; cim[r] = 0.f; }
    // set SET computes in the phases q with (q & 1) == SET (tile q - 1) and runs that tile's discriminator in phase q + 1
    for (int p = 0; p < nph; p += 2) {
        unsigned long long tp = stamp ? __builtin_readcyclecounter() : 0;
        const int qm = p + SET;                             // this set's MFMA phase of the pair
        if (SET == 1) {                                     // phase p: discriminator of tile p - 2 (computed in phase p - 1)
#ifndef DD_AB_NO_EPI
            if (p >= 2 && p - 2 < n) dd_ab_epilogue<NKS>(P, t_begin + p - 2, mw, lane, cre, cim, xrd);
#endif
            DD_AB_STAMP(1)
            __syncthreads();
            DD_AB_STAMP(2)
        }
        if (qm >= 1 && qm <= n) {
            __builtin_amdgcn_s_setprio(3);                  // MFMAs issue as soon as the pipe frees up
            const char* abase = smem + ((qm - 1) & 1) * A::PLANES_BYTES + aoff;
#ifndef DD_AB_NO_MFMA
            dd_ab_mfma_strip<NKS>(abase, tb, cre, cim);
#else
            (void)abase; (void)tb;
#endif
            dd_ab_publish(lane, cre, cim, x0, xa, x1, xb);
            __builtin_amdgcn_s_setprio(DD_AB_EPI_PRIO);
        }
        DD_AB_STAMP(0)
        __syncthreads();
        DD_AB_STAMP(2)
        if (SET == 0) {                                     // phase p + 1: discriminator of tile p - 1 (computed in phase p)
#ifndef DD_AB_NO_EPI
            if (p >= 1 && p - 1 < n) dd_ab_epilogue<NKS>(P, t_begin + p - 1, mw, lane, cre, cim, xrd);
#endif
            DD_AB_STAMP(1)
            __syncthreads();
            DD_AB_STAMP(2)
        }
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

// FM output only (the complex-output flavour stays on k_chain_mfma_ws).  U8: raw interleaved uint8 I,Q input.
template <int NKS, bool U8>
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
    if (taps.stamps && threadIdx.x == 0) taps.stamps[((size_t)wg * 16) * 8 + 4] = __builtin_amdgcn_s_memrealtime();
    const int nt = t_last - t_first;
    const int t_begin = t_first + (int)(((int64_t)wg * nt) / nwg);
    const int t_end = t_first + (int)(((int64_t)(wg + 1) * nt) / nwg);
    if (t_begin >= t_end) return;
    {
        v8h* tl = reinterpret_cast<v8h*>(smem + A::TAPS_OFF);
        for (int idx = threadIdx.x; idx < 2 * NKS * 64; idx += WS_THREADS) tl[idx] = taps.frag[idx];
        if (threadIdx.x < 4) reinterpret_cast<int*>(smem + A::NONUNIT_OFF)[threadIdx.x] = threadIdx.x == 0 ? 1 : 0;   // tile 0: read the true max
    }
    __syncthreads();
    const int nph = ((t_end - t_begin + 2 + 5) / 6) * 6;      // phases: a multiple of the vector loop's 3 and the matrix sets' 2
    if (threadIdx.x < 256) dd_ab_matrix<NKS, 0>(P, taps, smem, t_begin, t_end, nph);
    else if (threadIdx.x < 512) dd_ab_matrix<NKS, 1>(P, taps, smem, t_begin, t_end, nph);
    else dd_ab_vector<NKS, U8>(P, taps, smem, t_begin, t_end, nph);
}
