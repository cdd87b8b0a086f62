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
//     B[m][j] = g2[m - j]            Toeplitz band of the reversed taps (registers)
// K-dimension = 31 + (HALO+1) padded to 16*NKS.  With A = signal, the 32 lanes of
// one accumulator register hold 32 CONSECUTIVE outputs, so the FM discriminator's
// y[n-1] is one lane to the left and the result stores are full 128-byte lines.
//
// LDS image: four f16 planes (re_hi, re_lo, im_hi, im_lo) of the NCO-rotated,
// power-of-two-scaled tile; every 32 samples are followed by 16 B of padding so the
// 64-byte-strided ds_read_b128 of the A fragments is bank-conflict free
// (dword index 20 i + 4 h, distinct for the 16 lanes of every b128 lane group).
#include "dd_chain_kernels.h"

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
};

__device__ __forceinline__ float dd_pow2_scale_for(float m) {
    // power of two s with m*s in [2^13, 2^14)  (s = 1 for m == 0 / denormal)
    const uint32_t eb = (__float_as_uint(m) >> 23) & 0xff;
    int se = 267 - (int)eb;
    se = se > 254 ? 254 : (se < 1 ? 1 : se);
    return eb == 0 ? 1.0f : __uint_as_float((uint32_t)se << 23);
}

template <int NKS>
__global__ void __launch_bounds__(MF_THREADS, 2) k_chain_mfma(const DDChainParams P, const DDMfmaTaps taps) {
    constexpr int HALO = 16 * NKS - 32;
    constexpr int SPAN = MF_T + HALO;                     // staged samples (multiple of 32)
    constexpr int PLANE = SPAN * 2 + (SPAN / 32) * 16;    // bytes per f16 plane incl. padding
    constexpr int NIT = (SPAN / 2 + MF_THREADS - 1) / MF_THREADS;
    constexpr int NGRP = SPAN / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* planes = smem;
    float2* w2 = reinterpret_cast<float2*>(smem + 4 * PLANE);
    float* red = reinterpret_cast<float*>(w2 + NGRP);           // MF_WAVES floats
    float2* wlast = reinterpret_cast<float2*>(red + MF_WAVES);  // MF_WAVES float2

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = dd_xcd_tile(blockIdx.x, P.nblocks);
    const int64_t P0 = (int64_t)b * MF_ADV - 32;          // first FIR output computed by this tile
    const int64_t ns = P0 - HALO;                         // chunk-relative index of staged element 0
    const bool fm = (P.flags & DD_CHAIN_FM) != 0;
    const bool nco = (P.flags & DD_CHAIN_NCO) != 0;
    const int K = P.K;

    // ---- issue the tile's global loads (two consecutive samples per lane per step)
    float4 raw[NIT];
    const bool fast_ok = !(P.flags & DD_CHAIN_U8_INPUT) && ((reinterpret_cast<uintptr_t>(P.in) & 15) == 0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = 2 * (tid + MF_THREADS * it);
        const int64_t n = ns + e;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < SPAN) {
            if (fast_ok && n >= 0 && n + 1 < P.L) {
                v = *reinterpret_cast<const float4*>(reinterpret_cast<const float2*>(P.in) + n);
            } else {
                // slow path: stream edges, carried history (already NCO-rotated), u8 ingest
                DDChainParams Q = P;
                Q.flags &= ~DD_CHAIN_NCO;
                const float2 a = dd_load_sample(Q, n, make_float2(1.f, 0.f));
                const float2 c = dd_load_sample(Q, n + 1, make_float2(1.f, 0.f));
                v = make_float4(a.x, a.y, c.x, c.y);
            }
        }
        raw[it] = v;
    }

    // ---- per-64-sample NCO phasors and the tile's max |component|
    if (nco) {
        for (int g = tid; g < NGRP; g += MF_THREADS) {
            const uint64_t ph = (uint64_t)(P.abs0 + ns + (int64_t)g * 64) * P.cyc;
            w2[g] = dd_phasor(ph, P.nco_tbl);
        }
    }
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        m = fmaxf(m, fmaxf(fmaxf(fabsf(raw[it].x), fabsf(raw[it].y)), fmaxf(fabsf(raw[it].z), fabsf(raw[it].w))));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = dd_pow2_scale_for(m);
    const float inv_scale = 1.0f / scale;

    // ---- rotate, scale, split into f16 limbs, write the LDS planes
    {
        float2 w1a = make_float2(1.f, 0.f), w1b = make_float2(1.f, 0.f);
        if (nco) {
            w1a = dd_phasor((uint64_t)((2 * tid) & 63) * P.cyc, P.nco_tbl);
            w1b = dd_phasor((uint64_t)(((2 * tid) & 63) + 1) * P.cyc, P.nco_tbl);
        }
        const int64_t tail_first = P.L - (K - 1);       // first sample of the new history
        const bool tail_writer = (b == P.nblocks - 1) && P.tail_out != nullptr;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = 2 * (tid + MF_THREADS * it);
            if (e >= SPAN) continue;
            const int64_t n = ns + e;
            float2 pa = make_float2(scale, 0.f), pb = make_float2(scale, 0.f);
            if (nco) {
                const float2 g = w2[e >> 6];
                const float2 gs = make_float2(g.x * scale, g.y * scale);
                if (n >= 0) pa = dd_cmul(gs, w1a);
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
            *reinterpret_cast<v2h*>(planes + PLANE + off) = rl;
            *reinterpret_cast<v2h*>(planes + 2 * PLANE + off) = ih;
            *reinterpret_cast<v2h*>(planes + 3 * PLANE + off) = il;
        }
    }
    // ---- Toeplitz tap fragments -> registers (L2 resident, 16 B per lane, coalesced);
    //      issued here so they do not share the register file with the raw tile
    v8h bh[NKS], bl[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        bh[ks] = taps.frag[ks * 64 + lane];
        bl[ks] = taps.frag[(NKS + ks) * 64 + lane];
    }
    __syncthreads();

    // ---- Toeplitz GEMM: 6 MFMAs per k-step (3 limb products x re/im)
    v16f cre, cim;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cre[r] = 0.f; cim[r] = 0.f; }
    {
        const int i = lane & 31, h = lane >> 5;
        const int sb = wave * MF_STRIP;
        const char* abase = planes + (2 * sb + (sb >> 1)) + 80 * i + 16 * h;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int off = 32 * ks + 16 * (ks >> 1);
            const v8h arh = *reinterpret_cast<const v8h*>(abase + off);
            const v8h arl = *reinterpret_cast<const v8h*>(abase + PLANE + off);
            const v8h aih = *reinterpret_cast<const v8h*>(abase + 2 * PLANE + off);
            const v8h ail = *reinterpret_cast<const v8h*>(abase + 3 * PLANE + off);
            cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, bh[ks], cre, 0, 0, 0);
            cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, bh[ks], cim, 0, 0, 0);
            cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(arl, bh[ks], cre, 0, 0, 0);
            cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(ail, bh[ks], cim, 0, 0, 0);
            cre = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, bl[ks], cre, 0, 0, 0);
            cim = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, bl[ks], cim, 0, 0, 0);
        }
    }

    // ---- epilogue.  lane (j = lane & 31, h = lane >> 5), register r holds output
    //      p = P0 + 1024*wave + 32*row + j,  row = (r & 3) + 8 (r >> 2) + 4 h
    const int j = lane & 31, h = lane >> 5;
    const int64_t pw = P0 + (int64_t)wave * MF_STRIP;
    const int64_t p_lo = P0 + 32;                          // first output this tile owns
    const float unscale = inv_scale * taps.inv_tapscale;

    if (!fm) {
        float2* out = reinterpret_cast<float2*>(P.out);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int64_t p = pw + 32 * row + j;
            if (p >= p_lo && p < P.Ld) out[p] = make_float2(cre[r] * unscale, cim[r] * unscale);
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

    float* out = reinterpret_cast<float*>(P.out);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // neighbour y[p-1]: one lane to the left, except column 0 (lanes 0 and 32)
        float pre = __shfl_up(cre[r], 1);
        float pim = __shfl_up(cim[r], 1);
        // sources for column 0: row-1 lives in (r-1) of lane 31/63, or across the 4-row split
        float a_re, a_im, c_re, c_im;
        if ((r & 3) != 0) {
            a_re = __shfl(cre[r - 1], 31); a_im = __shfl(cim[r - 1], 31);     // lane 0  <- lane 31
            c_re = __shfl(cre[r - 1], 63); c_im = __shfl(cim[r - 1], 63);     // lane 32 <- lane 63
        } else {
            if (r > 0) { a_re = __shfl(cre[r - 1], 63); a_im = __shfl(cim[r - 1], 63); }   // rows 8,16,24 <- 7,15,23
            else { a_re = prev_strip.x; a_im = prev_strip.y; }                               // row 0 <- previous strip
            c_re = __shfl(cre[r + 3], 31); c_im = __shfl(cim[r + 3], 31);                    // rows 4,12,.. <- 3,11,..
        }
        if (lane == 0) { pre = a_re; pim = a_im; }
        if (lane == 32) { pre = c_re; pim = c_im; }
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int64_t p = pw + 32 * row + j;
        const float2 cur = make_float2(cre[r], cim[r]);
        if (p >= p_lo && p >= P.s && p < P.Ld) {
            out[p - P.s] = dd_fm_angle(cur, make_float2(pre, pim));
        }
        if (p == P.Ld - 1 && p >= p_lo - (b == 0 ? 0 : 0)) {
            *P.lasty_out = make_float2(cur.x * unscale, cur.y * unscale);
        }
    }
}

// ============================================================================
// host side
// ============================================================================
struct DDMfmaState {
    int K;
    int nks;
    v8h* frag;          // device
    float inv_tapscale;
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
    hipError_t e = hipMalloc((void**)&s->frag, frag.size() * sizeof(_Float16));
    if (e == hipSuccess) e = hipMemcpy(s->frag, frag.data(), frag.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (s->frag) hipFree(s->frag);
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
    hipFree(s->frag);
    delete s;
}

template <int NKS>
static int mfma_launch_t(const DDMfmaState* st, DDChainParams& P, hipStream_t s) {
    constexpr int HALO = 16 * NKS - 32;
    constexpr int SPAN = MF_T + HALO;
    constexpr int PLANE = SPAN * 2 + (SPAN / 32) * 16;
    const size_t lds = (size_t)4 * PLANE + sizeof(float2) * (SPAN / 64) + sizeof(float) * MF_WAVES +
                       sizeof(float2) * MF_WAVES + 16;
    static bool attr_set = false;
    if (!attr_set) {
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_mfma<NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    DDMfmaTaps t;
    t.frag = st->frag;
    t.inv_tapscale = st->inv_tapscale;
    hipLaunchKernelGGL(k_chain_mfma<NKS>, dim3(P.nblocks), dim3(MF_THREADS), lds, s, P, t);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int dd_mfma_launch(void* stv, const DDChainParams& Pin, hipStream_t s) {
    const DDMfmaState* st = reinterpret_cast<const DDMfmaState*>(stv);
    DDChainParams P = Pin;
    P.T = MF_T;
    P.nblocks = (int)((P.Ld + MF_ADV - 1) / MF_ADV);
    if (P.nblocks < 1) P.nblocks = 1;
    switch (st->nks) {
        case 6: return mfma_launch_t<6>(st, P, s);
        case 10: return mfma_launch_t<10>(st, P, s);
        case 12: return mfma_launch_t<12>(st, P, s);
        case 18: return mfma_launch_t<18>(st, P, s);
    }
    return DD_ERR_UNSUPPORTED;
}
