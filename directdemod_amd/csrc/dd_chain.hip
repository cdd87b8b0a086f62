// Fused hot path, f32 direct form:  offsetFreq (NCO) -> FIR (state carried) ->
// bwLim (integer decimation, only kept outputs are computed) -> demod_fm.
// Reference call sites: decode_noaa.py:623, decode_fm.py:64-68,
// decode_afsk1200.py:79-94, tutorial/3_chunking.py:24-38; operators comm.py:63-130,
// filters.py:53-75, demod_fm.py:29-51.
//
// Two kernels share the staging / epilogue code:
//   k_chain_dense  (M == 1)  256 threads x 8 contiguous outputs, register-tiled
//                            sliding window, planar skewed LDS (conflict-free b32)
//   k_chain_decim  (M >= 2)  one thread per kept output, float2 skewed LDS
// Taps are wave-uniform and come through the scalar cache (s_load), so the VALU
// issues only FMAs and LDS reads.
#include "dd_chain_kernels.h"
#include "dd_decimw.h"
#include <mutex>

#define DD_DENSE_R 8
#define DD_DENSE_THREADS 256
#define DD_DENSE_T (DD_DENSE_R * DD_DENSE_THREADS)
#ifndef DD_DECIM_THREADS
#define DD_DECIM_THREADS 256
#endif
#ifndef DD_DECIM_SPAN_MAX
#define DD_DECIM_SPAN_MAX 6144
#endif

// ============================================================================
// dense kernel
// ============================================================================
// LDS: sre[pos(e)], sim[pos(e)], pos(e) = e + (e >> 3)  (thread t reads 9t + ...)
__global__ void __launch_bounds__(DD_DENSE_THREADS) k_chain_dense(const DDChainParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int R = DD_DENSE_R;
    const int T = DD_DENSE_T;
    const int K = P.K;
    const int niter = (K + R - 1 + R - 1) / R;          // window elements / R, rounded up
    const int S = T + niter * R;                        // staged elements (incl. slack)
    const int SP = S + (S >> 3) + 8;                    // skewed length
    float* sre = reinterpret_cast<float*>(smem);
    float* sim = sre + SP;
    float2* w2 = reinterpret_cast<float2*>(sim + SP);   // per-64-group phasors
    float2* ylast = w2 + ((S + 63) / 64 + 1);           // per-thread last output

    const int t = threadIdx.x;
    const int b = dd_xcd_tile(blockIdx.x, P.nblocks);
    const int64_t pfirst = dd_tile_pfirst(P, b);
    const int64_t ns = pfirst - (K - 1);                // chunk-relative index of staged element 0
    const bool fm = (P.flags & DD_CHAIN_FM) != 0;

    // ---- per-group NCO phasors
    const int ngroups = (S + 63) / 64;
    if (P.flags & DD_CHAIN_NCO) {
        for (int g = t; g < ngroups; g += DD_DENSE_THREADS) {
            const uint64_t ph = (uint64_t)(P.abs0 + ns + (int64_t)g * 64) * P.cyc;
            w2[g] = dd_phasor(ph, P.nco_tbl);
        }
    }
    __syncthreads();
    // ---- stage the tile (coalesced), NCO applied on the way in
    {
        const float2 w1 = (P.flags & DD_CHAIN_NCO) ? dd_phasor((uint64_t)(t & 63) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
        // interior tile: batches of 8 unconditional loads in flight per lane (a predicated
        // load makes hipcc branch and wait for every element separately)
        const bool interior = !(P.flags & DD_CHAIN_U8_INPUT) && ns >= 0 && ns + S <= P.L;
        if (interior) {
            const float2* __restrict__ src = reinterpret_cast<const float2*>(P.in) + ns;
            for (int e0 = t; e0 < S; e0 += 8 * DD_DENSE_THREADS) {
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * DD_DENSE_THREADS;
                    v[u] = src[e < S ? e : S - 1];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * DD_DENSE_THREADS;
                    if (e < S) {
                        float2 x = v[u];
                        if (P.flags & DD_CHAIN_NCO) x = dd_cmul(x, dd_cmul(w2[e >> 6], w1));
                        const int p = e + (e >> 3);
                        sre[p] = x.x;
                        sim[p] = x.y;
                    }
                }
            }
        } else {
            for (int e = t; e < S; e += DD_DENSE_THREADS) {
                float2 ph = make_float2(1.f, 0.f);
                if (P.flags & DD_CHAIN_NCO) ph = dd_cmul(w2[e >> 6], w1);
                const float2 v = dd_load_sample(P, ns + e, ph);
                const int p = e + (e >> 3);
                sre[p] = v.x;
                sim[p] = v.y;
            }
        }
    }
    __syncthreads();

    // ---- new tail: the last tile holds the chunk's final K-1 (post-NCO) samples
    if (b == P.nblocks - 1 && P.tail_out) {
        for (int i = t; i < K - 1; i += DD_DENSE_THREADS) {
            const int64_t e = (P.L - (K - 1) + i) - ns;
            const int p = (int)e + ((int)e >> 3);
            P.tail_out[i] = make_float2(sre[p], sim[p]);
        }
    }

    // ---- register-tiled correlation: y[o] = sum_j g[j] s[o + j], o = tR + r
    float are[DD_DENSE_R], aim[DD_DENSE_R];
#pragma unroll
    for (int r = 0; r < R; ++r) { are[r] = 0.f; aim[r] = 0.f; }
    const float* __restrict__ G = P.taps_rev;           // G[i] = g[i - (R-1)], zero padded
    const int base = 9 * t;                             // pos(tR) = 8t + t
    for (int it = 0; it < niter; ++it) {
        float g[2 * DD_DENSE_R - 1];
#pragma unroll
        for (int i = 0; i < 2 * R - 1; ++i) g[i] = G[it * R + i];   // uniform -> s_load
        float vre[DD_DENSE_R], vim[DD_DENSE_R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            vre[i] = sre[base + it * 9 + i];
            vim[i] = sim[base + it * 9 + i];
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // element m = it*R + i pairs with tap j = m - r -> g[(R-1) + i - r]
                are[r] = fmaf(g[R - 1 + i - r], vre[i], are[r]);
                aim[r] = fmaf(g[R - 1 + i - r], vim[i], aim[r]);
            }
        }
    }

    // ---- epilogue
    const int64_t p0 = pfirst + (int64_t)t * R;         // FIR-output index of r = 0
    if (!fm) {
        float2* out = reinterpret_cast<float2*>(P.out);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t p = p0 + r;
            if (p < P.Ld) out[p] = make_float2(are[r], aim[r]);
        }
        if (P.lasty_out && p0 <= P.Ld - 1 && P.Ld - 1 < p0 + R) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (p0 + r == P.Ld - 1) *P.lasty_out = make_float2(are[r], aim[r]);
        }
        return;
    }
    if (p0 == -1) {            // y[-1] is the carried sample of the previous chunk
        const float2 ly = *P.lasty_in;
        are[0] = ly.x;
        aim[0] = ly.y;
    }
    ylast[t] = make_float2(are[R - 1], aim[R - 1]);
    __syncthreads();
    float2 prv = (t > 0) ? ylast[t - 1] : make_float2(0.f, 0.f);
    float* out = reinterpret_cast<float*>(P.out);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t p = p0 + r;
        const float2 cur = make_float2(are[r], aim[r]);
        if (p > pfirst && p < P.Ld) {
            out[p - P.s] = dd_fm_angle(cur, prv);
            if (p == P.Ld - 1) *P.lasty_out = cur;
        } else if (p == P.Ld - 1 && p >= 0) {
            *P.lasty_out = cur;       // single kept sample owned only as "previous"
        }
        prv = cur;
    }
}

// ============================================================================
// decimating kernel: one thread per kept output
// ============================================================================
// (device function: the stand-alone kernel k_chain_decim below runs it for every tile of a chunk without
// an interior run; k_chain_decim_p runs it in its trailing workgroups for the tiles around the interior run,
// so a chunk is ONE launch either way)
// in-launch hand-over of the carried state between the chunks of dd_chain_process_chunks (cdna_hip_programming.md,
// Guideline 16): the producer's stores are drained by every wave, the workgroup meets, one lane releases at agent scope
// and sets the flag with an agent-scope atomic; the consumer polls that one word relaxed from one lane, acquires once,
// the workgroup meets, then everybody reads with plain loads.  The flags are zeroed by a memset ahead of every launch.
// Producers have lower workgroup indices than their consumers and only two workgroups per chunk ever wait, so a
// waiting workgroup cannot keep its producer off the device; the spin is bounded all the same.
// A wait that gives up (the producer is not resident -- ascending dispatch order of workgroups is what the hardware does, not an
// architectural promise -- or is held up by a debugger or profiler) counts itself in *err and goes on with whatever the state
// buffers hold: the launch completes, its outputs are wrong, and the host turns the count into DD_ERR_TIMEOUT.
__device__ __forceinline__ void dd_seam_wait(unsigned int* flag, unsigned int* err, int spin_log2) {
    if (threadIdx.x == 0) {
        typedef __attribute__((address_space(1))) unsigned int gu32;
        gu32* f = (gu32*)flag;
        // default bound 2^19 polls of ~0.2 us (s_sleep 8) = ~0.1 s: three orders of magnitude above the ~100 us a producer tile of a
        // resident launch needs, and short enough that a launch whose producer is NOT resident (a GPU shared with another process)
        // fails within a second instead of holding the device for seconds per chunk (round 4's bound was 2^24)
        const unsigned bound = 1u << (spin_log2 > 0 && spin_log2 < 31 ? spin_log2 : 19);
        bool seen = false;
        for (unsigned spins = 0; spins < bound; ++spins) {
            if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { seen = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!seen && err) __hip_atomic_fetch_add((gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (a word of pinned host memory)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
__device__ __forceinline__ void dd_seam_post(unsigned int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave
    __syncthreads();
    if (threadIdx.x == 0) {
        typedef __attribute__((address_space(1))) unsigned int gu32;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (ROCm 7.2 may drop the fence's own wait)
        __hip_atomic_store((gu32*)flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void dd_decim_edge_tile(const DDChainParams& P, const int bid, char* smem) {
    const int K = P.K, M = P.M, T = P.T;
    const int S = (T - 1) * M + K + (M - 1);
    const int SP = S + 4;                                 // linear image: see the tap loop
    float2* sx = reinterpret_cast<float2*>(smem);
    float2* w2 = sx + SP;
    float2* yblk = w2 + ((S + 63) / 64 + 1);
    // taps, K floats (+ pad to 8), 16-byte aligned.  Offset arithmetic on the LDS base, not on a uintptr_t:
    // a pointer that went through an integer is a generic pointer to the compiler, its reads become
    // flat loads, and a flat load's vmcnt wait in the tap loop drains every tile load in flight
    float* gl = reinterpret_cast<float*>(smem + ((sizeof(float2) * ((size_t)SP + ((S + 63) / 64 + 1) + DD_DECIM_THREADS) + 15) & ~(size_t)15));

    const int t = threadIdx.x;
    int b = dd_xcd_tile(bid, P.nblocks - (P.skip_hi - P.skip_lo));
    if (b >= P.skip_lo) b += P.skip_hi - P.skip_lo;        // those tiles run in the persistent workgroups
    const int64_t pfirst = dd_tile_pfirst(P, b);
    const int64_t ns = (int64_t)P.off + pfirst * M - (K - 1);
    const bool fm = (P.flags & DD_CHAIN_FM) != 0;
    if (P.seam_wait && (ns < 0 || pfirst < 0)) dd_seam_wait(P.seam_wait, P.seam_err, P.seam_spin_log2);      // this tile reads the previous chunk's state

    const int ngroups = (S + 63) / 64;
    // interior tile (whole span inside the chunk, complex64 input): the WHOLE tile is requested at
    // once, before anything else -- up to 12 unconditional 16-byte loads (two samples each) in flight
    // per lane -- so a tile costs one HBM latency and that latency also covers the tap copy and the
    // phasor table fetches below.  (A predicated load -- edges, history, u8 -- makes hipcc branch and
    // wait for every element: those tiles take the loop further down.)
    typedef float v4f_a8 __attribute__((ext_vector_type(4), aligned(8)));     // 16-byte load on an 8-byte boundary
    constexpr int NV = (DD_DECIM_SPAN_MAX / 2 + DD_DECIM_THREADS - 1) / DD_DECIM_THREADS;
    const bool interior = !(P.flags & DD_CHAIN_U8_INPUT) && ns >= 0 && ns + S <= P.L && S <= DD_DECIM_SPAN_MAX;
    const int nq = S / 2;                                     // whole sample pairs in the span
    v4f_a8 v[NV];
    if (interior) {
        const float2* __restrict__ src = reinterpret_cast<const float2*>(P.in) + ns;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            int q = t + u * DD_DECIM_THREADS;
            q = q < nq ? q : nq - 1;                          // past the span: harmless re-read, never used
            v[u] = *reinterpret_cast<const v4f_a8*>(src + 2 * q);
        }
    }
    for (int j = t; j < ((K + 7) & ~7); j += DD_DECIM_THREADS) gl[j] = j < K ? P.taps_rev[(DD_DENSE_R - 1) + j] : 0.f;
    if (P.flags & DD_CHAIN_NCO) {
        for (int g = t; g < ngroups; g += DD_DECIM_THREADS) {
            const uint64_t ph = (uint64_t)(P.abs0 + ns + (int64_t)g * 64) * P.cyc;
            w2[g] = dd_phasor(ph, P.nco_tbl);
        }
    }
    __syncthreads();
    {
        const float2 w1 = (P.flags & DD_CHAIN_NCO) ? dd_phasor((uint64_t)(t & 63) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
        if (interior) {
            const float2* __restrict__ src = reinterpret_cast<const float2*>(P.in) + ns;
            float2 w1a = make_float2(1.f, 0.f), w1b = make_float2(1.f, 0.f);
            if (P.flags & DD_CHAIN_NCO) {
                w1a = dd_phasor((uint64_t)((2 * t) & 63) * P.cyc, P.nco_tbl);
                w1b = dd_phasor((uint64_t)(((2 * t) & 63) + 1) * P.cyc, P.nco_tbl);
            }
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int q = t + u * DD_DECIM_THREADS;
                const int e = 2 * q;
                if (q < nq) {
                    float2 xa = make_float2(v[u].x, v[u].y), xb = make_float2(v[u].z, v[u].w);
                    if (P.flags & DD_CHAIN_NCO) {
                        const float2 g = w2[e >> 6];
                        xa = dd_cmul(xa, dd_cmul(g, w1a));
                        xb = dd_cmul(xb, dd_cmul(g, w1b));
                    }
                    sx[e] = xa;
                    sx[e + 1] = xb;
                }
            }
            if ((S & 1) && t == 0) {                          // odd span: its last sample on its own
                const int e = S - 1;
                float2 x = src[e];
                if (P.flags & DD_CHAIN_NCO) x = dd_cmul(x, dd_cmul(w2[e >> 6], dd_phasor((uint64_t)(e & 63) * P.cyc, P.nco_tbl)));
                sx[e] = x;
            }
        } else {
            // edge tile (stream start with the carried history, chunk end, u8 input): batches of 8 UNCONDITIONAL loads
            // on clamped indices, the value selected afterwards.  A predicated load (dd_load_sample) makes hipcc branch
            // and wait per element: 24 dependent round trips, ~20 us for one tile -- which set the duration of the
            // whole launch for chunks of a few million samples (C3: 2^22-sample chunks, 21 us each)
            const bool u8 = (P.flags & DD_CHAIN_U8_INPUT) != 0;
            const bool need_tail = ns < 0;                             // block uniform
            const int K1 = K - 1;
            // the whole span in ONE round of loads (up to 24 per lane): an edge tile is on the critical path of every
            // launch -- each chunk has a first and a last one -- and in three rounds of eight its three memory round
            // trips were most of the ~9 us a launch costs before its first byte of payload
            constexpr int NB = DD_DECIM_SPAN_MAX / DD_DECIM_THREADS;   // 24
            constexpr int NT = 2;                                      // rounds whose samples may lie in the carried history (K - 1 <= 512)
            float2 x[NB], h[NT];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int64_t n = ns + t + u * DD_DECIM_THREADS;
                const int64_t nc = n < 0 ? 0 : (n >= P.L ? P.L - 1 : n);
                if (u8) {
                    const uchar2 q = reinterpret_cast<const uchar2*>(P.in)[nc];
                    x[u] = make_float2((float)q.x - 127.5f, (float)q.y - 127.5f);
                } else {
                    x[u] = reinterpret_cast<const float2*>(P.in)[nc];
                }
            }
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                h[u] = make_float2(0.f, 0.f);
                if (need_tail) {
                    const int64_t ti = ns + t + u * DD_DECIM_THREADS + K1;
                    h[u] = P.tail_in[ti < 0 ? 0 : (ti >= K1 ? (K1 > 0 ? K1 - 1 : 0) : ti)];
                }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int e = t + u * DD_DECIM_THREADS;
                if (e < S) {
                    const int64_t n = ns + e;
                    float2 v = x[u];
                    if (P.flags & DD_CHAIN_NCO) v = dd_cmul(v, dd_cmul(w2[e >> 6], w1));
                    // history samples are already rotated; before the history and past the chunk: zeros
                    if (n < 0) {
                        if (u < NT) v = (n + K1 >= 0) ? h[u] : make_float2(0.f, 0.f);
                        else v = (n + K1 >= 0) ? P.tail_in[n + K1] : make_float2(0.f, 0.f);     // (filters longer than 513 taps)
                    }
                    if (n >= P.L) v = make_float2(0.f, 0.f);
                    sx[e] = v;
                }
            }
            for (int e = t + NB * DD_DECIM_THREADS; e < S; e += DD_DECIM_THREADS) {     // (spans beyond DD_DECIM_SPAN_MAX: long filters)
                float2 ph = make_float2(1.f, 0.f);
                if (P.flags & DD_CHAIN_NCO) ph = dd_cmul(w2[e >> 6], w1);
                sx[e] = dd_load_sample(P, ns + e, ph);
            }
        }
    }
    __syncthreads();

    if (b == P.nblocks - 1 && P.tail_out) {
        for (int i = t; i < K - 1; i += DD_DECIM_THREADS) {
            const int64_t e = (P.L - (K - 1) + i) - ns;
            P.tail_out[i] = sx[(int)e];
        }
    }

    // Tap loop.  The LDS image is linear (no skew) on purpose: thread t's window starts at sample
    // t*M, so every read is base + immediate offset (ds_read2_b64) and a tap costs one packed FMA.
    // A skewed image avoided the 2-way bank conflict of the stride-M reads but cost ~4 vector
    // instructions of index arithmetic per tap.  Taps are read from their LDS copy: fetched from
    // global memory here they came as per-lane vector loads with a wait every 16 taps.
    float2 acc = make_float2(0.f, 0.f);
    const float* __restrict__ G = gl;                              // g[j] = h[K-1-j]
    if (t < T) {
        const float2* __restrict__ win = sx + t * M;
        int j = 0;
        if ((M & 1) == 0) {
            // even M: every window starts on a 16-byte boundary -> ds_read_b128 (two samples), whose
            // 16-lane groups make the stride-M reads bank-conflict free at 256 B/clk; as ds_read2_b64
            // (128 B/clk, 2-way conflicts) these reads kept the LDS array busy 70 % of the kernel
            // (PMC: SQ_LDS_IDX_ACTIVE, a third of it SQ_LDS_BANK_CONFLICT).
            const float4* __restrict__ win4 = reinterpret_cast<const float4*>(__builtin_assume_aligned(win, 16));
            const float4* __restrict__ G4 = reinterpret_cast<const float4*>(__builtin_assume_aligned(G, 16));
            for (; j + 8 <= K; j += 8) {
                const float4 x0 = win4[j / 2], x1 = win4[j / 2 + 1], x2 = win4[j / 2 + 2], x3 = win4[j / 2 + 3];
                const float4 c0 = G4[j / 4], c1 = G4[j / 4 + 1];
                acc.x = fmaf(c0.x, x0.x, acc.x); acc.y = fmaf(c0.x, x0.y, acc.y);
                acc.x = fmaf(c0.y, x0.z, acc.x); acc.y = fmaf(c0.y, x0.w, acc.y);
                acc.x = fmaf(c0.z, x1.x, acc.x); acc.y = fmaf(c0.z, x1.y, acc.y);
                acc.x = fmaf(c0.w, x1.z, acc.x); acc.y = fmaf(c0.w, x1.w, acc.y);
                acc.x = fmaf(c1.x, x2.x, acc.x); acc.y = fmaf(c1.x, x2.y, acc.y);
                acc.x = fmaf(c1.y, x2.z, acc.x); acc.y = fmaf(c1.y, x2.w, acc.y);
                acc.x = fmaf(c1.z, x3.x, acc.x); acc.y = fmaf(c1.z, x3.y, acc.y);
                acc.x = fmaf(c1.w, x3.z, acc.x); acc.y = fmaf(c1.w, x3.w, acc.y);
            }
        } else {
            for (; j + 8 <= K; j += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float2 v = win[j + u];
                    const float g = G[j + u];
                    acc.x = fmaf(g, v.x, acc.x);
                    acc.y = fmaf(g, v.y, acc.y);
                }
            }
        }
        for (; j < K; ++j) {
            const float2 v = win[j];
            const float g = G[j];
            acc.x = fmaf(g, v.x, acc.x);
            acc.y = fmaf(g, v.y, acc.y);
        }
    }
    const int64_t p = pfirst + t;
    const bool posts = P.seam_post && b == P.nblocks - 1;       // the tile that writes the state the next chunk of this launch reads
    if (!fm) {
        if (t < T && p < P.Ld) {
            reinterpret_cast<float2*>(P.out)[p] = acc;
            if (P.lasty_out && p == P.Ld - 1) *P.lasty_out = acc;
        }
        if (posts) dd_seam_post(P.seam_post);
        return;
    }
    if (p == -1) acc = *P.lasty_in;
    if (t < T) yblk[t] = acc;
    __syncthreads();
    if (t < T && p >= 0 && p < P.Ld) {
        if (t > 0) {
            reinterpret_cast<float*>(P.out)[p - P.s] = dd_fm_angle(acc, yblk[t - 1]);
        }
        if (p == P.Ld - 1) *P.lasty_out = acc;
    }
    if (posts) dd_seam_post(P.seam_post);
}

__global__ void __launch_bounds__(DD_DECIM_THREADS) k_chain_decim(const DDChainParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    dd_decim_edge_tile(P, (int)blockIdx.x, smem);
}

// ============================================================================
// decimating kernel, interior tiles, persistent
// ============================================================================
// One launch-long workgroup per third of a CU (LDS bound, 3 per CU) walks the interior tiles blockIdx.x, blockIdx.x + nwg,
// ... (round 4: the device then works on one moving window of nwg tiles, and the tile loads are non-temporal -- what
// tools/ubench/stream_2to1 found for streaming kernels: C3 / C4 one-chunk passes 0.110 / 0.113 -> 0.0985 / 0.102 ms = 0.69 / 0.67 of
// 8 TB/s, profiles/r04_decim_map.txt; -DDD_DECIM_CONTIG / -DDD_DECIM_NO_NT: a contiguous run per workgroup, plain loads) of the
// interior tiles: whole span inside the chunk, every output valid, complex64 input.  The next
// tile's samples are requested (12 x 16 B per lane) the moment the current tile has been staged
// into LDS -- into the same registers, which staging has just freed -- so they are in flight
// during the tap loop, the discriminator and both barriers, and the per-workgroup constants
// (taps in LDS, group and lane phasors) are built once per launch instead of once per tile.  NCO phase is TILE-RELATIVE: the factor exp(-j w (abs0 + ns)) is
// common to a tile, so it cancels in the discriminator (a tile recomputes the output before
// its first one, pairs never straddle tiles) and is applied to the FIR output only for
// complex output.
#define DD_DECIM_NV ((DD_DECIM_SPAN_MAX / 2 + DD_DECIM_THREADS - 1) / DD_DECIM_THREADS)
typedef float dd_v4f_a8 __attribute__((ext_vector_type(4), aligned(8)));     // 16-byte load on an 8-byte boundary

__device__ __forceinline__ void dd_decim_issue(const DDChainParams& P, int b, int nq, int t, dd_v4f_a8 (&v)[DD_DECIM_NV]) {
    const int64_t ns = (int64_t)P.off + dd_tile_pfirst(P, b) * P.M - (P.K - 1);
    const float2* __restrict__ src = reinterpret_cast<const float2*>(P.in) + ns;      // wave uniform
#pragma unroll
    for (int u = 0; u < DD_DECIM_NV; ++u) {
        int q = t + u * DD_DECIM_THREADS;
        q = q < nq ? q : nq - 1;                          // past the span: harmless re-read, never used
#ifndef DD_DECIM_NO_NT
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const dd_v4f_a8*>(src + 2 * q));
#else
        v[u] = *reinterpret_cast<const dd_v4f_a8*>(src + 2 * q);
#endif
    }
}

// raw u8 input (DD_CHAIN_U8_INPUT, 2 B/sample): a 16-byte load carries eight samples (I0 Q0 I1 Q1 ...)
#define DD_DECIM_NV8 ((DD_DECIM_SPAN_MAX / 8 + DD_DECIM_THREADS - 1) / DD_DECIM_THREADS)
typedef uint32_t dd_v4u_a4 __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load on a 4-byte boundary

__device__ __forceinline__ void dd_decim_issue_u8(const DDChainParams& P, int b, int nq8, int t, dd_v4u_a4 (&v)[DD_DECIM_NV8]) {
    const int64_t ns = (int64_t)P.off + dd_tile_pfirst(P, b) * P.M - (P.K - 1);
    const unsigned char* __restrict__ src = reinterpret_cast<const unsigned char*>(P.in) + 2 * ns;   // wave uniform
#pragma unroll
    for (int u = 0; u < DD_DECIM_NV8; ++u) {
        int q = t + u * DD_DECIM_THREADS;
        q = q < nq8 ? q : nq8 - 1;
#ifndef DD_DECIM_NO_NT
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const dd_v4u_a4*>(src + 16 * q));
#else
        v[u] = *reinterpret_cast<const dd_v4u_a4*>(src + 16 * q);
#endif
    }
}

// stage one tile of raw u8 samples: widen (source.py:117-118: value - 127.5), rotate, write the LDS image
__device__ __forceinline__ void dd_decim_stage_u8(const DDChainParams& P, int nq8, int t, const dd_v4u_a4 (&v)[DD_DECIM_NV8],
                                                  float2* sx, const float2* w2, const float2 (&w1)[8]) {
    const bool nco = (P.flags & DD_CHAIN_NCO) != 0;
#pragma unroll
    for (int u = 0; u < DD_DECIM_NV8; ++u) {
        const int q = t + u * DD_DECIM_THREADS;
        if (q < nq8) {
            const float2 g = nco ? w2[(8 * q) >> 6] : make_float2(1.f, 0.f);
            const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float2 xa = make_float2((float)(d[k] & 0xff) - 127.5f, (float)((d[k] >> 8) & 0xff) - 127.5f);
                float2 xb = make_float2((float)((d[k] >> 16) & 0xff) - 127.5f, (float)(d[k] >> 24) - 127.5f);
                if (nco) {
                    xa = dd_cmul(xa, dd_cmul(g, w1[2 * k]));
                    xb = dd_cmul(xb, dd_cmul(g, w1[2 * k + 1]));
                }
                *reinterpret_cast<float4*>(sx + 8 * q + 2 * k) = make_float4(xa.x, xa.y, xb.x, xb.y);
            }
        }
    }
}

template <bool U8>
__device__ __forceinline__ void dd_decim_tile(const DDChainParams& P, int b, const DDChainParams& Pn, int b_next, int nq, int t, dd_v4f_a8 (&v)[DD_DECIM_NV],
                                              dd_v4u_a4 (&v8)[DD_DECIM_NV8], const float2 (&w18)[8],
                                              float2* sx, const float2* w2, float2* yblk, const float* gl, float2 w1a, float2 w1b) {
    const int K = P.K, M = P.M, T = P.T;
    const bool nco = (P.flags & DD_CHAIN_NCO) != 0, fm = (P.flags & DD_CHAIN_FM) != 0;
    const int64_t pfirst = dd_tile_pfirst(P, b);
    float2 tilew = make_float2(1.f, 0.f);
    if (nco && !fm) {                                      // absolute phase of the tile start (table fetch issued early)
        const int64_t ns = (int64_t)P.off + pfirst * M - (K - 1);
        tilew = dd_phasor((uint64_t)(P.abs0 + ns) * P.cyc, P.nco_tbl);
    }
    if (U8) {
        dd_decim_stage_u8(P, nq, t, v8, sx, w2, w18);      // (nq counts octets here)
        if (b_next >= 0) dd_decim_issue_u8(Pn, b_next, nq, t, v8);
    } else {
        float2 g[DD_DECIM_NV];
        if (nco) {
#pragma unroll
            for (int u = 0; u < DD_DECIM_NV; ++u) {
                const int q = t + u * DD_DECIM_THREADS;
                g[u] = w2[(q < nq ? 2 * q : 0) >> 6];
            }
        }
#pragma unroll
        for (int u = 0; u < DD_DECIM_NV; ++u) {
            const int q = t + u * DD_DECIM_THREADS;
            if (q < nq) {
                float2 xa = make_float2(v[u].x, v[u].y), xb = make_float2(v[u].z, v[u].w);
                if (nco) {
                    xa = dd_cmul(xa, dd_cmul(g[u], w1a));
                    xb = dd_cmul(xb, dd_cmul(g[u], w1b));
                }
                *reinterpret_cast<float4*>(sx + 2 * q) = make_float4(xa.x, xa.y, xb.x, xb.y);
            }
        }
        if (b_next >= 0) dd_decim_issue(Pn, b_next, nq, t, v); // next tile (of this chunk or, in a multi-chunk launch, the next): in flight from here to the next staging
    }
    __syncthreads();
    float2 acc = make_float2(0.f, 0.f);
    if (t < T) {
        const float2* __restrict__ win = sx + t * M;
        int j = 0;
#ifndef DD_DECIM_LDS_TAPS                                  // (-DDD_DECIM_LDS_TAPS: the taps from LDS, one multiply-add per component, as in rounds 1-4)
        if ((M & 1) == 0) {
            // the taps are wave uniform: they come through the scalar cache (a third of the tap loop's LDS reads were theirs), and a
            // multiply-add handles re and im at once; two partial sums
            typedef float dd_v2f __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(4))) float* dd_cfp;
            const dd_cfp Gs = (dd_cfp)(P.taps_rev + (DD_DENSE_R - 1));
            const float4* __restrict__ win4 = reinterpret_cast<const float4*>(__builtin_assume_aligned(win, 16));
            dd_v2f a0 = (dd_v2f){0.f, 0.f}, a1 = (dd_v2f){0.f, 0.f};
            for (; j + 8 <= K; j += 8) {
                const float4 x0 = win4[j / 2], x1 = win4[j / 2 + 1], x2 = win4[j / 2 + 2], x3 = win4[j / 2 + 3];
                const float c0 = Gs[j], c1 = Gs[j + 1], c2 = Gs[j + 2], c3 = Gs[j + 3], c4 = Gs[j + 4], c5 = Gs[j + 5], c6 = Gs[j + 6], c7 = Gs[j + 7];
                a0 = __builtin_elementwise_fma((dd_v2f){c0, c0}, (dd_v2f){x0.x, x0.y}, a0);
                a1 = __builtin_elementwise_fma((dd_v2f){c1, c1}, (dd_v2f){x0.z, x0.w}, a1);
                a0 = __builtin_elementwise_fma((dd_v2f){c2, c2}, (dd_v2f){x1.x, x1.y}, a0);
                a1 = __builtin_elementwise_fma((dd_v2f){c3, c3}, (dd_v2f){x1.z, x1.w}, a1);
                a0 = __builtin_elementwise_fma((dd_v2f){c4, c4}, (dd_v2f){x2.x, x2.y}, a0);
                a1 = __builtin_elementwise_fma((dd_v2f){c5, c5}, (dd_v2f){x2.z, x2.w}, a1);
                a0 = __builtin_elementwise_fma((dd_v2f){c6, c6}, (dd_v2f){x3.x, x3.y}, a0);
                a1 = __builtin_elementwise_fma((dd_v2f){c7, c7}, (dd_v2f){x3.z, x3.w}, a1);
            }
            acc.x = a0.x + a1.x; acc.y = a0.y + a1.y;
        }
#else
        if ((M & 1) == 0) {                                // 16-byte aligned windows: conflict-free ds_read_b128 (see k_chain_decim)
            const float4* __restrict__ win4 = reinterpret_cast<const float4*>(__builtin_assume_aligned(win, 16));
            const float4* __restrict__ G4 = reinterpret_cast<const float4*>(__builtin_assume_aligned(gl, 16));
            for (; j + 8 <= K; j += 8) {
                const float4 x0 = win4[j / 2], x1 = win4[j / 2 + 1], x2 = win4[j / 2 + 2], x3 = win4[j / 2 + 3];
                const float4 c0 = G4[j / 4], c1 = G4[j / 4 + 1];
                acc.x = fmaf(c0.x, x0.x, acc.x); acc.y = fmaf(c0.x, x0.y, acc.y);
                acc.x = fmaf(c0.y, x0.z, acc.x); acc.y = fmaf(c0.y, x0.w, acc.y);
                acc.x = fmaf(c0.z, x1.x, acc.x); acc.y = fmaf(c0.z, x1.y, acc.y);
                acc.x = fmaf(c0.w, x1.z, acc.x); acc.y = fmaf(c0.w, x1.w, acc.y);
                acc.x = fmaf(c1.x, x2.x, acc.x); acc.y = fmaf(c1.x, x2.y, acc.y);
                acc.x = fmaf(c1.y, x2.z, acc.x); acc.y = fmaf(c1.y, x2.w, acc.y);
                acc.x = fmaf(c1.z, x3.x, acc.x); acc.y = fmaf(c1.z, x3.y, acc.y);
                acc.x = fmaf(c1.w, x3.z, acc.x); acc.y = fmaf(c1.w, x3.w, acc.y);
            }
        }
#endif
        for (; j + 8 <= K; j += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float2 x = win[j + u];
                const float c = gl[j + u];
                acc.x = fmaf(c, x.x, acc.x);
                acc.y = fmaf(c, x.y, acc.y);
            }
        }
        for (; j < K; ++j) {
            const float2 x = win[j];
            const float c = gl[j];
            acc.x = fmaf(c, x.x, acc.x);
            acc.y = fmaf(c, x.y, acc.y);
        }
    }
    const int64_t p = pfirst + t;
    if (!fm) {
        if (t < T) reinterpret_cast<float2*>(P.out)[p] = dd_cmul(acc, tilew);
        __syncthreads();                                   // tap loop done everywhere before the image is overwritten
        return;
    }
    if (t < T) yblk[t] = acc;
    __syncthreads();                                       // (also: tap loop done everywhere)
    if (t > 0 && t < T) reinterpret_cast<float*>(P.out)[p - P.s] = dd_fm_angle(acc, yblk[t - 1]);
}

template <bool U8>
__global__ void __launch_bounds__(DD_DECIM_THREADS, 3) k_chain_decim_p(const DDChainParams P, int b_lo, int b_hi, int nwg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // workgroups [nwg, gridDim.x): one edge tile each (stream start with the carried history, chunk end with the
    // new tail / last sample, partial tiles) -- resident beside the persistent ones from the start, so they cost
    // neither a launch of their own nor a tail after the persistent loop
    if ((int)blockIdx.x >= nwg) {
        dd_decim_edge_tile(P, (int)blockIdx.x - nwg, smem);
        return;
    }
    const int K = P.K, M = P.M, T = P.T;
    // span, rounded up to whole load granules: sample pairs (complex64) or octets (u8)
    const int S = U8 ? (((T - 1) * M + K + (M - 1) + 7) & ~7) : (((T - 1) * M + K + (M - 1) + 1) & ~1);
    const int nq = U8 ? S / 8 : S / 2;
    float2* sx = reinterpret_cast<float2*>(smem);
    float2* w2 = sx + S + 4;
    float2* yblk = w2 + (S / 64 + 2);
    float* gl = reinterpret_cast<float*>(smem + ((sizeof(float2) * ((size_t)S + 4 + (S / 64 + 2) + DD_DECIM_THREADS) + 15) & ~(size_t)15));   // (LDS offset arithmetic: see k_chain_decim)
    const int t = threadIdx.x;
    // contiguous run of tiles per workgroup, and per XCD (workgroups are dealt round-robin to the 8 XCDs)
    const int n = b_hi - b_lo;
#ifndef DD_DECIM_CONTIG
    // tiles blockIdx.x, blockIdx.x + nwg, ...: the device works on one moving window of nwg tiles (tools/ubench/stream_2to1)
    const int begin = b_lo + (int)blockIdx.x, end = b_hi, step = nwg;
#else
    const int wg = (nwg % 8 == 0) ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int begin = b_lo + (int)(((int64_t)wg * n) / nwg), end = b_lo + (int)(((int64_t)(wg + 1) * n) / nwg), step = 1;
#endif
    (void)n;
    if (begin >= end) return;

    dd_v4f_a8 v[DD_DECIM_NV];
    dd_v4u_a4 v8[DD_DECIM_NV8];
    if (U8) dd_decim_issue_u8(P, begin, nq, t, v8);        // in flight while the constants are built
    else dd_decim_issue(P, begin, nq, t, v);
    for (int j = t; j < ((K + 7) & ~7); j += DD_DECIM_THREADS) gl[j] = j < K ? P.taps_rev[(DD_DENSE_R - 1) + j] : 0.f;
    float2 w1a = make_float2(1.f, 0.f), w1b = make_float2(1.f, 0.f);
    if (P.flags & DD_CHAIN_NCO) {
        for (int g = t; g < S / 64 + 1; g += DD_DECIM_THREADS) w2[g] = dd_phasor((uint64_t)g * 64 * P.cyc, P.nco_tbl);
        w1a = dd_phasor((uint64_t)((2 * t) & 63) * P.cyc, P.nco_tbl);
        w1b = dd_phasor((uint64_t)(((2 * t) & 63) + 1) * P.cyc, P.nco_tbl);
    }
    float2 w18[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        w18[k] = (U8 && (P.flags & DD_CHAIN_NCO)) ? dd_phasor((uint64_t)(((8 * t) & 63) + k) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
    __syncthreads();
    for (int b = begin; b < end; b += step) dd_decim_tile<U8>(P, b, P, b + step < end ? b + step : -1, nq, t, v, v8, w18, sx, w2, yblk, gl, w1a, w1b);
}

// ---- several chunks, ONE launch (dd_chain_process_chunks) -------------------------------------------------------------
// The reference's chunk loops (decode_fm.py:54-70: 2^22-sample chunks; decode_noaa.py:614-624) make one call per chunk;
// a launch per chunk costs ~7 us of kernel boundary, fill and drain around ~15 us of work (C3).  Here the persistent
// workgroups walk the CONCATENATED list of every chunk's interior tiles, and every chunk's edge tiles (its first, with
// the carried history, and its last, which writes the new state) ride along as trailing workgroups.  Each tile is
// computed exactly as in the chunk's own launch -- same tile grid per chunk, same tile-relative constants -- so the
// outputs are bit-identical to the loop's; the state a chunk hands to the next travels through device memory behind
// the seam flags above.  Per chunk: its full parameter block (edge tiles) and the few fields an interior tile needs.
struct DDSeg {
    const void* in;
    void* out;
    int64_t abs0;
    int off, s, lo, pad;
};
template <bool U8>
__global__ void __launch_bounds__(DD_DECIM_THREADS, 3) k_chain_decim_multi(const DDChainParams* __restrict__ Pc, const DDSeg* __restrict__ seg,
                                                                           const int* __restrict__ ipre, const int* __restrict__ epre, int nchunks, int nwg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= nwg) {
        const int e = (int)blockIdx.x - nwg;
        int c = 0;
        while (c + 1 < nchunks && e >= epre[c + 1]) ++c;               // (uniform)
        const DDChainParams P = Pc[c];
        dd_decim_edge_tile(P, e - epre[c], smem);
        return;
    }
    DDChainParams P = Pc[0];                                           // K, M, T, flags, cyc, taps and tables are the same for every chunk
    const int K = P.K, M = P.M, T = P.T;
    const int S = U8 ? (((T - 1) * M + K + (M - 1) + 7) & ~7) : (((T - 1) * M + K + (M - 1) + 1) & ~1);
    const int nq = U8 ? S / 8 : S / 2;
    float2* sx = reinterpret_cast<float2*>(smem);
    float2* w2 = sx + S + 4;
    float2* yblk = w2 + (S / 64 + 2);
    float* gl = reinterpret_cast<float*>(smem + ((sizeof(float2) * ((size_t)S + 4 + (S / 64 + 2) + DD_DECIM_THREADS) + 15) & ~(size_t)15));
    const int t = threadIdx.x;
    const int n = ipre[nchunks];
#ifndef DD_DECIM_CONTIG
    const int begin = (int)blockIdx.x, end = n, step = nwg;
#else
    const int wg = (nwg % 8 == 0) ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int begin = (int)(((int64_t)wg * n) / nwg), end = (int)(((int64_t)(wg + 1) * n) / nwg), step = 1;      // global interior tile indices
#endif
    if (begin >= end) return;
    int c = 0;
    while (begin >= ipre[c + 1]) ++c;
    auto load_seg = [&](DDChainParams& Q, int cc) {
        const DDSeg g = seg[cc];
        Q.in = g.in; Q.out = g.out; Q.abs0 = g.abs0; Q.off = g.off; Q.s = g.s;
        return g.lo;
    };
    int lo = load_seg(P, c);
    DDChainParams Pn = P;
    dd_v4f_a8 v[DD_DECIM_NV];
    dd_v4u_a4 v8[DD_DECIM_NV8];
    if (U8) dd_decim_issue_u8(P, lo + (begin - ipre[c]), nq, t, v8);
    else dd_decim_issue(P, lo + (begin - ipre[c]), nq, t, v);
    for (int j = t; j < ((K + 7) & ~7); j += DD_DECIM_THREADS) gl[j] = j < K ? P.taps_rev[(DD_DENSE_R - 1) + j] : 0.f;
    float2 w1a = make_float2(1.f, 0.f), w1b = make_float2(1.f, 0.f);
    if (P.flags & DD_CHAIN_NCO) {
        for (int g = t; g < S / 64 + 1; g += DD_DECIM_THREADS) w2[g] = dd_phasor((uint64_t)g * 64 * P.cyc, P.nco_tbl);
        w1a = dd_phasor((uint64_t)((2 * t) & 63) * P.cyc, P.nco_tbl);
        w1b = dd_phasor((uint64_t)(((2 * t) & 63) + 1) * P.cyc, P.nco_tbl);
    }
    float2 w18[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        w18[k] = (U8 && (P.flags & DD_CHAIN_NCO)) ? dd_phasor((uint64_t)(((8 * t) & 63) + k) * P.cyc, P.nco_tbl) : make_float2(1.f, 0.f);
    __syncthreads();
    for (int g = begin; g < end; g += step) {
        const int b = lo + (g - ipre[c]);
        int b_next = -1, cn = c, lon = lo;
        if (g + step < end) {
            if (g + step >= ipre[c + 1]) {                            // the next tile opens the next chunk with an interior run
                do { ++cn; } while (g + step >= ipre[cn + 1]);
                lon = load_seg(Pn, cn);
            }
            b_next = lon + (g + step - ipre[cn]);
        }
        dd_decim_tile<U8>(P, b, Pn, b_next, nq, t, v, v8, w18, sx, w2, yblk, gl, w1a, w1b);
        if (cn != c) { P = Pn; c = cn; lo = lon; }
    }
}

// rare path (chunk without a kept sample) and shard priming: new tail only
__global__ void k_tail_update(const DDChainParams P) {
    const int K = P.K;
    for (int i = threadIdx.x; i < K - 1; i += blockDim.x) {
        const int64_t n = P.L - (K - 1) + i;
        float2 ph = make_float2(1.f, 0.f);
        if (P.flags & DD_CHAIN_NCO) ph = dd_phasor((uint64_t)(P.abs0 + n) * P.cyc, P.nco_tbl);
        P.tail_out[i] = dd_load_sample(P, n, ph);
    }
}

__global__ void k_fill_c64(float2* p, int n, float re, float im) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = make_float2(re, im);
}

// ============================================================================
// host side
// ============================================================================
// MFMA path hooks (dd_mfma.hip)
int dd_kernel_sel_decimp(void);     // (dd_mfma.hip) dd_debug_select_kernel("decimp"): the tile kernels for M > 1
int dd_mfma_supported(int K, int M, int flags);
int dd_mfma_create(void** st, const double* taps, int K);
void dd_mfma_destroy(void* st);
int dd_mfma_launch(void* st, const DDChainParams& P, hipStream_t s, int* kernel_id);

// ---------------------------------------------------------------- dd_fir (taps + history)
// ---- chunk-list launches: hand-overs that timed out (dd_seam_wait) become DD_ERR_TIMEOUT --------------------------------
static std::mutex g_seam_mu;
static std::vector<dd_fir*> g_seam_pending;          // filters that have made a chunk-list launch (their error word is looked at until they are destroyed)
static int g_seam_withhold = -1, g_seam_spin_log2 = 0;

extern "C" int dd_debug_seam(int withhold_chunk, int spin_log2) {
    std::lock_guard<std::mutex> lk(g_seam_mu);
    g_seam_withhold = withhold_chunk;
    g_seam_spin_log2 = spin_log2;
    return DD_OK;
}
// look at one filter's error word; caller holds g_seam_mu.  Round 5: the word is pinned host memory the kernel counts into directly
// (system-scope atomic, taken only by a wait that gives up) -- a chunk-list call used to carry a device-to-host copy of the count and
// an event behind every launch, two stream operations of ~3 us each around a 100 us kernel.  A look while the launch still runs may be
// early; the word is final once its stream has been synchronised, and every later fused launch and dd_stream_sync looks again.
static int seam_look(dd_fir* f, bool) {
    if (!f->seam_pending || !f->seam_err_host) return DD_OK;
    const unsigned n = *reinterpret_cast<volatile unsigned int*>(f->seam_err_host);
    if (n == 0) return DD_OK;
    f->state_invalid = 1;          // (the faulty launch also committed its carried state: nothing may continue from it)
    *reinterpret_cast<volatile unsigned int*>(f->seam_err_host) = 0;
    dd_set_error("chunk-list launch: %u in-launch hand-over wait(s) of the carried FIR / FM state timed out; the outputs of that "
                 "dd_*_process_chunks call are invalid (run the chunks one by one, or raise the bound with dd_debug_seam)", n);
    return DD_ERR_TIMEOUT;
}
int dd_seam_poll_all(void) {
    std::lock_guard<std::mutex> lk(g_seam_mu);
    int rc = DD_OK;
    const std::vector<dd_fir*> firs = g_seam_pending;
    for (dd_fir* f : firs) {
        const int r = seam_look(f, false);
        if (r != DD_OK) rc = r;
    }
    return rc;
}
static void seam_forget(dd_fir* f) {
    std::lock_guard<std::mutex> lk(g_seam_mu);
    for (size_t i = 0; i < g_seam_pending.size(); ++i)
        if (g_seam_pending[i] == f) { g_seam_pending.erase(g_seam_pending.begin() + i); break; }
    if (f->seam_err_host) { (void)hipDeviceSynchronize(); (void)hipHostFree(f->seam_err_host); }      // (no launch may still count into it)
    f->seam_ev = nullptr; f->seam_err_host = nullptr; f->seam_err = nullptr; f->seam_pending = 0;
}

extern "C" int dd_fir_create(dd_fir** h, const double* taps, int ntaps) {
    DD_REQUIRE(h && taps, "null argument");
    DD_REQUIRE(ntaps >= 1 && ntaps <= 4096, "ntaps must be in [1, 4096]");
    if (!dd_nco_table()) {
        dd_set_error("no usable GPU: the HIP path is mandatory (there is no CPU fallback)");
        return DD_ERR_NODEVICE;
    }
    dd_fir* f = new dd_fir();
    f->K = ntaps;
    f->taps.assign(taps, taps + ntaps);
    f->taps_rev = nullptr;
    f->tail[0] = f->tail[1] = nullptr;
    f->tail_const[0] = f->tail_const[1] = nullptr;
    f->tail_override = nullptr;
    f->parity = 0;
    f->mfma = nullptr;
    f->mfma_tried = 0;
    f->taps_dev = nullptr;
    f->hist[0] = f->hist[1] = nullptr;
    f->hpar = 0;
    f->hist_mode = DD_HIST_ONES;
    f->last_kernel = DD_KERNEL_NONE;
    f->launches = 0;
    f->dw_taps.dev = nullptr;
    f->dw_taps.key = -1;
    f->multi = nullptr;
    f->multi_bytes = 0;
    f->seam_err = nullptr;
    f->seam_err_host = nullptr;
    f->seam_ev = nullptr;
    f->seam_pending = 0;
    f->state_invalid = 0;
    const int R = DD_DENSE_R;
    const int K = ntaps;
    // G[i] = g[i-(R-1)], g[j] = h[K-1-j]; zero padded so every R-block read is in range
    const int niter = (K + R - 1 + R - 1) / R;
    const int len = niter * R + 6 * R;                    // (k_chain_decim_w reads whole trips of 32 taps: zeros behind the last one)
    std::vector<float> g(len, 0.f);
    for (int j = 0; j < K; ++j) g[j + R - 1] = (float)taps[K - 1 - j];
    const size_t tb = sizeof(float2) * (size_t)(K > 1 ? K - 1 : 1);
    hipError_t e = hipMalloc((void**)&f->taps_rev, len * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(f->taps_rev, g.data(), len * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&f->tail[0], tb);
    if (e == hipSuccess) e = hipMalloc((void**)&f->tail[1], tb);
    if (e == hipSuccess) e = hipMalloc((void**)&f->tail_const[0], tb);
    if (e == hipSuccess) e = hipMalloc((void**)&f->tail_const[1], tb);
    if (e == hipSuccess) {
        const int nc = K > 1 ? K - 1 : 1;
        hipLaunchKernelGGL(k_fill_c64, dim3((nc + 255) / 256), dim3(256), 0, 0, f->tail_const[0], nc, 0.f, 0.f);
        hipLaunchKernelGGL(k_fill_c64, dim3((nc + 255) / 256), dim3(256), 0, 0, f->tail_const[1], nc, 1.f, 0.f);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        dd_fir_destroy(f);
        dd_set_error("dd_fir_create: %s", hipGetErrorString(e));
        return DD_ERR_HIP;
    }
    int rc = dd_fir_reset(f, DD_HIST_ONES, nullptr, nullptr);
    if (rc != DD_OK) {
        dd_fir_destroy(f);
        return rc;
    }
    DD_HIP_CHECK(hipDeviceSynchronize());
    *h = f;
    return DD_OK;
}

extern "C" int dd_fir_destroy(dd_fir* f) {
    if (!f) return DD_OK;
    if (f->mfma) dd_mfma_destroy(f->mfma);
    seam_forget(f);
    (void)hipFree(f->multi);
    (void)hipFree(f->dw_taps.dev);
    (void)hipFree(f->taps_rev);
    (void)hipFree(f->tail[0]);
    (void)hipFree(f->tail[1]);
    (void)hipFree(f->tail_const[0]);
    (void)hipFree(f->tail_const[1]);
    (void)hipFree(f->taps_dev);
    (void)hipFree(f->hist[0]);
    (void)hipFree(f->hist[1]);
    delete f;
    return DD_OK;
}

int dd_fir_reset_f64(dd_fir* f, int mode, const float* hist_host, hipStream_t s);   // dd_fir.hip

extern "C" int dd_fir_reset(dd_fir* f, int mode, const float* hist_host, void* stream) {
    DD_REQUIRE(f, "h");
    DD_REQUIRE(mode == DD_HIST_ZEROS || mode == DD_HIST_ONES || mode == DD_HIST_GIVEN, "mode");
    hipStream_t s = dd_stream(stream);
    const int n = f->K - 1;
    if (n > 0) {
        if (mode == DD_HIST_GIVEN) {
            DD_REQUIRE(hist_host, "hist_host");
            DD_HIP_CHECK(hipMemcpyAsync(f->tail[f->parity], hist_host, sizeof(float2) * n, hipMemcpyHostToDevice, s));
            DD_HIP_CHECK(hipStreamSynchronize(s));
            f->tail_override = nullptr;
        } else {
            // DD_HIST_ONES: lfilter_zi(b,[1]) unscaled == history of 1.0+0j (filters.py:45, quirk Q1).
            // No launch: the next kernel simply reads the constant history buffer.
            f->tail_override = f->tail_const[mode == DD_HIST_ONES ? 1 : 0];
        }
    }
    f->hist_mode = mode;
    f->state_invalid = 0;
    f->launches = 0;
    return dd_fir_reset_f64(f, mode, hist_host, s);
}

// ---------------------------------------------------------------- dd_fm
extern "C" int dd_fm_create(dd_fm** h) {
    DD_REQUIRE(h, "h");
    dd_fm* f = new dd_fm();
    f->last = nullptr;
    f->parity = 0;
    f->has_last = 0;
    hipError_t e = hipMalloc((void**)&f->last, 2 * sizeof(float2));
    if (e == hipSuccess) e = hipMemset(f->last, 0, 2 * sizeof(float2));
    if (e != hipSuccess) {
        delete f;
        dd_set_error("dd_fm_create: %s", hipGetErrorString(e));
        return (e == hipErrorNoDevice) ? DD_ERR_NODEVICE : DD_ERR_HIP;
    }
    *h = f;
    return DD_OK;
}
extern "C" int dd_fm_destroy(dd_fm* h) {
    if (h) {
        (void)hipFree(h->last);
        delete h;
    }
    return DD_OK;
}
extern "C" int dd_fm_reset(dd_fm* h) {
    DD_REQUIRE(h, "h");
    h->has_last = 0;
    return DD_OK;
}

// ---------------------------------------------------------------- fused launch
static inline int64_t kept_count(int64_t L, int off, int M) {
    return (L > off) ? (L - off + M - 1) / M : 0;
}

int64_t dd_fused_out_count(const dd_fm* fm, int64_t n, int M, int off) {
    const int64_t Ld = kept_count(n, off, M);
    if (!fm) return Ld;
    const int64_t no = Ld - (fm->has_last ? 0 : 1);
    return no > 0 ? no : 0;
}

// geometry of one chunk through the decimating kernels (M > 1): tile size, tile count, LDS, and -- when the chunk has an
// interior run worth a persistent grid -- that run [lo, hi) (P.skip_lo / skip_hi) and the workgroups a CU holds
struct DDDecimPlan {
    size_t lds, lds_p;
    bool persistent;
    int lo, hi, per_cu;
};
static int decim_plan(DDChainParams& P, DDDecimPlan& pl) {
    const bool isfm = (P.flags & DD_CHAIN_FM) != 0;
    int T = (DD_DECIM_SPAN_MAX - P.K - (P.M - 1)) / P.M + 1;
    if (T > DD_DECIM_THREADS) T = DD_DECIM_THREADS;
    if (T < 2) T = 2;
    P.T = T;
    P.nblocks = isfm ? (int)((P.Ld - P.s + (P.T - 2)) / (P.T - 1)) : (int)((P.Ld + P.T - 1) / P.T);
    if (P.nblocks < 1) P.nblocks = 1;
    const int S = (T - 1) * P.M + P.K + (P.M - 1);
    const int SP = S + 4;
    const size_t lds = sizeof(float2) * ((size_t)SP + (S + 63) / 64 + 1 + DD_DECIM_THREADS) + sizeof(float) * ((P.K + 7) & ~7) + 16;
    DD_REQUIRE(lds <= 160 * 1024, "filter/decimation too large for the decimating kernel's LDS tile");
    if (lds > 64 * 1024)
        DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_decim, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    pl.lds = lds;
    pl.lds_p = lds;
    pl.persistent = false;
    pl.per_cu = 1;
    // interior tiles [b_lo, b_hi): span (rounded up to a sample pair) inside the chunk, all T outputs valid,
    // complex64 input, not the chunk's last tile (that one writes the carried state)
    P.skip_lo = P.skip_hi = P.nblocks;
    pl.lo = pl.hi = P.nblocks;
    const bool u8in = (P.flags & DD_CHAIN_U8_INPUT) != 0;
    const int S2 = u8in ? ((S + 7) & ~7) : ((S + 1) & ~1);
    if (S2 <= DD_DECIM_SPAN_MAX && (reinterpret_cast<uintptr_t>(P.in) & (u8in ? 3 : 7)) == 0) {
        const int64_t adv = isfm ? (P.T - 1) : P.T;                  // outputs a tile advances by
        const int64_t pf0 = isfm ? (int64_t)P.s - 1 : 0;             // pfirst of tile 0
        // ns(b) = off + (pf0 + b adv) M - (K-1) >= 0 ;  ns(b) + S2 <= L ;  pf0 + b adv >= 0 ;  pf0 + b adv + T <= Ld
        int64_t lo = 0;
        while (lo < P.nblocks && ((int64_t)P.off + (pf0 + lo * adv) * P.M - (P.K - 1) < 0 || pf0 + lo * adv < 0)) ++lo;
        int64_t hi = P.nblocks - 1;                                  // exclusive bound candidates, walk down
        while (hi > lo && ((int64_t)P.off + (pf0 + (hi - 1) * adv) * P.M - (P.K - 1) + S2 > P.L || pf0 + (hi - 1) * adv + P.T > P.Ld)) --hi;
        if (hi - lo >= 64) {
            P.skip_lo = (int)lo;
            P.skip_hi = (int)hi;
            pl.lo = (int)lo;
            pl.hi = (int)hi;
            pl.persistent = true;
            const size_t lds_p0 = sizeof(float2) * ((size_t)S2 + 4 + S2 / 64 + 2 + DD_DECIM_THREADS) + sizeof(float) * ((P.K + 7) & ~7) + 16;
            pl.lds_p = lds_p0 > lds ? lds_p0 : lds;                  // the edge workgroups of the same launch need `lds`
            static DDOncePerDevice attr_p;
            if (attr_p.need()) {
                DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_decim_p<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_decim_p<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_decim_multi<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_decim_multi<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_p.mark();
            }
            // every workgroup must be resident from the start (a persistent grid with queued workgroups
            // runs in rounds): ask the runtime how many fit (LDS and registers)
            static std::mutex occ_mu;
            static size_t occ_lds[2] = {0, 0};          // the answer depends on (flavour, LDS size) only: asked once per change
            static int occ_val[2] = {0, 0};
            std::lock_guard<std::mutex> lk(occ_mu);
            int per_cu = occ_val[u8in];
            if (occ_lds[u8in] != pl.lds_p || per_cu < 1) {
                if ((u8in ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chain_decim_p<true>, DD_DECIM_THREADS, pl.lds_p)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chain_decim_p<false>, DD_DECIM_THREADS, pl.lds_p)) != hipSuccess || per_cu < 1) per_cu = 1;
                occ_lds[u8in] = pl.lds_p;
                occ_val[u8in] = per_cu;
            }
            static const char* wg_env = DD_TUNE_ENV("DD_DECIM_WGS_PER_CU");            // tools: fewer persistent workgroups per CU than fit
            if (wg_env && atoi(wg_env) >= 1 && atoi(wg_env) < per_cu) per_cu = atoi(wg_env);
            pl.per_cu = per_cu;
        }
    }
    return DD_OK;
}

int dd_fused_launch(dd_fir* fir, dd_fm* fm, const DDFusedArgs& a, int64_t* n_out, hipStream_t s) {
    DD_REQUIRE(fir && a.n >= 0 && a.M >= 1 && a.off >= 0 && a.off < a.M, "fused arguments");
    {
        // a chunk-list launch whose in-launch hand-over timed out: reported here too, not only by dd_stream_sync, and the
        // state that launch committed is refused until the filter is reset
        // (only THIS filter's word decides this call -- a timeout on another filter marks that filter, whose next call, or
        //  dd_stream_sync, reports it)
        int sr = DD_OK;
        {
            std::lock_guard<std::mutex> lk(g_seam_mu);
            sr = seam_look(fir, false);
        }
        if (sr != DD_OK) return sr;
        if (fir->state_invalid) {
            dd_set_error("this filter's carried state comes from a chunk-list launch that timed out (DD_ERR_TIMEOUT was reported): reset it "
                         "(dd_fir_reset / dd_chain_reset / dd_chain_seek) before processing more samples");
            return DD_ERR_TIMEOUT;
        }
    }
    DDChainParams P;
    memset(&P, 0, sizeof(P));
    P.in = a.in;
    P.out = a.out;
    P.tail_in = fir->tail_override ? fir->tail_override : fir->tail[fir->parity];
    P.tail_out = a.commit ? fir->tail[fir->parity ^ 1] : nullptr;
    P.taps_rev = fir->taps_rev;
    P.nco_tbl = dd_nco_table();
    P.cyc = a.cyc;
    P.abs0 = a.start_index;
    P.L = a.n;
    P.K = fir->K;
    P.M = a.M;
    P.off = a.off;
    P.Ld = kept_count(a.n, a.off, a.M);
    P.flags = (a.nco ? DD_CHAIN_NCO : 0) | (fm ? DD_CHAIN_FM : 0) | (a.u8 ? DD_CHAIN_U8_INPUT : 0) | (a.tight ? DD_CHAIN_TIGHT : 0);
    const bool isfm = fm != nullptr;
    P.s = (isfm && !fm->has_last) ? 1 : 0;
    if (isfm) {
        P.lasty_in = fm->last + fm->parity;
        P.lasty_out = fm->last + (fm->parity ^ 1);
    }
    const int64_t no = dd_fused_out_count(fm, a.n, a.M, a.off);
    if (n_out) *n_out = no;
    fir->last_kernel = DD_KERNEL_NONE;
    if (a.n == 0) return DD_OK;
    DD_REQUIRE(a.in, "in");
    DD_REQUIRE(a.out || no == 0, "out");

    if (P.Ld == 0) {
        // no kept sample in this chunk: only the FIR history moves on
        if (a.commit && fir->K > 1) {
            if (a.M > 1 && !dd_kernel_sel_decimp() && dd_decimw_supported(P.K, P.M, P.flags, P.in)) {
                // (k_chain_decim_w's history is ITS value of a sample after the NCO: the same arithmetic for a chunk without a kept sample)
                int rc = dd_decimw_launch(P, fir->taps_rev + (DD_DENSE_R - 1), fir->taps.data(), &fir->dw_taps, s);
                if (rc != DD_OK) return rc;
            } else {
                hipLaunchKernelGGL(k_tail_update, dim3(1), dim3(256), 0, s, P);
            }
            DD_LAUNCH_CHECK();
            fir->parity ^= 1;
            fir->tail_override = nullptr;
        }
        return DD_OK;
    }

    int use_mfma = 0;
    if (!a.force_direct && dd_mfma_supported(fir->K, a.M, P.flags)) {
        if (!fir->mfma && !fir->mfma_tried) {
            fir->mfma_tried = 1;
            if (dd_mfma_create(&fir->mfma, fir->taps.data(), fir->K) != DD_OK) fir->mfma = nullptr;
        }
        use_mfma = fir->mfma != nullptr;
    }
    if (use_mfma) {
        int rc = dd_mfma_launch(fir->mfma, P, s, &fir->last_kernel);
        if (rc != DD_OK) return rc;
    } else if (a.M == 1) {
        P.T = DD_DENSE_T;
        P.nblocks = isfm ? (int)((P.Ld - P.s + (P.T - 2)) / (P.T - 1)) : (int)((P.Ld + P.T - 1) / P.T);
        if (P.nblocks < 1) P.nblocks = 1;
        const int R = DD_DENSE_R;
        const int niter = (P.K + R - 1 + R - 1) / R;
        const int S = P.T + niter * R;
        const int SP = S + (S >> 3) + 8;
        const size_t lds = (size_t)SP * 2 * sizeof(float) + sizeof(float2) * ((S + 63) / 64 + 1) +
                           sizeof(float2) * DD_DENSE_THREADS;
        DD_REQUIRE(lds <= 160 * 1024, "filter too long for the dense kernel's LDS tile");
        if (lds > 64 * 1024)
            DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_dense, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_chain_dense, dim3(P.nblocks), dim3(DD_DENSE_THREADS), lds, s, P);
        DD_LAUNCH_CHECK();
        fir->last_kernel = DD_KERNEL_DENSE_F32;
    } else if (!dd_kernel_sel_decimp() && dd_decimw_supported(P.K, P.M, P.flags, P.in)) {
        // even M in [8, 64], up to 256 taps: one wave per row of 64 kept outputs on the absolute decimation grid (dd_decimw.hip)
        int kid = DD_KERNEL_DECIM_WAVE;
        int rc = dd_decimw_launch(P, fir->taps_rev + (DD_DENSE_R - 1), fir->taps.data(), &fir->dw_taps, s, &kid);
        if (rc != DD_OK) return rc;
        fir->last_kernel = kid;
    } else {
        DDDecimPlan pl;
        int rc = decim_plan(P, pl);
        if (rc != DD_OK) return rc;
        const bool u8in = (P.flags & DD_CHAIN_U8_INPUT) != 0;
        if (pl.persistent) {
            // the tiles around the interior run ride along as trailing workgroups of the same launch; the
            // persistent grid leaves them their slots
            const int n_rest = P.nblocks - (pl.hi - pl.lo);
            const int slots = dd_cu_count() * pl.per_cu;
            int grid = n_rest < slots / 2 ? slots - n_rest : slots / 2;
            if (grid > pl.hi - pl.lo) grid = pl.hi - pl.lo;
            if (grid >= 8) grid &= ~7;
            if (u8in) hipLaunchKernelGGL(k_chain_decim_p<true>, dim3(grid + n_rest), dim3(DD_DECIM_THREADS), pl.lds_p, s, P, pl.lo, pl.hi, grid);
            else hipLaunchKernelGGL(k_chain_decim_p<false>, dim3(grid + n_rest), dim3(DD_DECIM_THREADS), pl.lds_p, s, P, pl.lo, pl.hi, grid);
            DD_LAUNCH_CHECK();
            fir->last_kernel = DD_KERNEL_DECIM_PERSISTENT;
        } else {
            // no interior run (short chunk, unaligned input): every tile through the stand-alone edge kernel
            hipLaunchKernelGGL(k_chain_decim, dim3(P.nblocks), dim3(DD_DECIM_THREADS), pl.lds, s, P);
            DD_LAUNCH_CHECK();
            fir->last_kernel = DD_KERNEL_DECIM_TILES;
        }
    }
    ++fir->launches;
    if (a.commit) {
        fir->parity ^= 1;
        fir->tail_override = nullptr;
    }
    if (isfm) {
        fm->parity ^= 1;
        fm->has_last = 1;
    }
    return DD_OK;
}

extern "C" int dd_fused_process(dd_fir* fir, dd_fm* fm, const void* in, void* out, int64_t n,
                                int nco, uint64_t cycles_q64, int64_t start_index, int decim, int offset,
                                int flags, int carry, int64_t* n_out, void* stream) {
    DD_REQUIRE(fir, "fir");
    DDFusedArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in;
    a.out = out;
    a.n = n;
    a.nco = nco;
    a.cyc = cycles_q64;
    a.start_index = start_index;
    a.M = decim;
    a.off = offset;
    a.u8 = (flags & DD_CHAIN_U8_INPUT) ? 1 : 0;
    a.commit = carry;
    a.force_direct = (flags & DD_CHAIN_FORCE_DIRECT) ? 1 : 0;
    a.tight = (flags & DD_CHAIN_TIGHT) ? 1 : 0;
    return dd_fused_launch(fir, fm, a, n_out, dd_stream(stream));
}

// ---------------------------------------------------------------- stand-alone rows
extern "C" int dd_fir_c64(dd_fir* f, const float* in_c64, float* out_c64, int64_t n, int carry, void* stream) {
    DD_REQUIRE(f && n >= 0, "h/n");
    DDFusedArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in_c64;
    a.out = out_c64;
    a.n = n;
    a.M = 1;
    a.commit = carry;
    return dd_fused_launch(f, nullptr, a, nullptr, dd_stream(stream));
}

// out[j] = angle(x[j+s] * conj(x[j+s-1])), x[-1] = *last ; s = 1 if no previous sample
__global__ void __launch_bounds__(256) k_fm(const float2* __restrict__ in, float* __restrict__ out, int64_t n_out, int s,
                                            const float2* __restrict__ last, float2* __restrict__ last_out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_out; j += stride) {
        const int64_t p = j + s;
        const float2 cur = in[p];
        const float2 prv = (p == 0) ? *last : in[p - 1];
        out[j] = dd_fm_angle(cur, prv);
    }
    if (last_out && blockIdx.x == 0 && threadIdx.x == 0) *last_out = in[n - 1];
}

extern "C" int dd_fm_discrim_c64(dd_fm* h, const float* in_c64, float* out, int64_t n, int carry,
                                 int64_t* n_out, void* stream) {
    DD_REQUIRE(h && n >= 0, "h/n");
    // demod_fm.py:44/48 index sig[-1]: an empty chunk is an IndexError in the reference
    DD_REQUIRE(!(carry && n == 0), "empty chunk with storeState (IndexError in the reference)");
    const int s = (carry && h->has_last) ? 0 : 1;
    const int64_t no = n - s > 0 ? n - s : 0;
    if (n_out) *n_out = no;
    if (n == 0) return DD_OK;
    DD_REQUIRE(in_c64 && (out || no == 0), "null buffer");
    int64_t g = (no + 255) / 256;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_fm, dim3((unsigned)g), dim3(256), 0, dd_stream(stream), (const float2*)in_c64, out, no, s,
                       h->last + h->parity, carry ? h->last + (h->parity ^ 1) : (float2*)nullptr, n);
    DD_LAUNCH_CHECK();
    if (carry) {
        h->parity ^= 1;
        h->has_last = 1;
    }
    return DD_OK;
}

// ---------------------------------------------------------------- dd_chain convenience handle
// Bundles a filter, an FM demodulator and the chunker variables of one stream
// (NCO sample index "freqoffset", decimation phase "bwlim", constants.py:38-39).
struct dd_chain {
    dd_fir* fir;
    dd_fm* fm;
    uint64_t cyc;
    int M, flags;
    int64_t abs_index;
    void* scratch;          // discarded outputs of dd_chain_prime
    size_t scratch_bytes;
};

extern "C" int dd_chain_create(dd_chain** h, const double* taps, int ntaps, uint64_t cycles_q64,
                               int decim, int flags) {
    DD_REQUIRE(h && taps, "null argument");
    DD_REQUIRE(decim >= 1, "decim must be >= 1");
    dd_chain* c = new dd_chain();
    c->fir = nullptr;
    c->fm = nullptr;
    c->cyc = cycles_q64;
    c->M = decim;
    c->flags = flags;
    c->abs_index = 0;
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    int rc = dd_fir_create(&c->fir, taps, ntaps);
    if (rc == DD_OK && (flags & DD_CHAIN_FM)) rc = dd_fm_create(&c->fm);
    if (rc != DD_OK) {
        dd_chain_destroy(c);
        return rc;
    }
    *h = c;
    return DD_OK;
}

extern "C" int dd_chain_destroy(dd_chain* c) {
    if (!c) return DD_OK;
    dd_fir_destroy(c->fir);
    dd_fm_destroy(c->fm);
    (void)hipFree(c->scratch);
    delete c;
    return DD_OK;
}

extern "C" int dd_chain_reset(dd_chain* c, void* stream) {
    DD_REQUIRE(c, "h");
    c->abs_index = 0;
    if (c->fm) dd_fm_reset(c->fm);
    return dd_fir_reset(c->fir, DD_HIST_ONES, nullptr, stream);
}

static inline int chain_off(const dd_chain* c) {
    // kept global indices are the multiples of M (comm.py:123-127, quirk Q4)
    const int64_t r = c->abs_index % c->M;
    return (int)((c->M - r) % c->M);
}

extern "C" int64_t dd_chain_out_count(const dd_chain* c, int64_t n) {
    if (!c || n < 0) return -1;
    return dd_fused_out_count(c->fm, n, c->M, chain_off(c));
}

extern "C" int dd_chain_path(const dd_chain* c) {
    if (!c) return DD_ERR_INVALID;
    if (c->flags & DD_CHAIN_FORCE_DIRECT) return 0;
    const int fl = (c->flags & (DD_CHAIN_NCO | DD_CHAIN_FM | DD_CHAIN_U8_INPUT));
    return (dd_mfma_supported(c->fir->K, c->M, fl) && (c->fir->mfma || !c->fir->mfma_tried)) ? 1 : 0;
}

extern "C" int dd_fir_last_kernel(const dd_fir* f) {
    if (!f) return DD_ERR_INVALID;
    return f->last_kernel;
}

extern "C" long long dd_fir_launch_count(const dd_fir* f) {
    return f ? f->launches : -1;
}

extern "C" int dd_chain_last_kernel(const dd_chain* c) {
    if (!c) return DD_ERR_INVALID;
    return c->fir->last_kernel;
}

extern "C" int dd_chain_process(dd_chain* c, const void* in, void* out, int64_t n, int64_t* n_out, void* stream) {
    DD_REQUIRE(c && n >= 0, "h/n");
    DDFusedArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in;
    a.out = out;
    a.n = n;
    a.nco = (c->flags & DD_CHAIN_NCO) ? 1 : 0;
    a.cyc = c->cyc;
    a.start_index = c->abs_index;
    a.M = c->M;
    a.off = chain_off(c);
    a.u8 = (c->flags & DD_CHAIN_U8_INPUT) ? 1 : 0;
    a.commit = 1;
    a.force_direct = (c->flags & DD_CHAIN_FORCE_DIRECT) ? 1 : 0;
    a.tight = (c->flags & DD_CHAIN_TIGHT) ? 1 : 0;
    int rc = dd_fused_launch(c->fir, c->fm, a, n_out, dd_stream(stream));
    if (rc == DD_OK) c->abs_index += n;
    return rc;
}

// Core of the chunk-list entry points: the chunks [bounds[i], bounds[i+1]) of `in` through (fir, fm) exactly as the
// per-chunk loop would take them -- chunk i starts at absolute index start_index + (bounds[i] - bounds[0]) with the
// decimation phase the previous chunk left (comm.py:123-125) -- in ONE launch when the chain decimates (M > 1) and every
// chunk keeps at least one sample.  Returns 1 when it has done so (state committed), 0 when the caller must loop, < 0 on error.
static int fused_chunks_one_launch(dd_fir* fir, dd_fm* fm, const void* in, void* out, const int64_t* bounds_host, int nchunks,
                                   int nco, uint64_t cyc, int64_t start_index, int M, int off0, int flags, int64_t* n_out_host, hipStream_t s) {
    const bool isfm = fm != nullptr;
    const bool u8 = (flags & DD_CHAIN_U8_INPUT) != 0;
    const size_t isz = u8 ? 2 : sizeof(float2), osz = isfm ? sizeof(float) : sizeof(float2);
    if (!(M > 1 && nchunks >= 2 && !(flags & DD_CHAIN_FORCE_DIRECT) && fir->K >= 2)) return 0;
    if (!dd_kernel_sel_decimp() && dd_decimw_supported(fir->K, M, flags, in)) {
        // k_chain_decim_w lays its rows on the ABSOLUTE decimation grid, which a chunk list continues from chunk to chunk (comm.py:123-125),
        // and a sample after the NCO is a pure function of its absolute index: the list is ONE chunk -- same outputs as the loop, bit for
        // bit, no hand-over inside the launch
        const int64_t n = bounds_host[nchunks] - bounds_host[0];
        int has_last = isfm ? fm->has_last : 0, off = off0;
        for (int i = 0; i < nchunks; ++i) {
            const int64_t ni = bounds_host[i + 1] - bounds_host[i];
            const int64_t Ld = kept_count(ni, off, M);
            if (ni == 0 || Ld == 0) return 0;
            if (n_out_host) n_out_host[i] = isfm ? Ld - (has_last ? 0 : 1) : Ld;
            off = (int)((M - (ni - off) % M) % M);
            if (isfm) has_last = 1;
        }
        DDFusedArgs a;
        memset(&a, 0, sizeof(a));
        a.in = in; a.out = out; a.n = n; a.nco = nco; a.cyc = cyc; a.start_index = start_index; a.M = M; a.off = off0;
        a.u8 = u8 ? 1 : 0; a.commit = 1;
        const int rc = dd_fused_launch(fir, fm, a, nullptr, s);
        return rc == DD_OK ? 1 : rc;
    }
    int withhold = -1, spin_log2 = 0;
    {
        // an earlier chunk-list launch through this filter whose hand-over timed out: say so now, before anything is enqueued
        std::lock_guard<std::mutex> lk(g_seam_mu);
        const int rc0 = seam_look(fir, false);
        if (rc0 != DD_OK) return rc0;
        if (fir->state_invalid) {
            // (the look above zeroes the word when it reports -- a SECOND chunk-list call without a reset must not go on from the
            //  state the faulty launch committed)
            dd_set_error("this filter's carried state comes from a chunk-list launch that timed out (DD_ERR_TIMEOUT was reported): reset it "
                         "(dd_fir_reset / dd_chain_reset / dd_chain_seek) before processing more samples");
            return DD_ERR_TIMEOUT;
        }
        withhold = g_seam_withhold;
        g_seam_withhold = -1;                  // (one launch)
        spin_log2 = g_seam_spin_log2;
    }
    if (!fir->seam_err) {
        // [0]: waits that gave up, counted by the kernel; [1]: where dd_debug_seam sends a withheld flag.  Pinned, mapped host memory.
        DD_HIP_CHECK(hipHostMalloc((void**)&fir->seam_err_host, 2 * sizeof(unsigned int), hipHostMallocMapped));
        fir->seam_err_host[0] = fir->seam_err_host[1] = 0;
        DD_HIP_CHECK(hipHostGetDevicePointer((void**)&fir->seam_err, fir->seam_err_host, 0));
    }
    {
        // (the multi kernels' dynamic LDS limit: a chunk list whose chunks have no interior run never passes through
        // decim_plan's persistent branch, where it used to be raised)
        static DDOncePerDevice attr_m;
        if (attr_m.need()) {
            DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_decim_multi<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            DD_HIP_CHECK(hipFuncSetAttribute((const void*)k_chain_decim_multi<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_m.mark();
        }
    }
    std::vector<DDChainParams> Pv;
    std::vector<DDDecimPlan> plv;
    std::vector<int64_t> nout(nchunks, 0);
    int64_t abs_index = start_index, opos = 0;
    int has_last = isfm ? fm->has_last : 0, off = off0;
    for (int i = 0; i < nchunks; ++i) {
        const int64_t n = bounds_host[i + 1] - bounds_host[i];
        DDChainParams P;
        memset(&P, 0, sizeof(P));
        P.in = reinterpret_cast<const char*>(in) + isz * (size_t)(bounds_host[i] - bounds_host[0]);
        P.out = reinterpret_cast<char*>(out) + osz * (size_t)opos;
        P.taps_rev = fir->taps_rev;
        P.nco_tbl = dd_nco_table();
        P.cyc = cyc;
        P.abs0 = abs_index;
        P.L = n;
        P.K = fir->K;
        P.M = M;
        P.off = off;
        P.Ld = kept_count(n, off, M);
        P.flags = (nco ? DD_CHAIN_NCO : 0) | (u8 ? DD_CHAIN_U8_INPUT : 0) | (isfm ? DD_CHAIN_FM : 0);
        P.s = (isfm && !has_last) ? 1 : 0;
        if (n == 0 || P.Ld == 0) return 0;
        DDDecimPlan pl;
        int rc = decim_plan(P, pl);
        if (rc != DD_OK) return rc;
        nout[i] = isfm ? P.Ld - P.s : P.Ld;
        if (nout[i] < 0) nout[i] = 0;
        Pv.push_back(P);
        plv.push_back(pl);
        opos += nout[i];
        abs_index += n;
        off = (int)((M - (n - off) % M) % M);                    // nextOff of comm.py:124
        if (isfm) has_last = 1;
    }
    // device image: [flags, 16-byte padded][parameter blocks][segments][interior prefix][edge prefix][seam tails][seam last samples]
    const int K1 = fir->K - 1;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_par = al(sizeof(unsigned int) * nchunks), o_seg = o_par + al(sizeof(DDChainParams) * nchunks);
    const size_t o_ipre = o_seg + al(sizeof(DDSeg) * nchunks), o_epre = o_ipre + al(sizeof(int) * (nchunks + 1));
    const size_t o_tail = o_epre + al(sizeof(int) * (nchunks + 1)), o_last = o_tail + al(sizeof(float2) * (size_t)K1 * nchunks);
    const size_t total = o_last + al(sizeof(float2) * nchunks);
    if (total > fir->multi_bytes) {
        DD_HIP_CHECK(hipStreamSynchronize(s));
        (void)hipFree(fir->multi);
        fir->multi = nullptr;
        fir->multi_bytes = 0;
        DD_HIP_CHECK(hipMalloc((void**)&fir->multi, total));
        fir->multi_bytes = total;
    }
    unsigned int* seam_flags = reinterpret_cast<unsigned int*>(fir->multi);
    float2* seam_tail = reinterpret_cast<float2*>(fir->multi + o_tail);
    float2* seam_last = reinterpret_cast<float2*>(fir->multi + o_last);
    std::vector<char> img(o_tail, 0);                         // (from the seam flags, which start as zeros, to the edge prefix: ONE copy)
    DDChainParams* hP = reinterpret_cast<DDChainParams*>(img.data() + o_par);
    DDSeg* hS = reinterpret_cast<DDSeg*>(img.data() + o_seg);
    int* hI = reinterpret_cast<int*>(img.data() + o_ipre);
    int* hE = reinterpret_cast<int*>(img.data() + o_epre);
    hI[0] = hE[0] = 0;
    size_t lds_p = 0;
    int per_cu = 0;
    for (int i = 0; i < nchunks; ++i) {
        DDChainParams& P = Pv[i];
        P.tail_in = i == 0 ? (fir->tail_override ? fir->tail_override : fir->tail[fir->parity]) : seam_tail + (size_t)K1 * (i - 1);
        P.tail_out = i == nchunks - 1 ? fir->tail[fir->parity ^ 1] : seam_tail + (size_t)K1 * i;
        if (isfm) {
            P.lasty_in = i == 0 ? fm->last + fm->parity : seam_last + (i - 1);
            P.lasty_out = i == nchunks - 1 ? fm->last + (fm->parity ^ 1) : seam_last + i;
        }
        P.seam_wait = i > 0 ? seam_flags + (i - 1) : nullptr;
        P.seam_post = i < nchunks - 1 ? seam_flags + i : nullptr;
        if (P.seam_post && i == withhold) P.seam_post = fir->seam_err + 1;       // (dd_debug_seam: this chunk's flag is never set)
        P.seam_err = fir->seam_err;
        P.seam_spin_log2 = spin_log2;
        hP[i] = P;
        hS[i].in = P.in; hS[i].out = P.out; hS[i].abs0 = P.abs0; hS[i].off = P.off; hS[i].s = P.s; hS[i].lo = plv[i].lo; hS[i].pad = 0;
        hI[i + 1] = hI[i] + (plv[i].hi - plv[i].lo);
        hE[i + 1] = hE[i] + (P.nblocks - (plv[i].hi - plv[i].lo));
        if (plv[i].lds_p > lds_p) lds_p = plv[i].lds_p;
        if (plv[i].persistent && (per_cu == 0 || plv[i].per_cu < per_cu)) per_cu = plv[i].per_cu;
    }
    const int n_int = hI[nchunks], n_edge = hE[nchunks];
    if (per_cu < 1) per_cu = 1;
    DD_HIP_CHECK(hipMemcpyAsync(fir->multi, img.data(), img.size(), hipMemcpyHostToDevice, s));      // (pageable source: staged before the call returns)
    (void)seam_flags;
    const int slots = dd_cu_count() * per_cu;
    int grid = n_edge < slots / 2 ? slots - n_edge : slots / 2;
    if (grid > n_int) grid = n_int;
    if (grid >= 8) grid &= ~7;
    if (grid < 0) grid = 0;
    const DDChainParams* dP = reinterpret_cast<const DDChainParams*>(fir->multi + o_par);
    const DDSeg* dS = reinterpret_cast<const DDSeg*>(fir->multi + o_seg);
    const int* dI = reinterpret_cast<const int*>(fir->multi + o_ipre);
    const int* dE = reinterpret_cast<const int*>(fir->multi + o_epre);
    if (u8) hipLaunchKernelGGL(k_chain_decim_multi<true>, dim3(grid + n_edge), dim3(DD_DECIM_THREADS), lds_p, s, dP, dS, dI, dE, nchunks, grid);
    else hipLaunchKernelGGL(k_chain_decim_multi<false>, dim3(grid + n_edge), dim3(DD_DECIM_THREADS), lds_p, s, dP, dS, dI, dE, nchunks, grid);
    DD_LAUNCH_CHECK();
    {
        // the kernel counts waits that gave up into fir->seam_err_host; every later fused launch and dd_stream_sync looks at it
        std::lock_guard<std::mutex> lk(g_seam_mu);
        if (!fir->seam_pending) g_seam_pending.push_back(fir);
        fir->seam_pending = 1;
    }
    fir->last_kernel = DD_KERNEL_DECIM_MULTI;
    ++fir->launches;
    fir->parity ^= 1;
    fir->tail_override = nullptr;
    if (isfm) {
        fm->parity ^= 1;
        fm->has_last = 1;
    }
    if (n_out_host) for (int i = 0; i < nchunks; ++i) n_out_host[i] = nout[i];
    return 1;
}

// The chunks [bounds[i], bounds[i+1]) of `in` (sample offsets, ascending, nchunks + 1 of them) as dd_chain_process would
// take them one after the other -- same outputs, bit for bit, concatenated at `out`, same state afterwards.
extern "C" int dd_chain_process_chunks(dd_chain* c, const void* in, void* out, const int64_t* bounds_host, int nchunks,
                                       int64_t* n_out_host, void* stream) {
    DD_REQUIRE(c && bounds_host && nchunks >= 0, "arguments");
    hipStream_t s = dd_stream(stream);
    const bool isfm = c->fm != nullptr;
    const bool u8 = (c->flags & DD_CHAIN_U8_INPUT) != 0;
    const size_t isz = u8 ? 2 : sizeof(float2), osz = isfm ? sizeof(float) : sizeof(float2);
    for (int i = 0; i < nchunks; ++i) DD_REQUIRE(bounds_host[i + 1] >= bounds_host[i], "bounds must ascend");
    if (nchunks == 0) return DD_OK;
    const char* in0 = reinterpret_cast<const char*>(in) + isz * (size_t)bounds_host[0];
    int rc = fused_chunks_one_launch(c->fir, c->fm, in0, out, bounds_host, nchunks, (c->flags & DD_CHAIN_NCO) ? 1 : 0, c->cyc, c->abs_index,
                                     c->M, chain_off(c), c->flags & (DD_CHAIN_U8_INPUT | DD_CHAIN_FORCE_DIRECT), n_out_host, s);
    if (rc < 0) return rc;
    if (rc == 1) {
        c->abs_index += bounds_host[nchunks] - bounds_host[0];
        return DD_OK;
    }
    int64_t opos = 0;
    for (int i = 0; i < nchunks; ++i) {
        int64_t got = 0;
        rc = dd_chain_process(c, reinterpret_cast<const char*>(in) + isz * (size_t)bounds_host[i],
                              reinterpret_cast<char*>(out) + osz * (size_t)opos, bounds_host[i + 1] - bounds_host[i], &got, stream);
        if (rc != DD_OK) return rc;
        if (n_out_host) n_out_host[i] = got;
        opos += got;
    }
    return DD_OK;
}

// The same for the object-model form (dd_fused_process): chunk i of `in` (which points at the first chunk's first sample)
// is bounds[i+1] - bounds[i] samples long; start_index / offset are the chunker variables of the FIRST chunk, the later
// chunks' follow by the reference's own carry rules (comm.py:75-76, 123-125).  carry must be 1 (storeState).
extern "C" int dd_fused_process_chunks(dd_fir* fir, dd_fm* fm, const void* in, void* out, const int64_t* bounds_host, int nchunks,
                                       int nco, uint64_t cycles_q64, int64_t start_index, int decim, int offset, int flags,
                                       int64_t* n_out_host, void* stream) {
    DD_REQUIRE(fir && bounds_host && nchunks >= 0 && decim >= 1 && offset >= 0 && offset < decim, "arguments");
    hipStream_t s = dd_stream(stream);
    const bool u8 = (flags & DD_CHAIN_U8_INPUT) != 0;
    const size_t isz = u8 ? 2 : sizeof(float2), osz = fm ? sizeof(float) : sizeof(float2);
    for (int i = 0; i < nchunks; ++i) DD_REQUIRE(bounds_host[i + 1] >= bounds_host[i], "bounds must ascend");
    if (nchunks == 0) return DD_OK;
    int rc = fused_chunks_one_launch(fir, fm, in, out, bounds_host, nchunks, nco, cycles_q64, start_index, decim, offset,
                                     flags & (DD_CHAIN_U8_INPUT | DD_CHAIN_FORCE_DIRECT), n_out_host, s);
    if (rc != 0) return rc < 0 ? rc : DD_OK;
    int64_t opos = 0, abs_index = start_index;
    int off = offset;
    for (int i = 0; i < nchunks; ++i) {
        const int64_t n = bounds_host[i + 1] - bounds_host[i];
        int64_t got = 0;
        rc = dd_fused_process(fir, fm, reinterpret_cast<const char*>(in) + isz * (size_t)(bounds_host[i] - bounds_host[0]),
                              reinterpret_cast<char*>(out) + osz * (size_t)opos, n, nco, cycles_q64, abs_index, decim, off, flags, 1, &got, stream);
        if (rc != DD_OK) return rc;
        if (n_out_host) n_out_host[i] = got;
        opos += got;
        abs_index += n;
        off = (int)((decim - (n - off) % decim) % decim);
    }
    return DD_OK;
}

extern "C" int dd_chain_seek(dd_chain* c, int64_t abs_index, void* stream) {
    DD_REQUIRE(c && abs_index >= 0, "arguments");
    c->abs_index = abs_index;
    if (c->fm) dd_fm_reset(c->fm);
    return dd_fir_reset(c->fir, abs_index == 0 ? DD_HIST_ONES : DD_HIST_ZEROS, nullptr, stream);   // launch-free
}

extern "C" int dd_chain_prime(dd_chain* c, const void* halo_in, int64_t n_halo, int64_t abs_index, void* stream) {
    DD_REQUIRE(c && n_halo >= 0 && abs_index >= 0, "arguments");
    DD_REQUIRE(n_halo <= abs_index, "halo longer than the samples that precede abs_index");
    hipStream_t s = dd_stream(stream);
    if (abs_index == 0) return dd_chain_reset(c, stream);
    DD_REQUIRE(halo_in, "halo_in");
    const bool at_start = (n_halo == abs_index);
    DD_REQUIRE(at_start || n_halo >= (int64_t)c->fir->K - 1 + c->M,
               "halo must hold at least ntaps-1+decim samples (or reach back to the stream start)");
    // Replay the halo as a chunk whose outputs are discarded.  If the halo reaches
    // the stream start the history is the reference's ones (Q1); otherwise it is
    // irrelevant: the last kept output and the last K-1 inputs of the halo have
    // their full windows inside the halo.
    c->abs_index = abs_index - n_halo;
    if (c->fm) dd_fm_reset(c->fm);
    int rc = dd_fir_reset(c->fir, at_start ? DD_HIST_ONES : DD_HIST_ZEROS, nullptr, stream);
    if (rc != DD_OK) return rc;
    const int64_t no = dd_chain_out_count(c, n_halo);
    const size_t ob = (size_t)(no > 0 ? no : 1) * ((c->flags & DD_CHAIN_FM) ? sizeof(float) : sizeof(float2));
    if (ob > c->scratch_bytes) {
        DD_HIP_CHECK(hipStreamSynchronize(s));
        (void)hipFree(c->scratch);
        c->scratch = nullptr;
        c->scratch_bytes = 0;
        DD_HIP_CHECK(hipMalloc(&c->scratch, ob));
        c->scratch_bytes = ob;
    }
    return dd_chain_process(c, halo_in, c->scratch, n_halo, nullptr, stream);
}

// dd_code_warmup (dd_runtime.hip): the runtime loads a translation unit's code object when one of its kernels is first named
int dd_code_touch_chain(void) {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, (const void*)k_chain_dense) == hipSuccess ? DD_OK : DD_ERR_HIP;
}
