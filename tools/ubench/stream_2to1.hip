// What can MI355X stream when a kernel reads 8 B and writes 4 B per sample (the traffic of the C2 chain: complex64 in,
// float32 angle out), independently of the FFT kernel?  (the "0.60 of 8 TB/s" ceiling of DESIGN 4.2c
// rested on k_chain_fft1k's own -DFF_NO_COMPUTE build.)
//
//   out[k] = x[k].re + x[k].im    over N = 2^26 samples, persistent waves, swept over
//     waves per CU            8, 12, 16, 24, 32             (256-thread workgroups, grid = CUs x waves / 4)
//     U  16-byte loads in flight per wave   2, 4, 6, 8     (one "unit" = one wave-wide dwordx4 load = 128 samples)
//     DB                      1: the next U loads are issued before the current U units are summed and stored
//                                (what k_chain_fft1k does: six loads per block fly during its tail), 0: load, wait, store
//     S  store width          4, 8, 16 bytes per lane
//     off                     output pointer 0 or -4 bytes off a 512-byte boundary (a stream START drops one angle, demod_fm.py:43-49)
//     map                     0: one contiguous run of units per wave (dd_fftfir.hip's q_begin..q_end), 1: block-cyclic
//                                (wave w takes pieces w, w + W, ...: the live working set is one moving window),
//                             2: block-cyclic per XCD (workgroup b lives on XCD b % 8: each XCD walks its own eighth)
//     NT                      non-temporal loads and stores
//   plus read-only, write-only and a float4 copy (the guide's 6.29 TB/s figure) as calibration.
//
// hipcc --offload-arch=gfx950 -O3 -o bin/stream_2to1 stream_2to1.hip ; ./bin/stream_2to1 [quick]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ v4f ld16(const v4f* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT, typename T> __device__ __forceinline__ void st(T* p, T v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// piece p (U units) -> where its first unit lives, for the three mappings
struct Map {
    long npieces;       // total
    int nwaves;         // total waves of the launch
    int mode;
};
__device__ __forceinline__ void my_range(const Map& m, int gw, int wg, int wave, int nwg, long& first, long& stride, long& count) {
    if (m.mode == 0) {                                   // contiguous run per wave
        const long b = m.npieces * gw / m.nwaves, e = m.npieces * (gw + 1) / m.nwaves;
        first = b; stride = 1; count = e - b;
    } else if (m.mode == 1) {                            // block-cyclic over all waves
        first = gw; stride = m.nwaves; count = (m.npieces - gw + m.nwaves - 1) / m.nwaves;
    } else {                                             // block-cyclic inside the XCD's eighth (workgroup b runs on XCD b % 8)
        const int xcd = wg & 7, wx = (wg >> 3) * 4 + wave, nwx = ((nwg + 7 - xcd) >> 3) * 4;      // waves of this XCD
        const long b = m.npieces * xcd / 8, e = m.npieces * (xcd + 1) / 8;
        first = b + wx; stride = nwx; count = (e - b - wx + nwx - 1) / nwx;
        if (e - b - wx <= 0) count = 0;
    }
}

template <int U, int S>
__device__ __forceinline__ void sum_store(const v4f (&x)[U], float* o, int lane, bool nt) {
    // o = the piece's first output
    if (S == 4) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // (two dword stores of 256 contiguous bytes each: what a 4-byte-per-lane layout would issue)
            if (nt) { __builtin_nontemporal_store(x[u].x + x[u].y, o + 128 * u + lane); __builtin_nontemporal_store(x[u].z + x[u].w, o + 128 * u + 64 + lane); }
            else { o[128 * u + lane] = x[u].x + x[u].y; o[128 * u + 64 + lane] = x[u].z + x[u].w; }
        }
    } else if (S == 8) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const v2f v = {x[u].x + x[u].y, x[u].z + x[u].w};
            if (nt) __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(o + 128 * u + 2 * lane));
            else *reinterpret_cast<v2f*>(o + 128 * u + 2 * lane) = v;
        }
    } else {
#pragma unroll
        for (int u = 0; u + 1 < U; u += 2) {
            const v4f v = {x[u].x + x[u].y, x[u].z + x[u].w, x[u + 1].x + x[u + 1].y, x[u + 1].z + x[u + 1].w};
            if (nt) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(o + 128 * u + 4 * lane));
            else *reinterpret_cast<v4f*>(o + 128 * u + 4 * lane) = v;
        }
    }
}

// S == 0: read-only (sums kept, stored never)
template <int U, int S, bool DB, bool NT>
__global__ void __launch_bounds__(256) k_s21(const v4f* __restrict__ in, float* __restrict__ out, Map m) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    long first, stride, count;
    my_range(m, gw, blockIdx.x, wave, gridDim.x, first, stride, count);
    if (count <= 0) return;
    v4f a[U], b[U];
    float acc = 0.f;
    if (DB) {
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = ld16<NT>(in + (first * U + u) * 64 + lane);
    }
    for (long i = 0; i < count; ++i) {
        const long p = first + i * stride;
        if (DB) {
            const long pn = (i + 1 < count) ? p + stride : p;          // (the last one re-reads itself: no branch)
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = ld16<NT>(in + (pn * U + u) * 64 + lane);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = ld16<NT>(in + (p * U + u) * 64 + lane);
        }
        if (S == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc += a[u].x + a[u].y + a[u].z + a[u].w;
        } else {
            sum_store<U, S == 0 ? 8 : S>(a, out + p * U * 128, lane, NT);
        }
        if (DB) {
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = b[u];
        }
    }
    if (S == 0 && acc == 123.456f) out[0] = acc;
}

// write-only: 4 B per sample, S bytes per lane
template <int S>
__global__ void __launch_bounds__(256) k_wr(float* __restrict__ out, Map m) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    long first, stride, count;
    my_range(m, gw, blockIdx.x, wave, gridDim.x, first, stride, count);
    for (long i = 0; i < count; ++i) {
        float* o = out + (first + i * stride) * 4 * 128;          // pieces of 4 units
        if (S == 8) {
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<v2f*>(o + 128 * u + 2 * lane) = (v2f){(float)i, (float)lane};
        } else {
#pragma unroll
            for (int u = 0; u < 4; u += 2) *reinterpret_cast<v4f*>(o + 128 * u + 4 * lane) = (v4f){(float)i, (float)lane, 1.f, 2.f};
        }
    }
}

// float4 copy (the guide's calibration: 6.29 TB/s), grid-stride, U loads then U stores
template <int U>
__global__ void __launch_bounds__(256) k_copy(const v4f* __restrict__ in, v4f* __restrict__ out, long n4) {
    const long nthreads = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += nthreads * U) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long j = i + u * nthreads; v[u] = j < n4 ? in[j] : (v4f){0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long j = i + u * nthreads; if (j < n4) out[j] = v[u]; }
    }
}

static hipEvent_t e0, e1;
template <typename F>
static float time_ms(F launch, int warm, int reps) {
    for (int i = 0; i < warm; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

struct Row { int U, S, DB, NT, off, map; float tbs[5]; int waves[5]; };
static const int WAVES[5] = {8, 12, 16, 24, 32};
static int g_cus = 256;
static const long N = 1L << 26;
static v4f* d_in; static float* d_out;
static int g_warm = 10, g_reps = 40;

template <int U, int S, bool DB, bool NT>
static void run_cfg(std::vector<Row>& rows, int off, int map) {
    Row r; r.U = U; r.S = S; r.DB = DB; r.NT = NT; r.off = off; r.map = map;
    int maxwg = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxwg, k_s21<U, S, DB, NT>, 256, 0));
    const long units = N / 128, npieces = (units + U - 1) / U;
    for (int w = 0; w < 5; ++w) {
        int wgs = WAVES[w] / 4;
        if (wgs > maxwg) wgs = maxwg;               // (register-limited: reported with the waves it really runs)
        r.waves[w] = wgs * 4;
        if (w > 0 && r.waves[w] == r.waves[w - 1]) { r.tbs[w] = r.tbs[w - 1]; continue; }
        Map m; m.npieces = npieces; m.nwaves = g_cus * wgs * 4; m.mode = map;
        float* o = d_out + 128 - off;
        const float ms = time_ms([&] { hipLaunchKernelGGL((k_s21<U, S, DB, NT>), dim3(g_cus * wgs), dim3(256), 0, 0, d_in, o, m); }, g_warm, g_reps);
        const double bytes = (S == 0 ? 8.0 : 12.0) * (double)N;
        r.tbs[w] = (float)(bytes / (ms * 1e-3) / 1e12);
    }
    rows.push_back(r);
}

template <int U, int S>
static void run_db_nt(std::vector<Row>& rows, int off, int map, bool nt_too) {
    run_cfg<U, S, true, false>(rows, off, map);
    run_cfg<U, S, false, false>(rows, off, map);
    if (nt_too) { run_cfg<U, S, true, true>(rows, off, map); }
}
template <int S>
static void run_u(std::vector<Row>& rows, int off, int map, bool nt_too) {
    run_db_nt<2, S>(rows, off, map, nt_too);
    run_db_nt<4, S>(rows, off, map, nt_too);
    run_db_nt<6, S>(rows, off, map, nt_too);
    run_db_nt<8, S>(rows, off, map, nt_too);
}

static void print_rows(const std::vector<Row>& rows, size_t from) {
    for (size_t i = from; i < rows.size(); ++i) {
        const Row& r = rows[i];
        printf("map %d off %2d S %2d U %d DB %d NT %d |", r.map, -4 * r.off, r.S, r.U, r.DB, r.NT);
        for (int w = 0; w < 5; ++w) printf(" %5.2f%s", r.tbs[w], r.waves[w] == WAVES[w] ? " " : "*");
        printf("\n");
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    printf("# %s, %d CUs; N = 2^26 samples: 512 MiB complex64 in, 256 MiB float32 out (>> 256 MiB Infinity Cache); TB/s of the 12 B/sample\n", prop.name, g_cus);
    printf("# (8 TB/s spec peak: 0.70 = 5.60, 0.60 = 4.80); columns = waves per CU %d %d %d %d %d (* = register-limited below that)\n", WAVES[0], WAVES[1], WAVES[2], WAVES[3], WAVES[4]);
    CK(hipMalloc(&d_in, N * 8 + (1 << 20)));
    CK(hipMalloc(&d_out, N * 4 + (1 << 20)));
    CK(hipMemset(d_in, 0, N * 8 + (1 << 20)));
    CK(hipMemset(d_out, 0, N * 4 + (1 << 20)));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (quick) { g_warm = 3; g_reps = 10; }

    // ---- calibration: copy, read-only, write-only
    for (int wgs : {2, 4, 8}) {
        const long n4 = N * 8 / 16 / 2;          // 256 MiB -> 256 MiB
        float ms = time_ms([&] { hipLaunchKernelGGL((k_copy<4>), dim3(g_cus * wgs), dim3(256), 0, 0, d_in, reinterpret_cast<v4f*>(d_out), n4); }, g_warm, g_reps);
        printf("copy float4 256 MiB -> 256 MiB, %2d waves/CU, 4 loads then 4 stores: %.4f ms = %.2f TB/s (read + written)\n", wgs * 4, ms, 2.0 * n4 * 16 / (ms * 1e-3) / 1e12);
    }
    {
        std::vector<Row> rows;
        printf("# read-only (8 B/sample, 512 MiB), TB/s of the bytes read\n");
        for (int map = 0; map < 3; ++map) { run_cfg<4, 0, true, false>(rows, 0, map); run_cfg<8, 0, true, false>(rows, 0, map); run_cfg<8, 0, false, false>(rows, 0, map); }
        print_rows(rows, 0);
        printf("# write-only (4 B/sample, 256 MiB), TB/s of the bytes written; 8 then 16 bytes per lane; maps 0 1 2; waves/CU 8 16 32\n");
        for (int S : {8, 16})
            for (int map = 0; map < 3; ++map) {
                printf("write-only S %2d map %d |", S, map);
                for (int wgs : {2, 4, 8}) {
                    Map m; m.npieces = N / 128 / 4; m.nwaves = g_cus * wgs * 4; m.mode = map;
                    float ms = S == 8 ? time_ms([&] { hipLaunchKernelGGL((k_wr<8>), dim3(g_cus * wgs), dim3(256), 0, 0, d_out, m); }, g_warm, g_reps)
                                      : time_ms([&] { hipLaunchKernelGGL((k_wr<16>), dim3(g_cus * wgs), dim3(256), 0, 0, d_out, m); }, g_warm, g_reps);
                    printf(" %5.2f", 4.0 * N / (ms * 1e-3) / 1e12);
                }
                printf("\n");
            }
    }
    // ---- the sweep
    std::vector<Row> rows;
    for (int map = 0; map < 3; ++map)
        for (int off = 0; off < 2; ++off) {
            const size_t from = rows.size();
            const bool nt = !quick;
            run_u<4>(rows, off, map, nt);
            run_u<8>(rows, off, map, nt);
            run_u<16>(rows, off, map, nt);
            print_rows(rows, from);
        }
    // ---- best configurations
    std::vector<std::pair<float, std::pair<int, int>>> best;
    for (size_t i = 0; i < rows.size(); ++i)
        for (int w = 0; w < 5; ++w) best.push_back({rows[i].tbs[w], {(int)i, w}});
    std::sort(best.begin(), best.end(), [](auto& a, auto& b) { return a.first > b.first; });
    printf("# top 12\n");
    for (int k = 0; k < 12 && k < (int)best.size(); ++k) {
        const Row& r = rows[best[k].second.first];
        printf("%.2f TB/s = %.3f of 8 TB/s: map %d off %d S %d U %d DB %d NT %d waves/CU %d\n", best[k].first, best[k].first / 8.0, r.map, -4 * r.off, r.S, r.U, r.DB, r.NT, r.waves[best[k].second.second]);
    }
    // the product kernel's shape: contiguous run per wave, 12 waves/CU, 6 loads ahead, 8-byte stores
    for (const Row& r : rows)
        if (r.map == 0 && r.S == 8 && r.U == 6 && r.DB == 1 && r.NT == 0)
            printf("# k_chain_fft1k's shape (map 0, S 8, U 6, DB 1, 12 waves/CU), off %d: %.2f TB/s\n", -4 * r.off, r.tbs[1]);
    return 0;
}
