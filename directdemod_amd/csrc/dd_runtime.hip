// Runtime plumbing (memory, streams, events) and the element-wise rows of the
// hot path: S1 (u8 ingest), N1 (NCO), R1 (decimation gather), D1 (FM
// discriminator).  gfx950 only.
#include "dd_common.h"
#include <vector>
#include <mutex>
#include <stdarg.h>
#include <mutex>

// ---------------------------------------------------------------- error state
static thread_local char g_err[512] = "";

void dd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dd_last_error(void) { return g_err; }
extern "C" const char* dd_version(void) { return "directdemod_hip 0.1 (gfx950)"; }

extern "C" int dd_device_count(int* count) {
    DD_REQUIRE(count, "count");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        dd_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return DD_ERR_NODEVICE;
    }
    *count = c;
    return c > 0 ? DD_OK : DD_ERR_NODEVICE;
}

extern "C" int dd_set_device(int device) {
    DD_HIP_CHECK(hipSetDevice(device));
    return DD_OK;
}

extern "C" int dd_get_device(int* device) {
    DD_REQUIRE(device, "device");
    DD_HIP_CHECK(hipGetDevice(device));
    return DD_OK;
}

extern "C" int dd_device_name(char* buf, int buflen) {
    DD_REQUIRE(buf && buflen > 0, "buf");
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t p;
    DD_HIP_CHECK(hipGetDeviceProperties(&p, dev));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return DD_OK;
}

extern "C" int dd_malloc(void** dptr, size_t bytes) {
    DD_REQUIRE(dptr, "dptr");
    *dptr = nullptr;
    if (bytes == 0) bytes = 16;
    DD_HIP_CHECK(hipMalloc(dptr, bytes));
    return DD_OK;
}
extern "C" int dd_free(void* dptr) {
    if (dptr) DD_HIP_CHECK(hipFree(dptr));
    return DD_OK;
}
extern "C" int dd_memset(void* dptr, int value, size_t bytes, void* stream) {
    if (bytes) DD_HIP_CHECK(hipMemsetAsync(dptr, value, bytes, dd_stream(stream)));
    return DD_OK;
}
extern "C" int dd_host_alloc_pinned(void** hptr, size_t bytes) {
    DD_REQUIRE(hptr, "hptr");
    *hptr = nullptr;
    if (bytes == 0) bytes = 16;
    DD_HIP_CHECK(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return DD_OK;
}
extern "C" int dd_host_free_pinned(void* hptr) {
    if (hptr) DD_HIP_CHECK(hipHostFree(hptr));
    return DD_OK;
}
// pin a range of the caller's own memory in place (an array, or a file mapping of the recording): a
// hipMemcpyAsync out of it is then a true asynchronous DMA with no staging copy in the runtime
extern "C" int dd_host_register(void* hptr, size_t bytes) {
    DD_REQUIRE(hptr && bytes, "hptr/bytes");
    hipError_t e = hipHostRegister(hptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dd_set_error("hipHostRegister(%p, %zu): %s", hptr, bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? DD_ERR_NOMEM : DD_ERR_UNSUPPORTED;   // e.g. a read-only mapping: the caller falls back
    }
    return DD_OK;
}
extern "C" int dd_host_unregister(void* hptr) {
    if (hptr) DD_HIP_CHECK(hipHostUnregister(hptr));
    return DD_OK;
}
// ---- scratch buffers keyed by (device, stream) (dd_common.h, DDScratchLock)
struct DDScratchEntry {
    std::mutex mu;
    int dev;
    hipStream_t stream;
    char* buf = nullptr;
    size_t bytes = 0;
    unsigned long long last_use = 0;
};
static std::mutex g_scr_mu;                                   // the table only; never held while an entry is in use
static std::vector<DDScratchEntry*> g_scr;
static unsigned long long g_scr_clock = 0;
#define DD_SCRATCH_MAX_ENTRIES 16

int DDScratchLock::get(size_t bytes, hipStream_t s) {
    int dev = 0;
    DD_HIP_CHECK(hipGetDevice(&dev));
    DDScratchEntry* e = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_scr_mu);
        for (DDScratchEntry* c : g_scr)
            if (c->dev == dev && c->stream == s) { e = c; break; }
        if (!e) {
            if (g_scr.size() >= DD_SCRATCH_MAX_ENTRIES) {
                // streams come and go (a caller may destroy one without telling us): drop the least recently used
                // entry nobody holds; its stream may be dead, so nothing is synchronised -- hipFree waits for the device
                size_t victim = g_scr.size();
                for (size_t i = 0; i < g_scr.size(); ++i)
                    if ((victim == g_scr.size() || g_scr[i]->last_use < g_scr[victim]->last_use) && g_scr[i]->mu.try_lock()) {
                        if (victim != g_scr.size()) g_scr[victim]->mu.unlock();
                        victim = i;
                    }
                if (victim != g_scr.size()) {
                    DDScratchEntry* v = g_scr[victim];
                    if (v->buf) (void)hipFree(v->buf);
                    v->mu.unlock();
                    delete v;
                    g_scr.erase(g_scr.begin() + (long)victim);
                }
            }
            e = new DDScratchEntry();
            e->dev = dev;
            e->stream = s;
            g_scr.push_back(e);
        }
        e->last_use = ++g_scr_clock;
    }
    e->mu.lock();
    entry = e;
    if (e->bytes < bytes) {
        if (e->buf) {
            const hipError_t fe = hipFree(e->buf);              // (waits for the device: earlier users of the buffer are done)
            e->buf = nullptr;
            e->bytes = 0;
            DD_HIP_CHECK(fe);
        }
        const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        DD_HIP_CHECK(hipMalloc((void**)&e->buf, want));
        e->bytes = want;
    }
    ptr = e->buf;
    return DD_OK;
}
DDScratchLock::~DDScratchLock() {
    if (entry) reinterpret_cast<DDScratchEntry*>(entry)->mu.unlock();
}
void dd_scratch_forget_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_scr_mu);
    for (size_t i = 0; i < g_scr.size();) {
        DDScratchEntry* c = g_scr[i];
        if (c->stream == s && c->mu.try_lock()) {
            if (c->buf) (void)hipFree(c->buf);
            c->mu.unlock();
            delete c;
            g_scr.erase(g_scr.begin() + (long)i);
        } else {
            ++i;
        }
    }
}

// Diagnostic: leave every compute unit's LDS holding `pattern` (the hardware does not clear LDS between workgroups, so a
// kernel that reads an LDS word it never wrote sees what the previous workgroup on that CU left there).  The tests run
// the chain kernels after a fill with NaN patterns and after a fill with zeros and demand bit-identical outputs.
__global__ void __launch_bounds__(256) k_fill_lds(uint32_t pattern, int words, int spin) {
    extern __shared__ uint32_t dd_fill_words[];
    for (int i = threadIdx.x; i < words; i += 256) dd_fill_words[i] = pattern;
    __syncthreads();
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);   // keep the CU occupied until every CU has received a workgroup
    uint32_t v = dd_fill_words[(threadIdx.x * 97) % words];
    asm volatile("" ::"v"(v));
}
extern "C" int dd_debug_fill_lds(uint32_t pattern, void* stream) {
    const int bytes = 160 * 1024;                              // the whole LDS of a CU: one workgroup per CU at a time
    static DDOncePerDevice once;
    if (once.need()) {
        DD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fill_lds), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        once.mark();
    }
    hipLaunchKernelGGL(k_fill_lds, dim3(4 * dd_cu_count()), dim3(256), bytes, dd_stream(stream), pattern, bytes / 4, 200);
    DD_HIP_CHECK(hipGetLastError());
    return DD_OK;
}
extern "C" int dd_memcpy_h2d(void* dst, const void* src_host, size_t bytes, void* stream) {
    if (bytes) DD_HIP_CHECK(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, dd_stream(stream)));
    return DD_OK;
}
extern "C" int dd_memcpy_d2h(void* dst_host, const void* src, size_t bytes, void* stream) {
    if (bytes) DD_HIP_CHECK(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, dd_stream(stream)));
    return DD_OK;
}
extern "C" int dd_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes) DD_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, dd_stream(stream)));
    return DD_OK;
}
// the first host-to-device copy of a process costs ~90 ms inside the runtime whatever its size (tools/debug/first_copy.py), and the first one of
// a megabyte or more another 7-8 ms (its other copy path; tools/debug/first_copy2.py: 1 MB 7.5 ms, the 8 MB after it 0.22 ms): a 4 KB and a 1 MB
// synchronous copy on the CALLING thread's device, for a helper thread to make while the caller opens its recording.  Touches no stream of
// the library and looks at no seam word (dd_stream_sync does).
extern "C" int dd_copy_warmup(void) {
    const size_t big = (size_t)1 << 20;
    void* d = nullptr;
    DD_HIP_CHECK(hipMalloc(&d, big));
    // (kept for the life of the process: the runtime pins the source of a copy this large in place, and returning pinned pages to the system
    //  has stalled later copies -- DESIGN.md 4.5, round 6)
    static char* const zeros = static_cast<char*>(calloc(1, big));
    if (!zeros) { (void)hipFree(d); dd_set_error("dd_copy_warmup: out of host memory"); return DD_ERR_NOMEM; }
    hipError_t e = hipMemcpy(d, zeros, 4096, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d, zeros, big, hipMemcpyHostToDevice);
    (void)hipFree(d);
    DD_HIP_CHECK(e);
    return DD_OK;
}
// The runtime loads the code object of a translation unit at the first launch (or query) of one of its kernels: 1-4 ms each, in the caller's
// first call (the decimating chain's first launch 3.4 ms against 0.2, the crude tail's 2.9 against 0.6, the accurate sync's 8 against 2.5:
// tools/debug/cold_c4_trace.py).  This names one kernel per unit -- for a helper thread to call once the copy path is up.
extern "C" int dd_code_warmup(void) {
    int rc = dd_code_touch_decimw();
    if (rc == DD_OK) rc = dd_code_touch_audio();
    if (rc == DD_OK) rc = dd_code_touch_chain();
    if (rc == DD_OK) rc = dd_code_touch_cosfir();
    if (rc == DD_OK) rc = dd_code_touch_fftfir();
    if (rc == DD_OK) rc = dd_code_touch_mfma();
    if (rc == DD_OK) rc = dd_code_touch_fir();
    if (rc == DD_OK) rc = dd_code_touch_afsk();
    if (rc != DD_OK) dd_set_error("dd_code_warmup: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}
extern "C" int dd_stream_create(void** stream) {
    DD_REQUIRE(stream, "stream");
    hipStream_t s;
    DD_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return DD_OK;
}
extern "C" int dd_stream_destroy(void* stream) {
    if (stream) {
        DD_HIP_CHECK(hipStreamSynchronize(dd_stream(stream)));
        dd_scratch_forget_stream(dd_stream(stream));
        dd_audio_forget_stream(dd_stream(stream));
        DD_HIP_CHECK(hipStreamDestroy(dd_stream(stream)));
    }
    return DD_OK;
}
extern "C" int dd_stream_sync(void* stream) {
    DD_HIP_CHECK(hipStreamSynchronize(dd_stream(stream)));
    return dd_seam_poll_all();          // a chunk-list launch whose in-launch hand-over timed out (DD_ERR_TIMEOUT)
}
extern "C" int dd_event_create(void** ev) {
    DD_REQUIRE(ev, "ev");
    hipEvent_t e;
    DD_HIP_CHECK(hipEventCreate(&e));
    *ev = (void*)e;
    return DD_OK;
}
extern "C" int dd_event_destroy(void* ev) {
    if (ev) DD_HIP_CHECK(hipEventDestroy((hipEvent_t)ev));
    return DD_OK;
}
extern "C" int dd_event_record(void* ev, void* stream) {
    DD_HIP_CHECK(hipEventRecord((hipEvent_t)ev, dd_stream(stream)));
    return DD_OK;
}
extern "C" int dd_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
    DD_REQUIRE(ms, "ms");
    DD_HIP_CHECK(hipEventSynchronize((hipEvent_t)ev_stop));
    DD_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return DD_OK;
}

extern "C" int dd_event_sync(void* ev) {
    DD_HIP_CHECK(hipEventSynchronize((hipEvent_t)ev));
    return DD_OK;
}
extern "C" int dd_stream_wait_event(void* stream, void* ev) {
    DD_HIP_CHECK(hipStreamWaitEvent(dd_stream(stream), (hipEvent_t)ev, 0));
    return DD_OK;
}

// ---------------------------------------------------------------- NCO table
__global__ void k_fill_nco_table(float2* t) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < DD_NCO_TSIZE) {
        double s, c;
        sincospi(2.0 * (double)k / (double)DD_NCO_TSIZE, &s, &c);
        t[k] = make_float2((float)c, (float)(-s));
    }
}

static std::mutex g_tbl_mu;
static float2* g_tbl[64] = {nullptr};

const float2* dd_nco_table(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_tbl_mu);
    if (!g_tbl[dev]) {
        float2* p = nullptr;
        if (hipMalloc((void**)&p, sizeof(float2) * DD_NCO_TSIZE) != hipSuccess) return nullptr;
        hipLaunchKernelGGL(k_fill_nco_table, dim3(DD_NCO_TSIZE / 256), dim3(256), 0, 0, p);
        if (hipDeviceSynchronize() != hipSuccess) {
            (void)hipFree(p);
            return nullptr;
        }
        g_tbl[dev] = p;
    }
    return g_tbl[dev];
}

// ---------------------------------------------------------------- S1: u8 -> c64
// 16 B per lane in (8 samples), 4 x 16 B per lane out.
__global__ void __launch_bounds__(256) k_u8iq_to_c64(const uint8_t* __restrict__ in, float2* __restrict__ out, int64_t n, int vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n8 = vec ? n >> 3 : 0;                   // unaligned buffers (a view into a recording): pair by pair
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const uint4 v = reinterpret_cast<const uint4*>(in)[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        float4* o = reinterpret_cast<float4*>(out + i * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 r;
            r.x = (float)(w[j] & 0xff) - 127.5f;
            r.y = (float)((w[j] >> 8) & 0xff) - 127.5f;
            r.z = (float)((w[j] >> 16) & 0xff) - 127.5f;
            r.w = (float)(w[j] >> 24) - 127.5f;
            o[j] = r;
        }
    }
    // tail
    const int64_t base = n8 << 3;
    for (int64_t i = base + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        out[i] = make_float2((float)in[2 * i] - 127.5f, (float)in[2 * i + 1] - 127.5f);
    }
}

static inline int dd_grid_for(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

extern "C" int dd_u8iq_to_c64(const uint8_t* in_iq, float* out_c64, int64_t n, void* stream) {
    DD_REQUIRE(n >= 0, "n");
    if (n == 0) return DD_OK;
    DD_REQUIRE(in_iq && out_c64, "null buffer");
    const int vec = !(((uintptr_t)in_iq & 15) || ((uintptr_t)out_c64 & 15));
    hipLaunchKernelGGL(k_u8iq_to_c64, dim3(dd_grid_for(vec ? n / 8 + 1 : n, 256)), dim3(256), 0, dd_stream(stream),
                       in_iq, (float2*)out_c64, n, vec);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// ---------------------------------------------------------------- N1: NCO
__global__ void __launch_bounds__(256) k_nco_c64(const float2* __restrict__ in, float2* __restrict__ out, int64_t n,
                                                 uint64_t cyc, int64_t start, const float2* __restrict__ tbl) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t ph = (uint64_t)(start + i) * cyc;
        out[i] = dd_cmul(in[i], dd_phasor(ph, tbl));
    }
}

extern "C" int dd_nco_c64(const float* in_c64, float* out_c64, int64_t n, uint64_t cycles_q64,
                          int64_t start_index, void* stream) {
    DD_REQUIRE(n >= 0, "n");
    if (n == 0) return DD_OK;
    DD_REQUIRE(in_c64 && out_c64, "null buffer");
    const float2* tbl = dd_nco_table();
    if (!tbl) {
        dd_set_error("NCO table initialisation failed (no GPU?)");
        return DD_ERR_NODEVICE;
    }
    hipLaunchKernelGGL(k_nco_c64, dim3(dd_grid_for(n, 256)), dim3(256), 0, dd_stream(stream),
                       (const float2*)in_c64, (float2*)out_c64, n, cycles_q64, start_index, tbl);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// ---------------------------------------------------------------- N1 with a per-sample frequency (comm.py:77 with an
// array freqOffset: the Doppler correction of decode_funcube.py:228).  phase = f[n] (n0+n) / fs is
// formed and reduced in float64, like the reference's np.exp argument; 1e-9-grade phase, the product
// is rounded once to complex64.
__global__ void __launch_bounds__(256) k_nco_c64_freqs(const float2* __restrict__ in, float2* __restrict__ out, int64_t n,
                                                       const double* __restrict__ f, double fs, int64_t start) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double cyc = f[i] * (double)(start + i) / fs;
        const double fr = cyc - floor(cyc);
        double sn, cs;
        sincospi(2.0 * fr, &sn, &cs);
        const float2 x = in[i];
        const double re = (double)x.x * cs + (double)x.y * sn;      // x * (cos - j sin)
        const double im = (double)x.y * cs - (double)x.x * sn;
        out[i] = make_float2((float)re, (float)im);
    }
}

extern "C" int dd_nco_c64_freqs(const float* in_c64, float* out_c64, int64_t n, const double* freqs_hz, double samp_rate,
                                int64_t start_index, void* stream) {
    DD_REQUIRE(n >= 0 && samp_rate > 0, "n / samp_rate");
    if (n == 0) return DD_OK;
    DD_REQUIRE(in_c64 && out_c64 && freqs_hz, "null buffer");
    hipLaunchKernelGGL(k_nco_c64_freqs, dim3(dd_grid_for(n, 256)), dim3(256), 0, dd_stream(stream),
                       (const float2*)in_c64, (float2*)out_c64, n, freqs_hz, samp_rate, start_index);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// ---------------------------------------------------------------- R1: decimation gather
template <typename T>
__global__ void __launch_bounds__(256) k_decimate(const T* __restrict__ in, T* __restrict__ out, int64_t n_out, int m, int off) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride) {
        out[i] = in[(int64_t)off + i * m];
    }
}

extern "C" int dd_decimate(const void* in, void* out, int64_t n, int m, int offset, int elem_bytes,
                           int64_t* n_out, void* stream) {
    DD_REQUIRE(n >= 0 && m >= 1 && offset >= 0, "n/m/offset");
    const int64_t no = (n > offset) ? (n - offset + m - 1) / m : 0;
    if (n_out) *n_out = no;
    if (no == 0) return DD_OK;
    DD_REQUIRE(in && out, "null buffer");
    const dim3 g(dd_grid_for(no, 256)), b(256);
    hipStream_t s = dd_stream(stream);
    switch (elem_bytes) {
        case 4: hipLaunchKernelGGL(k_decimate<float>, g, b, 0, s, (const float*)in, (float*)out, no, m, offset); break;
        case 8: hipLaunchKernelGGL(k_decimate<float2>, g, b, 0, s, (const float2*)in, (float2*)out, no, m, offset); break;
        case 16: hipLaunchKernelGGL(k_decimate<double2>, g, b, 0, s, (const double2*)in, (double2*)out, no, m, offset); break;
        default: dd_set_error("dd_decimate: elem_bytes must be 4, 8 or 16"); return DD_ERR_INVALID;
    }
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// ---------------------------------------------------------------- widening / narrowing
__global__ void __launch_bounds__(256) k_f32_to_f64(const float* __restrict__ in, double* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (double)in[i];
}
__global__ void __launch_bounds__(256) k_f64_to_f32(const double* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (float)in[i];
}
extern "C" int dd_f32_to_f64(const float* in, double* out, int64_t n, void* stream) {
    if (n <= 0) return DD_OK;
    hipLaunchKernelGGL(k_f32_to_f64, dim3(dd_grid_for(n, 256)), dim3(256), 0, dd_stream(stream), in, out, n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}
extern "C" int dd_f64_to_f32(const double* in, float* out, int64_t n, void* stream) {
    if (n <= 0) return DD_OK;
    hipLaunchKernelGGL(k_f64_to_f32, dim3(dd_grid_for(n, 256)), dim3(256), 0, dd_stream(stream), in, out, n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// ---------------------------------------------------------------- np.abs (demod_am.demod_amFLT, demod_am.py:62)
// kind 0: float64 -> |x| ; 1: complex128 ; 2: complex64 (both -> float64 magnitude, hypot like NumPy)
template <int KIND>
__global__ void __launch_bounds__(256) k_abs_f64(const void* __restrict__ in, double* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (KIND == 0) out[i] = fabs(reinterpret_cast<const double*>(in)[i]);
        else if (KIND == 1) { const double2 v = reinterpret_cast<const double2*>(in)[i]; out[i] = hypot(v.x, v.y); }
        else { const float2 v = reinterpret_cast<const float2*>(in)[i]; out[i] = hypot((double)v.x, (double)v.y); }
    }
}
extern "C" int dd_abs_f64(const void* in, int kind, double* out, int64_t n, void* stream) {
    DD_REQUIRE(kind >= 0 && kind <= 2 && n >= 0, "dd_abs_f64: kind / n");
    if (n == 0) return DD_OK;
    DD_REQUIRE(in && out, "dd_abs_f64: null buffer");
    const dim3 g(dd_grid_for(n, 256)), b(256);
    if (kind == 0) hipLaunchKernelGGL(k_abs_f64<0>, g, b, 0, dd_stream(stream), in, out, n);
    else if (kind == 1) hipLaunchKernelGGL(k_abs_f64<1>, g, b, 0, dd_stream(stream), in, out, n);
    else hipLaunchKernelGGL(k_abs_f64<2>, g, b, 0, dd_stream(stream), in, out, n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}
