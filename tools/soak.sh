#!/bin/bash
# soak: many launches of the persistent kernels (barrier / LDS-counter hand-overs), bounded by timeouts
timeout 120 python bench.py --no-cpu-baseline --steps 20000 --warmup 10 --ramp-ms 0 --steady-ms 100 --no-side | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['kernel'], '20000 steps:', d['value'], d['ms_per_step'], d['extra']['output_rms_rad'])"
timeout 600 python - <<'PY'
# the three M = 1 FM kernels (k_chain_cos1k, the default for Hamming 255 since round 5, k_chain_fft1k and k_chain_mfma_ab): the whole output must be bit-identical
# from launch to launch (no atomics on the data path: a race in the LDS hand-overs -- exchange images, boundary tables, range
# slots, plane buffers -- would show as a changing checksum)
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from directdemod_amd import _hip
import bench
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 26
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
taps = np.ascontiguousarray(0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(255) / 254))
for kern, u8 in (("cos1k", False), ("cos1k", True), ("fft1k", False), ("fft1k", True), ("ab", False), ("ab", True)):
    _hip.select_kernel(kern)
    want = {"cos1k": _hip.DD_KERNEL_COS_RS, "fft1k": _hip.DD_KERNEL_FFT_OS, "ab": _hip.DD_KERNEL_MFMA_AB}[kern]
    x = bench.make_input(torch, n, 0, dev, 11)
    if u8:
        x = (x + 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
    out = torch.empty(n, dtype=torch.float32, device=dev)
    h = C.c_void_p()
    _hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 255, _hip.cycles_q64(25000.0, 2400000), 1,
                                   _hip.DD_CHAIN_NCO | _hip.DD_CHAIN_FM | (_hip.DD_CHAIN_U8_INPUT if u8 else 0)), "create")
    got = C.c_int64(0)
    ref = None
    t0 = time.time()
    for i in range(3001):
        lib.dd_chain_reset(h, stream)
        _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
        if i % 500 == 0:
            torch.cuda.synchronize()
            assert lib.dd_chain_last_kernel(h) == want
            s = (int(out[:n - 1].view(torch.int32).to(torch.int64).sum()), float(out[:n - 1].double().abs().max()))
            assert ref is None or s == ref, (u8, i, s, ref)
            ref = s
    torch.cuda.synchronize()
    lib.dd_chain_destroy(h)
    print("%s%s: 3000 launches, bit-identical output (checksum %d), %.1f s" % (kern, " (u8 input)" if u8 else "", ref[0], time.time() - t0))
    del x, out
PY
timeout 400 python - <<'PY'
# round 6: k_chain_decim_b (block sums on the matrix pipe; hand-placed LDS waits) -- the C4 and C3 shapes from complex64 and from raw u8, 3000 launches each: the WHOLE
# output must be bit-identical from launch to launch (a read that is waited for one instruction too late would show as a changing checksum)
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch, scipy.signal as ss
from directdemod_amd import _hip
import bench
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 26
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = bench.make_input(torch, n, 0, dev, 7)
x8 = (x + 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
k = np.arange(151)
bh = 0.35875 - 0.48829 * np.cos(2 * np.pi * k / 150) + 0.14128 * np.cos(4 * np.pi * k / 150) - 0.01168 * np.cos(6 * np.pi * k / 150)
rz = ss.remez(127, [0, 100e3, 150e3, 4999999], [1, 0], fs=1e7)
for name, taps, M, f, fs in (("C4 BH151 /34", bh, 34, 30000.0, 2048000), ("C3 remez127 /50", rz, 50, 250000.0, 10000000)):
    taps = np.ascontiguousarray(taps, dtype=np.float64)
    for u8 in (False, True):
        h = C.c_void_p()
        _hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), len(taps), _hip.cycles_q64(f, fs), M,
                                       _hip.DD_CHAIN_NCO | _hip.DD_CHAIN_FM | (_hip.DD_CHAIN_U8_INPUT if u8 else 0)), "create")
        out = torch.zeros(n // M + 8, dtype=torch.float32, device=dev)
        src = x8 if u8 else x
        got = C.c_int64(0)
        ref = None
        t0 = time.time()
        for i in range(3001):
            lib.dd_chain_reset(h, stream)
            _hip.check(lib.dd_chain_process(h, src.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
            if i % 250 == 0:
                torch.cuda.synchronize()
                assert lib.dd_chain_last_kernel(h) == _hip.DD_KERNEL_DECIM_BLOCKS
                sm = (int(out[:got.value].view(torch.int32).to(torch.int64).sum()), int((out[:got.value].view(torch.int32).to(torch.int64) * torch.arange(got.value, device=dev) % 1000003).sum()))
                assert ref is None or sm == ref, (name, u8, i, sm, ref)
                ref = sm
        lib.dd_chain_destroy(h)
        print("%s%s (k_chain_decim_b): 3000 launches, bit-identical output (checksums %d, %d), %.1f s" % (name, " from raw u8" if u8 else "", ref[0], ref[1], time.time() - t0))
PY
timeout 300 python - <<'PY'
# the chunk-list launch (k_chain_decim_w: the list as one chunk on the absolute sample grid; DD_MFMA_KERNEL=decimp: k_chain_decim_multi, carried state
# handed from chunk to chunk through device memory and flags inside ONE launch) -- 4000 launches of the C3 shape (16 chunks) and of a ragged list, output bit-identical every time and equal to the loop's
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch, scipy.signal as ss
from directdemod_amd import _hip
import bench
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 26
x = bench.make_input(torch, n, 0, dev, 7)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rz = np.ascontiguousarray(ss.remez(127, [0, 100e3, 150e3, 4999999], [1, 0], fs=1e7))
for name, bounds in (("16 x 2^22", [i << 22 for i in range(17)]), ("ragged", [0, 70001, 70001 + 4096, 1 << 20, (1 << 20) + 100, 1 << 25, (1 << 25) + 1234567, n])):
    h = C.c_void_p()
    _hip.check(lib.dd_chain_create(C.byref(h), rz.ctypes.data_as(C.POINTER(C.c_double)), 127, _hip.cycles_q64(250000.0, 10000000), 50, _hip.DD_CHAIN_NCO | _hip.DD_CHAIN_FM), "create")
    nch = len(bounds) - 1
    cb = (C.c_int64 * (nch + 1))(*bounds)
    cn = (C.c_int64 * nch)()
    out = torch.zeros(n // 50 + 16, dtype=torch.float32, device=dev)
    loop = torch.zeros_like(out)
    lib.dd_chain_reset(h, stream)
    pos = 0
    for i in range(nch):
        got = C.c_int64(0)
        _hip.check(lib.dd_chain_process(h, x.data_ptr() + 8 * bounds[i], loop.data_ptr() + 4 * pos, bounds[i + 1] - bounds[i], C.byref(got), stream), "process")
        pos += got.value
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(4001):
        lib.dd_chain_reset(h, stream)
        _hip.check(lib.dd_chain_process_chunks(h, x.data_ptr(), out.data_ptr(), cb, nch, cn, stream), "chunks")
        if i % 500 == 0:
            torch.cuda.synchronize()
            assert lib.dd_chain_last_kernel(h) in (_hip.DD_KERNEL_DECIM_WAVE, _hip.DD_KERNEL_DECIM_BLOCKS, _hip.DD_KERNEL_DECIM_MULTI) and sum(cn) == pos
            assert torch.equal(out.view(torch.int32), loop.view(torch.int32)), (name, i)
    torch.cuda.synchronize()
    lib.dd_chain_destroy(h)
    print("chunk list %s: 4000 launches, every checked output bit-identical to the chunk loop, %.1f s" % (name, time.time() - t0))
PY
# round 6: the IIR block passes (counted vmcnt waits around LDS-DMA steps): outputs bit-identical from launch to launch
timeout 600 python tools/debug/iir_soak.py
