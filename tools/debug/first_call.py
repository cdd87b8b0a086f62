import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from directdemod_amd import _hip
import bench
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 26
x = bench.make_input(torch, n, 0, dev, 3)
out = torch.zeros(n, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
taps = np.ascontiguousarray(0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(255) / 254))
for rep in range(2):
    h = C.c_void_p()
    t0 = time.perf_counter()
    _hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 255, _hip.cycles_q64(25000.0, 2400000), 1, _hip.DD_CHAIN_NCO | _hip.DD_CHAIN_FM), "create")
    t1 = time.perf_counter()
    got = C.c_int64(0)
    lib.dd_chain_reset(h, stream)
    t2 = time.perf_counter()
    _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
    t5 = time.perf_counter()
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    print("handle %d: create %.3f ms, reset %.3f, first process call %.3f (+ %.3f to completion), second call %.3f (+ %.3f)" % (rep, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3, (t6-t5)*1e3))
