"""Experiment: IIR over 2^26 complex128 as one call against k calls over sub-ranges with carried state (does the second
pass of a sub-range find its input in the 256 MB infinity cache?)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import scipy.signal as ss
from directdemod_amd import _hip
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 26
x = torch.randn(n, 2, dtype=torch.float64, device=dev)
y = torch.empty_like(x)
b, a = ss.butter(6, 100000.0 / 1.2e6)
b = np.ascontiguousarray(b); a = np.ascontiguousarray(a)
zi = np.ascontiguousarray(ss.lfilter_zi(b, a))
h = C.c_void_p()
dp = C.POINTER(C.c_double)
_hip.check(lib.dd_iir_create(C.byref(h), b.ctypes.data_as(dp), a.ctypes.data_as(dp), len(b), zi.ctypes.data_as(dp)), "create")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ref = None
for log2c in (26, 25, 24, 23, 22, 21):
    c = 1 << log2c
    def run():
        # fresh state per pass: recreate is host-side only; instead run carry=1 over chunks after a reset through create
        for s in range(0, n, c):
            _hip.check(lib.dd_iir_f64(h, x.data_ptr() + 16 * s, y.data_ptr() + 16 * s, c, 1, 1, stream), "iir")
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    print("chunks of 2^%d: %.3f ms per 2^26 samples" % (log2c, e0.elapsed_time(e1) / 5), flush=True)
