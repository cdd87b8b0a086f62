"""20 calls of dd_iir_f64 over 2^26 complex128 (for rocprofv3 --kernel-trace --stats)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import scipy.signal as ss
from directdemod_amd import _hip
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << int(os.environ.get("LOG2N", "26"))
x = torch.randn(n, 2, dtype=torch.float64, device=dev)
y = torch.empty_like(x)
b, a = ss.butter(6, 100000.0 / 1.2e6)
b = np.ascontiguousarray(b); a = np.ascontiguousarray(a)
h = C.c_void_p()
dp = C.POINTER(C.c_double)
_hip.check(lib.dd_iir_create(C.byref(h), b.ctypes.data_as(dp), a.ctypes.data_as(dp), len(b), None), "create")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(20):
    _hip.check(lib.dd_iir_f64(h, x.data_ptr(), y.data_ptr(), n, 1, 1, stream), "iir")
torch.cuda.synchronize()
