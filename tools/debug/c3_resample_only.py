import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from directdemod_amd import _hip
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
nch = 16
lens = [83886 if i % 2 == 0 else 83887 for i in range(nch)]
nums = [4624] * nch
x = torch.randn(sum(lens), dtype=torch.float32, device=dev)
out = torch.empty(sum(nums), dtype=torch.float64, device=dev)
A = C.c_int64 * nch
ioff = A(*[sum(lens[:i]) for i in range(nch)]); ln = A(*lens); ooff = A(*[sum(nums[:i]) for i in range(nch)]); nm = A(*nums)
for _ in range(5):
    _hip.check(lib.dd_resample_fft_chunks(x.data_ptr(), 1, ioff, ln, out.data_ptr(), ooff, nm, nch, stream), "rs")
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    _hip.check(lib.dd_resample_fft_chunks(x.data_ptr(), 1, ioff, ln, out.data_ptr(), ooff, nm, nch, stream), "rs")
e1.record(); torch.cuda.synchronize()
print("resample 16 chunks: %.4f ms per call" % (e0.elapsed_time(e1) / 20))
