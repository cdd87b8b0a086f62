import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, scipy.signal
from directdemod_amd import _hip as hip
import bench
hip.require_gpu(); lib = hip.lib(); dev = torch.device("cuda", 0)
n = 1 << 26
x = bench.make_input(torch, n, 0, dev, 1)
taps = np.ascontiguousarray(scipy.signal.remez(127, [0, 100e3, 150e3, 4999999], [1, 0], fs=1e7))
h = C.c_void_p()
hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 127, hip.cycles_q64(250000.0, 1e7), 50, hip.DD_CHAIN_NCO | hip.DD_CHAIN_FM))
out = torch.empty(n // 50 + 8, dtype=torch.float32, device=dev)
got = C.c_int64(0)
ch = 1 << 22
for rep in range(20):
    lib.dd_chain_reset(h, None)
    o = 0
    for a in range(0, n, ch):
        hip.check(lib.dd_chain_process(h, x.data_ptr() + 8 * a, out.data_ptr() + 4 * o, ch, C.byref(got), None))
        o += got.value
torch.cuda.synchronize()
