"""Launch-to-launch determinism of the headline kernel: where do two launches over the same input differ?"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from directdemod_amd import _hip
import bench
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << int(os.environ.get("LOG2N", "26"))
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
taps = np.ascontiguousarray(0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(255) / 254))
x = bench.make_input(torch, n, 0, dev, 11)
out = torch.empty(n, dtype=torch.float32, device=dev)
h = C.c_void_p()
_hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 255, _hip.cycles_q64(25000.0, 2400000), 1,
                               _hip.DD_CHAIN_NCO | _hip.DD_CHAIN_FM), "create")
got = C.c_int64(0)
ref = None
nbad = 0
for i in range(int(os.environ.get("LAUNCHES", "400"))):
    lib.dd_chain_reset(h, stream)
    _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
        continue
    d = (out[:n - 1].view(torch.int32) != ref[:n - 1].view(torch.int32)).nonzero().flatten()
    if d.numel():
        nbad += 1
        idx = d.cpu().numpy()
        a = out[d].cpu().numpy(); b = ref[d].cpu().numpy()
        tl = (idx + 32) // 4064
        tiles, cnt = np.unique(tl, return_counts=True)
        within = (idx + 32) - 4064 * tl
        hist = np.bincount(within // 256, minlength=16)
        print("  tiles differing %d; counts/tile min %d max %d; first tiles %s; mod2 %s mod3 %s; within-tile/256 hist %s" %
              (tiles.size, cnt.min(), cnt.max(), tiles[:12], np.bincount(tiles % 2, minlength=2), np.bincount(tiles % 3, minlength=3), hist))
        gaps = np.diff(tiles)
        print("  tile-id gaps histogram (1..8):", np.bincount(np.minimum(gaps, 9), minlength=10)[1:], " lanes(idx%64) hist nonzero:", np.count_nonzero(np.bincount(idx % 64, minlength=64)))
        print("launch %d: %d samples differ, tiles(8192) %s, idx range [%d..%d], max|diff| %.3e, first: idx %s got %s ref %s" %
              (i, idx.size, tiles[:4], idx.min(), idx.max(), np.abs(a - b).max(), idx[:6], a[:6], b[:6]), flush=True)
        if nbad >= 12:
            break
print("launches with differences:", nbad)
