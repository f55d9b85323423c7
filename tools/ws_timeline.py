"""Where a k-tile's cycles go in the producer / consumer kernel (experiments/gemmws.hip): `make -C neurons_amd/csrc experiments STAMP=1`, then
NR_LIB_VARIANT=exp python tools/ws_timeline.py.  Per role (producer 0, consumer 0 of every workgroup), cycles per k-tile spent waiting at barrier A,
issuing DMA / reading fragments, waiting at barrier B, waiting for the landing / in the MFMAs (medians over workgroups)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ.setdefault("NR_LIB_VARIANT", "exp")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

lib = ops._lib.load()
lib.nr_ws_stamp_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
lib.nr_ws_stamp_read.restype = C.c_int
dev = torch.device("cuda", 0)
os.environ["NR_IGEMM_WS"] = "2"
os.environ["NR_IGEMM_WS_SPLITK"] = "1"
nimg, H, W, N, Cin = 160, 16, 16, 640, 640
x = torch.randn(nimg, H, W, Cin, device=dev).to(torch.bfloat16)
w = (torch.randn(N, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(torch.bfloat16)
wt = w.reshape(N, 9, Cin // 64, 64).permute(0, 2, 1, 3).contiguous()
b = torch.randn(N, device=dev)
out = torch.empty(nimg, H, W, N, dtype=torch.bfloat16, device=dev)
nk = 9 * Cin // 64
for (nc, npd, ns) in ((4, 4, 3), (8, 4, 3), (8, 8, 3), (8, 8, 4), (4, 8, 4)):
    os.environ["NR_IGEMM_WS_NCONS"], os.environ["NR_IGEMM_WS_NPROD"], os.environ["NR_IGEMM_WS_NS"] = str(nc), str(npd), str(ns)
    for _ in range(3):
        ops._lib.check(lib.nr_op_conv3x3_tap_inner(ops._stream(), ops._ptr(x), Cin, nimg, H, W, ops._ptr(wt), ops._ptr(b), None, 1, None, ops._ptr(out), N))
    torch.cuda.synchronize()
    buf = np.zeros((512, 2, 4), dtype=np.uint64)
    assert lib.nr_ws_stamp_read(buf.ctypes.data, buf.nbytes, 1) == 0
    ops._lib.check(lib.nr_op_conv3x3_tap_inner(ops._stream(), ops._ptr(x), Cin, nimg, H, W, ops._ptr(wt), ops._ptr(b), None, 1, None, ops._ptr(out), N))
    torch.cuda.synchronize()
    assert lib.nr_ws_stamp_read(buf.ctypes.data, buf.nbytes, 0) == 0
    st = buf.astype(np.float64)
    st = st[st[:, 1, 3] > 0] / nk
    med = np.median(st, axis=0)
    print(f"c{nc}p{npd}s{ns}: producer  wait A {med[0,0]:6.0f}  issue {med[0,1]:6.0f}  wait B {med[0,2]:6.0f}  landing {med[0,3]:6.0f} | "
          f"consumer  wait A {med[1,0]:6.0f}  reads {med[1,1]:6.0f}  wait B {med[1,2]:6.0f}  MFMAs {med[1,3]:6.0f} | per k-tile {med[1].sum():6.0f} cycles", flush=True)
