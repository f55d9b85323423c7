"""In-kernel timeline of the fused cross-attention block (diagnostic build `make -C neurons_amd/csrc stamp`, NR_LIB_VARIANT=stamp): shader-clock stamps
of wave 0 of the first 256 workgroups at the stage boundaries of xattn.hip.  Medians over workgroups, in shader cycles, per head:
  q_wait   = arrival at the q stage -> behind its DMA wait + barrier        q_gemm = the 60 projection MFMAs (+ 10 prefetch pieces)
  kv_wait  = same for the kv stage                                          attn   = scores, softmax, P.V for two row tiles (+ 8 prefetch pieces)
  o_wait   = same for the o stage                                           o_gemm = the 80 out-projection MFMAs (+ 5 prefetch pieces)
Usage (GPU box): NR_LIB_VARIANT=stamp python tools/xattn_timeline.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ.setdefault("NR_LIB_VARIANT", "stamp")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.nr_xattn_stamp_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
lib.nr_xattn_stamp_read.restype = C.c_int
dev = torch.device("cuda", 0)
nimg, hw, ipc, Lk, Cc = 32, 1024, 16, 77, 320
g = torch.Generator(device=dev).manual_seed(0)
t = torch.randn(nimg * hw, Cc, generator=g, device=dev).to(torch.bfloat16)
gamma, beta, bo = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
wq, wo = (torch.randn(Cc, Cc, generator=g, device=dev) * Cc ** -0.5 for _ in range(2))
kv = torch.randn(2 * Lk, 2 * Cc, generator=g, device=dev).to(torch.bfloat16)
buf = np.zeros((256, 64), dtype=np.uint64)
rows = []
for it in range(8):
    ops.xattn_fused(t, nimg, hw, ipc, gamma, beta, wq, wo, bo, kv, Lk, reuse_streams=it > 0)
    torch.cuda.synchronize()
    assert lib.nr_xattn_stamp_read(buf.ctypes.data, buf.nbytes, 1) == 0
    if it < 3:
        continue
    st = buf.astype(np.int64)
    st = st[st[:, 0] > 0]
    d = {"prologue": st[:, 1] - st[:, 0], "epilogue": st[:, 51] - st[:, 50], "total": st[:, 51] - st[:, 0]}
    for k, (a, b) in {"q_wait": (2, 3), "q_gemm": (3, 4), "kv_wait": (4, 5), "attn": (5, 6), "o_wait": (6, 7)}.items():
        d[k] = np.mean([st[:, b + 6 * h] - st[:, a + 6 * h] for h in range(8)], axis=0)
    d["o_gemm"] = np.mean([(st[:, 2 + 6 * (h + 1)] if h < 7 else st[:, 50]) - st[:, 7 + 6 * h] for h in range(8)], axis=0)
    rows.append({k: float(np.median(v)) for k, v in d.items()})
med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
print("xattn_fused M=32768 (256 workgroups), median cycles: " + " ".join(f"{k}={v:.0f}" for k, v in med.items()))
print("per head: " + " ".join(f"{k}={med[k]:.0f}" for k in ("q_wait", "q_gemm", "kv_wait", "attn", "o_wait", "o_gemm")) +
      f" | sum {sum(med[k] for k in ('q_wait', 'q_gemm', 'kv_wait', 'attn', 'o_wait', 'o_gemm')):.0f}")
