"""In-kernel timeline of the tiled igemm on the small-M Linears of the sgm keyframe path (diagnostic build `make -C neurons_amd/csrc stamp`,
loaded through NR_LIB_VARIANT=stamp): shader-clock stamps of wave 0 of every workgroup at kernel entry, after the prologue's DMA
issue, behind every k-tile's barrier, at the end of the k-loop and after the epilogue's stores have retired.  Weights rotate through a
pool larger than the Infinity Cache so that every launch streams them from HBM, as in the denoiser.
Usage (GPU box): NR_LIB_VARIANT=stamp python tools/igemm_timeline.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ.setdefault("NR_LIB_VARIANT", "stamp")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
lib.nr_stamp_read.argtypes = [C.c_void_p, C.c_size_t]
lib.nr_stamp_read.restype = C.c_int
SLOTS = 48


def stamps():
    buf = np.zeros((512, SLOTS), dtype=np.uint64)
    assert lib.nr_stamp_read(buf.ctypes.data, buf.nbytes) == 0
    return buf.astype(np.int64)


def run(M, N, K, res, pool_mb, force=None, geglu=False):
    npool = max(1, int(pool_mb * 1e6 / (N * K * 2)))
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(npool)]
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N // 2 if geglu else N, device=dev).to(torch.bfloat16) if res else None
    if force:
        os.environ["NR_IGEMM_FORCE"] = force
    else:
        os.environ.pop("NR_IGEMM_FORCE", None)
    rows = []
    for it in range(min(npool, 12) + 3):
        w = ws[it % npool]
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.gemm(a, w, b, r, geglu=geglu)
        e.record()
        torch.cuda.synchronize()
        st = stamps()
        if it < 3:
            continue
        live = st[:, 0] > 0
        st = st[live]
        nk = int(((st[0, 4:] > 0)).sum())
        t0 = st[:, 0].min()
        rows.append(dict(ev_us=s.elapsed_time(e) * 1e3, nwg=int(live.sum()), nk=nk,
                         entry_spread=(st[:, 0].max() - t0), prologue=np.median(st[:, 1] - st[:, 0]),
                         first_tile=np.median(st[:, 4] - st[:, 1]),
                         per_tile=np.median((st[:, 4 + nk - 1] - st[:, 4]) / max(nk - 1, 1)),
                         loop=np.median(st[:, 2] - st[:, 1]), epilogue=np.median(st[:, 3] - st[:, 2]),
                         wg_total=np.median(st[:, 3] - st[:, 0]), kernel_span=(st[:, 3].max() - t0)))
    keys = list(rows[0].keys())
    med = {k: float(np.median([r_[k] for r_ in rows])) for k in keys}
    print(f"M={M} N={N} K={K} res={int(res)} geglu={int(geglu)} pool={pool_mb}MB force={force}: " +
          " ".join(f"{k}={med[k]:.0f}" if k != "ev_us" else f"{k}={med[k]:.1f}" for k in keys), flush=True)


# memtime ticks: 100 MHz on gfx950 (constant rate)?  Print the ratio against wall time so the unit is explicit.
for pool in (0, 600):
    run(512, 1280, 1280, True, pool)
    run(512, 1280, 1280, True, pool, force="64,64,1,2,-1,4")
    run(512, 1280, 1280, True, pool, force="64,32,1,8,-1,4")
    run(512, 3840, 1280, False, pool)
    run(512, 1280, 5120, True, pool)
    run(2048, 1280, 1280, True, pool)
    run(2048, 640, 640, True, pool)
    run(8192, 640, 640, True, pool)
