"""In-kernel timelines of the small-M Linears (diagnostic build `make -C neurons_amd/csrc stamp`, loaded through NR_LIB_VARIANT=stamp):
shader-clock stamps of wave 0 of every workgroup.  Tiled igemm: kernel entry, after the prologue's DMA issue, behind every k-tile's
barrier, end of the k-loop, after the epilogue's stores have retired.  smallm: entry, panel DMA issued, weight loads issued, panel landed
+ barrier, end of the MFMA loop, stores retired.  Weights rotate through a pool larger than the Infinity Cache (pool MB > 0) so that every
launch streams them from HBM, as in the denoiser.  All numbers are medians over workgroups and launches, in shader cycles.
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
HAS_SMALLM = True
try:
    lib.nr_smallm_stamp_read
except AttributeError:      # a build without -DNR_STAMP in smallm.hip
    HAS_SMALLM = False
for fn in (lib.nr_stamp_read,) + ((lib.nr_smallm_stamp_read,) if HAS_SMALLM else ()):
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    fn.restype = C.c_int
SLOTS = 48


def stamps(smallm):
    buf = np.zeros((512, 40 if smallm else SLOTS), dtype=np.uint64)
    f = lib.nr_smallm_stamp_read if smallm else lib.nr_stamp_read
    assert f(buf.ctypes.data, buf.nbytes, 1) == 0
    return buf.astype(np.int64)


def run(M, N, K, res, pool_mb, smallm, force=None):
    npool = max(1, int(pool_mb * 1e6 / (N * K * 2)))
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(npool)]
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
    os.environ["NR_SMALLM"] = "2" if smallm else "0"
    if HAS_SMALLM and hasattr(lib, "nr_op_fm_cache_clear"):
        torch.cuda.synchronize()
        lib.nr_op_fm_cache_clear()          # NR_SMALLM_FM=1: the fragment-major copies are cached per weight pointer
    if force:
        os.environ["NR_IGEMM_FORCE"] = force
    else:
        os.environ.pop("NR_IGEMM_FORCE", None)
    rows = []
    stamps(smallm)
    for it in range(min(npool, 12) + 3):
        w = ws[it % npool]
        torch.cuda.synchronize()
        if os.environ.get("TIMELINE_GEGLU"):      # the GEGLU projection's epilogue and tile rules (no bias / residual)
            ops.gemm(a, w, geglu=True)
        else:
            ops.gemm(a, w, b, r)
        torch.cuda.synchronize()
        st = stamps(smallm)
        if it < 3:
            continue
        st = st[st[:, 0] > 0]
        t0 = st[:, 0].min()
        rt0, rt1 = (6, 7) if smallm else (44, 45)      # 100 MHz chip-wide counter at entry / exit: 24 shader cycles per tick at 2.4 GHz
        spread_us = float(st[:, rt0].max() - st[:, rt0].min()) / 100.0
        span_us = float(st[:, rt1].max() - st[:, rt0].min()) / 100.0
        if smallm:
            rows.append(dict(nwg=len(st), entry_spread_us=spread_us, dma_issued=np.median(st[:, 1] - st[:, 0]),
                             w_issued=np.median(st[:, 2] - st[:, 1]), panel_landed=np.median(st[:, 3] - st[:, 2]),
                             mfma_loop=np.median(st[:, 4] - st[:, 3]), epilogue=np.median(st[:, 5] - st[:, 4]),
                             wg_total=np.median(st[:, 5] - st[:, 0]), wg_max=(st[:, 5] - st[:, 0]).max(), kernel_span_us=span_us))
            nst = int((st[0, 8:38:3] > 0).sum())          # per step: wait (from the previous step's end), barrier, MFMAs + refill issue
            prev = st[:, 1]
            steps = []
            for t_ in range(nst):
                steps.append("%d/%d/%d" % (np.median(st[:, 8 + 3 * t_] - prev), np.median(st[:, 9 + 3 * t_] - st[:, 8 + 3 * t_]), np.median(st[:, 10 + 3 * t_] - st[:, 9 + 3 * t_])))
                prev = st[:, 10 + 3 * t_]
            rows[-1]["_steps"] = " ".join(steps)
        else:
            nk = int((st[0, 4:40] > 0).sum())
            rows.append(dict(nwg=len(st), nk=nk, entry_spread_us=spread_us, p_tileidx=np.median(st[:, 40] - st[:, 0]), p_rows=np.median(st[:, 41] - st[:, 40]),
                             p_kt=np.median(st[:, 42] - st[:, 41]), p_setup=np.median(st[:, 43] - st[:, 42]), p_issue=np.median(st[:, 1] - st[:, 43]), prologue=np.median(st[:, 1] - st[:, 0]),
                             first_tile=np.median(st[:, 4] - st[:, 1]),
                             per_tile=np.median((st[:, 4 + nk - 1] - st[:, 4]) / max(nk - 1, 1)),
                             loop=np.median(st[:, 2] - st[:, 1]), epilogue=np.median(st[:, 3] - st[:, 2]),
                             wg_total=np.median(st[:, 3] - st[:, 0]), wg_max=(st[:, 3] - st[:, 0]).max(), kernel_span_us=span_us))
    step_txt = rows[-1].pop("_steps", None)
    for r_ in rows:
        r_.pop("_steps", None)
    keys = list(rows[0].keys())
    med = {k: float(np.median([r_[k] for r_ in rows])) for k in keys}
    print(f"{'smallm' if smallm else 'igemm '} M={M} N={N} K={K} res={int(res)} pool={pool_mb}MB force={force}: " +
          " ".join(f"{k}={med[k]:.2f}" if k.endswith("_us") else f"{k}={med[k]:.0f}" for k in keys) +
          (f" steps(wait/barrier/mfma)=[{step_txt}]" if step_txt else ""), flush=True)


if os.environ.get("TIMELINE_SHAPES"):      # "M,N,K,res;M,N,K,res;..." : the tiled igemm only, hot and HBM-cold weights
    for pool in (0, 600):
        for spec in os.environ["TIMELINE_SHAPES"].split(";"):
            M_, N_, K_, r_ = (int(v) for v in spec.split(","))
            run(M_, N_, K_, bool(r_), pool, False, os.environ.get("TIMELINE_FORCE") or None)
            if HAS_SMALLM and not os.environ.get("TIMELINE_IGEMM_ONLY"):
                run(M_, N_, K_, bool(r_), pool, True)
    sys.exit(0)
for pool in (0, 600):
    for smallm in (False,) if (os.environ.get("TIMELINE_IGEMM_ONLY") or not HAS_SMALLM) else (False, True):
        run(512, 1280, 1280, True, pool, smallm)
        run(128, 1280, 1280, True, pool, smallm)
        run(512, 3840, 1280, False, pool, smallm)
        run(512, 640, 640, True, pool, smallm)
        if not smallm:
            run(8192, 1920, 640, False, pool, smallm)
            run(2048, 1280, 1280, True, pool, smallm)
