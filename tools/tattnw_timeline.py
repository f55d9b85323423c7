"""In-kernel timeline of the temporal attention head kernel (tattnw.hip; diagnostic build `make -C neurons_amd/csrc stamp`, NR_LIB_VARIANT=stamp):
shader-clock stamps of wave 0 of the first 512 workgroups.  Per shape: medians over workgroups (cycles) of prologue, per-stage wait / barrier /
compute, the epilogue (fold + attention + stores), workgroup lifetime, the span first entry -> last exit, and which XCDs ran the heads of one pixel group.
Usage (GPU box): NR_LIB_VARIANT=stamp python tools/tattnw_timeline.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ.setdefault("NR_LIB_VARIANT", "stamp")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.nr_tattnw_stamp_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
lib.nr_tattnw_stamp_read.restype = C.c_int
dev = torch.device("cuda", 0)
if len(sys.argv) > 1 and sys.argv[1] == "xattn":
    # the cross-attention head kernel (xattnw.hip): same stamps, + the K | V image wait behind the loop
    lib.nr_xattnw_stamp_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lib.nr_xattnw_stamp_read.restype = C.c_int
    for Cc, nimg, hw in ((640, 32, 256), (1280, 32, 64)):
        S = Cc // 32
        g = torch.Generator(device=dev).manual_seed(0)
        t = torch.randn(nimg * hw, Cc, generator=g, device=dev).to(torch.bfloat16)
        gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
        wq = torch.randn(Cc, Cc, generator=g, device=dev) * Cc ** -0.5
        kv = torch.randn(2 * 77, 2 * Cc, generator=g, device=dev).to(torch.bfloat16)
        buf = np.zeros((512, 128), dtype=np.uint64)
        rows = []
        for it in range(8):
            ops.xattn_head(t, nimg, hw, 16, gamma, beta, wq, kv, 77, reuse_streams=it > 0)
            torch.cuda.synchronize()
            assert lib.nr_xattnw_stamp_read(buf.ctypes.data, buf.nbytes, 1) == 0
            if it < 3:
                continue
            st = buf.astype(np.int64)
            st = st[st[:, 0] > 0]
            wait = np.mean([st[:, 2 + 3 * s] - (st[:, 1] if s == 0 else st[:, 4 + 3 * (s - 1)]) for s in range(S)], axis=0)
            bar = np.mean([st[:, 3 + 3 * s] - st[:, 2 + 3 * s] for s in range(S)], axis=0)
            comp = np.mean([st[:, 4 + 3 * s] - st[:, 3 + 3 * s] for s in range(S)], axis=0)
            rows.append(dict(wgs=len(st), prologue=np.median(st[:, 1] - st[:, 0]), wait=np.median(wait), barrier=np.median(bar), compute=np.median(comp),
                             loop=np.median(st[:, 1 + 3 * S] - st[:, 1]), kv_issue=np.median(st[:, 123] - st[:, 1 + 3 * S]), kv_wait=np.median(st[:, 124] - st[:, 123]),
                             attention=np.median(st[:, 125] - st[:, 124]), life=np.median(st[:, 125] - st[:, 0])))
        med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
        print(f"xattn_head C={Cc} M={nimg * hw} ({int(med['wgs'])} stamped workgroups, {S} stages): " + " ".join(f"{k}={v:.0f}" for k, v in med.items() if k != "wgs"))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "lin160p":
    # the register-panel form of lin160.hip (LayerNorm-folded wide projections): the first 40 stages of a workgroup are stamped; "block_gap" = from the last
    # stage of a column block to the first wait of the next one (the block's epilogue)
    lib.nr_lin160_stamp_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lib.nr_lin160_stamp_read.restype = C.c_int
    for M, K, N, geglu in ((8192, 640, 5120, True), (8192, 640, 1920, False), (2048, 1280, 10240, True), (2048, 1280, 3840, False)):
        S = 5                             # stages per column block (128 channels; K = 1280: 256)
        g = torch.Generator(device=dev).manual_seed(0)
        a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        w = torch.randn(N, K, generator=g, device=dev) * K ** -0.5
        bias = torch.zeros(N, device=dev)
        gamma, beta = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        buf = np.zeros((512, 128), dtype=np.uint64)
        rows = []
        for it in range(8):
            ops.ln_gemm(a, w, gamma, beta, bias=bias, geglu=geglu)
            torch.cuda.synchronize()
            assert lib.nr_lin160_stamp_read(buf.ctypes.data, buf.nbytes, 1) == 0
            if it < 3:
                continue
            st = buf.astype(np.int64)
            st = st[st[:, 0] > 0]
            NST = 40 if st[:, 2 + 3 * 39].min() > 0 else S * int((st[0, 2:122:3] > 0).sum() // S)
            inblock = [s_ for s_ in range(NST) if s_ % S != 0]
            wait = np.mean([st[:, 2 + 3 * s_] - st[:, 4 + 3 * (s_ - 1)] for s_ in inblock], axis=0)
            gaps = [st[:, 2 + 3 * s_] - st[:, 4 + 3 * (s_ - 1)] for s_ in range(S, NST, S)]
            bar = np.mean([st[:, 3 + 3 * s_] - st[:, 2 + 3 * s_] for s_ in range(NST)], axis=0)
            comp = np.mean([st[:, 4 + 3 * s_] - st[:, 3 + 3 * s_] for s_ in range(NST)], axis=0)
            # stamps of stage g: 2 + 3 g = in front of the barrier of stage g + 1 (which sits before the LAST k-step of stage g), 3 + 3 g behind it, 4 + 3 g = stage g's MFMAs issued.
            # "k012" = end of stage g - 1 -> that barrier (k-steps 0 .. 2 of stage g + the DMA wait), "k3" = the last k-step; "block_first" = k012 of the first stage of a
            # block (it holds the previous block's epilogue)
            rows.append(dict(wgs=len(st), stamped_stages=NST, prologue=np.median(st[:, 1] - st[:, 0]), pro_dma_issue=np.median(st[:, 122] - st[:, 0]), pro_table=np.median(st[:, 123] - st[:, 122]),
                             pro_x_issue=np.median(st[:, 124] - st[:, 123]), pro_stats=np.median(st[:, 127] - st[:, 124]), pro_barrier=np.median(st[:, 1] - st[:, 127]),
                             k012=np.median(wait), barrier=np.median(bar),
                             k3=np.median(comp), block_first=np.median(np.mean(gaps, axis=0)) if gaps else 0.0,
                             stage_avg=np.median((st[:, 4 + 3 * (NST - 1)] - st[:, 1]) / NST), life=np.median(st[:, 126] - st[:, 0])))
            if it == 7 and (buf[256:512, 64] > 0).any():
                # fine stamps of stage 11: wave 0 (rows < 256) and its SIMD-mate wave 4, relative to wave 0's k-step 0 begin
                full = buf.astype(np.int64)
                w0, w4 = full[:256], full[256:512]
                ok = (w0[:, 64] > 0) & (w4[:, 64] > 0)
                t0 = w0[ok, 64]
                names = ["k0", "k0 dma", "k0 mfma", "k1", "k1 dma", "k1 mfma", "k2", "k2 dma", "k2 mfma", "k3", "k3 dma", "k3 mfma", "pre-wait", "post-wait", "post-barrier"]
                print("   stage 11, medians rel. to wave 0's k0: " + "  ".join(f"{n_}: {np.median(w0[ok, 64 + i_] - t0):.0f}/{np.median(w4[ok, 64 + i_] - t0):.0f}" for i_, n_ in enumerate(names)))
        med = {k: float(np.median([q[k] for q in rows])) for k in rows[0]}
        print(f"lin160p M={M} N={N} K={K} geglu={int(geglu)} ({int(med['wgs'])} stamped workgroups, {S} stages per block): " + " ".join(f"{k}={v:.0f}" for k, v in med.items() if k != "wgs"))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "lin160":
    # the short-K Linear kernel (lin160.hip): 64-channel stages
    lib.nr_lin160_stamp_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lib.nr_lin160_stamp_read.restype = C.c_int
    for M, N, K in ((8192, 640, 640), (2048, 1280, 1280)):
        S = K // 64
        g = torch.Generator(device=dev).manual_seed(0)
        a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
        r = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
        bias = torch.zeros(N, device=dev)
        buf = np.zeros((512, 128), dtype=np.uint64)
        for res in (None, r):
            rows = []
            for it in range(8):
                ops.gemm(a, w, bias=bias, res=res)
                torch.cuda.synchronize()
                assert lib.nr_lin160_stamp_read(buf.ctypes.data, buf.nbytes, 1) == 0
                if it < 3:
                    continue
                st = buf.astype(np.int64)
                st = st[st[:, 0] > 0]
                wait = np.mean([st[:, 2 + 3 * s] - (st[:, 1] if s == 0 else st[:, 4 + 3 * (s - 1)]) for s in range(S)], axis=0)
                bar = np.mean([st[:, 3 + 3 * s] - st[:, 2 + 3 * s] for s in range(S)], axis=0)
                comp = np.mean([st[:, 4 + 3 * s] - st[:, 3 + 3 * s] for s in range(S)], axis=0)
                rows.append(dict(wgs=len(st), prologue=np.median(st[:, 1] - st[:, 0]), first_wait=np.median(st[:, 2] - st[:, 1]), wait=np.median(wait), barrier=np.median(bar),
                                 compute=np.median(comp), loop=np.median(st[:, 125] - st[:, 1]), epilogue=np.median(st[:, 126] - st[:, 125]), life=np.median(st[:, 126] - st[:, 0])))
            med = {k: float(np.median([q[k] for q in rows])) for k in rows[0]}
            print(f"lin160 M={M} N={N} K={K} res={int(res is not None)} ({int(med['wgs'])} stamped workgroups, {S} stages): " + " ".join(f"{k}={v:.0f}" for k, v in med.items() if k != "wgs"))
    sys.exit(0)
for Cc, nbatch, hw in ((640, 2, 256), (1280, 2, 64), (1280, 2, 16)):
    S = Cc // 32
    g = torch.Generator(device=dev).manual_seed(0)
    t = torch.randn(nbatch * 16 * hw, Cc, generator=g, device=dev).to(torch.bfloat16)
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    wq, wk, wv = (torch.randn(Cc, Cc, generator=g, device=dev) * Cc ** -0.5 for _ in range(3))
    buf = np.zeros((512, 128), dtype=np.uint64)
    rows = []
    for it in range(8):
        ops.tattn_head(t, nbatch, hw, gamma, beta, wq, wk, wv, reuse_stream=it > 0)
        torch.cuda.synchronize()
        assert lib.nr_tattnw_stamp_read(buf.ctypes.data, buf.nbytes, 1) == 0
        if it < 3:
            continue
        st = buf.astype(np.int64)
        st = st[st[:, 0] > 0]
        wait = np.mean([st[:, 2 + 3 * s] - (st[:, 1] if s == 0 else st[:, 4 + 3 * (s - 1)]) for s in range(S)], axis=0)
        bar = np.mean([st[:, 3 + 3 * s] - st[:, 2 + 3 * s] for s in range(S)], axis=0)
        comp = np.mean([st[:, 4 + 3 * s] - st[:, 3 + 3 * s] for s in range(S)], axis=0)
        first = st[:, 2] - st[:, 1]
        rows.append(dict(wgs=len(st), prologue=np.median(st[:, 1] - st[:, 0]), first_wait=np.median(first), wait=np.median(wait), barrier=np.median(bar),
                         compute=np.median(comp), loop=np.median(st[:, 125] - st[:, 1]), epilogue=np.median(st[:, 126] - st[:, 125]),
                         life=np.median(st[:, 126] - st[:, 0]), span=float(st[:, 126].max() - st[:, 0].min())))
        xcc = st[:, 127]
    med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
    print(f"tattn_head C={Cc} M={nbatch * 16 * hw} ({int(med['wgs'])} stamped workgroups, {S} stages): " + " ".join(f"{k}={v:.0f}" for k, v in med.items() if k != "wgs"))
    print(f"   per stage: wait {med['wait']:.0f} + barrier {med['barrier']:.0f} + compute {med['compute']:.0f} = {med['wait'] + med['barrier'] + med['compute']:.0f} cycles;"
          f" XCC ids of workgroups 0..15: {xcc[:16].tolist()}")
