"""A/B of the panel-resident small-M GEMM (smallm.hip, fragment-major weights) against the tiled igemm on the M <= 512 Linears of the U-Net 4x4 level and the sgm
keyframe model: same inputs, NR_SMALLM=0 vs 2, max |diff| of the outputs, and the time per launch inside a replayed graph of 48 launches that
walk a pool of distinct weight tensors larger than the Infinity Cache (every launch streams its weights from HBM, as in the denoiser).
Usage (GPU box): python tools/smallm_ab.py > gpurun_out/smallm_ab.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurons_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
# (M, N, K, kind) kind: plain | res | ln | geglu | lngeglu | rv
SHAPES = [(512, 1280, 1280, "res"), (512, 1280, 1280, "ln"), (512, 3840, 1280, "ln"), (512, 10240, 1280, "lngeglu"), (512, 1280, 5120, "res"),
          (512, 1280, 6400, "res"), (512, 1280, 2560, "plain"), (512, 1280, 1280, "rv"), (500, 1280, 1280, "res"), (128, 1280, 1280, "res"),
          (512, 640, 640, "res"), (512, 1920, 640, "ln"), (512, 5120, 640, "geglu"), (2048, 640, 640, "res"), (2048, 1280, 1280, "res")]
NCALL = 48
if os.environ.get("SMALLM_AB_ONLY"):          # comma-separated indices into SHAPES
    SHAPES = [SHAPES[int(i)] for i in os.environ["SMALLM_AB_ONLY"].split(",")]


def make(M, N, K, kind, npool):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(npool)]
    b = torch.randn(N, device=dev)
    nout = N // 2 if "geglu" in kind else N
    r = torch.randn(M, nout, device=dev).to(torch.bfloat16)
    gamma, beta = 1.0 + 0.1 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev)
    rv = torch.randn(16, N, device=dev)
    if kind == "plain":
        return [lambda w=w: ops.gemm(a, w, b, None) for w in ws]
    if kind == "res":
        return [lambda w=w: ops.gemm(a, w, b, r) for w in ws]
    if kind == "geglu":
        return [lambda w=w: ops.gemm(a, w, b, None, geglu=True) for w in ws]
    if kind == "rv":
        return [lambda w=w: ops.gemm_ex(a, w, b, rowvec=rv, rowvec_div=1, rowvec_mod=16, res=r, out_scale=0.5) for w in ws]
    # LayerNorm folded: prepare the folded operands once per weight (the engine does this at plan time)
    fns = []
    lib = ops._lib.load()
    for w in ws:
        wf = w.float()
        wsc = (wf * gamma[None]).to(torch.bfloat16).contiguous()
        c = wsc.float().sum(dim=1).contiguous()
        bb = ((wf.double() @ beta.double()).float() + b).contiguous()

        def fn(wsc=wsc, c=c, bb=bb):
            out = torch.empty(M, nout, dtype=torch.bfloat16, device=dev)
            ops._lib.check(lib.nr_op_gemm_ex(ops._stream(), ops._ptr(a), K, ops._ptr(wsc), ops._ptr(bb), ops._ptr(c), 1e-5, None, 1, 0, 0,
                                             None if "geglu" in kind else ops._ptr(r), nout, ops._ptr(out), nout, M, N, K, 1 if "geglu" in kind else 0, 0, 1.0))
            return out
        fns.append(fn)
    return fns


def graph_time(fns):
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for f in fns[:3]:
            f()
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(NCALL):
                fns[i % len(fns)]()
    torch.cuda.synchronize()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * NCALL) * 1e3


os.environ["NR_OP_FM_CACHE"] = "1"       # one fragment-major copy per weight tensor of the pool, packed outside the capture
for (M, N, K, kind) in SHAPES:
    npool = max(2, min(NCALL, int(600e6 / (N * K * 2))))
    fns = make(M, N, K, kind, npool)
    os.environ["NR_SMALLM"] = "0"
    ref = fns[0]().float()
    t0 = graph_time(fns)
    os.environ["NR_SMALLM"] = "2"
    for f in fns:
        f()
    out = fns[0]().float()
    t1 = graph_time(fns)
    err = (out - ref).abs().max().item()
    rel = ((out - ref).norm() / ref.norm()).item()
    print(f"M={M:5d} N={N:5d} K={K:5d} {kind:8s} igemm {t0:6.1f}us | smallm {t1:6.1f}us x{t0/t1:4.2f} max|d|={err:.3g} rel={rel:.1e}", flush=True)
    torch.cuda.synchronize()
    del fns
    ops._lib.load().nr_op_fm_cache_clear()
