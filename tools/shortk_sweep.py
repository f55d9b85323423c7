"""Round 5: plan sweep of the short-K Linears (K = 640 / 1280) as the engine replays them -- a chain of identical launches in ONE hipGraph, weights
L2 / Infinity-Cache hot -- over tile / ring-depth / wave-count candidates of the tiled igemm (NR_IGEMM_FORCE), with the asm LDS-DMA form enabled
from 4 k-tiles (NR_IGEMM_ADMA_MINK, read once per process: run the tool once per value).  us per launch, boundary included.
Usage (GPU box): NR_IGEMM_ADMA_MINK=4 python tools/shortk_sweep.py > gpurun_out/r05_shortk_sweep_adma4.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neurons_amd import ops  # noqa: E402

SHAPES = [(8192, 640, 640, 0), (8192, 1920, 640, 0), (8192, 5120, 640, 1), (8192, 640, 3200, 0), (2048, 1280, 1280, 0), (2048, 3840, 1280, 0),
          (2048, 10240, 1280, 1), (2048, 1280, 6400, 0), (512, 1280, 1280, 0), (512, 3840, 1280, 0)]
if os.environ.get("SWEEP_SHAPES"):          # "M,N,K,geglu;..."
    SHAPES = [tuple(int(v) for v in spec.split(",")) for spec in os.environ["SWEEP_SHAPES"].split(";")]
CANDS = ["", "128,160,1,2,-1,4", "128,160,1,3,-1,4", "128,160,1,4,-1,4", "128,128,1,2,-1,8", "128,128,1,3,-1,8", "128,128,1,4,-1,8", "128,128,1,2,-1,4",
         "128,128,1,4,-1,4", "128,64,1,2,-1,8", "128,64,1,3,-1,8", "128,64,1,4,-1,8", "128,64,1,4,-1,4", "64,64,1,2,-1,4", "64,64,1,4,-1,4", "64,64,1,6,-1,4",
         "64,32,1,4,-1,4", "64,32,1,8,-1,4", "256,128,1,2,-1,8", "256,160,1,2,-1,4"]
if os.environ.get("SWEEP_CANDS"):           # ";"-separated NR_IGEMM_FORCE strings (empty = heuristic)
    CANDS = os.environ["SWEEP_CANDS"].split(";")
CHAIN, REPS = 48, 20


POOL_MB = float(os.environ.get("SWEEP_POOL_MB", "0"))      # > 0: the chain walks a pool of distinct weight tensors of this size (HBM-cold weights,
                                                            # as in the denoiser, when the pool exceeds the 256 MB Infinity Cache)


def launch(a, w, kind):
    """kind 0 plain, 1 GEGLU, 2 LayerNorm folded, 3 LayerNorm folded + GEGLU (timing only: the folded operands are dummies)"""
    if kind < 2:
        return ops.gemm(a, w, geglu=bool(kind))
    M, K = a.shape
    N = w.shape[0]
    nout = N // 2 if kind == 3 else N
    if N not in _LN:
        _LN[N] = (torch.ones(N, device=a.device), torch.zeros(N, device=a.device))
    c, b = _LN[N]
    out = torch.empty(M, nout, dtype=torch.bfloat16, device=a.device)
    ops._lib.check(ops._lib.load().nr_op_ln_gemm(ops._stream(), ops._ptr(a), K, ops._ptr(w), ops._ptr(c), ops._ptr(b), 1e-5, None, nout, ops._ptr(out), nout,
                                                 M, N, K, 1 if kind == 3 else 0, 0))
    return out


_LN = {}


def chain_us(a, ws, geglu):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        keep = []
        with torch.cuda.graph(g):
            for i in range(CHAIN):
                keep.append(launch(a, ws[i % len(ws)], geglu))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / REPS / CHAIN


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ops.g8p_mode(0)          # the tiled igemm only: this sweep is about its plan space
    gen = torch.Generator(device=dev).manual_seed(0)
    print(f"weight pool {POOL_MB:.0f} MB; NR_IGEMM_ADMA_MINK={os.environ.get('NR_IGEMM_ADMA_MINK', '(default 24)')}; chain of {CHAIN} launches per graph, {REPS} replays; us per launch")
    for (M, N, K, geglu) in SHAPES:
        npool = max(1, min(CHAIN, int(POOL_MB * 1e6 / (N * K * 2))))
        w = [(torch.randn(N, K, generator=gen, device=dev) * 0.03).to(torch.bfloat16) for _ in range(npool)]
        a = torch.randn(M, K, generator=gen, device=dev).to(torch.bfloat16)
        res = []
        for c in CANDS:
            rowwave = c.endswith(",41")             # 128 x 160 as 4 x 1 waves: GEGLU / LayerNorm-capable
            if c.startswith("128,160") and (N % 160 or (geglu and not rowwave)):
                continue
            if c.startswith("256,160") and (N % 160 or geglu):
                continue
            if c.startswith("64,32") and geglu:
                continue
            if geglu >= 2 and not rowwave and (c.startswith("256") or c.startswith("128,160") or (c and c.split(",")[3] not in ("2", "4")) or (c and c.split(",")[2] != "1")):
                continue                                   # LayerNorm-folded instantiations: 2-stage tiles up to 128x128, 4-stage 64x64
            if c:
                os.environ["NR_IGEMM_FORCE"] = c
            else:
                os.environ.pop("NR_IGEMM_FORCE", None)
            try:
                launch(a, w[0], geglu)                     # a plan the launcher refuses must fail HERE, not inside the capture
                torch.cuda.synchronize()
                res.append((chain_us(a, w, geglu), c or "heuristic"))
            except Exception as e:          # a plan the launcher refuses
                res.append((float("inf"), (c or "heuristic") + " (" + str(e)[:40] + ")"))
        os.environ.pop("NR_IGEMM_FORCE", None)
        base = [r for r in res if r[1] == "heuristic"][0][0]
        res.sort()
        gf = 2.0 * M * N * K / 1e9
        print(f"M={M} N={N} K={K} geglu={geglu} ({gf:.1f} GFLOP): heuristic {base:.2f} us ({gf / base / 1e3:.0f} TF) | " +
              ", ".join(f"[{c}] {u:.2f}" for u, c in res[:6]), flush=True)


if __name__ == "__main__":
    main()
