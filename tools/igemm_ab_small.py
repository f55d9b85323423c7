"""In-situ A/B (whole DDIM step under hipGraph, grouped SparseCtrl schedule) of igemm plans for the latency-bound mid-size Linears:
6.7-27 GFLOP each, 10-20 k-tiles per tile.  Same mechanism as tools/igemm_ab_shapes.py.
Usage (GPU box): python tools/igemm_ab_small.py > gpurun_out/igemm_ab_small.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(2048, 1280, 1280), (8192, 640, 640), (8192, 1920, 640), (2048, 3840, 1280), (8192, 640, 3200), (2048, 1280, 6400), (32768, 320, 1600),
          (8192, 1280, 1280), (8192, 3840, 1280)]
CANDS = ["64,64,1,4,-1,4", "64,64,1,3,-1,4", "64,64,1,2,-1,4", "128,64,1,3,-1,8", "128,64,1,3,-1,4", "128,64,1,4,-1,4", "128,64,1,2,-1,8", "128,64,1,2,-1,4",
         "128,128,1,2,-1,8", "128,128,1,3,-1,8", "128,160,1,2,-1,4", "128,160,1,3,-1,4", "128,64,2,2,-1,4", "128,128,2,2,-1,8"]


def run(env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-psnr", "--steps", "1", "--warmup", "1"],
                         env=e, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    return json.loads(out)["config"]["ms_per_ddim_step"]


base = [run({}) for _ in range(3)]
print("base ms/step", base, flush=True)
b = sorted(base)[1]
for (M, N, K) in SHAPES:
    res = []
    for c in CANDS:
        if c.startswith("128,160") and N % 160 != 0:
            continue
        ms = run({"NR_IGEMM_FORCE": c, "NR_IGEMM_FORCE_MAXM": str(M), "NR_IGEMM_FORCE_MINM": str(M), "NR_IGEMM_FORCE_N": str(N),
                  "NR_IGEMM_FORCE_K": str(K), "NR_IGEMM_FORCE_KS": "1"})
        res.append((ms - b, c))
    res.sort()
    print(f"M={M} N={N} K={K}: " + ", ".join(f"[{c}] {d:+.3f}" for d, c in res[:5]) + f"  | worst [{res[-1][1]}] {res[-1][0]:+.3f}", flush=True)
print("base again", run({}), flush=True)
