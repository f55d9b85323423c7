"""In-situ per-shape A/B of igemm plans: for each (M, N, K, ks) class of BASELINE config 2, force candidate plans on exactly
that class (NR_IGEMM_FORCE + _MAXM/_MINM/_N/_K/_KS filters) and time the whole DDIM step under hipGraph.
Usage (GPU box): python tools/igemm_ab_shapes.py > gpurun_out/igemm_ab_shapes.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(32768, 960, 320, 1), (32768, 2560, 320, 1), (32768, 320, 320, 1), (32768, 320, 1280, 1),
          (8192, 1920, 640, 1), (8192, 5120, 640, 1), (8192, 640, 640, 1), (8192, 640, 2560, 1),
          (2048, 3840, 1280, 1), (2048, 10240, 1280, 1), (2048, 1280, 1280, 1), (2048, 1280, 5120, 1),
          (32768, 320, 2880, 3), (32768, 320, 5760, 3), (8192, 640, 5760, 3), (8192, 640, 11520, 3)]
CANDS = ["128,128,1,2,-1,8", "128,128,1,2,-1,4", "128,64,1,2,-1,4", "128,64,1,2,-1,8", "128,160,1,2,-1,4", "64,64,1,2,-1,4", "128,64,2,2,-1,4",
         "128,128,2,2,-1,8", "128,160,2,2,-1,4"]


def run(env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-psnr", "--steps", "1", "--warmup", "1"],
                         env=e, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    return json.loads(out)["config"]["ms_per_ddim_step"]


base = [run({}) for _ in range(2)]
print("base ms/step", base, flush=True)
b = sum(base) / len(base)
for (M, N, K, ks) in SHAPES:
    res = []
    for c in CANDS:
        if c.startswith("128,160") and N % 160 != 0:
            continue
        if "geglu" and N in (2560, 5120, 10240) and c.startswith("128,160"):
            continue
        ms = run({"NR_IGEMM_FORCE": c, "NR_IGEMM_FORCE_MAXM": str(M), "NR_IGEMM_FORCE_MINM": str(M), "NR_IGEMM_FORCE_N": str(N),
                  "NR_IGEMM_FORCE_K": str(K), "NR_IGEMM_FORCE_KS": str(ks)})
        res.append((ms - b, c))
    res.sort()
    print(f"M={M} N={N} K={K} ks={ks}: " + ", ".join(f"[{c}] {d:+.3f}" for d, c in res[:4]) + f"  | worst {res[-1][0]:+.3f}", flush=True)
print("base again", run({}), flush=True)
