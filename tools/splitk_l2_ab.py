"""In-launch split-K reduction (gemm.hip `l2red`, NR_SPLITK_L2=1: all K-slices of an output tile on ONE XCD, fp32 slabs meet in that XCD's L2, the last
arriver sums them in slice order and runs the epilogue) against the two-pass form (slabs + splitk_reduce_kernel), VERDICT r5 next #6.

One ARM per process (the switch is read once): `python tools/splitk_l2_ab.py arm 0|1 out.pt` runs, for every split-K shape of the headline and the
keyframe path,
  * timing: a chain of 40 launches on a pool of HBM-cold weights, replayed 20 x, us per launch (events on the launch stream);
  * bits: the output of a seeded problem, saved so the parent can compare the arms bit for bit;
  * staleness (arm 1 matters): 100 repetitions of TWO alternating shapes that re-use the SAME slab workspace while a second stream streams
    1 GB copies (its own L2 traffic on every XCD): every repetition must reproduce the first result bit for bit — a stale slab line read by the
    last arriver, or a lost counter update, shows up as a difference.
`python tools/splitk_l2_ab.py` (no arguments) runs both arms as child processes and prints the table (profiles/r06_splitk_xcd_ab.txt)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (kind, M or (nimg, H, W), N, K or Cin): the split-K launches of BASELINE config 2 (4x4 / 8x8 levels) and of the keyframe model (16x16 level = 512 rows)
SHAPES = [("conv", (32, 4, 4), 1280, 1280), ("conv", (32, 4, 4), 1280, 2560), ("conv", (32, 8, 8), 1280, 1280), ("conv", (32, 8, 8), 1280, 2560),
          ("conv", (32, 8, 8), 1280, 1920), ("lin", 2048, 1280, 6400), ("lin", 2048, 1280, 2560), ("lin", 1024, 1280, 5120), ("conv", (2, 16, 16), 1280, 1280)]


def make(kind, m, N, K, seed, dev, torch):
    g = torch.Generator(device=dev).manual_seed(seed)
    if kind == "conv":
        nimg, H, W = m
        x = torch.randn(nimg, H, W, K, generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, 3, 3, K, generator=g, device=dev) * (9 * K) ** -0.5).to(torch.bfloat16)
        res = torch.randn(nimg, H, W, N, generator=g, device=dev).to(torch.bfloat16)
    else:
        x = torch.randn(m, K, generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
        res = torch.randn(m, N, generator=g, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev)
    return x, w, bias, res


def run(kind, x, w, bias, res, ops):
    return ops.conv3x3(x, w, bias=bias, res=res) if kind == "conv" else ops.gemm(x, w, bias=bias, res=res)


def arm(which, out_path):
    os.environ["NR_SPLITK_L2"] = which
    os.environ.setdefault("NR_LIB_VARIANT", "exp")          # the in-launch form lives in the experiments library only (make -C neurons_amd/csrc experiments)
    import torch
    from neurons_amd import ops
    dev = torch.device("cuda", 0)
    result = {"arm": which, "rows": []}
    for kind, m, N, K in SHAPES:
        x, w, bias, res = make(kind, m, N, K, 7, dev, torch)
        ref = run(kind, x, w, bias, res, ops).clone()
        # timing on a pool of weights (HBM-cold: 40 different weight tensors per chain)
        pool = [w.clone() for _ in range(40)]
        s = torch.cuda.current_stream()
        for wi in pool[:4]:
            run(kind, x, wi, bias, res, ops)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(5):
            e0.record(s)
            for _ in range(4):
                for wi in pool:
                    run(kind, x, wi, bias, res, ops)
            e1.record(s)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000.0 / 160)
        result["rows"].append({"shape": f"{kind} {m} N={N} K={'9x' if kind == 'conv' else ''}{K}", "us": best, "out": ref.cpu()})
        del pool
    # staleness: two shapes alternate on the same workspace, a second stream hammers the memory system
    side = torch.cuda.Stream()
    big_a, big_b = torch.empty(256 << 20, dtype=torch.float32, device=dev), torch.empty(256 << 20, dtype=torch.float32, device=dev)
    probs = [(k, make(k, m, N, K, 11 + i, dev, torch)) for i, (k, m, N, K) in enumerate(SHAPES[:2] + SHAPES[5:7])]
    firsts = [run(k, *p, ops).clone() for k, p in probs]
    bad = 0
    for rep in range(100):
        with torch.cuda.stream(side):
            big_b.copy_(big_a)
        for (k, p), f in zip(probs, firsts):
            if not torch.equal(run(k, *p, ops), f):
                bad += 1
    torch.cuda.synchronize()
    result["stale_mismatches"] = bad
    torch.save(result, out_path)


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "arm":
        return arm(sys.argv[2], sys.argv[3])
    import torch
    outs = []
    for a in ("0", "1"):
        path = f"/tmp/splitk_arm{a}.pt"
        subprocess.run([sys.executable, os.path.abspath(__file__), "arm", a, path], check=True)
        outs.append(torch.load(path))
    print("in-launch split-K reduction (arm 1: NR_SPLITK_L2=1) vs slabs + splitk_reduce_kernel (arm 0); us per launch, HBM-cold weights, eager chain of 160 launches")
    for r0, r1 in zip(outs[0]["rows"], outs[1]["rows"]):
        same = torch.equal(r0["out"], r1["out"])
        print(f"  {r0['shape']:38s} two-pass {r0['us']:7.2f} us   in-launch {r1['us']:7.2f} us   x{r0['us'] / r1['us']:.2f}   bit-identical: {same}")
    print(f"  staleness test (4 shapes alternating on one workspace, 100 repetitions, concurrent 1-GB copies on a second stream): "
          f"two-pass mismatches {outs[0]['stale_mismatches']}, in-launch mismatches {outs[1]['stale_mismatches']}")


if __name__ == "__main__":
    main()
