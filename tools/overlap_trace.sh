#!/bin/bash
# How much of SparseCtrl really hides behind the U-Net (reconcile "0.7 ms hides" with "2.1 ms of concurrency-stretched durations").
# One rocprofv3 kernel trace of a 10-step clip (hipGraph replay, two engine streams) -> per queue: sum of kernel durations, wall time covered,
# time where BOTH queues have a kernel in flight; then the same clip with --no-controlnet for the wall-time cost of SparseCtrl.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/ovl_trace -o g --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-psnr --no-op-profile --no-end-to-end > $R/gpurun_out/ovl_trace.log 2>&1
cd $R
python3 - <<'PY' > gpurun_out/overlap_trace.txt
import csv, glob, collections
f = glob.glob("gpurun_out/ovl_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "anonymous namespace" in r["Kernel_Name"] and "_pack_kernel" not in r["Kernel_Name"] and "fold_linear" not in r["Kernel_Name"]]
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
by = collections.defaultdict(list)
for r in rows:
    by[r[key]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
qs = sorted(by, key=lambda q: -len(by[q]))[:2]
# the timed clip = the last 10 DDIM steps: take the window that starts at the last cfg_ddim_step minus 10 steps
steps = sorted(s for q in by for s, e, n in by[q] if "cfg_ddim_step" in n)
t0, t1 = steps[-11], steps[-1]
def clip(ks): return sorted((max(s, t0), min(e, t1)) for s, e, _ in ks if e > t0 and s < t1)
def union(iv):
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
def inter(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s: tot += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
iv = {q: clip(by[q]) for q in qs}
un = {q: union(iv[q]) for q in qs}
n = 10.0
print(f"window: 10 DDIM steps = {(t1 - t0) / 1e6:.2f} ms -> {(t1 - t0) / 1e6 / n:.3f} ms per step")
for q in qs:
    names = collections.Counter(nm.split("::")[-1][:24] for s, e, nm in by[q] if e > t0 and s < t1).most_common(2)
    print(f"queue {q}: {len(iv[q]) / n:.0f} kernels per step, sum of durations {sum(e - s for s, e in iv[q]) / 1e6 / n:.3f} ms per step, wall time covered {sum(e - s for s, e in un[q]) / 1e6 / n:.3f} ms per step  {names}")
both = inter(un[qs[0]], un[qs[1]])
either = sum(e - s for s, e in union(sorted(un[qs[0]] + un[qs[1]])))
print(f"both queues busy {both / 1e6 / n:.3f} ms per step; at least one busy {either / 1e6 / n:.3f} ms per step; idle (launch boundaries) {((t1 - t0) - either) / 1e6 / n:.3f} ms per step")
PY
find gpurun_out/ovl_trace -name "*.csv" -size +1M -delete
for a in "" "--no-controlnet"; do python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr --no-end-to-end $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench $a:', d['config']['ms_per_ddim_step'], 'ms per DDIM step')" >> gpurun_out/overlap_trace.txt; done
cat gpurun_out/overlap_trace.txt
