#!/bin/bash
# Kernel-to-kernel gaps inside the replayed hipGraphs of one DDIM step (rocprofv3 --kernel-trace timestamps), per stream.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/gap_trace -o g --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-psnr --no-op-profile > $R/gpurun_out/gap_trace.log 2>&1
cd $R
python3 - <<'PY' > gpurun_out/gap_trace_summary.txt
import csv, glob, collections
f = glob.glob("gpurun_out/gap_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
rows = [r for r in rows if "anonymous namespace" in r["Kernel_Name"]]
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
by = collections.defaultdict(list)
for r in rows:
    by[r[key]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in by.items():
    ks.sort()
    # second half of the run = the timed clip (graphs replayed)
    ks = ks[len(ks) // 2:]
    busy = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g < 50000]
    span = ks[-1][1] - ks[0][0]
    print(f"{key} {q}: {len(ks)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, gaps<50us: n={len(small)} sum {sum(small)/1e6:.2f} ms mean {sum(small)/max(1,len(small))/1e3:.2f} us; "
          f"overlapped(neg) {sum(1 for g in gaps if g < 0)}; big gaps {sum(1 for g in gaps if g >= 50000)} sum {sum(g for g in gaps if g >= 50000)/1e6:.2f} ms")
    hist = collections.Counter(min(20, g // 500) for g in small)
    print("   gap histogram (0.5 us bins):", dict(sorted(hist.items())))
PY
find gpurun_out/gap_trace -name "*.csv" -size +1M -delete
cat gpurun_out/gap_trace_summary.txt
