#!/bin/bash
# GroupNorm A/B: slab kernel vs stats+apply passes; kernel durations from rocprofv3 (python launch overhead exceeds the kernels)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gn_slab -o gn -- python3 $R/tools/gn_bench.py > $R/gpurun_out/gn_slab.log 2>&1
export NR_GN_SLAB=0
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gn_old -o gn -- python3 $R/tools/gn_bench.py > $R/gpurun_out/gn_old.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("gn_slab", "gn_old"):
    f = glob.glob(f"gpurun_out/{d}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "gn_" in r["Kernel_Name"]]
    # group consecutive launches by (kernel, grid)
    agg = collections.OrderedDict()
    for r in rows:
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"], r["Workgroup_Size"])
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(d)
    for k, a in agg.items():
        print(f"  {k[0]:42s} grid={k[1]:>8s} n={a[0]:3d} avg={a[1]/a[0]:7.2f} us")
PY
