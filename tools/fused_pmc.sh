#!/bin/bash
# rocprofv3 PMC passes over the two fused level-0 kernels (ffpanel.hip, tattn.hip), driven by their standalone timing tools.
# Usage (GPU box): bash tools/fused_pmc.sh <tag>
R=$GRAFT_REPO_ROOT
TAG=${1:-fused}
cd /tmp && export TMPDIR=/tmp
for drv in ff_bench tattn_bench; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmc1_$drv -o p --output-format csv -- python3 $R/tools/$drv.py > $R/gpurun_out/${TAG}_pmc1_$drv.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/${TAG}_pmc2_$drv -o p --output-format csv -- python3 $R/tools/$drv.py > $R/gpurun_out/${TAG}_pmc2_$drv.log 2>&1
done
cd $R
python3 - "$TAG" <<'PY' > gpurun_out/${TAG}_pmc_summary.txt
import csv, glob, collections, sys
tag = sys.argv[1]
for d in sorted(glob.glob(f"gpurun_out/{tag}_pmc[12]_*")):
    if not d.endswith(("ff_bench", "tattn_bench")): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in rows:
            k = r["Kernel_Name"]
            if not any(s in k for s in ("ff_fused", "tattn_fused", "rowpanel", "igemm", "attn_fwd")): continue
            key = (k[:70], r["Grid_Size"])
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(key, r["Counter_Name"])] += 1
        for key, v in agg.items():
            print(d.split("/")[-1], key, {a: round(b / max(1, n[(key, a)])) for a, b in v.items()})
PY
find gpurun_out -path "*${TAG}_pmc*" -name "*.csv" -size +1M -delete
cat gpurun_out/${TAG}_pmc_summary.txt
