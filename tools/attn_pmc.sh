#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $R/gpurun_out/attn_pmc1 -o a --output-format csv -- python3 $R/tools/attn_pmc.py > $R/gpurun_out/attn_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES -d $R/gpurun_out/attn_pmc2 -o a --output-format csv -- python3 $R/tools/attn_pmc.py > $R/gpurun_out/attn_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE -d $R/gpurun_out/attn_pmc3 -o a --output-format csv -- python3 $R/tools/attn_pmc.py > $R/gpurun_out/attn_pmc3.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/attn_pmc1", "gpurun_out/attn_pmc2", "gpurun_out/attn_pmc3"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "attn_fwd_shared" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        print({k: round(v / n[k]) for k, v in agg.items()})
PY
