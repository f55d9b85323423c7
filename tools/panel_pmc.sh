# two SQ PMC passes over the register-panel kernel (standalone driver tools/panel_pmc.py): MFMA busy, LDS bank conflicts, wave / wait cycles, LDS instruction counts
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/panel_pmc1 -o p --output-format csv -- python3 $R/tools/panel_pmc.py > $R/gpurun_out/panel_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/panel_pmc2 -o p --output-format csv -- python3 $R/tools/panel_pmc.py > $R/gpurun_out/panel_pmc2.log 2>&1
cd $R
python3 - <<'PY' | tee gpurun_out/panel_pmc.txt
import csv, glob, collections
for d in ("gpurun_out/panel_pmc1", "gpurun_out/panel_pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in rows:
            k = r["Kernel_Name"]
            if "lin128q_kernel" not in k: continue
            key = (k[k.index("lin128q_kernel"):][:40], r["Grid_Size"])
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        for key, v in sorted(agg.items()):
            print(d.split("/")[-1], key, {a: round(b / 4) for a, b in sorted(v.items())})
PY
