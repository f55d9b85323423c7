"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass over bench.py into per-kernel MFMA utilisation:
util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCD-summed samples)  -- same formula as the v4 note in
profiles/README.md (GRBM_GUI_ACTIVE is reported summed over the 8 XCDs).   Usage: python tools/pmc_mfma.py <dir> <out.json>"""
import csv
import glob
import json
import os
import sys
import collections

d, out = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
busy = collections.defaultdict(float)
act = collections.defaultdict(float)
calls = collections.defaultdict(int)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[name] += float(r["Counter_Value"]); calls[name] += 1
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[name] += float(r["Counter_Value"])
res = {}
tb = ta = 0.0
for k in busy:
    if act[k] <= 0:
        continue
    util = busy[k] / (1024.0 * act[k] / 8.0)
    if "igemm" in k or "g8p_kernel" in k or "rowpanel" in k or "ff_fused" in k or "tattn_fused" in k or "tattn_head" in k or "xattn_fused" in k or "xattn_head" in k or "lin160_kernel" in k or "lin128q_kernel" in k or "smallm_kernel" in k:
        tb += busy[k]; ta += act[k]
    res[k] = {"launches": calls[k], "mfma_util": round(util, 4)}
res["_igemm_class"] = {"mfma_util": round(tb / (1024.0 * ta / 8.0), 4) if ta else None}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get("launches") or 0))[:14]:
    print(k, v)
print("igemm class", res["_igemm_class"])
