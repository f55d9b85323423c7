"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of
`bench.py --steps 1 --warmup 0 --ddim-steps N --no-cpu-baseline` into profiles/rNN_*_traffic_pmc.json.
Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <forwards> <out.json>
  forwards = network evaluations in the profiled run (ddim steps + 1 per-op profiling pass)."""
import csv
import glob
import json
import os
import sys


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    tot, ig, n_ig = 0.0, 0.0, 0
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            v = float(r["Counter_Value"])
            tot += v
            if "igemm_bf16_kernel" in r["Kernel_Name"] or "splitk_reduce" in r["Kernel_Name"]:
                ig += v
                n_ig += 1
    return tot, ig, n_ig


fetch_dir, write_dir, forwards, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
f_tot, f_ig, n_ig = load(fetch_dir, "FETCH_SIZE")
w_tot, w_ig, _ = load(write_dir, "WRITE_SIZE")
# units: KB (x1024); FETCH_SIZE is doubled on gfx950 (it reports half of wide coalesced reads) per MI355X_MICROARCH.md
res = {
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 1 --warmup 0 --ddim-steps N "
            f"--no-cpu-baseline: {forwards:g} forwards of each network; FETCH_SIZE doubled per MI355X_MICROARCH.md; units KB*1024",
    "igemm_launches_per_ddim_step": n_ig / forwards,
    "igemm_hbm_bytes_per_ddim_step": (2.0 * f_ig + w_ig) * 1024.0 / forwards,
    "igemm_fetch_kb_raw": f_ig / forwards,
    "igemm_write_kb": w_ig / forwards,
    "whole_step_hbm_bytes": (2.0 * f_tot + w_tot) * 1024.0 / forwards,
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
