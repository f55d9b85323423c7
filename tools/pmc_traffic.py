"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of
`bench.py --steps 1 --warmup 0 --ddim-steps N --no-cpu-baseline` into profiles/rNN_*_traffic_pmc.json.
Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <forwards> <out.json>
  forwards = network evaluations in the profiled run (ddim steps + 1 per-op profiling pass)."""
import csv
import glob
import json
import os
import re
import sys

PER = {}


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    tot, ig, n_ig = 0.0, 0.0, 0
    per = PER.setdefault(counter, {})
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            v = float(r["Counter_Value"])
            name = r["Kernel_Name"]
            # the denoising path = this library's kernels; weight conversion (fold_linear_pair_kernel) and torch's setup kernels
            # (random weights, copies) run before the timed region and are not part of a DDIM step
            if "anonymous namespace" not in name or "fold_linear_pair" in name or "at::" in name or "_pack_kernel" in name:
                continue                      # (plan-time weight packing: fragment-major copies of smallm.hip, the xattn / tattn weight streams)
            tot += v
            m = re.search(r"::(\w+)(<[^>]*>)?", name)
            key = (m.group(1) + (m.group(2) or "")) if m else name[:60]
            e = per.setdefault(key, [0.0, 0])
            e[0] += v
            e[1] += 1
            if ("igemm_bf16_kernel" in name or "splitk_reduce" in name or "rowpanel_kernel" in name or "ff_fused_kernel" in name
                    or "tattn_fused_kernel" in name or "tattn_head_kernel" in name or "xattn_head_kernel" in name or "lin160_kernel" in name or "lin128q_kernel" in name or "g8p_kernel" in name or "smallm_kernel" in name or "xattn_fused_kernel" in name):
                ig += v
                n_ig += 1
    return tot, ig, n_ig


fetch_dir, write_dir, forwards, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
f_tot, f_ig, n_ig = load(fetch_dir, "FETCH_SIZE")
w_tot, w_ig, _ = load(write_dir, "WRITE_SIZE")
# units: KB (x1024); FETCH_SIZE is doubled on gfx950 (it reports half of wide coalesced reads) per MI355X_MICROARCH.md
res = {
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 1 --warmup 0 --ddim-steps N "
            f"--no-cpu-baseline: {forwards:g} forwards of each network; FETCH_SIZE doubled per MI355X_MICROARCH.md; units KB*1024; "
            "library kernels only (setup-time weight conversion and torch kernels excluded); igemm class = igemm + ping-pong (g8p, round 4) + rowpanel + split-K reduce + the fused FeedForward / temporal- / cross-attention block kernels + the panel-resident small-M kernel (round 5) + the temporal attention head kernel (round 6)",
    "igemm_launches_per_ddim_step": n_ig / forwards,
    "igemm_hbm_bytes_per_ddim_step": (2.0 * f_ig + w_ig) * 1024.0 / forwards,
    "igemm_fetch_kb_raw": f_ig / forwards,
    "igemm_write_kb": w_ig / forwards,
    "whole_step_hbm_bytes": (2.0 * f_tot + w_tot) * 1024.0 / forwards,
}
# per kernel: GB per step (fetch doubled + write), launches per step
keys = set(PER.get("FETCH_SIZE", {})) | set(PER.get("WRITE_SIZE", {}))
rows = []
for k in keys:
    f = PER.get("FETCH_SIZE", {}).get(k, [0.0, 0])
    w = PER.get("WRITE_SIZE", {}).get(k, [0.0, 0])
    rows.append((round((2.0 * f[0] + w[0]) * 1024.0 / forwards / 1e9, 3), round(2.0 * f[0] * 1024.0 / forwards / 1e9, 3),
                 round(w[0] * 1024.0 / forwards / 1e9, 3), round(max(f[1], w[1]) / forwards, 1), k))
rows.sort(reverse=True)
res["per_kernel_gb_per_step"] = [{"kernel": k, "total_gb": t, "fetch_gb": f, "write_gb": w, "launches": n} for t, f, w, n, k in rows[:24]]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
