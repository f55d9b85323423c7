"""Per-shape GroupNorm kernel time from the rocprofv3 traces written by tools/gn_ab.sh."""
import csv, glob, re
SH = [(32, 1024, 320, 0), (32, 1024, 640, 320), (32, 1024, 320, 320), (32, 256, 640, 0), (32, 256, 320, 0), (32, 256, 1280, 640),
      (32, 256, 1280, 0), (32, 256, 640, 640), (32, 64, 1280, 0), (32, 64, 1280, 1280), (32, 64, 640, 0), (32, 16, 1280, 0), (32, 16, 1280, 1280)]
res = {}
for d in ("gn_slab", "gn_slab512", "gn_old"):
    fs = glob.glob(f"gpurun_out/{d}/**/*kernel_trace.csv", recursive=True)
    if not fs:
        continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "gn_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    calls = []
    for r in rows:
        n = re.search(r"(gn_\w+)", r["Kernel_Name"]).group(1)
        t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if n in ("gn_slab_kernel", "gn_stats_kernel", "gn_fused_small_kernel"):
            calls.append([n, t, r["Workgroup_Size_X"]])
        else:
            calls[-1][1] += t
    for si, sh in enumerate(SH):
        c = calls[si * 55 + 5:(si + 1) * 55]
        res.setdefault(sh, {})[d] = (sum(x[1] for x in c) / len(c), c[0][0], c[0][2])
for sh, v in res.items():
    mb = sh[0] * sh[1] * (sh[2] + sh[3]) * 4 / 1e6
    print(sh, "  ".join(f"{k}: {x[0]:5.1f} us ({x[1][3:8]},{x[2]}) {mb / x[0]:4.2f} TB/s" for k, x in v.items()))
