"""Bucket a per-op CSV (tools/per_op_profile.py) by kernel kind and row count: launches, total ms, TFLOP/s.  Usage: per_op_buckets.py a.csv [top]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(float(r["ms"]) for r in rows)
print(f"{sys.argv[1]}: {len(rows)} ops, {tot:.3f} ms, {sum(float(r['gflop']) for r in rows) / tot:.1f} TFLOP/s overall")
b = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    d = r["desc"]
    m = re.search(r"M=(\d+)", d)
    M = int(m.group(1)) if m else 0
    kind = d.split(" ")[0] if d else "other"
    if kind == "igemm":
        kind = "igemm" + re.search(r"ks=(\d)", d).group(1) + ("-geglu" if "geglu=1" in d else "")
        nk = re.search(r"N=(\d+) K=(\d+)", d)
        kind += f" N={nk.group(1)} K={nk.group(2)}"
    if kind == "attention":
        kind += " " + re.search(r"mode=(\d)", d).group(0) + " " + re.search(r"Lq=(\d+)", d).group(0)
    if kind == "groupnorm":
        M = int(re.search(r"hw=(\d+)", d).group(1))
    b[(kind, M)][0] += 1
    b[(kind, M)][1] += float(r["ms"])
    b[(kind, M)][2] += float(r["gflop"])
for k, v in sorted(b.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0]:40s} M/hw={k[1]:8d} n={v[0]:3d} ms={v[1]:8.3f} ({100 * v[1] / tot:4.1f} %) TF={v[2] / v[1] if v[1] else 0:7.1f}")
