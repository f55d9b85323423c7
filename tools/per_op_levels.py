"""Per resolution level of a per-op CSV (tools/per_op_profile.py): launches, ms, TFLOP/s, keyed by the row count of the level
(BASELINE config 2: 512 / 2048 / 8192 / 32768 rows).  Usage: per_op_levels.py a.csv [rows_per_level ...]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
levels = [int(x) for x in sys.argv[2:]] or [512, 2048, 8192, 32768]
hw_of = {m: m // 32 for m in levels}          # rows = 32 frame-images x hw
acc = {m: [0, 0.0, 0.0] for m in levels}
other = [0, 0.0, 0.0]
for r in rows:
    d = r["desc"]
    m = re.search(r"\bM=(\d+)", d)
    M = int(m.group(1)) if m else None
    if M is None:
        g = re.search(r"groupnorm nimg=(\d+) hw=(\d+)", d)
        a = re.search(r"attention mode=(\d) nbatch=(\d+) heads=\d+ d=\d+ Lq=(\d+)", d)
        if g:
            M = int(g.group(1)) * int(g.group(2))
        elif a:
            M = int(a.group(2)) * int(a.group(3))
        else:
            b = re.search(r"rows=(\d+)", d)
            M = 2 * int(b.group(1)) if b else None
    tgt = acc.get(M)
    if tgt is None and M is not None:
        tgt = acc.get(2 * M)          # CFG de-duplicated launches run on half a level's rows
    if tgt is None:
        tgt = other
    tgt[0] += 1; tgt[1] += float(r["ms"]); tgt[2] += float(r["gflop"])
tot = sum(float(r["ms"]) for r in rows)
for m in levels:
    n, ms, gf = acc[m]
    print(f"{m:6d} rows: {n:4d} launches {ms:7.3f} ms {gf / ms if ms else 0:7.1f} TFLOP/s")
print(f" other     : {other[0]:4d} launches {other[1]:7.3f} ms;  total {len(rows)} launches {tot:.3f} ms {sum(float(r['gflop']) for r in rows) / tot:.1f} TFLOP/s")
