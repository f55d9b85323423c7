"""Static check (no GPU): for every kernel of the library, how many separate scalar-load round trips (s_load ... s_waitcnt lgkmcnt) sit between the
kernel entry and its first vector-memory instruction?  Lazy kernel-argument loads serialise the prologue of short kernels (common.h nr_pin).
Usage: python tools/kernarg_report.py [objdir]   (default neurons_amd/csrc/build)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
objdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "neurons_amd", "csrc", "build")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
for obj in sorted(glob.glob(os.path.join(objdir, "*.o"))):
    subprocess.run([OBJDUMP, "--offloading", obj], capture_output=True, cwd=objdir)
    dev = glob.glob(obj + ".0.hipv4-*")
    if not dev:
        continue
    asm = subprocess.run([OBJDUMP, "-d", dev[0]], capture_output=True, text=True).stdout
    for f in glob.glob(obj + ".0.*"):
        os.remove(f)
    name, rows = None, []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            state = dict(trips=0, pending=False, insts=0, done=False)
            rows.append((name, state))
            continue
        if name is None or not rows or rows[-1][1]["done"]:
            continue
        st = rows[-1][1]
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        st["insts"] += 1
        if ins.startswith("s_load"):
            st["pending"] = True
        elif ins.startswith("s_waitcnt") and "lgkmcnt" in ins and st["pending"]:
            st["trips"] += 1
            st["pending"] = False
        elif re.match(r"(global_load|buffer_load|global_store|buffer_store|global_atomic|flat_)", ins):
            st["done"] = True
    print(os.path.basename(obj))
    for n, st in rows:
        if st["trips"] >= 1:
            print(f"  {st['trips']:2d} scalar round trips, {st['insts']:5d} instructions before the first vector-memory op  {n[:150]}")
