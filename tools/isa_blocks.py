"""Per basic block of one kernel in a hipcc -save-temps .s file: instruction counts by kind (MFMA, LDS reads, LDS-DMA, accvgpr copies, waits).
Usage: isa_blocks.py file.s kernel-name-substring"""
import re
import sys

s = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
lines = s[start:end]
labels = [0] + [i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)]
for a, b in zip(labels, labels[1:] + [len(lines)]):
    seg = [l for l in lines[a:b] if l.startswith("\t") and not l.strip().startswith((";", "."))]
    cnt = lambda p: sum(1 for l in seg if re.search(p, l))
    valu = cnt(r'^\s*v_(?!mfma|accvgpr)')
    br = [l.strip() for l in seg if "s_cbranch" in l or "s_branch" in l]
    print(f"{lines[a][:14]:14s} insts={len(seg):5d} mfma={cnt('v_mfma'):4d} ds_read={cnt('ds_read'):4d} glds={cnt('global_load_lds'):3d} gload={cnt('global_load_dword'):3d} "
          f"acc_cp={cnt('v_accvgpr'):4d} wait={cnt('s_waitcnt'):3d} bar={cnt('s_barrier'):2d} dot2={cnt('v_dot2'):3d} valu={valu:4d} nop={cnt('s_nop'):3d} {br[:2]}")
