"""Scan gfx950 ISA (hipcc -S output or llvm-objdump -d) for the accumulator pattern hipcc 7.2 miscompiles: an MFMA whose destination
PARTIALLY overlaps its SrcC (neither identical nor disjoint) while that SrcC was itself written by an MFMA a few instructions earlier, e.g.

    v_mfma_f32_16x16x32_bf16 v[0:3], v[86:89], v[18:21], v[2:5]
    v_mfma_f32_16x16x32_bf16 v[2:5], v[90:93], v[22:25], v[0:3]

The register allocator forms such chains under -amdgpu-mfma-vgpr-form when it un-ties an accumulator; no wait states are inserted between
the dependent MFMAs and on MI355X the chain returned wrong sums (attention.hip, ONES instantiation, round 4: 0.44 relative error; the same
source with the accumulator tied through inline asm is exact).  Partial overlap with a SrcC that ordinary VALU code wrote (a zero-initialised
accumulator) is fine and is not reported.
Usage: python tools/check_mfma_overlap.py file.s [...]   (exit status 1 when a chain is found)"""
import re
import sys

_MFMA = re.compile(r"^\s*(v_mfma_\w+)\s+([va])\[(\d+):(\d+)\],\s*[^,]+,\s*[^,]+,\s*(?:([va])\[(\d+):(\d+)\]|\S+)")
_DST = re.compile(r"^\s*(?:v_|ds_read|ds_bpermute|ds_permute|global_load|buffer_load|scratch_load|flat_load)\w*\s+v(?:\[(\d+):(\d+)\]|(\d+))\b")
WINDOW = 24      # instructions: far more than the wait states any MFMA -> MFMA SrcC hazard needs


def scan(asm_text):
    """Returns a list of (line_number, text) of MFMAs that close a partially overlapping accumulator chain."""
    found = []
    last_mfma = {}          # (bank, register) -> instruction index of the MFMA that wrote it last
    idx = 0
    for ln, line in enumerate(asm_text.splitlines(), 1):
        body = line.split("//")[0].split(";")[0]
        if not body.strip() or body.lstrip().startswith((".", "#")) or body.rstrip().endswith(":"):
            continue
        idx += 1
        m = _MFMA.match(body)
        if m:
            bank, d0, d1 = m.group(2), int(m.group(3)), int(m.group(4))
            if m.group(5) is not None:
                cb, c0, c1 = m.group(5), int(m.group(6)), int(m.group(7))
                overlap = cb == bank and not (d1 < c0 or c1 < d0) and (d0, d1) != (c0, c1)
                if overlap and any(idx - last_mfma.get((cb, r), -10 ** 9) <= WINDOW for r in range(c0, c1 + 1)):
                    found.append((ln, body.strip()))
            for r in range(d0, d1 + 1):
                last_mfma[(bank, r)] = idx
            continue
        d = _DST.match(body)
        if d:
            lo, hi = (int(d.group(1)), int(d.group(2))) if d.group(1) is not None else (int(d.group(3)), int(d.group(3)))
            for r in range(lo, hi + 1):
                last_mfma.pop(("v", r), None)
    return found


if __name__ == "__main__":
    bad = 0
    for path in sys.argv[1:]:
        hits = scan(open(path).read())
        for ln, text in hits:
            print(f"{path}:{ln}: {text}")
        bad += len(hits)
    print(f"{bad} partially overlapping MFMA accumulator chain(s)")
    sys.exit(1 if bad else 0)
