#!/usr/bin/env python
"""Dev tool: list compiler-generated `s_waitcnt vmcnt` (outside inline-asm blocks) per kernel, flagging the ones
inside loops - those drain the hand-counted LDS-DMA rings.  usage: vmcnt_audit.py file.s"""
import re, sys
kern, in_asm, loop = None, False, False
out = {}
for ln, line in enumerate(open(sys.argv[1]), 1):
    s = line.strip()
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kern = m.group(1); out[kern] = []
    if s.startswith(";;#ASMSTART"): in_asm = True
    elif s.startswith(";;#ASMEND"): in_asm = False
    elif s.startswith(".LBB") or s.startswith("; %bb."):
        loop = "in Loop" in s or "Loop Header" in s
    elif kern and not in_asm and "s_waitcnt" in s and "vmcnt" in s:
        out[kern].append((ln, s, loop))
for k, v in out.items():
    inl = [x for x in v if x[2]]
    print(f"{k[:70]}: {len(v)} compiler vmcnt waits, {len(inl)} in loops")
    for ln, s, _ in inl:
        print(f"    line {ln}: {s}")
