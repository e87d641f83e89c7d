#!/usr/bin/env python3
"""Counts the instructions of a kernel's ISA listing (hipcc -S) by issue port, for a line range or the whole text:
tools/isa_count.py file.s [first_line last_line].  s_nop / s_waitcnt / branches are listed apart (they take scalar issue
slots but no ALU work)."""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
if len(sys.argv) > 3:
    lines = lines[int(sys.argv[2]) - 1:int(sys.argv[3])]
c = collections.Counter()
detail = collections.Counter()
for l in lines:
    l = l.strip()
    if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
        continue
    op = l.split()[0]
    if op.startswith("v_"):
        k = "VALU"
        if "readlane" in op or "readfirstlane" in op or "writelane" in op: k = "VALU(lane<->sgpr)"
    elif op.startswith("s_nop"): k = "s_nop"
    elif op.startswith("s_waitcnt"): k = "s_waitcnt"
    elif op.startswith(("s_cbranch", "s_branch")): k = "branch"
    elif op.startswith("s_barrier"): k = "barrier"
    elif op.startswith(("s_load", "s_buffer")): k = "SMEM"
    elif op.startswith("s_"): k = "SALU"
    elif op.startswith("ds_"): k = "LDS"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): k = "VMEM"
    else: k = "other"
    c[k] += 1
    if k in ("SALU",): detail[re.sub(r"_b(32|64)$|_u32$|_i32$", "", op)] += 1
print(dict(c))
print("SALU detail:", detail.most_common(14))
