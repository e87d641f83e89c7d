#!/usr/bin/env python3
"""Per basic block of one kernel's ISA listing (hipcc -S --cuda-device-only; cut the kernel out of the file first): line, instruction
counts by issue port (V vector ALU, S scalar ALU, smem, lds, vmem, nop, wait, br, bar) and the branches that leave it --
the raw material of profiles/r04_isa_breakdown_stream.txt.  tools/isa_blocks.py kernel.s"""
import sys,re,collections
lines=open(sys.argv[1]).read().split("\n")
blocks=[];cur=["entry",collections.Counter(),[]]
def cat(op):
    if op.startswith("v_"):
        return "V"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_cbranch","s_branch")): return "br"
    if op.startswith("s_barrier"): return "bar"
    if op.startswith(("s_load","s_buffer")): return "smem"
    if op.startswith("s_"): return "S"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_","buffer_","flat_","scratch_")): return "vmem"
    return "o"
for i,l in enumerate(lines):
    t=l.strip()
    m=re.match(r"^(\.LBB\d+_\d+):",t)
    m2=re.match(r"^; %bb\.(\d+):",t)
    if m or m2:
        blocks.append(cur); cur=[m.group(1) if m else "bb."+m2.group(1),collections.Counter(),[] ,i+1]
        continue
    if not t or t.startswith((";",".","//")) or t.endswith(":"): continue
    op=t.split()[0]
    cur[1][cat(op)]+=1
    if cat(op)=="br": cur[2].append(t)
blocks.append(cur)
for b in blocks:
    print(b[0], (b[3] if len(b)>3 else 0), dict(b[1]), b[2])
