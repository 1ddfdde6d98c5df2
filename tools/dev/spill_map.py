#!/usr/bin/env python3
"""Where does a kernel spill?  Per basic block of one kernel in a hipcc -S listing: MFMAs, scratch stores / loads, v_writelane / v_readlane
(SGPR spills), instructions.  usage: spill_map.py file.s kernel-substring"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith('_Z') and key in l and l.rstrip().split(';')[0].strip().endswith(':'))
end = next(i for i in range(start, len(s)) if '.end_amdhsa_kernel' in s[i] or (i > start and s[i].startswith('.Lfunc_end')))
segs = [['entry', 0, 0, 0, 0, 0, 0]]
for l in s[start + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        segs.append([m.group(1), 0, 0, 0, 0, 0, 0]); continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    g = segs[-1]
    g[6] += 1
    if t.startswith('v_mfma'): g[1] += 1
    elif t.startswith('scratch_store'): g[2] += 1
    elif t.startswith('scratch_load'): g[3] += 1
    elif t.startswith('v_writelane'): g[4] += 1
    elif t.startswith('v_readlane'): g[5] += 1
print(f"{'block':>12} {'mfma':>5} {'sst':>5} {'sld':>5} {'wlane':>5} {'rlane':>5} {'instr':>6}")
for g in segs:
    if g[6]: print(f"{g[0]:>12} {g[1]:5d} {g[2]:5d} {g[3]:5d} {g[4]:5d} {g[5]:5d} {g[6]:6d}")
