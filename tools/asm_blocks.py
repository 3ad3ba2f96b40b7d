#!/usr/bin/env python3
"""Basic-block census of one kernel in a hipcc -S listing.  usage: asm_blocks.py file.s kernel-substring [dump.s]"""
import re, sys
s = open(sys.argv[1]).read()
start = s.index(sys.argv[2]); i = s.index(':\n', start); j = s.index('s_endpgm', i)
body = s[i:j].split('\n')
if len(sys.argv) > 3: open(sys.argv[3], 'w').write('\n'.join(body))
blocks = []; cur = ['entry', []]; blocks.append(cur)
for l in body:
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t): cur = [t.split(':')[0], []]; blocks.append(cur)
    elif t and not t.startswith((';', '.')): cur[1].append(t)
tot = 0
for name, ins in blocks:
    v = sum(1 for x in ins if x.startswith('v_') and not x.startswith(('v_mfma', 'v_cmp', 'v_readlane', 'v_readfirst')))
    print(f"{name:12s} n={len(ins):4d} valu={v:4d} vcmp={sum(1 for x in ins if x.startswith('v_cmp')):3d} mfma={sum(1 for x in ins if x.startswith('v_mfma')):3d} "
          f"ds={sum(1 for x in ins if x.startswith('ds_')):3d} vmem={sum(1 for x in ins if x.startswith(('global_','buffer_'))):2d} salu={sum(1 for x in ins if x.startswith('s_')):3d}  {ins[-1][:44] if ins else ''}")
