#!/usr/bin/env python3
"""Common-path instruction census of a kernel listing (asm_blocks.py dump): walks from a start label, takes every
s_cbranch_execz (the rare paths are skipped), follows s_branch, and for other conditional branches uses the policy
given as label=t|n arguments (default: not taken).  Stops when it returns to the start label or hits `stop`.
usage: asm_path.py dump.s START [STOP] [Lxx=t ...] [-v]"""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
start = sys.argv[2]
args = sys.argv[3:]
verbose = '-v' in args
policy = dict(a.split('=') for a in args if '=' in a)
stops = [a for a in args if '=' not in a and a != '-v']
lab = {}
for i, l in enumerate(src):
    m = re.match(r'^(\.LBB\d+_\d+):', l.strip())
    if m: lab[m.group(1)] = i
i = lab[start] + 1
cnt = collections.Counter(); ops = collections.Counter()
steps = 0
while steps < 20000:
    steps += 1
    t = src[i].strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        if m.group(1) == start or m.group(1) in stops: break
        i += 1; continue
    if not t or t.startswith((';', '.')): i += 1; continue
    op = t.split()[0]
    if verbose: print(t)
    cat = ('mfma' if op.startswith('v_mfma') else 'vcmp' if op.startswith('v_cmp') else
           'valu' if op.startswith('v_') else 'ds' if op.startswith('ds_') else
           'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else 'nop' if op == 's_nop' else
           'wait' if op == 's_waitcnt' else 'salu' if op.startswith('s_') else 'other')
    cnt[cat] += 1; ops[op] += 1
    if op == 's_cbranch_execz' or op == 's_branch':
        i = lab[t.split()[1]]; continue
    if op.startswith('s_cbranch'):
        tgt = t.split()[1]
        key = f"{i+1}"
        take = policy.get(tgt, policy.get(key, 'n')) == 't'
        print(f"  [line {i+1}] {t}  -> {'taken' if take else 'not taken'}")
        if take: i = lab[tgt]; continue
    i += 1
print(dict(cnt)); print(sum(cnt.values()))
for k, v in ops.most_common(45): print(f"  {k:28s} {v}")
