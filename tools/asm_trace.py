#!/usr/bin/env python3
"""Hot-path census of a kernel listing: walks from START, at every conditional branch uses the policy (dict line->'t'/'n',
default per kind below), prints per-segment counts between marker lines.
usage: asm_trace.py dump.s START_LABEL policy.txt [-v]
policy file lines: "<lineno> t|n"  (line numbers of the listing); "mark <lineno> <name>" starts a new segment at that line."""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
start = sys.argv[2]
pol = {}; marks = {}
for l in open(sys.argv[3]):
    l = l.split('#')[0].split()
    if not l: continue
    if l[0] == 'mark': marks[int(l[1])] = l[2]
    else: pol[int(l[0])] = l[1]
verbose = '-v' in sys.argv
lab = {}
for i, l in enumerate(src):
    m = re.match(r'^(\.LBB\d+_\d+):', l.strip())
    if m: lab[m.group(1)] = i
def cat(op):
    return ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'ds' if op.startswith('ds_') else
            'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'nop' if op == 's_nop' else
            'wait' if op == 's_waitcnt' else 'smem' if op.startswith('s_load') else 'salu' if op.startswith('s_') else 'other')
i = lab[start] + 1
seg = 'top'; segs = collections.OrderedDict(); ops = collections.defaultdict(collections.Counter)
steps = 0
while steps < 40000:
    steps += 1
    if (i + 1) in marks: seg = marks[i + 1]
    t = src[i].strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        if m.group(1) == start: break
        i += 1; continue
    if not t or t.startswith((';', '.')): i += 1; continue
    op = t.split()[0]
    segs.setdefault(seg, collections.Counter())[cat(op)] += 1; ops[seg][op] += 1
    if verbose: print(f"{i+1:6d} [{seg}] {t}")
    if op == 's_endpgm': break
    if op == 's_branch': i = lab[t.split()[1]]; continue
    if op.startswith('s_cbranch'):
        tgt = t.split()[1]
        d = pol.get(i + 1)
        if d is None:
            d = 'n' if op in ('s_cbranch_execz',) else 'n'
            print(f"  ?? [line {i+1}] {t} -> default not taken")
        if d == 't': i = lab[tgt]; continue
    i += 1
tot = collections.Counter()
for s, c in segs.items():
    print(f"{s:14s} " + "  ".join(f"{k} {c[k]:4d}" for k in ('valu', 'mfma', 'salu', 'smem', 'ds', 'vmem', 'nop', 'wait')))
    tot.update(c)
print(f"{'TOTAL':14s} " + "  ".join(f"{k} {tot[k]:4d}" for k in ('valu', 'mfma', 'salu', 'smem', 'ds', 'vmem', 'nop', 'wait')))
if '-o' in sys.argv:
    for s in ops:
        print(s, ops[s].most_common(12))
