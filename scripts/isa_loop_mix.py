#!/usr/bin/env python3
"""Instruction mix of every loop of one kernel in a `hipcc -S` listing: MFMA, transcendental, other
VALU, LDS, global memory, waits, barriers, scalar.  usage: isa_loop_mix.py file.s <substring of the kernel's mangled name>"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(s) if l.startswith('_Z') and key in l and l.split(':')[0].endswith(l.split(':')[0]) and ':' in l][0]
end = [i for i in range(start, len(s)) if s[i].strip().startswith('s_endpgm')][0]
body = s[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i


def mix(lines):
    c = Counter()
    for l in lines:
        l = l.strip()
        if not l or l.startswith(('.', ';')) or l.endswith(':'):
            continue
        op = l.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt')): c['trans'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem'] += 1
        elif op.startswith('s_waitcnt'): c['wait'] += 1
        elif op.startswith('s_barrier'): c['barrier'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        else: c[op] += 1
    return dict(c)


print("whole kernel:", len(body), "lines", mix(body))
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            print("loop", t, "lines", i - labels[t], mix(body[labels[t]:i + 1]))
            if len(sys.argv) > 3:
                vc = Counter(x.strip().split()[0] for x in body[labels[t]:i + 1] if x.strip().startswith('v_') and not x.strip().startswith('v_mfma'))
                print("   ", vc.most_common(40))
