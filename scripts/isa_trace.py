#!/usr/bin/env python3
"""One character per instruction of a kernel's `hipcc -S` listing, block by block: M matrix, E transcendental,
v other vector, L LDS read, W LDS write, G global memory, S scratch, . wait, |B| barrier, > branch, s scalar.
usage: isa_trace.py file.s <substring of the mangled kernel name> [first label]"""
import re
import sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = [i for i, l in enumerate(s) if l.startswith('_Z') and key in l and ':' in l][0]
end = [i for i in range(start, len(s)) if s[i].strip().startswith('s_endpgm')][0]
body = s[start:end]
out = []
for l in body:
    t = l.strip()
    if re.match(r'^\.LBB', l):
        out.append('\n' + t.split(':')[0] + (' (loop)' if 'Loop' in l else '') + ' ')
        continue
    if not t or t.startswith(('.', ';')):
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): out.append('M')
    elif op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt')): out.append('E')
    elif op.startswith('v_'): out.append('v')
    elif op.startswith(('ds_read', 'ds_load')): out.append('L')
    elif op.startswith('ds_'): out.append('W')
    elif op.startswith('scratch'): out.append('S')
    elif op.startswith(('global', 'buffer', 'flat')): out.append('G')
    elif op.startswith('s_waitcnt'): out.append('.')
    elif op.startswith('s_barrier'): out.append('|B|')
    elif op.startswith(('s_cbranch', 's_branch')): out.append('>')
    else: out.append('s')
print(''.join(out))
