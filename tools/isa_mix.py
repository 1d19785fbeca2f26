#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: isa_mix.py file.s substring-of-mangled-name"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if re.match(r'^_Z\w+:', l) and key in l.split(':')[0])
ops = []
for l in txt[start + 1:]:
    if 's_endpgm' in l:
        break
    if l.startswith('\t') and not l.strip().startswith(('.', ';')):
        ops.append(l.split()[0])
c = collections.Counter(ops)
valu = sum(v for k, v in c.items() if k.startswith('v_'))
print(txt[start].split(':')[0][:100])
print('instructions', len(ops), 'VALU', valu, 'SALU', sum(v for k, v in c.items() if k.startswith('s_')),
      'VMEM', sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))),
      'LDS', sum(v for k, v in c.items() if k.startswith('ds_')))
for k, v in c.most_common(25):
    print('  %-28s %d' % (k, v))
