#!/usr/bin/env python3
"""Static instruction mix of the kernels in a hipcc -save-temps .s file.
Usage: isa_mix.py <file.s> <substring of the mangled kernel name> [...]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
funcs = re.split(r'\n(_ZN8himg_dev[^\n:]+):', s)
for i in range(1, len(funcs), 2):
    name = funcs[i]
    body = funcs[i + 1].split('.Lfunc_end')[0]
    if not any(p in name for p in sys.argv[2:]):
        continue
    c = collections.Counter()
    for line in body.splitlines():
        line = line.strip()
        if not line or line.startswith(';') or line.startswith('.') or line.endswith(':'):
            continue
        c[line.split()[0]] += 1
    tot = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    salu = sum(v for k, v in c.items() if k.startswith('s_'))
    print(name[:70], 'total', tot, 'valu', valu, 'salu', salu)
    print('  ', c.most_common(45))
