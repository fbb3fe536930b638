#!/usr/bin/env python3
"""Instruction mix per kernel of a hipcc -save-temps .s file:  python tools/asm_mix.py file.s [name-filter]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for idx, (i, name) in enumerate(starts):
    if flt not in name:
        continue
    j = next((k for k in range(i, len(lines)) if lines[k].startswith('.Lfunc_end')), len(lines))
    ins = [l.strip().split()[0] for l in lines[i:j] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = Counter(ins)
    print(name[:120], '| total', len(ins), '| mfma', sum(v for k, v in c.items() if 'mfma' in k),
          '| valu', sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k),
          '| ds', sum(v for k, v in c.items() if k.startswith('ds_')), '| vmem', sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'scratch_'))))
    print('    ' + ', '.join(f'{k}:{v}' for k, v in c.most_common(32)))
