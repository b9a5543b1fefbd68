#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing (tools/isa_stats.sh writes /tmp/isa_<file>.s).
usage: tools/isa_mix.py /tmp/isa_kernels_wf_primary.s <substring of the mangled name> [n most common]"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):\s*(?:;.*)?\n', txt, re.M):
    start = m.end()
    end = txt.index('.Lfunc_end', start)
    c = Counter()
    for ln in txt[start:end].split('\n'):
        ln = ln.strip()
        if not ln or ln.startswith(('.', ';', '//')) or ln.endswith(':'):
            continue
        c[ln.split()[0]] += 1
    tot = sum(c.values())
    print(m.group(1)[:100])
    print(' total', tot, 'valu', sum(v for k, v in c.items() if k.startswith('v_')), 'salu', sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith(('s_load', 's_buffer', 's_waitcnt', 's_nop'))),
          's_load', sum(v for k, v in c.items() if k.startswith(('s_load', 's_buffer'))), 'vmem', sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_'))),
          'lane moves', c['v_readlane_b32'] + c['v_writelane_b32'], 'scratch', sum(v for k, v in c.items() if k.startswith('scratch_')))
    print(' ', ', '.join(f'{k} {v}' for k, v in c.most_common(n)))
