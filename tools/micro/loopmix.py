"""Instruction mix of the innermost loop of every kernel in a hipcc -save-temps .s file (micro-benchmarks: tools/micro)."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
parts = re.split(r'\n(_Z\w+):[^\n]*\n', s)
for k in range(1, len(parts), 2):
    name, body = parts[k], parts[k + 1]
    if '.end_amdhsa_kernel' not in body:
        continue
    body = body[:body.find('.end_amdhsa_kernel')]
    lines = [l.strip() for l in body.split('\n')]
    idx = [i for i, l in enumerate(lines) if l.startswith('v_mfma')]
    if not idx:
        continue
    # the loop = the backward branch closest behind the last MFMA and its target label
    lab = {l.split(':')[0]: i for i, l in enumerate(lines) if l.startswith('.LBB')}
    seg = None
    for i in range(idx[-1], len(lines)):
        if lines[i].startswith('s_cbranch'):
            t = lines[i].split()[-1]
            if t in lab and lab[t] < idx[0] + 1 and lab[t] < i:
                seg = lines[lab[t]:i + 1]
                break
    if seg is None:
        seg = lines[idx[0]:idx[-1] + 1]
    c = Counter()
    for l in seg:
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
            continue
        c[l.split()[0]] += 1
    pick = lambda p: sum(v for kk, v in c.items() if kk.startswith(p))
    print(name, 'loop instrs', sum(c.values()), 'mfma', pick('v_mfma'), 'valu', pick('v_') - pick('v_mfma'), 'ds', pick('ds_'),
          'scratch', pick('scratch'), 'salu', pick('s_'))
    print('  ', dict(c.most_common(18)))
