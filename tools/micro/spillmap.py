"""Where a kernel's scratch traffic sits: python3 tools/micro/spillmap.py file.s kernel-name-substring [buckets]"""
import re
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 30
parts = re.split(r'\n(_Z\w+):[^\n]*\n', s)
for k in range(1, len(parts), 2):
    if key not in parts[k] or '.end_amdhsa_kernel' not in parts[k + 1]:
        continue
    body = parts[k + 1][:parts[k + 1].find('.end_amdhsa_kernel')]
    lines = body.split('\n')
    n = len(lines)
    print(parts[k], n, 'lines')
    for b in range(B):
        seg = lines[b * n // B:(b + 1) * n // B]
        cnt = lambda w: sum(1 for l in seg if w in l)
        regs = set()
        for l in seg:
            for m in re.finditer(r'v\[(\d+):(\d+)\]', l):
                regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
            for m in re.finditer(r'\bv(\d+)\b', l):
                regs.add(int(m.group(1)))
        print(b, 'scratch_store', cnt('scratch_store'), 'scratch_load', cnt('scratch_load'), 'mfma', cnt('v_mfma'), 'barrier', cnt('s_barrier'), 'ds', cnt('ds_'), 'nregs', len(regs))
