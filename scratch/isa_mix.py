"""Instruction mix of one kernel from a `hipcc -S --cuda-device-only` listing: python scratch/isa_mix.py file.s name-substring"""
import re, sys, collections
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    if sys.argv[2] not in m.group(1):
        continue
    lines = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    cnt = collections.Counter()
    for l in lines:
        op = l.split()[0]
        if op.endswith(':'):
            continue
        k = ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'ds' if op.startswith('ds_') else
             'salu' if op.startswith('s_') else 'vmem' if op.startswith(('global', 'buffer', 'flat', 'scratch')) else 'other')
        cnt[k] += 1
    print(m.group(1)[:90], dict(cnt))
    ops = collections.Counter(l.split()[0] for l in lines if l.split()[0].startswith('v_'))
    print('  ', ops.most_common(14))
