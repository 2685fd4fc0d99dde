"""Every s_waitcnt vmcnt of a kernel with the instruction behind it and the distance (in instructions) to the nearest earlier global load:
python scratch/vm_waits_list.py file.s kernel-substring"""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    if sys.argv[2] not in m.group(1):
        continue
    lines = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith(';')]
    print(m.group(1)[:100])
    last_load = None
    for k, l in enumerate(lines):
        if l.startswith('global_load'):
            last_load = k
        if re.match(r's_waitcnt.*vmcnt', l):
            nxt = lines[k + 1] if k + 1 < len(lines) else ''
            print('   %5d  %-26s dist %-5s | %s' % (k, l, (k - last_load) if last_load is not None else '-', nxt[:70]))
