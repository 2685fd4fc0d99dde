"""Waits that sit right behind the loads they wait for: per kernel of a `hipcc -S` listing, every s_waitcnt vmcnt(N) found within
WINDOW lines after a global load in the same basic-block run, with the next instruction (a v_mov of a loaded register = a phi copy).
python scratch/vm_after_load.py file.s [window]"""
import re, sys
s = open(sys.argv[1]).read()
W = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    lines = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith(';')]
    # main loop region: after the first 's_barrier'
    try:
        first_bar = next(k for k, l in enumerate(lines) if l.startswith('s_barrier'))
    except StopIteration:
        continue
    hits = []
    for k, l in enumerate(lines):
        if k < first_bar or not l.startswith('global_load'):
            continue
        for j in range(k + 1, min(k + 1 + W, len(lines))):
            if lines[j].startswith(('.LBB', 's_cbranch', 's_branch', 's_barrier')):
                break
            if re.match(r's_waitcnt.*vmcnt', lines[j]):
                hits.append((k, lines[j], lines[j + 1] if j + 1 < len(lines) else ''))
                break
    if hits:
        print(m.group(1)[:110])
        for h in hits[:8]:
            print('    line %5d  %-28s then  %s' % h)
