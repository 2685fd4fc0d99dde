"""How far behind its issue is every global load first waited for?  For each kernel in a `hipcc -S` listing: per load, the
instruction distance to the first s_waitcnt vmcnt(N) that covers it (straight-line scan, branches ignored) and the MFMA / LDS /
barrier instructions in between.  Short distances inside the main loop = an exposed HBM round trip.
python scratch/vm_waits.py file.s [max_distance] [kernel-substring]"""
import re, sys
s = open(sys.argv[1]).read()
D = int(sys.argv[2]) if len(sys.argv) > 2 else 80
sub = sys.argv[3] if len(sys.argv) > 3 else ''
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M):
    if sub not in m.group(1):
        continue
    lines = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
    ops = [l for l in lines if not l.split()[0].endswith(':')]
    hits = []
    for k, l in enumerate(ops):
        if not l.startswith(('global_load', 'buffer_load')):
            continue
        later = 0
        for j in range(k + 1, len(ops)):
            o = ops[j]
            if o.startswith(('global_', 'buffer_', 'scratch_')):
                later += 1
            mm = re.match(r's_waitcnt.*vmcnt\((\d+)\)', o)
            if mm and int(mm.group(1)) <= later:
                mid = ops[k + 1:j]
                hits.append((k, j - k, sum(x.startswith('v_mfma') for x in mid), sum(x.startswith('ds_') for x in mid), sum(x.startswith('s_barrier') for x in mid), o))
                break
    short = [h for h in hits if h[1] <= D]
    print(m.group(1)[:110], len(ops), 'instructions,', len(hits), 'loads,', len(short), 'waited for within', D)
    for h in short:
        print('   load at %5d: waited +%-4d (mfma %d, ds %d, barriers %d)  %s' % h)
