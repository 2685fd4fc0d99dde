import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('phase_ms_per_step'))
P = d.get('kernel_profile_period', 1) or 1
for k in d['kernels'][:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{k['kernel']:28s} {k['ms']*P/d['steps']:7.2f} ms/iter  {k['launches']}")
