#!/bin/bash
# cost of bench.py's own HIP-event brackets inside the timed region: bash scratch/ab_prof.sh [rounds]
R=${1:-3}
for r in $(seq $R); do for f in "--profile-period 8" "--profile-period 24" "--no-kernel-profile"; do
  python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-fp32-record $f > /tmp/abp.json 2>/tmp/abp.err || { tail -5 /tmp/abp.err; exit 1; }
  python - "$f" <<'PY'
import json,sys
d=json.loads(open('/tmp/abp.json').read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('%-22s' % sys.argv[1], 'value %.0f' % d['value'], 'rollout %.2f update %.2f ms' % (d['phase_ms_per_step']['rollout'], d['phase_ms_per_step']['update']), 'dominant %.1f us x %s' % ((r.get('avg_launch_ms') or 0)*1e3, r.get('launches')))
PY
done; done
