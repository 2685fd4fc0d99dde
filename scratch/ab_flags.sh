#!/bin/bash
# A/B of one library under several mi_debug_flags values through bench.py (one box): bash scratch/ab_flags.sh <rounds> <flags>...
R=$1; shift
for r in $(seq $R); do for f in "$@"; do
  python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-fp32-record --debug-flags $f > /tmp/abf.json 2>/tmp/abf.err || { tail -5 /tmp/abf.err; exit 1; }
  python - "$f" <<'PY'
import json,sys
d=json.loads(open('/tmp/abf.json').read().strip().splitlines()[-1])
print('flags', sys.argv[1], 'value %.0f' % d['value'], 'rollout %.2f update %.2f ms' % (d['phase_ms_per_step']['rollout'], d['phase_ms_per_step']['update']))
PY
done; done
