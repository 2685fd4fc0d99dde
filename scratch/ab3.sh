#!/bin/bash
# A/B/C of library builds on one box: bash scratch/ab3.sh <kernel-substring> <rounds> lib1 lib2 ...
K=$1; R=$2; shift 2
L=train-procgen-pytorch_amd/mi355/libmi355ppo.so
cp $L /tmp/lib_orig.so
for r in $(seq $R); do for v in "$@"; do
  cp $v $L
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-record --profile-period 2 > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; cp /tmp/lib_orig.so $L; exit 1; }
  python - "$v" "$K" <<'PY'
import json,sys
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
ks=[k for k in d['kernels'] if sys.argv[2] in k['kernel']]
print(sys.argv[1], 'update %.2f ms' % d['phase_ms_per_step']['update'], ' '.join('%s %.1f us' % (k['kernel'], k['ms']/k['launches']*1e3) for k in ks), 'loss', d['loss_total'])
PY
done; done
cp /tmp/lib_orig.so $L
