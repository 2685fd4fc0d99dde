#!/bin/bash
# A/B of two builds of the library on ONE box: bash scratch/ab_libs.sh <libA.so> <libB.so> <kernel-name-substring> [rounds]
# (alternates the builds; prints update-phase ms and the per-launch time of the named kernel class from bench.py's own HIP-event profile)
A=$1; B=$2; K=$3; R=${4:-2}
L=train-procgen-pytorch_amd/mi355/libmi355ppo.so
cp $L /tmp/lib_orig.so
for r in $(seq $R); do for v in $A $B; do
  cp $v $L
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-record --profile-period 2 > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; cp /tmp/lib_orig.so $L; exit 1; }
  python - "$v" "$K" <<'PY'
import json,sys
d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
ks=[k for k in d['kernels'] if sys.argv[2] in k['kernel']]
print(sys.argv[1], 'update %.2f ms' % d['phase_ms_per_step']['update'], ' '.join('%s %.1f us' % (k['kernel'], k['ms']/k['launches']*1e3) for k in ks))
PY
done; done
cp /tmp/lib_orig.so $L
