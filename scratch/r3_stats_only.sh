#!/bin/bash
# kernel stats of the default bench command + two default bench lines (gpurun -- 'bash scratch/r3_stats_only.sh <tag>')
set -o pipefail
T=${1:-r03_run10}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python bench.py > $O/${T}_bf16_bench.json 2> $O/${T}_bench.err || exit 1
python bench.py --no-cpu-baseline --no-fp32-record > $O/${T}b_bf16_bench.json 2>> $O/${T}_bench.err || exit 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_prof -o runc -- python3 $R/bench.py --no-cpu-baseline --no-fp32-record --steps 3 --warmup 1 > $O/${T}_bf16_bench_under_rocprof.json 2> $O/r3_prof.err || exit 3
cp $(find $O/r3_prof -name '*kernel_stats.csv' | head -1) $O/${T}_bf16_kernel_stats.csv
rm -rf $O/r3_prof
cd $R
python - $O/${T}_bf16_bench.json $O/${T}b_bf16_bench.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'value %.0f' % d['value'], 'ms %.2f' % d['ms_per_step'], d['phase_ms_per_step'], 'roofline', {k:d['roofline'][k] for k in ('achieved','frac','kernel','avg_launch_ms')}, 'fp32', d.get('fp32'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
