#!/bin/bash
# Round-3 profile collection on the GPU box (gpurun -- 'bash scratch/r3_profiles.sh'): kernel stats of the default bench command, then the
# counter passes (each its own run: --kernel-trace + --pmc only) on scratch/pmc_workload.py, summarised into gpurun_out/.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_prof -o runc -- python3 $R/bench.py --no-cpu-baseline --no-fp32-record --steps 3 --warmup 1 > $O/r03_run9_bf16_bench_under_rocprof.json 2> $O/r3_prof.err || exit 1
cp $(find $O/r3_prof -name '*kernel_stats.csv' | head -1) $O/r03_run9_bf16_kernel_stats.csv
rm -rf $O/r3_prof
W="python3 $R/scratch/pmc_workload.py 2 bf16"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $O/r3_pmc_sq -o runc -- $W > $O/r3_pmc_sq.log 2>&1 || exit 2
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/r3_pmc_fetch -o runc -- $W > $O/r3_pmc_fetch.log 2>&1 || exit 3
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/r3_pmc_write -o runc -- $W > $O/r3_pmc_write.log 2>&1 || exit 4
rocprofv3 --kernel-trace --output-format csv --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/r3_pmc_p1 -o runc -- $W > $O/r3_pmc_p1.log 2>&1 || exit 5
rocprofv3 --kernel-trace --output-format csv --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS -d $O/r3_pmc_p2 -o runc -- $W > $O/r3_pmc_p2.log 2>&1 || exit 6
cd $R
python scratch/pmc_summary.py $O/r3_pmc_sq $O/r3_pmc_fetch $O/r3_pmc_write $O/r03_pmc_bf16.json > $O/r03_pmc_bf16.txt 2>&1 || exit 7
python scratch/pmc_pipes.py $O/r3_pmc_p1 $O/r3_pmc_p2 $O/r03_pipes_bf16.json > $O/r03_pipes_bf16.txt 2>&1 || exit 8
rm -rf $O/r3_pmc_sq $O/r3_pmc_fetch $O/r3_pmc_write $O/r3_pmc_p1 $O/r3_pmc_p2
python scratch/kstats.py $O 3 20 2>/dev/null | head -3
head -12 $O/r03_pmc_bf16.txt; head -12 $O/r03_pipes_bf16.txt
