#!/bin/bash
# side-stream test + two-stream kernel trace of one minibatch (gaps around the fork points)
python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "side_stream" > gpurun_out/t_ev.log 2>&1; tail -2 gpurun_out/t_ev.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/mbt -o runc -- python3 $R/scratch/pmc_workload.py 5 bf16 > $R/gpurun_out/mbt.log 2>&1 || { tail -5 $R/gpurun_out/mbt.log; exit 1; }
cd $R && python scratch/mb_timeline2.py gpurun_out/mbt > gpurun_out/tl_ev.txt && rm -rf gpurun_out/mbt && sed -n 9,17p gpurun_out/tl_ev.txt && tail -8 gpurun_out/tl_ev.txt
