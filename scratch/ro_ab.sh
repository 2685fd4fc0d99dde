#!/bin/bash
# per-kernel durations of one env group's chain (rocprofv3 kernel trace of scratch/rollout_pipe.py) for several library builds: bash scratch/ro_ab.sh <lib.so>...
L=train-procgen-pytorch_amd/mi355/libmi355ppo.so
R=$GRAFT_REPO_ROOT
cp $R/$L /tmp/lib_orig.so
export TMPDIR=/tmp
for v in "$@"; do
  cp $R/$v $R/$L
  rm -rf /tmp/rot; (cd /tmp && GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --output-format csv -d /tmp/rot -o runc -- python3 $R/scratch/rollout_pipe.py 4 256 1 0 > /tmp/rot.log 2>&1) || { tail -3 /tmp/rot.log; break; }
  echo "== $v"; python $R/scratch/ro_timeline4.py /tmp/rot | tail -8
done
cp /tmp/lib_orig.so $R/$L
