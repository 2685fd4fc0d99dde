#!/bin/bash
# A/B of library builds on the pipelined rollout alone (one box): bash scratch/ab_roll.sh <flags> <rounds> <lib.so>...
F=$1; R=$2; shift 2
L=train-procgen-pytorch_amd/mi355/libmi355ppo.so
cp $L /tmp/lib_orig.so
for r in $(seq $R); do for v in "$@"; do
  cp $v $L
  echo -n "$v: "; GPU_MAX_HW_QUEUES=8 python scratch/rollout_pipe.py 4 256 1 $F 2>&1 | tail -1
done; done
cp /tmp/lib_orig.so $L
