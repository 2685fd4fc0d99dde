#!/bin/bash
# runtime environment switches against the pipelined rollout (one box): bash scratch/env_ab.sh [rounds]
R=${1:-2}
for r in $(seq $R); do
  echo -n "base:                     "; GPU_MAX_HW_QUEUES=8 python scratch/rollout_pipe.py 4 256 1 0 2>&1 | tail -1
  echo -n "HIP_FORCE_DEV_KERNARG=1:  "; HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=8 python scratch/rollout_pipe.py 4 256 1 0 2>&1 | tail -1
  echo -n "HIP_FORCE_DEV_KERNARG=0:  "; HIP_FORCE_DEV_KERNARG=0 GPU_MAX_HW_QUEUES=8 python scratch/rollout_pipe.py 4 256 1 0 2>&1 | tail -1
  echo -n "HSA_ENABLE_INTERRUPT=0:   "; HSA_ENABLE_INTERRUPT=0 GPU_MAX_HW_QUEUES=8 python scratch/rollout_pipe.py 4 256 1 0 2>&1 | tail -1
  echo -n "both:                     "; HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0 GPU_MAX_HW_QUEUES=8 python scratch/rollout_pipe.py 4 256 1 0 2>&1 | tail -1
done
