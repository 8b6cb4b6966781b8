#!/bin/bash
# same-box A/B of the shipped library against tools/ab/liblnx_prev.so (an older build: LNX_LIB_OLDER=1); N alternations
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in new prev; do
    if [ $v = prev ]; then export LNX_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/liblnx_prev.so LNX_LIB_OLDER=1; else unset LNX_LIB_PATH LNX_LIB_OLDER; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline --profile-steps 0 --no-sched-calibration 2>gpurun_out/ab_$v.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['ms_per_step'], d['config3_n1']['ms_per_step'], d['loss'])" || { tail -5 gpurun_out/ab_$v.err; exit 1; }
  done
done
