#!/bin/bash
# runtime settings of the HIP runtime beside the headline run, alternated (kernel arguments in device memory, hardware queues)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_env_ab.txt
: > $out
run() {
  tag="$1"; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-44s' % '$tag', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])" | tee -a $out
}
for rnd in 1 2; do
  run "default" A=1
  run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
  run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
  run "GPU_MAX_HW_QUEUES=4" GPU_MAX_HW_QUEUES=4
  run "default (again)                            " A=2
done
