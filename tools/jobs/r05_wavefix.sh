#!/bin/bash
# round 5: the wave-quantisation split (halo_wave_fix) in the THROUGHPUT geometry too? cfg3 / cfg5 with four / two scenes in flight;
# + the transposes of the training path on the HIP kernels (NchwToRowsFunction): tests and the training step
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "rows_transpose" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_modules.py -x -q -k "training or train" 2>&1 | tail -2
for wl in cfg3_arkit cfg5_arkit_large; do
n=${wl%%_*}
for f in 1 2 1 2; do
SGC_TUNE="halo_wave_fix=$f" timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-strict-fp32 --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n halo_wave_fix $f', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])"
done
done
timeout 600 python tools/train_step_bench.py --steps 10 2>/dev/null | tail -1
