#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_modules.py tests/test_gpu_kernels.py -x -q -m gpu -k "masked or topk" 2>&1 | tail -15
for mt in 0 1; do
SGC_MASKED_TAIL=$mt timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 --sustain 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('masked_tail $mt', d['value'], 'scenes/s', d['ms_per_step'], 'ms self_check', d['self_check']['mismatching'])"
done
