#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_modules.py -m gpu -x -q 2>&1 | tail -6
for w in cfg3_arkit cfg4_scannet200_large cfg5_arkit_large; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
timeout 300 python bench.py --workload cfg2_scannet --views 100 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2@100views', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
