#!/bin/bash
python tools/halo_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_halo4.log
python -m pytest tests/test_gpu_conv3d.py -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('self_check'), d.get('roofline_mfma'))"
