#!/bin/bash
SGC_AB_KNOB=igemm_nbuf1 timeout 300 python tools/igemm_ab.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('self_check')['mismatching'], d['sustained']['value'])"
