#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_conv3d.py -x -q 2>&1 | tail -2
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['self_check']['mismatching'], d['sustained']['value'], d['roofline_mfma']['avg_launch_us'])"; done
