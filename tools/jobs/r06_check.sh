#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r06_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('driver cmd', d['value'], 'sustained', d['sustained']['value'], 'gather', d['roofline']['frac'], 'self_check', d['self_check'])"
