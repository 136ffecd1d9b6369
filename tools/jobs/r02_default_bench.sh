#!/bin/bash
for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['steps'], d['warmup'], d['ms_per_step'], d['self_check']['mismatching'], d['sustained']['value'], d['strict_fp32']['value'], d['roofline']['frac'])"; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['steps'])"
