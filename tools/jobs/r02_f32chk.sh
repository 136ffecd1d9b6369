#!/bin/bash
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['strict_fp32']['value'], d['strict_fp32']['ms_per_step'], d['sustained']['value'])"
done
