#!/bin/bash
for w in cfg3_arkit cfg4_scannet200_large; do
timeout 600 python tools/train_step_bench.py --workload $w --steps 3 2>&1 | tail -1
SGC_TRAIN_CONV=library timeout 600 python tools/train_step_bench.py --workload $w --steps 3 2>&1 | tail -1
done
