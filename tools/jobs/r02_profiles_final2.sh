#!/bin/bash
timeout 900 python bench.py > gpurun_out/r02_bench_cfg2.json 2> gpurun_out/r02_bench_cfg2.err; echo bench rc $?
timeout 300 python bench.py --conv-mode f32 --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 > gpurun_out/r02_bench_cfg2_f32.json 2>/dev/null; echo f32 rc $?
timeout 300 python tools/train_step_bench.py --steps 5 --profile > gpurun_out/r02_train_step.json 2> gpurun_out/r02_train_step_kernels.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02_bench_cfg2.json").readline())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["strict_fp32"]["value"], d["sustained"]["value"], d["self_check"]["mismatching"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print(open("gpurun_out/r02_bench_cfg2_f32.json").readline()[:200])
print(open("gpurun_out/r02_train_step.json").readline())
PY
