#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv3d.py tests/test_gpu_modules.py -x -q -k "fp16 or bf16_mode or reduced or options or stagger or strict" 2>&1 | tail -5
timeout 600 python bench.py --conv-mode fp16 --no-cpu-baseline > gpurun_out/r04_bench_cfg2_fp16.json 2>gpurun_out/r04_bench_cfg2_fp16.err; echo rc $?
timeout 600 python bench.py --conv-mode bf16 --no-cpu-baseline > gpurun_out/r04_bench_cfg2_bf16.json 2>/dev/null; echo rc $?
timeout 600 python bench.py --conv-mode f32 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r04_bench_cfg2_f32.json 2>/dev/null; echo rc $?
python - <<'PY'
import json
for n in ("cfg2_fp16", "cfg2_bf16", "cfg2_f32"):
    d = json.loads(open(f"gpurun_out/r04_bench_{n}.json").readline())
    print(n, d["value"], "gather", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "mfma", (d.get("roofline_mfma") or {}).get("frac"), d["self_check"]["mismatching"], d["dtype"][:40])
PY
