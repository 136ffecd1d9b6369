#!/bin/bash
# opt-in plain-bf16 line (cfg2 and cfg5) + the test of its bound
python -m pytest tests/test_gpu_conv3d.py -m gpu -x -q -k "plain_bf16" 2>&1 | tail -3
python bench.py --conv-mode bf16 --steps 60 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_cfg2_bf16.json 2> gpurun_out/r03_bench_cfg2_bf16.err
python bench.py --conv-mode bf16 --storage bf16 --steps 60 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_cfg2_bf16_bf16store.json 2>/dev/null
python bench.py --conv-mode bf16 --workload cfg5_arkit_large --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r03_bench_cfg5_bf16.json 2>/dev/null
python bench.py --steps 60 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_cfg2_d.json 2>/dev/null
for f in r03_bench_cfg2_bf16 r03_bench_cfg2_bf16_bf16store r03_bench_cfg5_bf16 r03_bench_cfg2_d; do python - <<PY
import json
d = json.load(open("gpurun_out/$f.json"))
print("$f", d["value"], "sustained", d["sustained"]["value"], "self_check", d["self_check"]["mismatching"], d["dtype"][:40], "mfma frac", d["roofline_mfma"]["frac"] if d.get("roofline_mfma") else None)
PY
done
