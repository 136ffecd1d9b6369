#!/bin/bash
# last job of round 5: full -m gpu suite, smoke, the training-step lines + kernel table, the default bench line
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r05_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python tools/train_step_bench.py --steps 40 2>/dev/null | tail -1 > gpurun_out/r05_train_step.json
SGC_TRAIN_PACK_BATCH=0 SGC_BN_FUSE_TAIL=0 timeout 600 python tools/train_step_bench.py --steps 40 2>/dev/null | tail -1 >> gpurun_out/r05_train_step.json
timeout 600 python tools/train_step_bench.py --steps 10 --profile >> gpurun_out/r05_train_step.json 2> gpurun_out/r05_train_step_kernels.raw; echo train rc $?
grep -v "amdgpu.ids\|warn\|Warning" gpurun_out/r05_train_step_kernels.raw | cut -c1-200 > gpurun_out/r05_train_step_kernels.txt; rm -f gpurun_out/r05_train_step_kernels.raw
timeout 900 python bench.py > gpurun_out/r05_bench_cfg2.json 2> /dev/null; echo bench rc $?
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_cfg2_driver_cmd.json 2>/dev/null; echo driver-cmd rc $?
python - <<'PY'
import json
for n in ("cfg2", "cfg2_driver_cmd"):
    d = json.loads(open(f"gpurun_out/r05_bench_{n}.json").readline())
    print(n, d["value"], "sustained", d["sustained"]["value"], "gather", d["roofline"]["frac"], "self_check", d["self_check"]["mismatching"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
cat gpurun_out/r05_train_step.json
