#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
python __graft_entry__.py --smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
python bench.py --steps 20 --warmup 5 --breakdown 2>gpurun_out/bench.err | tee gpurun_out/bench.json
grep -v Warn gpurun_out/bench.err | tail -14
