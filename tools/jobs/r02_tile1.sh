#!/bin/bash
# round 2, first GPU visit: parity of the tiled gather + sweep of its window parameters
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tiled or headmajor or projection" 2>&1 | tail -15 > gpurun_out/r02_tile_tests.log
cat gpurun_out/r02_tile_tests.log
timeout 900 python tools/tile_bench.py cfg2 64x80 ring > gpurun_out/r02_tile_cfg2.log 2>&1; tail -12 gpurun_out/r02_tile_cfg2.log
timeout 900 python tools/tile_bench.py cfg4 59x80 ring > gpurun_out/r02_tile_cfg4.log 2>&1; tail -12 gpurun_out/r02_tile_cfg4.log
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r02_gpu_tests_1.log; cat gpurun_out/r02_gpu_tests_1.log
