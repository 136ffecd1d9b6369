#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -30 > gpurun_out/r02_gpu_tests_2.log; cat gpurun_out/r02_gpu_tests_2.log
