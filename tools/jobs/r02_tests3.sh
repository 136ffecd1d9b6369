#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -40 > gpurun_out/r02_gpu_tests_3.log; cat gpurun_out/r02_gpu_tests_3.log
