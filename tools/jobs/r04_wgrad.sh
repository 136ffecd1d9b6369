#!/bin/bash
# halo-form weight gradient: parity tests + A/B against the tile kernel
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv3d.py -x -q -m gpu -k wgrad 2>&1 | tail -15 > gpurun_out/r04_wgrad_tests.txt
timeout 600 python tools/wgrad_ab.py > gpurun_out/r04_wgrad_ab.txt 2>&1
cat gpurun_out/r04_wgrad_tests.txt gpurun_out/r04_wgrad_ab.txt
