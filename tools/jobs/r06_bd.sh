#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "block_diagonal or projected_query" 2>&1 | tail -4
python3 tools/blockdiag_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_blockdiag_bench.txt
