#!/bin/bash
# the LDS-tiled DFA3D backward alone: parity, then bins / halos and the timing knobs (1 no phase B, 2 no flush, 4 no value loads, 8 no depth atomics, 16 no row arithmetic)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "binned_backward or training_level or item_list or tiled_gather" 2>&1 | tail -5
export SGC_DIAG=1
SGC_BWD_CONFIGS="8,11,2,2,4;8,11,2,2,4,1;8,11,2,2,4,2;8,11,2,2,4,16;8,11,2,2,4,17;16,11,3,2,4;8,8,2,2,4;8,16,2,2,4;12,11,2,2,4;8,11,3,3,4" timeout 900 python tools/bwd_tile_bench.py cfg2 2>&1 | grep -v "amdgpu.ids\|warn" | tee gpurun_out/r06_bwd_tile_bench_cfg2.txt
SGC_BWD_CONFIGS="16,11,3,2,4;8,11,2,2,4" timeout 900 python tools/bwd_tile_bench.py cfg4 2>&1 | grep -v "amdgpu.ids\|warn" | tee gpurun_out/r06_bwd_tile_bench_cfg4.txt
