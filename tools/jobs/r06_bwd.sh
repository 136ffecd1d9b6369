#!/bin/bash
# round-6 job 2: the LDS-tiled DFA3D backward (parity tests, training step A/B, bin sweep) and the trimmed phase 1 of the tiled gather
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
filt() { grep -v "amdgpu.ids\|warn\|Warning"; }
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "binned_backward or binned or item_list or tiled_gather or bf16_storage or training_level" 2>&1 | tail -15 > gpurun_out/r06_bwd_tests.txt; echo tests rc $?
cat gpurun_out/r06_bwd_tests.txt
# tiled gather: one window test (tile_ds 1) vs the round-5 form (0), waves per workgroup, at the Cm = 16 and Cm = 32 production configs
SGC_TILE_CONFIGS="16,22,3,3,1,8,1,1,1,0,1;16,22,3,3,1,8,1,1,1,0,0;16,22,3,3,1,16,1,1,1,0,1;16,22,3,3,1,8,1,1,1,0,1;16,22,3,3,1,8,1,1,1,0,0" timeout 600 python tools/tile_bench.py cfg4 64x80 2>&1 | filt > gpurun_out/r06_tile_ds_cfg4.txt; echo tile cfg4 rc $?
SGC_TILE_CONFIGS="16,22,3,3,0,8,1,1,1,1,1;16,22,3,3,0,8,1,1,1,1,0;16,22,3,3,0,8,1,1,1,1,1" timeout 600 python tools/tile_bench.py cfg2 64x80 2>&1 | filt > gpurun_out/r06_tile_ds_cfg2.txt; echo tile cfg2 rc $?
cat gpurun_out/r06_tile_ds_cfg4.txt gpurun_out/r06_tile_ds_cfg2.txt
# training step: item kernel vs tiled backward, bins / halos
for spec in 0 "16,22,3,3" "16,11,3,3" "16,11,2,2" "16,22,2,2" "8,11,2,2" 0 "16,22,3,3"; do
  echo "SGC_TRAIN_BWD=$spec: $(SGC_TRAIN_BWD=$spec timeout 600 python tools/train_step_bench.py --steps 40 2>/dev/null | tail -1)"
done > gpurun_out/r06_train_bwd_ab.txt 2>&1
cat gpurun_out/r06_train_bwd_ab.txt
timeout 600 python tools/train_step_bench.py --steps 10 --profile > gpurun_out/r06_train_step_a.json 2> gpurun_out/r06_train_step_kernels_a.raw; echo train rc $?
grep -v "amdgpu.ids\|warn\|Warning" gpurun_out/r06_train_step_kernels_a.raw | cut -c1-200 > gpurun_out/r06_train_step_kernels_a.txt; rm -f gpurun_out/r06_train_step_kernels_a.raw
head -30 gpurun_out/r06_train_step_kernels_a.txt
timeout 600 python bench.py --workload cfg4_scannet200_large --no-cpu-baseline --no-strict-fp32 2>/dev/null | cut -c1-1200
timeout 600 python bench.py --workload cfg5_arkit_large --no-cpu-baseline --no-strict-fp32 2>/dev/null | cut -c1-1200
