#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tiled or headmajor" 2>&1 | tail -15 > gpurun_out/r02_tile3_tests.log
cat gpurun_out/r02_tile3_tests.log
export SGC_TILE_DIAG=1
# bw,bh,hx,hy,dl,nw,shift,nbuf,hg
C2="20,16,3,3,0,16,1,2,0;20,16,3,3,0,16,1,2,4;20,16,3,3,0,8,1,2,0;16,16,3,3,1,16,1,2,0;27,22,3,3,0,16,1,1,0;27,32,3,3,0,16,1,1,0;27,32,3,3,0,16,1,1,1;27,32,3,3,0,16,1,1,2;16,22,3,3,0,16,1,2,0;40,16,3,3,0,16,1,1,0;20,11,3,3,0,16,1,2,0;20,16,4,4,0,16,0,2,0;16,11,3,3,1,16,1,2,0"
SGC_TILE_CONFIGS="$C2" timeout 600 python tools/tile_bench.py cfg2 64x80 ring > gpurun_out/r02_tile3_cfg2.log 2>&1; cat gpurun_out/r02_tile3_cfg2.log
C4="27,11,3,3,1,16,1,1,0;27,11,3,3,1,16,1,2,0;20,20,3,3,1,16,1,1,0;27,22,3,3,1,16,1,1,0;40,30,3,3,1,16,1,1,0;20,15,3,3,1,16,1,2,0;27,30,3,3,1,16,1,1,0;27,30,3,3,1,16,1,1,2;27,30,3,3,1,8,1,1,0;40,59,2,2,1,16,1,1,0"
SGC_TILE_CONFIGS="$C4" timeout 600 python tools/tile_bench.py cfg4 59x80 ring > gpurun_out/r02_tile3_cfg4.log 2>&1; cat gpurun_out/r02_tile3_cfg4.log
