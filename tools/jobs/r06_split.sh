#!/bin/bash
# tile kernel: splits of whole K steps against tap groups (tools/split_free_ab.py)
mkdir -p gpurun_out
python3 tools/split_free_ab.py 8,32 targets=128,192,256,384,512,768,1024 > gpurun_out/r06_split_free_targets.txt 2>&1
echo "rc $?"; grep -v amdgpu.ids gpurun_out/r06_split_free_targets.txt
