#!/bin/bash
export SGC_HALO_VARIANTS=0
echo "== product"; python tools/halo_bench.py 2>&1 | grep -v amdgpu.ids | head -2
for v in nobar nob nobna; do
echo "== diag $v"; SGC_DIAG_LIB=tools/diag/libsgc_$v.so python tools/halo_bench.py 2>&1 | grep -v amdgpu.ids | head -2
done
