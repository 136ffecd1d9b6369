#!/bin/bash
export SGC_HALO_VARIANTS=0
echo "== 32x32x16 (product)"; python tools/halo_bench.py 2>&1 | grep -v amdgpu.ids
echo "== 16x16x32 (diag, garbage results)"; SGC_DIAG_LIB=tools/diag/libsgc_mfma16.so python tools/halo_bench.py 2>&1 | grep -v amdgpu.ids
echo "== 32x32x16 again"; python tools/halo_bench.py 2>&1 | grep -v amdgpu.ids
