#!/bin/bash
echo "== product"; timeout 120 python tools/linear_bench.py 2>&1 | grep -v amdgpu.ids
for v in ig_noepi ig_noloads ig_nomfma ig_noloads_noepi; do echo "== $v"; SGC_DIAG_LIB=tools/diag/libsgc_$v.so timeout 120 python tools/linear_bench.py 2>&1 | grep -v amdgpu.ids; done
