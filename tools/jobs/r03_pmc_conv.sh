#!/bin/bash
# SQ counters of the two convolution kernels after the round-3 addressing work: the halo kernel on the 90-GF layer (256 -> 256 at
# 40x40x16) and the tile kernel on the 1024 -> 1024 layer at 10x10x4.  Separate --pmc passes (no trace domains with them).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name, kernel substring, conv_one args...
  n=$1; k=$2; shift 2
  timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_${n}a -- python3 $R/tools/conv_one.py "$@" > /dev/null 2>&1; echo rc $?
  timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_${n}b -- python3 $R/tools/conv_one.py "$@" > /dev/null 2>&1; echo rc $?
  python3 - "$n" "$k" <<PY > $R/gpurun_out/r03_pmc_conv_${n}.json
import json, subprocess, sys
n, k = sys.argv[1], sys.argv[2]
out = {}
for part in "ab":
    out.update(json.loads(subprocess.run(["python3", "$R/tools/pmc_summary.py", f"/tmp/pmc_{n}{part}", k, "2"], capture_output=True, text=True).stdout))
print(json.dumps(out, indent=1))
PY
  rm -rf /tmp/pmc_${n}a /tmp/pmc_${n}b
}
run halo conv3d_halo 256 256 40 40 16
run tile conv3d_igemm 1024 1024 10 10 4
cd $R; cat gpurun_out/r03_pmc_conv_halo.json gpurun_out/r03_pmc_conv_tile.json
