#!/bin/bash
# SQ / LDS / TA counters of the tiled gather: tools/tile_bench.py with ONE configuration per run
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; WHICH=$2; HW=$3; export SGC_TILE_CONFIGS="$4"
run() { timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$N -- python3 $R/tools/tile_bench.py $WHICH $HW ring > $R/gpurun_out/pmc_${TAG}_$N.log 2>&1; echo rc $?; }
N=1; run SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_BUSY_CYCLES
N=2; run GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
N=3; run TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR
cd $R
for n in 1 2 3; do python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$n dfa3d_fwd_tile_kernel 2 > gpurun_out/pmc_${TAG}_$n.json; done
python - <<PY
import json
d = {}
for n in (1, 2, 3):
    d.update(json.load(open("gpurun_out/pmc_${TAG}_%d.json" % n)))
json.dump(d, open("gpurun_out/pmc_${TAG}.json", "w"), indent=1)
print(json.dumps(d))
PY
rm -rf gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2 gpurun_out/pmc_${TAG}_3
