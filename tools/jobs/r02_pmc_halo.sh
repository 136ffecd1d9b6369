#!/bin/bash
# SQ counters + durations of the 256 -> 256 halo convolution at 40x40x16, register-staged weights (halo_ring 0) vs LDS-DMA
# ring (halo_ring 2): separate --pmc passes, a kernel-trace pass for the durations
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 2; do
export SGC_TUNE="halo_ring=$v"
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_h${v}a -- python3 $R/tools/conv_one.py 256 256 40 40 16 > /dev/null 2>&1; echo rc $?
timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pmc_h${v}b -- python3 $R/tools/conv_one.py 256 256 40 40 16 > /dev/null 2>&1; echo rc $?
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_h${v}t -- python3 $R/tools/conv_one.py 256 256 40 40 16 3 1 30 > /dev/null 2>&1; echo rc $?
python3 $R/tools/pmc_summary.py /tmp/pmc_h${v}a conv3d_halo 2 > $R/gpurun_out/r02_pmc_halo_ring${v}_a.json
python3 $R/tools/pmc_summary.py /tmp/pmc_h${v}b conv3d_halo 2 > $R/gpurun_out/r02_pmc_halo_ring${v}_b.json
f=$(find /tmp/pmc_h${v}t -name "*kernel_stats.csv" | head -1); grep conv3d_halo $f | cut -c1-200 > $R/gpurun_out/r02_pmc_halo_ring${v}_t.csv
done
cd $R; cat gpurun_out/r02_pmc_halo_ring*_t.csv; cat gpurun_out/r02_pmc_halo_ring0_a.json gpurun_out/r02_pmc_halo_ring2_a.json | head -40
