#!/bin/bash
# HBM-side traffic of the tiled gather (finest level of config 2) under the two workgroup -> XCD deals:
# tile_xcd=0 (camera, bin, head) order = head h on XCD h; tile_xcd=1 camera n on XCD n % 8.  One rocprofv3 --pmc pass per counter.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for xcd in 0 1; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_x
    SGC_TUNE=tile_xcd=$xcd timeout 400 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_x -- python3 $R/bench.py --graph tail --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-strict-fp32 --sustain 0 > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py /tmp/pmc_x dfa3d_fwd_tile_kernel 2 > $R/gpurun_out/r03_pmc_gather_xcd${xcd}_${ctr}.json
    echo xcd $xcd $ctr: $(python3 -c "import json;d=json.load(open('$R/gpurun_out/r03_pmc_gather_xcd${xcd}_${ctr}.json'));print(d)")
  done
done
