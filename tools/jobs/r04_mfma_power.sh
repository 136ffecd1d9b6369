#!/bin/bash
# the matrix pipe under the package power limit: back-to-back bf16 MFMAs on every SIMD, clocks and power sampled beside it
mkdir -p gpurun_out
[ -x tools/probe/mfma_power ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_power.hip -o tools/probe/mfma_power 2>/dev/null
smi() { for i in $(seq 1 $1); do /opt/rocm/bin/rocm-smi --showclocks --showpower --json 2>/dev/null | python3 -c "
import json,sys
try:
    c=json.load(sys.stdin)['card0']; print('   smi:', c.get('sclk clock speed:'), c.get('Current Socket Graphics Package Power (W)'), 'W')
except Exception as e: print('   smi: n/a')
"; sleep 0.4; done; }
for mode in ${MODES:-"1 5 2" "0 5 2" "1 5 1"}; do
  echo "== mfma_power $mode (random data?, seconds, waves per SIMD, operand order)"
  smi 12 & S=$!
  timeout 60 tools/probe/mfma_power $mode
  wait $S
done 2>&1 | tee gpurun_out/r04_mfma_power.txt
