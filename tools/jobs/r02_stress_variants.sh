#!/bin/bash
# the whole GPU suite with kernel variants forced: the LDS-DMA ring on every halo layer, the staged form everywhere
SGC_TUNE="halo_ring=2" timeout 1500 python -m pytest tests -x -q -m gpu -k "not ring_form" 2>&1 | tail -3
SGC_TUNE="halo_ring=0" timeout 1500 python -m pytest tests/test_gpu_modules.py tests/test_gpu_conv3d.py -x -q -k "not ring_form" 2>&1 | tail -3
