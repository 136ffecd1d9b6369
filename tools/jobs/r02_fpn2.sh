#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_modules.py -x -q -k "fpn_on_hip or channels_last" 2>&1 | tail -8
