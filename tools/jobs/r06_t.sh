#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_conv3d.py -q -m gpu -x 2>&1 | tail -8
