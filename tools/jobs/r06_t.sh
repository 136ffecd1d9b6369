#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_modules.py -q -m gpu -x -k "test_full_size_config2_scene_against_the_oracle" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_conv3d.py -q -m gpu -x -k "train_weight_planes" 2>&1 | tail -3
