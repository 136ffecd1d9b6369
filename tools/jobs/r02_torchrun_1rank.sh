#!/bin/bash
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-strict-fp32 --sustain 0 2>gpurun_out/dist1.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['n_gpus'], d['value'], d['scaling'], d['self_check']['mismatching'])"
tail -3 gpurun_out/dist1.err
