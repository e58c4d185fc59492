#!/bin/bash
# round 5, GPU session 5: the sharded CLI's rank-0 tail at C4's size (8 ranks share the one GPU), then a smaller run with the
# one-process comparison (byte-identical files)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
nproc; free -g | head -2; df -h /tmp | tail -1
timeout 1500 python tools/sharded_c4.py --ranks 8 --reads 2000000 --one-process 1 --out gpurun_out/r05_sharded_8x2M.txt > gpurun_out/r5_5_small.log 2>&1
tail -25 gpurun_out/r5_5_small.log
timeout 4500 python tools/sharded_c4.py --ranks 8 --reads 20000000 --out gpurun_out/r05_sharded_c4.txt > gpurun_out/r5_5_c4.log 2>&1
tail -25 gpurun_out/r5_5_c4.log
