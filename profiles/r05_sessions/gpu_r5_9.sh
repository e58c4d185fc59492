#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python tools/ab_multi.py --rounds 3 cur= w2_8=build_var/w2_8.so > gpurun_out/r5_9_ab.txt 2>&1
tail -4 gpurun_out/r5_9_ab.txt
