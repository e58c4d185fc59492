#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python tools/ab_multi.py --rounds 3 cur= b1_32=build_var/b1_32.so b1_128=build_var/b1_128.so > gpurun_out/r5_12_ab.txt 2>&1
tail -5 gpurun_out/r5_12_ab.txt
