#!/bin/bash
# session 26: a sample in which 60 % of the reads are 32-50 nt (the two-word group is the bulk group): selected vs indexed words
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python tools/ab_multi.py --rounds 3 --bench-args "--long-frac 0.6" cur= wide_index=build_var/wide_index.so > gpurun_out/r5_26_ab.txt 2>&1
tail -4 gpurun_out/r5_26_ab.txt
