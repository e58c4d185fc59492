#!/bin/bash
# round 5, GPU session 1: parity of the exact-walk cascade, then an interleaved A/B of the build variants
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu \
   -k "exact_passes or long_reads_under or vs_oracle or vs_bruteforce or edge_cases or golden_cascade or one_call or low_complexity or random_cascade or reads_longer_than_255" 2>&1 | tail -15 ) > gpurun_out/r5_1_tests.txt 2>&1
cat gpurun_out/r5_1_tests.txt | tail -5
timeout 2400 python tools/ab_multi.py --rounds 2 r4=build_var/r4.so cur= walks0=,MIRGE_EXACT_WALKS=0 ride0=build_var/ride0.so launder=build_var/launder.so \
   dedup512=build_var/dedup512.so dedup256=build_var/dedup256.so agg2=build_var/agg2.so agg4=build_var/agg4.so > gpurun_out/r5_1_ab.txt 2>&1
tail -12 gpurun_out/r5_1_ab.txt
