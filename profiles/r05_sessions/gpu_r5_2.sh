#!/bin/bash
# round 5, GPU session 2: early small-group cascades + adopted collapse variants: parity, then A/B against round 4's library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu \
   -k "exact_passes or long_reads_under or vs_oracle or vs_bruteforce or edge_cases or golden or one_call or low_complexity or random_cascade or random_collapse or reads_longer_than_255 or partitioned_collapse or overflow or linked_and_anchored or random_trimming or properties_at_scale or full_size_c3 or bench_single" 2>&1 | tail -15 ) > gpurun_out/r5_2_tests.txt 2>&1
tail -5 gpurun_out/r5_2_tests.txt
timeout 2400 python tools/ab_multi.py --rounds 3 r4=build_var/r4.so cur= early0=,MIRGE_EARLY_SMALL=0 walks0=,MIRGE_EXACT_WALKS=0 ride0=build_var/ride0.so > gpurun_out/r5_2_ab.txt 2>&1
tail -8 gpurun_out/r5_2_ab.txt
