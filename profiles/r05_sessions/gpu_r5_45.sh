#!/bin/bash
# session 45: k_part_dedup merging equal keys inside a wave before the LDS table (2 rounds = default, 1, 0): parity, then A/B on C3 and the Zipf sample
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "collapse or partition or full_size_c3 or one_call or skew" > gpurun_out/r5_45_tests.txt 2>&1; tail -2 gpurun_out/r5_45_tests.txt
timeout 1500 python tools/ab_multi.py --rounds 4 m2= m0=build_var/merge0.so m1=build_var/merge1.so > gpurun_out/r5_45_ab_c3.txt 2>&1
tail -4 gpurun_out/r5_45_ab_c3.txt
timeout 1500 python tools/ab_multi.py --rounds 3 --bench-args "--pool 600000" m2= m0=build_var/merge0.so > gpurun_out/r5_45_ab_zipf.txt 2>&1
tail -3 gpurun_out/r5_45_ab_zipf.txt
