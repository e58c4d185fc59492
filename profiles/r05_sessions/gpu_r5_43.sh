#!/bin/bash
# session 43: k_part_dedup with 2 (default) / 1 / 4 records per thread and trip, their loads issued together: parity, then A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "collapse or partition or full_size_c3 or one_call" > gpurun_out/r5_43_tests.txt 2>&1; tail -2 gpurun_out/r5_43_tests.txt
timeout 1500 python tools/ab_multi.py --rounds 4 u2= u1=build_var/dedup_u1.so u4=build_var/dedup_u4.so > gpurun_out/r5_43_ab_c3.txt 2>&1
tail -4 gpurun_out/r5_43_ab_c3.txt
