#!/bin/bash
# session 47: k_part_dedup reserving its output per wave (no block scan, no broadcast barrier): parity, then A/B on C3 and the Zipf sample
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "collapse or partition or full_size_c3 or one_call or skew or sorted_order or baking" > gpurun_out/r5_47_tests.txt 2>&1; tail -2 gpurun_out/r5_47_tests.txt
timeout 1500 python tools/ab_multi.py --rounds 4 wave= block=build_var/block_reserve.so > gpurun_out/r5_47_ab_c3.txt 2>&1
tail -3 gpurun_out/r5_47_ab_c3.txt
timeout 1500 python tools/ab_multi.py --rounds 3 --bench-args "--pool 600000" wave= block=build_var/block_reserve.so > gpurun_out/r5_47_ab_zipf.txt 2>&1
tail -3 gpurun_out/r5_47_ab_zipf.txt
