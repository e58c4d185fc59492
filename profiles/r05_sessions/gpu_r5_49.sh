#!/bin/bash
# session 49: the look-back with relaxed accesses: parity, then A/B against the global cursor
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "collapse or partition or full_size or one_call or skew or sorted_order or baking or properties_at_scale" > gpurun_out/r5_49_tests.txt 2>&1; tail -2 gpurun_out/r5_49_tests.txt
timeout 900 python tools/ab_multi.py --rounds 4 lookback= cursor=,MIRGE_DEDUP_LOOKBACK=0 > gpurun_out/r5_49_ab_c3.txt 2>&1
tail -3 gpurun_out/r5_49_ab_c3.txt
timeout 600 python tools/ab_multi.py --rounds 2 --bench-args "--pool 600000" lookback= cursor=,MIRGE_DEDUP_LOOKBACK=0 > gpurun_out/r5_49_ab_zipf.txt 2>&1
tail -3 gpurun_out/r5_49_ab_zipf.txt
