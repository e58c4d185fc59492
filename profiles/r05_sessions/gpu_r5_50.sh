#!/bin/bash
# session 50: the look-back with pauses between looks: A/B against the global cursor
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "collapse_vs_oracle or partitioned_collapse or one_call or full_size_c3" > gpurun_out/r5_50_tests.txt 2>&1; tail -2 gpurun_out/r5_50_tests.txt
timeout 900 python tools/ab_multi.py --rounds 3 lookback= cursor=,MIRGE_DEDUP_LOOKBACK=0 > gpurun_out/r5_50_ab_c3.txt 2>&1
tail -3 gpurun_out/r5_50_ab_c3.txt
timeout 600 python tools/ab_multi.py --rounds 2 --bench-args "--pool 600000" lookback= cursor=,MIRGE_DEDUP_LOOKBACK=0 > gpurun_out/r5_50_ab_zipf.txt 2>&1
tail -3 gpurun_out/r5_50_ab_zipf.txt
