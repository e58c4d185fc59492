#!/bin/bash
# session 48: k_part_dedup's output offsets by a decoupled look-back over the buckets instead of one global cursor: parity, then A/B (MIRGE_DEDUP_LOOKBACK=0 = the cursor)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "collapse or partition or full_size or one_call or skew or sorted_order or baking or properties_at_scale" > gpurun_out/r5_48_tests.txt 2>&1; tail -2 gpurun_out/r5_48_tests.txt
timeout 900 python tools/ab_multi.py --rounds 4 lookback= cursor=,MIRGE_DEDUP_LOOKBACK=0 > gpurun_out/r5_48_ab_c3.txt 2>&1
tail -3 gpurun_out/r5_48_ab_c3.txt
timeout 900 python tools/ab_multi.py --rounds 3 --bench-args "--pool 600000" lookback= cursor=,MIRGE_DEDUP_LOOKBACK=0 > gpurun_out/r5_48_ab_zipf.txt 2>&1
tail -3 gpurun_out/r5_48_ab_zipf.txt
