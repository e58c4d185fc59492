#!/bin/bash
# session 52: k_part_dedup's output through eight cursors into a staging area + k_part_compact, against the one global cursor (MIRGE_DEDUP_SHARDED=0)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "collapse or partition or one_call or full_size_c3 or skew or sorted_order" > gpurun_out/r5_52_tests.txt 2>&1; tail -2 gpurun_out/r5_52_tests.txt
timeout 700 python tools/ab_multi.py --rounds 3 sharded= cursor=,MIRGE_DEDUP_SHARDED=0 > gpurun_out/r5_52_ab_c3.txt 2>&1
tail -3 gpurun_out/r5_52_ab_c3.txt
timeout 400 python tools/ab_multi.py --rounds 2 --bench-args "--pool 600000" sharded= cursor=,MIRGE_DEDUP_SHARDED=0 > gpurun_out/r5_52_ab_zipf.txt 2>&1
tail -3 gpurun_out/r5_52_ab_zipf.txt
