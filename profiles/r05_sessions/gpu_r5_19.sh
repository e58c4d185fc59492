#!/bin/bash
# session 19: skewed segments in k_cascade_bulk (MIRGE_BULK_SKEW=1): parity first, then the interleaved A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
MIRGE_BULK_SKEW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "full_size_c3 or full_size_c4 or vs_oracle or one_call or exact_passes or random_cascade" > gpurun_out/r5_19_tests.txt 2>&1
tail -3 gpurun_out/r5_19_tests.txt
timeout 1200 python tools/ab_multi.py --rounds 4 cur= skew=,MIRGE_BULK_SKEW=1 > gpurun_out/r5_19_ab.txt 2>&1
tail -6 gpurun_out/r5_19_ab.txt
