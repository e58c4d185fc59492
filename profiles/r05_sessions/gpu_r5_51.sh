#!/bin/bash
# session 51: half the buckets (twice the reads each, the 4096-slot table): half the reservations on the output cursor
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
MIRGE_PART_HALF_B=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "collapse_vs_oracle or partitioned_collapse or one_call" > gpurun_out/r5_51_tests.txt 2>&1; tail -2 gpurun_out/r5_51_tests.txt
timeout 900 python tools/ab_multi.py --rounds 3 b8192= b4096=,MIRGE_PART_HALF_B=1 > gpurun_out/r5_51_ab_c3.txt 2>&1
tail -3 gpurun_out/r5_51_ab_c3.txt
