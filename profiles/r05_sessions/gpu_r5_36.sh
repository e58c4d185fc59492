#!/bin/bash
# session 36: the adaptive level-1 bin count at the sizes where it changes (20 M and 50 M reads), and the partition tests
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "full_size or partitioned_collapse or collapse_vs_oracle or one_call" > gpurun_out/r5_36_tests.txt 2>&1; tail -2 gpurun_out/r5_36_tests.txt
