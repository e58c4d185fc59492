#!/bin/bash
# session 35: level-1 bins of the partitioned collapse at C4's per-rank size (20 M reads) and C5's (50 M): 64 / 128 / 256
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
MIRGE_PART_NB1=128 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "partitioned_collapse or collapse_vs_oracle or full_size_c4" > gpurun_out/r5_35_tests.txt 2>&1; tail -2 gpurun_out/r5_35_tests.txt
timeout 1500 python tools/ab_multi.py --rounds 3 --bench-args "--workload c4 --reads 20000000" b64= b128=,MIRGE_PART_NB1=128 b256=,MIRGE_PART_NB1=256 > gpurun_out/r5_35_ab_c4.txt 2>&1
tail -4 gpurun_out/r5_35_ab_c4.txt
timeout 1500 python tools/ab_multi.py --rounds 2 --bench-args "--workload c5 --reads 50000000" b64= b128=,MIRGE_PART_NB1=128 b256=,MIRGE_PART_NB1=256 > gpurun_out/r5_35_ab_c5.txt 2>&1
tail -4 gpurun_out/r5_35_ab_c5.txt
