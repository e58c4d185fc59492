#!/bin/bash
# session 17: RCCL with one rank through bench.py's N > 1 route; the alternate-path runs at five fuzz seeds
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --durations=8 -k "rccl_with_one_rank or bench_two_ranks or staged_cascade_for_every_group" > gpurun_out/r5_17_tests.txt 2>&1
tail -25 gpurun_out/r5_17_tests.txt; cat gpurun_out/rccl_one_rank.json
