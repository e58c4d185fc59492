#!/bin/bash
# session 55: the full-size tests and the fuzzers' collapse part on the round's last tree
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "full_size or random_collapse or umi_route_at_scale or properties_at_scale" > gpurun_out/r5_55_tests.txt 2>&1; tail -2 gpurun_out/r5_55_tests.txt
