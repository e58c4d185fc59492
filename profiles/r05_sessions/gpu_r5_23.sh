#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "isotype" > gpurun_out/r5_23_tests.txt 2>&1
tail -12 gpurun_out/r5_23_tests.txt
