#!/bin/bash
# session 15: the borrowed-library rule (whole-read tables only through the owning context)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "borrowed or whole_read or exact_passes or hooks or golden_dropin" > gpurun_out/r5_15_tests.txt 2>&1
tail -5 gpurun_out/r5_15_tests.txt
