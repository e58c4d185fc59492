#!/bin/bash
# session 39: the full-size C3 test with its new isomiR-typing part; the shortened test selections
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --durations=5 -k "full_size_c3 or general_instance_at_both or staged_cascade_for_every_group" > gpurun_out/r5_39_tests.txt 2>&1
tail -10 gpurun_out/r5_39_tests.txt
