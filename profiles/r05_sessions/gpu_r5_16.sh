#!/bin/bash
# session 16: the whole GPU suite with per-test durations (what the driver's round-end run has to fit)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -q -m gpu --durations=60 > gpurun_out/r5_16_tests.txt 2>&1
tail -70 gpurun_out/r5_16_tests.txt
