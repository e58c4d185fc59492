#!/bin/bash
# session 31: -gff through the CLI's route with the writer indexing the run's table of unique reads; the golden -gff files
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gff" > gpurun_out/r5_31_tests.txt 2>&1
tail -3 gpurun_out/r5_31_tests.txt
bash tools/gpu_r5_30.sh
