#!/bin/bash
# session 28: k_trim's general instance with one run-time-parameterised search (no seven inlined variants), 32-row and 64-row heights
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "trim or adapter or linked or anchored or cutadapt" > gpurun_out/r5_28_tests.txt 2>&1
tail -3 gpurun_out/r5_28_tests.txt
MIRGE_TRIM_TALL=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "trim or adapter or linked or anchored" > gpurun_out/r5_28_tests_tall.txt 2>&1
tail -3 gpurun_out/r5_28_tests_tall.txt
for tall in 0 1; do
MIRGE_TRIM_TALL=$tall python bench.py --steps 3 --warmup 2 --cpu-baseline 0 --pmc 0 --cli-path 0 --read-sets 0 --two-in-flight 0 --min-seconds 0.2 2>gpurun_out/r5_28.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['fastq_trim_path']; print('tall=$tall', {k:f[k] for k in ('parse_trim_ms','parse_trim_general_instance_ms','reads_kept','reads_kept_general_instance')})" | tee -a gpurun_out/r5_28_trim_general.txt
done
