#!/bin/bash
# session 22: k_isotype's two forms on the device (equal records), the -gff tests with MIRGE_ISO_FAST=0, the driver's command
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "isotype or gff or bench_single" > gpurun_out/r5_22_tests.txt 2>&1
tail -3 gpurun_out/r5_22_tests.txt
MIRGE_ISO_FAST=0 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gff" > gpurun_out/r5_22_tests_array_form.txt 2>&1
tail -3 gpurun_out/r5_22_tests_array_form.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_22_bench_driver_cmd.json 2> gpurun_out/r5_22_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_22_bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('gff_typing'))
PY
