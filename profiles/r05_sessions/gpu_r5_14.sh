#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bench_single or bench_two" 2>&1 | tail -4 ) > gpurun_out/r5_14_tests.txt 2>&1
tail -3 gpurun_out/r5_14_tests.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_14_bench_driver_cmd.json 2> gpurun_out/r5_14_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5_14_bench_driver_cmd.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["collapsed_reads_annotation_on_host"], d["host_buffer_path"]["ms"])
PY
