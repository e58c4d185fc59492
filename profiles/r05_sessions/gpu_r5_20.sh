#!/bin/bash
# session 20: the gff_typing leg of the bench line; the bowtie command lines as argument lists (a2i, crosscheck harness, backend)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bench_single or a2i_report or crosscheck_harness or backend_bowtie or shim" > gpurun_out/r5_20_tests.txt 2>&1
tail -5 gpurun_out/r5_20_tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_20_bench_driver_cmd.json 2> gpurun_out/r5_20_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_20_bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('gff_typing'))
PY
