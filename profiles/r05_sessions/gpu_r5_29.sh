#!/bin/bash
# session 29: what the driver runs at round end, on the current tree: smoke, the whole GPU suite (with durations), the bench command
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_29_smoke.txt 2>&1; tail -1 gpurun_out/r5_29_smoke.txt
timeout 1500 python -m pytest tests/ -q -m gpu --durations=12 > gpurun_out/r5_29_tests.txt 2>&1
tail -18 gpurun_out/r5_29_tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_29_bench_driver_cmd.json 2> gpurun_out/r5_29_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_29_bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v.get('frac') for k,v in d['roofline_stages'].items()}, d['gff_typing'].get('k_isotype_ms'), d['fastq_trim_path'].get('parse_trim_general_instance_ms'))
PY
