#!/bin/bash
# session 53: what the driver runs at round end, on the tree with k_part_dedup's sharded output: smoke, the whole GPU suite, the bench command
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_53_smoke.txt 2>&1; tail -1 gpurun_out/r5_53_smoke.txt
timeout 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_53_tests.txt 2>&1
tail -6 gpurun_out/r5_53_tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_53_bench_driver_cmd.json 2> gpurun_out/r5_53_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_53_bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v.get('frac') for k,v in d['roofline_stages'].items()}, {k:(v.get('ms_per_step')) for k,v in d['read_sets'].items() if isinstance(v,dict)})
PY
