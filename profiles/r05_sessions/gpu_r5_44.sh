#!/bin/bash
# session 44: what the driver runs at round end, on the round's last tree: smoke, the whole GPU suite, the bench command
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_44_smoke.txt 2>&1; tail -1 gpurun_out/r5_44_smoke.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=6 > gpurun_out/r5_44_tests.txt 2>&1
tail -12 gpurun_out/r5_44_tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_44_bench_driver_cmd.json 2> gpurun_out/r5_44_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_44_bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v.get('frac') for k,v in d['roofline_stages'].items()}, d['gff_typing'].get('k_isotype_ms'), d['cli_path'].get('gff_libraries_resident',{}).get('wall_s'))
PY
