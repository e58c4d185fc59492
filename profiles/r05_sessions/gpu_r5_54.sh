#!/bin/bash
# session 54 (the round's last minutes of GPU): the adaptive choice of k_part_dedup's sharded output -- collapse tests, the forced-shards test, the bench command
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "collapse_vs_oracle or partitioned_collapse or one_call or sharded_output" > gpurun_out/r5_54_tests.txt 2>&1; tail -2 gpurun_out/r5_54_tests.txt
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --cli-path 0 --cpu-baseline 0 --two-in-flight 0 > gpurun_out/r5_54_bench.json 2> gpurun_out/r5_54_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_54_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v.get('frac') for k,v in d['roofline_stages'].items()}, {k:(v.get('ms_per_step')) for k,v in d['read_sets'].items() if isinstance(v,dict)}, [k for k in d['kernels'] if 'compact' in k])
PY
