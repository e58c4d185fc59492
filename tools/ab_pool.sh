#!/bin/bash
# A/B of the default build against a variant on the Zipf-pool sample (skewed duplicates): bash tools/ab_pool.sh build_var/x.so
for so in "" "$1"; do
  MIRGE_NATIVE_SO=$so python bench.py --pool 600000 --steps 30 --warmup 3 --cpu-baseline 0 --pmc 0 --cli-path 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('${so:-default}', d['value'], d['ms_per_step'], {n: k[n]['avg_ms'] for n in k if 'k_part' in n})"
done
