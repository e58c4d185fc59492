#!/bin/bash
# A/B of the default build against a variant on the C5 workload (variant tally): bash tools/ab_c5.sh build_var/x.so [reads]
for so in "" "$1"; do
  MIRGE_NATIVE_SO=$so python bench.py --workload c5 --reads ${2:-20000000} --steps 10 --warmup 2 --cpu-baseline 0 --pmc 0 --cli-path 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('${so:-default}', d['value'], d['ms_per_step'], {n: round(k[n]['avg_ms'],4) for n in k if 'tally' in n})"
done
