#!/bin/bash
# occupancy of k_cascade_bulk: builds with amdgpu_waves_per_eu(7|8, 8) (build_var/waves7.so, waves8.so) at 7 / 8 workgroups per CU
# against the in-tree build (6), interleaved:   bash tools/ab_waves.sh
cd "$(dirname "$0")/.."
for r in 1 2; do
  for v in 6 7 8; do
    if [ $v = 6 ]; then unset MIRGE_NATIVE_SO; else export MIRGE_NATIVE_SO=$PWD/build_var/waves$v.so; fi
    MIRGE_WG_PER_CU=$v python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --pmc 0 --cli-path 0 | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('wg_per_cu $v', d['ms_per_step'], d['timing']['step_ms_rank0']['median'], {k: round(x['avg_ms'],3) for k,x in d['kernels'].items() if 'bulk' in k})"
  done
done
