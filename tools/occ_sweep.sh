# usage: bash tools/occ_sweep.sh <libmirge_native variant .so> "<list of workgroups per CU>"
export MIRGE_NATIVE_SO=$1
for w in $2; do
  MIRGE_WG_PER_CU=$w python bench.py --steps 20 --warmup 2 --pmc 0 --cpu-baseline 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', d['ms_per_step'], {k[6:]: round(v['avg_ms'],3) for k,v in d['kernels'].items() if k.endswith('.w1') and 'pass' in k})"
done
