# wall time per step vs the sum of the bulk group's kernel times, repeated runs on one box
for i in 1 2 3 4 5 6; do
  if [ $i -gt 3 ]; then export MIRGE_WG_PER_CU=6; fi
  python bench.py --steps 30 --warmup 3 --pmc 0 --cpu-baseline 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
main=sum(v['avg_ms'] for n,v in k.items() if n.endswith('.w1'))
print('run $i env=${MIRGE_WG_PER_CU:-none}', 'ms/step', d['ms_per_step'], 'bulk-group kernels', round(main,3), 'join', round(sum(v['avg_ms']*v['launches'] for n,v in k.items() if n=='k_join')/max(1,k['k_part_agg.w1']['launches']),3))"
done
