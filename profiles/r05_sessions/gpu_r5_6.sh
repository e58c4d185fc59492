#!/bin/bash
# round 5, GPU session 6: small groups' collapse chains side by side + wider insert grid: parity, A/B, timeline; sharded C4 again; other configs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu \
   -k "collapse or one_call or overflow or weighted or multi_sample or two_ranks or bench or backend_bowtie or documented_linked or baking or properties_at_scale or full_size_c3 or full_size_c4 or golden or random_cascade or exact_passes or edge_cases" 2>&1 | tail -15 ) > gpurun_out/r5_6_tests.txt 2>&1
tail -4 gpurun_out/r5_6_tests.txt
timeout 1500 python tools/ab_multi.py --rounds 3 cur= spread0=,MIRGE_SPREAD_SMALL=0 ins2=,MIRGE_INSERT_WG_PER_CU=2 both0=,MIRGE_SPREAD_SMALL=0,MIRGE_INSERT_WG_PER_CU=2 r4=build_var/r4.so > gpurun_out/r5_6_ab.txt 2>&1
tail -7 gpurun_out/r5_6_ab.txt
KT_ONLY=1 BENCH_ARGS="--cli-path 0" bash profiles/collect.sh r05b > gpurun_out/r5_6_collect.log 2>&1
python profiles/timeline.py r05b > gpurun_out/r05b_timeline.txt 2>&1
rm -rf gpurun_out/prof_r05b
tail -22 gpurun_out/r05b_timeline.txt
timeout 3000 python tools/sharded_c4.py --ranks 8 --reads 20000000 --out gpurun_out/r05_sharded_c4_b.txt > gpurun_out/r5_6_c4.log 2>&1
tail -22 gpurun_out/r5_6_c4.log
for w in "c2" "c4 --reads 20000000" "c5 --reads 50000000"; do
  timeout 1500 python bench.py --workload $w --cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 > gpurun_out/r05_bench_$(echo $w | cut -d' ' -f1).json 2>/dev/null
done
timeout 900 python bench.py --pool 600000 --cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 --read-sets 0 > gpurun_out/r05_bench_zipf_pool.json 2>/dev/null
python - <<'PY'
import json
for n in ("c2", "c4", "c5", "zipf_pool"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r05_bench_{n}.json") if l.startswith("{")][-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["workload"][:60])
    except Exception as e:
        print(n, "failed", e)
PY
