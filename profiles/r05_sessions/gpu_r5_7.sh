#!/bin/bash
# round 5, GPU session 7: the small groups' join rows on aux: A/B; every config's bench line with the final build
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu \
   -k "one_call or golden or join or bench or two_ranks or properties_at_scale or full_size or random_count_join or multi_sample" 2>&1 | tail -8 ) > gpurun_out/r5_7_tests.txt 2>&1
tail -3 gpurun_out/r5_7_tests.txt
timeout 1500 python tools/ab_multi.py --rounds 3 cur= joinmain=,MIRGE_JOIN_SMALL_ON_AUX=0 ins2=,MIRGE_INSERT_WG_PER_CU=2 r4=build_var/r4.so > gpurun_out/r5_7_ab.txt 2>&1
tail -6 gpurun_out/r5_7_ab.txt
for w in "c2" "c4 --reads 20000000" "c5 --reads 50000000"; do
  timeout 1500 python bench.py --workload $w --cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 > gpurun_out/r05_bench_$(echo $w | cut -d' ' -f1).json 2>/dev/null
done
timeout 900 python bench.py --pool 600000 --cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 --read-sets 0 > gpurun_out/r05_bench_zipf_pool.json 2>/dev/null
MIRGE_NATIVE_SO=build_var/r4.so timeout 900 python bench.py --workload c4 --reads 20000000 --cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 --read-sets 0 > gpurun_out/r05_bench_c4_r4lib.json 2>/dev/null
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_7_bench_driver_cmd.json 2> gpurun_out/r5_7_bench_driver_cmd.err
python - <<'PY'
import json
for n in ("r05_bench_c2", "r05_bench_c4", "r05_bench_c4_r4lib", "r05_bench_c5", "r05_bench_zipf_pool", "r5_7_bench_driver_cmd"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{n}.json") if l.startswith("{")][-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["workload"][:50], {k: round(v["avg_ms"], 4) for k, v in d["kernels"].items() if v["avg_ms"] > 0.08})
    except Exception as e:
        print(n, "failed", e)
PY
