#!/bin/bash
# round 5, GPU session 11: the one-level partition (k_part_dedup_fat): parity, A/B against the two-level partition and round 4
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 2700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu \
   -k "one_level or partitioned_collapse or overflow or one_call or random_collapse or full_size_c3 or full_size_c2 or properties_at_scale or collapse_vs_oracle or multi_sample or bench_single or golden_cascade" 2>&1 | tail -12 ) > gpurun_out/r5_11_tests.txt 2>&1
tail -4 gpurun_out/r5_11_tests.txt
timeout 1500 python tools/ab_multi.py --rounds 3 cur= fat0=,MIRGE_FAT_DEDUP=0 r4=build_var/r4.so > gpurun_out/r5_11_ab.txt 2>&1
tail -5 gpurun_out/r5_11_ab.txt
for m in 1 2; do
  MIRGE_FAT_DEDUP=$m timeout 900 python bench.py --workload c4 --reads 20000000 --cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 --read-sets 0 > gpurun_out/r5_11_c4_fat$m.json 2>/dev/null
done
MIRGE_FAT_DEDUP=2 timeout 900 python bench.py --workload c5 --reads 50000000 --cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 --read-sets 0 > gpurun_out/r5_11_c5_fat2.json 2>/dev/null
python - <<'PY'
import json
for n in ("r5_11_c4_fat1", "r5_11_c4_fat2", "r5_11_c5_fat2"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{n}.json") if l.startswith("{")][-1])
        print(n, d["value"], d["ms_per_step"], {k: round(v["avg_ms"], 4) for k, v in d["kernels"].items() if k.startswith("k_part")})
    except Exception as e:
        print(n, "failed", e)
PY
