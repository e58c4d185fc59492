#!/bin/bash
# session 33: the other configurations on the round's last tree (C2, C4's per-rank shape at 20 M reads, C5 at 50 M reads, the Zipf sample)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
A="--cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 --read-sets 0"
timeout 900 python bench.py --workload c2 $A > gpurun_out/r5_33_c2.json 2>/dev/null
timeout 900 python bench.py --workload c4 --reads 20000000 $A > gpurun_out/r5_33_c4.json 2>/dev/null
timeout 900 python bench.py --workload c5 --reads 50000000 $A > gpurun_out/r5_33_c5.json 2>/dev/null
timeout 900 python bench.py --pool 600000 $A > gpurun_out/r5_33_zipf.json 2>/dev/null
python - <<'PY'
import json
for f in ("c2","c4","c5","zipf"):
    try:
        d=json.loads(open(f"gpurun_out/r5_33_{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d.get("roofline_stages",{}).get("c5_tally",{}).get("ms"))
    except Exception as e: print(f, "failed", e)
PY
