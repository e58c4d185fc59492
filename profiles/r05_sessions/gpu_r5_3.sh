#!/bin/bash
# round 5, GPU session 3: the whole GPU suite on the new tree, A/B of the bucketized agg cache, the driver's bench command
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r5_3_tests.txt 2>&1
tail -6 gpurun_out/r5_3_tests.txt
timeout 1200 python tools/ab_multi.py --rounds 3 cur= bucket0=build_var/bucket0.so r4=build_var/r4.so > gpurun_out/r5_3_ab.txt 2>&1
tail -5 gpurun_out/r5_3_ab.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_3_bench_driver_cmd.json 2> gpurun_out/r5_3_bench_driver_cmd.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5_3_bench_driver_cmd.json") if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "collapsed_reads_per_s_M", "cascade_walks")})
print(json.dumps(d.get("roofline_stages"), indent=0)[:1500])
print(json.dumps(d.get("read_sets"), indent=0)[:1500])
print(d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("cpu_baseline", {}).get("value"), d.get("parity_on_cpu_sample"))
PY
