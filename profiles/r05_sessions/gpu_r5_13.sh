#!/bin/bash
# round 5, GPU session 13: what the driver runs at round end, on the final tree: smoke(), the whole GPU suite, the bench command
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 ) > gpurun_out/r5_13_smoke.txt 2>&1
cat gpurun_out/r5_13_smoke.txt
( timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r5_13_tests.txt 2>&1
tail -3 gpurun_out/r5_13_tests.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_13_bench_driver_cmd.json 2> gpurun_out/r5_13_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5_13_bench_driver_cmd.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["timing"]["step_ms_rank0"], d["parity_on_cpu_sample"])
PY
