#!/bin/bash
# round 5, GPU session 8: the final tree: whole GPU suite, the driver's bench command, the rocprofv3 evidence again
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r5_8_tests.txt 2>&1
tail -3 gpurun_out/r5_8_tests.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_8_bench_driver_cmd.json 2> gpurun_out/r5_8_bench_driver_cmd.err
timeout 900 python tools/ab_multi.py --rounds 3 cur= r4=build_var/r4.so > gpurun_out/r5_8_ab.txt 2>&1
tail -3 gpurun_out/r5_8_ab.txt
BENCH_ARGS="--cli-path 0" bash profiles/collect.sh r05 > gpurun_out/r5_8_collect.log 2>&1
bash profiles/collect_mem.sh r05 > gpurun_out/r5_8_collect_mem.log 2>&1
python profiles/summarize.py r05 > gpurun_out/r5_8_summarize.log 2>&1
python profiles/timeline.py r05 > gpurun_out/r05_timeline.txt 2>&1
python profiles/mem_summary.py gpurun_out/prof_r05_mem > gpurun_out/r05_mem_counters.txt 2>&1
cp gpurun_out/prof_r05/bench_kt.json gpurun_out/r05_bench_under_rocprof.json
cp profiles/r05_kernel_stats.csv profiles/r05_pmc.csv gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_r05 gpurun_out/prof_r05_mem
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5_8_bench_driver_cmd.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic"), d["timing"]["step_ms_rank0"])
print({k: (v["ms"], v["frac"]) for k, v in d["roofline_stages"].items()})
print({k: (v["ms_per_step"], v["U_over_N"]) for k, v in d["read_sets"].items()})
PY
tail -14 gpurun_out/r05_timeline.txt
