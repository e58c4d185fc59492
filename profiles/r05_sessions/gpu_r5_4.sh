#!/bin/bash
# round 5, GPU session 4: the whole GPU suite, then the rocprofv3 evidence of the round (kernel trace + stats, PMC passes, memory counters)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r5_4_tests.txt 2>&1
tail -4 gpurun_out/r5_4_tests.txt
BENCH_ARGS="--cli-path 0" bash profiles/collect.sh r05 > gpurun_out/r5_4_collect.log 2>&1
bash profiles/collect_mem.sh r05 > gpurun_out/r5_4_collect_mem.log 2>&1
python profiles/summarize.py r05 > gpurun_out/r5_4_summarize.log 2>&1
python profiles/timeline.py r05 > gpurun_out/r05_timeline.txt 2>&1
python profiles/mem_summary.py gpurun_out/prof_r05_mem > gpurun_out/r05_mem_counters.txt 2>&1
cp gpurun_out/prof_r05/bench_kt.json gpurun_out/r05_bench_under_rocprof.json
cp profiles/r05_kernel_stats.csv profiles/r05_pmc.csv gpurun_out/ 2>/dev/null
head -30 gpurun_out/r05_timeline.txt
# keep the merged-back scratch small: the raw traces stay on the box
rm -rf gpurun_out/prof_r05 gpurun_out/prof_r05_mem
