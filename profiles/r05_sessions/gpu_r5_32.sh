#!/bin/bash
# session 32: the round's rocprofv3 evidence again on the final tree (kernel trace + stats, PMC passes, memory counters, timeline)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
BENCH_ARGS="--cli-path 0" bash profiles/collect.sh r05 > gpurun_out/r5_32_collect.log 2>&1
bash profiles/collect_mem.sh r05 > gpurun_out/r5_32_collect_mem.log 2>&1
python profiles/summarize.py r05 > gpurun_out/r5_32_summarize.log 2>&1
python profiles/timeline.py r05 > gpurun_out/r05_timeline.txt 2>&1
python profiles/mem_summary.py gpurun_out/prof_r05_mem > gpurun_out/r05_mem_counters.txt 2>&1
cp gpurun_out/prof_r05/bench_kt.json gpurun_out/r05_bench_under_rocprof.json
cp profiles/r05_kernel_stats.csv profiles/r05_pmc.csv gpurun_out/ 2>/dev/null
head -12 gpurun_out/r05_kernel_stats.csv | cut -c1-160
grep -n "k_isotype\|k_trim" gpurun_out/r05_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/prof_r05 gpurun_out/prof_r05_mem
