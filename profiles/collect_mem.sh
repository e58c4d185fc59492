#!/bin/bash
# Memory-pipeline counters of the step's kernels (texture addresser, vector L1, address translation), separate --pmc
# passes as profiles/collect.sh:   bash profiles/collect_mem.sh r02   -> gpurun_out/prof_<tag>_mem/pmc*/
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_mem
mkdir -p "$OUT"
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
"$PY" "$REPO/__graft_entry__.py" || exit 1
cd /tmp && export TMPDIR=/tmp
BENCH="$PY $REPO/bench.py --steps 5 --warmup 2 --cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 --read-sets 0 --min-seconds 0.1 --spinup 0.1"
i=0
for CTRS in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
            "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/bench_pmc$i.json" 2> "$OUT/pmc$i.err"
done
ls -R "$OUT" | head -20
