#!/bin/bash
# LDS counters of the collapse kernels on the default and on the Zipf-pool sample:  bash profiles/collect_lds.sh r02
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_lds
mkdir -p "$OUT"
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
cd /tmp && export TMPDIR=/tmp
for W in default pool; do
  EXTRA=""; [ $W = pool ] && EXTRA="--pool 600000"
  BENCH="$PY $REPO/bench.py --steps 5 --warmup 2 --cpu-baseline 0 --pmc 0 --cli-path 0 $EXTRA"
  i=0
  for CTRS in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
              "SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
              "SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/$W$i" -- $BENCH > "$OUT/bench_$W$i.json" 2> "$OUT/$W$i.err"
  done
done
