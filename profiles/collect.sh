#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r01
# 1) --kernel-trace --stats of the default bench command (no PMC), 2..n) separate --pmc passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with sys/hip traces).
# Raw output goes to gpurun_out/ (scratch); profiles/summarize.py writes the committed summaries.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
# the interpreter itself, not a launcher script: whatever follows `--` must not exec another program once the
# profiler's preloaded library has initialised the GPU (a pyenv/conda shim or `#!/usr/bin/env` script would)
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
# everything is compiled BEFORE the first profiled run (bench.py would otherwise spawn hipcc/gcc under rocprofv3)
"$PY" "$REPO/__graft_entry__.py" || exit 1
cd /tmp && export TMPDIR=/tmp
BENCH="$PY $REPO/bench.py --steps 5 --warmup 2 --cpu-baseline 0 --pmc 0 --two-in-flight 0 --read-sets 0 --min-seconds 0.1 --spinup 0.1 ${BENCH_ARGS:-}"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/bench_kt.json" 2> "$OUT/kt.err"
if [ -n "${KT_ONLY:-}" ]; then exit 0; fi
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
            "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/bench_pmc$i.json" 2> "$OUT/pmc$i.err"
done
ls -R "$OUT" | head -40
