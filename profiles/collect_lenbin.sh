#!/bin/bash
# Counters of the cascade's passes on unique reads in collapse order vs grouped by length (tools/len_bin_experiment.py),
# one order per rocprofv3 --pmc pass:   bash profiles/collect_lenbin.sh   -> gpurun_out/prof_lenbin/<order>_<n>/ + summary
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_lenbin
mkdir -p "$OUT"
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
"$PY" "$REPO/__graft_entry__.py" || exit 1
cd /tmp && export TMPDIR=/tmp
for ORDER in collapse grouped; do
  i=0
  for CTRS in "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/${ORDER}_$i" -- $PY $REPO/tools/len_bin_experiment.py 10000000 2 $ORDER > "$OUT/${ORDER}_$i.txt" 2> "$OUT/${ORDER}_$i.err"
  done
done
"$PY" - "$OUT" <<'PYEOF'
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
for order in ("collapse", "grouped"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, order + "_*", "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        gmax = defaultdict(int)
        for r in rows:
            m = re.search(r"k_pass<1, (\d+)>", r["Kernel_Name"])
            if m: gmax[m.group(1)] = max(gmax[m.group(1)], int(r["Grid_Size"]))
        for r in rows:
            m = re.search(r"k_pass<1, (\d+)>", r["Kernel_Name"])
            if m and int(r["Grid_Size"]) == gmax[m.group(1)]:
                acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for p in sorted(acc, key=int):
        a = {k: sum(v) / len(v) for k, v in acc[p].items()}
        util = a.get("SQ_THREAD_CYCLES_VALU", 0) / max(a.get("SQ_ACTIVE_INST_VALU", 1), 1) / 64 * 4  # lanes per issued VALU cycle
        print(order, "k_pass<1,%s>" % p, " ".join(f"{k}={v:.3g}" for k, v in sorted(a.items())), f"lane_util~{util:.2f}")
PYEOF
