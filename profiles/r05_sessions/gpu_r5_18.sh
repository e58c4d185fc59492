#!/bin/bash
# session 18: memory-pipeline counters PER PASS -- the staged route (MIRGE_BULK_FUSED=0: one k_pass launch per pass, no whole-read
# tables there) under separate --pmc passes; which pass the bulk kernel's L1->L2 requests and L2 misses belong to
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r5_18_perpass
mkdir -p "$OUT"
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
"$PY" "$REPO/__graft_entry__.py" || exit 1
cd /tmp && export TMPDIR=/tmp
export MIRGE_BULK_FUSED=0
BENCH="$PY $REPO/bench.py --steps 5 --warmup 2 --cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 --read-sets 0 --min-seconds 0.1 --spinup 0.1"
i=0
for CTRS in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
            "TCC_MISS_sum TCC_HIT_sum" \
            "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/bench_pmc$i.json" 2> "$OUT/pmc$i.err"
done
cd "$REPO"
python3 - <<'PY' > gpurun_out/r5_18_per_pass_counters.txt
import csv, glob, os, re
from collections import defaultdict
root = "gpurun_out/prof_r5_18_perpass"
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "pmc*", "*", "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", row["Kernel_Name"]))
        if not name.startswith(("k_pass", "k_cascade", "k_resolve")): continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
avg = lambda v: sum(v) / len(v) if v else 0.0
cols = ["n", "avg_ms", "L1_acc_M", "L2_req_M", "TCC_miss_M", "TCC_hit_M", "L2_lat", "pend_stall", "TA_busy", "VALU_M", "VMEM_RD_M", "LDS_M", "waves_k"]
print(f"{'kernel':30s}" + "".join(f"{c:>12s}" for c in cols))
for name in sorted(acc, key=lambda n: n):
    c = {k: avg(v) for k, v in acc[name].items()}
    per_xcd = c.get("GRBM_GUI_ACTIVE", 0) / 8
    req = c.get("TCP_TCC_READ_REQ_sum", 0)
    vals = [max(len(v) for v in acc[name].values()), avg(dur[name]), c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / 1e6, req / 1e6,
            c.get("TCC_MISS_sum", 0) / 1e6, c.get("TCC_HIT_sum", 0) / 1e6, c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / req if req else 0,
            c.get("TCP_PENDING_STALL_CYCLES_sum", 0) / 256 / per_xcd if per_xcd else 0, c.get("TA_TA_BUSY_sum", 0) / 256 / per_xcd if per_xcd else 0,
            c.get("SQ_INSTS_VALU", 0) / 1e6, c.get("SQ_INSTS_VMEM_RD", 0) / 1e6, c.get("SQ_INSTS_LDS", 0) / 1e6, c.get("SQ_WAVES", 0) / 1e3]
    print(f"{name:30s}" + "".join(f"{v:12.3f}" for v in vals))
PY
cat gpurun_out/r5_18_per_pass_counters.txt
python3 -c "
import json
d=json.loads(open('$OUT/bench_pmc2.json').read().strip().splitlines()[-1])
print({k:v for k,v in d['kernels'].items() if k.startswith('k_pass') or k.startswith('k_casc')})
" 2>&1 | cut -c1-3000
