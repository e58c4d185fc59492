#!/bin/bash
# session 38: memory counters of the partition kernels at C5's size (50 M reads): why k_part_split grows faster than its input
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r5_38; mkdir -p "$OUT"
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
"$PY" "$REPO/__graft_entry__.py" || exit 1
cd /tmp && export TMPDIR=/tmp
BENCH="$PY $REPO/bench.py --workload c5 --reads 50000000 --steps 3 --warmup 1 --cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 --read-sets 0 --min-seconds 0.05 --spinup 0.1"
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/bench_pmc$i.json" 2> "$OUT/pmc$i.err"
done
cd "$REPO"
python3 - <<'PY' | tee gpurun_out/r5_38_partition_counters_50M.txt
import csv, glob, os, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for f in sorted(glob.glob("gpurun_out/prof_r5_38/pmc*/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", row["Kernel_Name"]))
        if not name.startswith("k_part"): continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
avg = lambda v: sum(v) / len(v) if v else 0.0
for name in sorted(acc):
    c = {k: round(avg(v), 1) for k, v in acc[name].items()}
    print(name, "avg_ms", round(avg(dur[name]), 4), c)
PY
rm -rf gpurun_out/prof_r5_38
