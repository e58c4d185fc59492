#!/bin/bash
# session 46: LDS counters of the partition kernels (is k_part_dedup's record loop waiting for LDS?)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r5_46; mkdir -p "$OUT"
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
"$PY" "$REPO/__graft_entry__.py" || exit 1
cd /tmp && export TMPDIR=/tmp
BENCH="$PY $REPO/bench.py --steps 5 --warmup 2 --cpu-baseline 0 --pmc 0 --two-in-flight 0 --cli-path 0 --read-sets 0 --min-seconds 0.1 --spinup 0.1"
i=0
for CTRS in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_VALU" "SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/bench_pmc$i.json" 2> "$OUT/pmc$i.err"
  tail -2 "$OUT/pmc$i.err" | cut -c1-200
done
cd "$REPO"
python3 - <<'PY' | tee gpurun_out/r5_46_lds_counters.txt
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for f in sorted(glob.glob("gpurun_out/prof_r5_46/pmc*/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", row["Kernel_Name"]))
        if not name.startswith(("k_part", "k_cascade_bulk<1, false")): continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
avg = lambda v: sum(v) / len(v) if v else 0.0
for name in sorted(acc):
    print(name, "avg_ms", round(avg(dur[name]), 4), {k: round(avg(v), 1) for k, v in acc[name].items()})
PY
rm -rf gpurun_out/prof_r5_46
