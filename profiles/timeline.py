#!/usr/bin/env python3
"""Per-step timeline of the bench from a rocprofv3 kernel trace: start/end of every dispatch of the LAST
complete step relative to the step's first kernel, per HW queue, with the idle gap before it.
usage: python profiles/timeline.py r01 [> profiles/r01_timeline.txt]"""
import csv
import glob
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.abspath(__file__))
files = glob.glob(os.path.join(root, "..", "gpurun_out", f"prof_{tag}", "kt", "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "k_part_agg" in r["Kernel_Name"]]
# the timed steps are the shortest run of consecutive collapses: take the shortest interval
ts = [int(rows[i]["Start_Timestamp"]) for i in first]
j = min(range(len(first) - 1), key=lambda i: ts[i + 1] - ts[i])
a, b = first[j], first[j + 1]
# the small groups' collapse kernels are enqueued before k_part_agg: start the step at the previous k_join's end
while a > 0 and "k_join" not in rows[a - 1]["Kernel_Name"]:
    a -= 1
while b > 0 and "k_join" not in rows[b - 1]["Kernel_Name"]:
    b -= 1
# what sits right behind the previous k_join belongs to that step (its read-back, the clear of its tables for the next
# call); the host's pause between two steps follows: start behind the largest gap in front of this step's k_part_agg
fa = next(i for i in range(a, b) if "k_part_agg" in rows[i]["Kernel_Name"])
if fa > a:
    gaps = [(int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]), i) for i in range(a + 1, fa + 1)]
    g, at = max(gaps)
    if g > 100_000:
        a = at
t0 = int(rows[a]["Start_Timestamp"])
end_by_q, busy_by_q = {}, {}
print(f"# one step = dispatches {a}..{b - 1}; times in us from the step's first kernel")
print(f"{'start':>9} {'end':>9} {'dur':>8} {'gap':>8}  q  {'grid':>9}  kernel")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r["Queue_Id"]
    gap = s - end_by_q.get(q, s)
    end_by_q[q] = e
    busy_by_q[q] = busy_by_q.get(q, 0) + e - s
    m = re.match(r"(?:void )?(\w+)(<[^>]*>)?", r["Kernel_Name"])
    nm = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"]
    print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:8.1f}  {q}  {r['Grid_Size_X']:>9}  {nm[:60]}")
span = max(int(r["End_Timestamp"]) for r in rows[a:b]) - t0
print(f"# step span {span / 1e3:.1f} us; busy per queue: " + ", ".join(f"q{q} {v / 1e3:.1f} us" for q, v in sorted(busy_by_q.items())))
