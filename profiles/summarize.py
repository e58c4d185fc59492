#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of profiles/collect.sh into the small committed summaries:
  profiles/<tag>_kernel_stats.csv   per-kernel launch statistics (the --stats table, names shortened)
  profiles/<tag>_pmc.csv            per-kernel PMC averages per launch (all passes merged)
usage: python profiles/summarize.py r01
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(root, "..", "gpurun_out", f"prof_{tag}")


def short(name):
    m = re.match(r"(?:void )?(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name


stats = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)
# the distribution behind the --stats averages, from the kernel trace itself: a process's first launches of the cascade
# kernels take tens of milliseconds (the whole device stalls while the first step's buffers are mapped), and one of them in
# ~160 launches moves an average by 20 %; median / p10 / p90 and the average of the launches within 1.5 x the median say what a
# launch takes once the process is warm (the launches bench.py times with HIP events)
durs = defaultdict(list)
for f in glob.glob(os.path.join(src, "kt", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))


def dist(name):
    v = sorted(durs.get(name, []))
    if not v:
        return ["", "", "", "", ""]
    med = v[len(v) // 2]
    warm = [x for x in v if x <= 1.5 * med]
    return [med, v[int(0.1 * (len(v) - 1))], v[int(0.9 * (len(v) - 1))], len(warm), f"{sum(warm) / len(warm):.0f}"]


if stats:
    with open(stats[0]) as fh, open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w", newline="") as out:
        rd = csv.DictReader(fh)
        wr = csv.writer(out)
        wr.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent", "min_ns", "max_ns", "median_ns", "p10_ns", "p90_ns",
                     "calls_within_1.5x_median", "avg_ns_of_those"])
        for r in rd:
            wr.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], f"{float(r['AverageNs']):.0f}",
                         r['Percentage'], r['MinNs'], r['MaxNs']] + dist(short(r['Name'])))
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
if acc:
    ctrs = sorted({c for k in acc for c in acc[k]})
    with open(os.path.join(root, f"{tag}_pmc.csv"), "w", newline="") as out:
        wr = csv.writer(out)
        wr.writerow(["kernel", "launches"] + ctrs)
        for k in sorted(acc):
            n = max(len(v) for v in calls[k].values())
            wr.writerow([k, n] + [f"{acc[k][c] / max(len(calls[k][c]), 1):.1f}" if c in acc[k] else "" for c in ctrs])
print("wrote", [f for f in os.listdir(root) if f.startswith(tag)])
