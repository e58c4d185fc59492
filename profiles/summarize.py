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
if stats:
    with open(stats[0]) as fh, open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w", newline="") as out:
        rd = csv.DictReader(fh)
        wr = csv.writer(out)
        wr.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent", "min_ns", "max_ns"])
        for r in rd:
            wr.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], f"{float(r['AverageNs']):.0f}",
                         r['Percentage'], r['MinNs'], r['MaxNs']])
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
