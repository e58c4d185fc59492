"""Per-kernel averages of the memory-pipeline counters profiles/collect_mem.sh collects.

    python profiles/mem_summary.py gpurun_out/prof_r02_mem > profiles/r02_mem_counters.txt

One row per kernel of the step (template arguments kept, so every cascade pass is its own row): launches, average
duration, and per-launch averages of each counter; derived columns: TA busy share of the kernel's cycles per CU
(TA_TA_BUSY_sum / 256 CUs / GRBM_GUI_ACTIVE per XCD), vector-L1 accesses, L1->L2 read requests, UTCL1 translation misses,
and the average L2 read latency per request (TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum).
"""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "pmc*", "*", "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if not name.startswith(("k_", "void k_")): continue
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name)
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)

def avg(v): return sum(v) / len(v) if v else 0.0

cols = ["launches", "avg_ms", "TA_busy", "TA_stall_TC", "L1_acc_M", "L2_rd_req_M", "L1_pend_stall", "TLB_miss_M", "TLB_req_M",
        "L2_lat_cyc"]
print(f"{'kernel':34s}" + "".join(f"{c:>14s}" for c in cols))
for name in sorted(acc, key=lambda n: -avg(dur[n])):
    c = {k: avg(v) for k, v in acc[name].items()}
    n_launch = max(len(v) for v in acc[name].values())
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; TA_*_sum over all 256 CUs' texture addressers
    per_xcd = gui / 8 if gui else 0.0
    ta = c.get("TA_TA_BUSY_sum", 0.0) / 256 / per_xcd if per_xcd else 0.0
    tastall = c.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0.0) / 256 / per_xcd if per_xcd else 0.0
    pend = c.get("TCP_PENDING_STALL_CYCLES_sum", 0.0) / 256 / per_xcd if per_xcd else 0.0
    req = c.get("TCP_TCC_READ_REQ_sum", 0.0)
    lat = c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / req if req else 0.0
    vals = [n_launch, avg(dur[name]), ta, tastall, c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / 1e6, req / 1e6, pend,
            c.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0.0) / 1e6, c.get("TCP_UTCL1_REQUEST_sum", 0.0) / 1e6, lat]
    print(f"{name[:34]:34s}" + f"{vals[0]:14d}" + "".join(f"{v:14.4f}" for v in vals[1:]))
