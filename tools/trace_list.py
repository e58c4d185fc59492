"""List the long launches (> 2 ms) and every launch of the isomiR pass of a rocprofv3 --kernel-trace run, in time order:
python tools/trace_list.py gpurun_out/<dir>   (used to find the one 28 ms launch of profiles/r02_kernel_stats.csv)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 2.0 or "k_pass<1, 8>" in n or "k_pass<1,8>" in n:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e6:10.2f} ms  {d:8.3f} ms  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>9s}  {n[:60]}")
