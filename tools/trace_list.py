"""Print a rocprofv3 kernel trace (CSV) as runs of identical (kernel, grid) launches in time order: count and average
duration per run.  Usage: python tools/trace_list.py <kernel_trace.csv> [substring ...]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
subs = sys.argv[2:]
runs = []
for r in rows:
    name = r["Kernel_Name"]
    if subs and not any(s in name for s in subs):
        continue
    key = (name[:70], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    runs.append((key, d))
agg = {}
order = []
# aggregate per repeating pattern position: consecutive blocks of the same key set
for key, d in runs:
    if key not in agg:
        agg[key] = []
        order.append(key)
    agg[key].append(d)
for key in order:
    v = agg[key]
    print(f"{key[0]:70s} grid {key[1]:>8s}x{key[2]:>3s} lds {key[3]:>6s} vgpr {key[4]:>4s} n={len(v):4d} avg {sum(v)/len(v):9.1f} us  min {min(v):9.1f}")
