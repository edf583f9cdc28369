"""Per-kernel call counts / average microseconds from a rocprofv3 --kernel-trace --stats output directory.
Usage: python tools/kstat.py <dir> [substring ...]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
subs = sys.argv[2:]
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(f"total kernel time {sum(float(r['TotalDurationNs']) for r in rows) / 1e6:.2f} ms ({f})")
for r in rows:
    n = r["Name"]
    if subs and not any(s in n for s in subs):
        continue
    print(f"{float(r['TotalDurationNs']) / 1e6:9.3f} ms  calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {n[:100]}")
