"""Average shader clock per kernel from a `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` counter file: the counter sums the busy
cycles of the 8 XCDs, so cycles / 8 / duration = the clock the chip held during that kernel (short kernels read low: ramp-up).
Usage: python tools/kernel_clock.py <p_counter_collection.csv> [min_us]"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in rows:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if us < min_us:
        continue
    a = agg[r["Kernel_Name"].split("(")[0][:70]]
    a[0] += float(r["Counter_Value"]) / 8.0
    a[1] += us
    a[2] += 1
for k, (cyc, us, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{cyc / us / 1e3:5.2f} GHz  n={n:5d}  {us / n:8.1f} us avg  {k}")
