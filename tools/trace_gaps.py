"""Idle time between consecutive kernels in a rocprofv3 kernel trace (one stream): python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# keep the steady-state part: drop the first 20 % and the last 5 %
ev = ev[len(ev) // 5: len(ev) * 95 // 100]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = [max(0, ev[i + 1][0] - ev[i][1]) for i in range(len(ev) - 1)]
gaps_sorted = sorted(gaps)
print(f"{len(ev)} kernels, span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), "
      f"gaps: sum {sum(gaps) / 1e6:.1f} ms, median {gaps_sorted[len(gaps) // 2] / 1e3:.2f} us, p90 {gaps_sorted[len(gaps) * 9 // 10] / 1e3:.2f} us, "
      f"max {gaps_sorted[-1] / 1e3:.1f} us")
big = sorted(((g, ev[i][2][:50], ev[i + 1][2][:50]) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, a, b in big:
    print(f"  {g / 1e3:8.1f} us between {a} -> {b}")
