"""Device busy fraction of a multi-stream run from a rocprofv3 kernel trace: union of the kernel intervals over the wall span
(first 20 % / last 5 % of the launches dropped).  python tools/trace_union.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
ev = ev[len(ev) // 5: len(ev) * 95 // 100]
span = max(e for _, e in ev) - ev[0][0]
busy, cur_s, cur_e, summed = 0, ev[0][0], ev[0][1], 0
for s, e in ev:
    summed += e - s
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{len(ev)} kernels over {span / 1e6:.1f} ms: some kernel running {100 * busy / span:.1f} % of the time; "
      f"sum of kernel durations {summed / 1e6:.1f} ms = {summed / span:.2f}x the span (kernels overlapping)")
