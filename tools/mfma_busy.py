"""MFMA-pipe occupancy per kernel of an engine run from ONE `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA
SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace` pass: per kernel, sum over its launches of
  matrix-pipe busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)   (SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD, summed)
and VALU instructions per MFMA instruction.  Usage: python tools/mfma_busy.py <p_counter_collection.csv> [--md]"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"],)
    if key not in seen:
        seen.add(key)
        calls[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = []
for k, c in agg.items():
    if c.get("SQ_INSTS_MFMA", 0) <= 0:
        continue
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    rows.append((dur[k], k, calls[k], c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]))
md = "--md" in sys.argv
if md:
    print("| kernel | launches | time under the counters (ms) | matrix pipes busy | VALU instructions per MFMA |\n|---|---|---|---|---|")
for d, k, n, busy, vpm in sorted(rows, reverse=True):
    if md:
        print(f"| {k} | {n} | {d / 1e3:.1f} | {100 * busy:.1f} % | {vpm:.2f} |")
    else:
        print(f"{100 * busy:5.1f} % busy  {vpm:6.2f} VALU/MFMA  n={n:5d}  {d / 1e3:8.1f} ms  {k}")
