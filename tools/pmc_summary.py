"""Summarise a rocprofv3 --pmc counter_collection.csv: mean per dispatch of each counter for kernels whose
name contains a substring.  Usage: python tools/pmc_summary.py <csv> [<csv> ...] --kernel conv_gemm"""
import collections
import csv
import sys

files = [a for a in sys.argv[1:] if a.endswith(".csv")]
kern = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else ""
for f in files:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"{k:32s} n={len(v):4d} mean={sum(v) / len(v):.6g}")
