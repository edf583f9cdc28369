"""Turn gpurun_out/prof_final (written by tools/refresh_profiles.sh on the GPU box) into the tracked profiles/ files.
Usage: python tools/make_profile_summaries.py [round-tag, default r02]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_final")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
GEMM = ("conv_gemm_kernel", "wino_gemm_kernel")          # the fp32-MFMA conv GEMMs (the roofline's dominant kernels)


def last_json(path):
    """The bench line: the last line of the file that is a JSON object (rocprofv3 appends its own log lines)."""
    for line in reversed(open(path).read().strip().split("\n")):
        if line.startswith("{"):
            return json.loads(line)
    raise ValueError(f"no JSON line in {path}")


def stats_csv(sub):
    f = glob.glob(os.path.join(SRC, sub, "**", "r_kernel_stats.csv"), recursive=True)
    return f[0] if f else None


def short(name):
    return name.replace("void ", "").split("(")[0]


bench = last_json(os.path.join(SRC, "bench_default.json"))
shutil.copy(os.path.join(SRC, "bench_default.json"), os.path.join(DST, f"{tag}_bench_default.json"))
s1 = last_json(os.path.join(SRC, "bench_streams1.json"))
roof = bench["roofline"]

# ---- kernel trace summary (solo launches)
src = stats_csv("trace")
stats = list(csv.DictReader(open(src)))
shutil.copy(src, os.path.join(DST, f"{tag}_bench_streams1_kernel_stats.csv"))
tot = sum(float(r["TotalDurationNs"]) for r in stats)
trace_bench = last_json(os.path.join(SRC, "trace.log"))
frames = trace_bench["config"]["frames_per_step"] * (trace_bench["steps"] + trace_bench["warmup"] + 1)   # + the solo determinism re-run
gemm = [r for r in stats if any(g in r["Name"] for g in GEMM)]
calls = sum(int(r["Calls"]) for r in gemm)
ns = sum(float(r["TotalDurationNs"]) for r in gemm)
wi = [r for r in stats if "wino_input_kernel" in r["Name"]]
wi_ns = sum(float(r["TotalDurationNs"]) for r in wi)
rd_ns = sum(float(r["TotalDurationNs"]) for r in stats if "conv_reduce" in r["Name"])
lines = [f"# rocprofv3 --kernel-trace --stats - {tag}, final engine of the round (solo launches)", "",
         "Command (GPU box): `cd /tmp && STCN_LOOKAHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py "
         "--streams 1 --steps 2 --warmup 1 --no-profile --cpu-frames 0 --no-r2 --no-config3 --no-memread-roofline`",
         f"{frames // trace_bench['config']['frames_per_step']} videos x {trace_bench['config']['frames_per_step']} propagated frames (480x854, k=1, "
         "mem_freq=5), one video in flight, no side stream: the same solo launches bench.py's roofline leg times with HIP events.", "",
         f"Total kernel time {tot / 1e6:.1f} ms = {tot / 1e6 / frames:.2f} ms per propagated frame.", "",
         "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in stats[:24]:
    lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |")
ex, al = roof["executed_flop_per_launch_avg"], roof["algorithmic_flop_per_launch_avg"]
lines += ["", f"Conv GEMMs (conv_gemm_kernel all variants + wino_gemm_kernel): {calls} launches, {ns / 1e6:.1f} ms, average {ns / calls / 1e3:.2f} us per launch; "
          f"with {ex / 1e9:.3f} GFLOP executed / {al / 1e9:.3f} GFLOP algorithmic per launch on average (bench roofline leg) = "
          f"{ex / (ns / calls * 1e-9) / 1e12:.1f} TFLOP/s executed on the matrix cores.",
          f"Including the Winograd input transforms ({wi_ns / 1e6:.1f} ms) and split-K reduces ({rd_ns / 1e6:.1f} ms): "
          f"{al * calls / ((ns + wi_ns + rd_ns) * 1e-9) / 1e12:.1f} TFLOP/s algorithmic.",
          f"bench.py default run of the same build: value {bench['value']:.1f} frames/s ({bench['config']['streams_per_gpu']} videos in flight), "
          f"roofline leg avg_launch_ms {roof['avg_launch_ms'] * 1e3:.2f} us, achieved {roof['achieved']:.1f} TFLOP/s executed = {roof['frac']:.3f} of the "
          f"fp32 MFMA peak, {roof['algorithmic_tflops_incl_transforms']:.1f} TFLOP/s algorithmic; one video in flight: {s1['value']:.1f} frames/s."]
open(os.path.join(DST, f"{tag}_bench_streams1_kernel_stats.md"), "w").write("\n".join(lines) + "\n")

# ---- trace of the default command (3 videos in flight + the solo roofline leg in one process)
td = stats_csv("trace_default")
if td:
    st = list(csv.DictReader(open(td)))
    shutil.copy(td, os.path.join(DST, f"{tag}_bench_default_kernel_stats.csv"))
    cv = [r for r in st if any(g in r["Name"] for g in GEMM)]
    c2 = sum(int(r["Calls"]) for r in cv)
    n2 = sum(float(r["TotalDurationNs"]) for r in cv)
    bd = last_json(os.path.join(SRC, "trace_default.log"))
    open(os.path.join(DST, f"{tag}_bench_default_kernel_stats.md"), "w").write(
        f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-frames 0 --no-r2 --no-config3 --no-memread-roofline ({tag})\n\n"
        f"The default timed region: {bd['config']['streams_per_gpu']} videos in flight ({bd['value']:.1f} frames/s under the "
        f"profiler) plus the solo roofline leg, in one process.\nConv GEMMs (conv_gemm_kernel + wino_gemm_kernel): {c2} launches, {n2 / 1e6:.1f} ms, "
        f"average {n2 / c2 / 1e3:.2f} us per launch.  Kernels of concurrent videos overlap here, so this average is NOT the "
        f"kernel's solo duration: while three conv kernels share the chip each one takes longer.  The roofline uses solo launches "
        f"(roofline leg avg {bd['roofline']['avg_launch_ms'] * 1e3:.2f} us = the `--streams 1` trace in "
        f"`{tag}_bench_streams1_kernel_stats.md`).\n\nFull table: `{tag}_bench_default_kernel_stats.csv`.\n")

# ---- memory-read bench trace
tm = stats_csv("trace_memread")
if tm:
    st = list(csv.DictReader(open(tm)))
    shutil.copy(tm, os.path.join(DST, f"{tag}_memread_kernel_stats.csv"))
    log = [l for l in open(os.path.join(SRC, "trace_memread.log")).read().split("\n") if l[:5].strip().isdigit() or l.strip().startswith("T ")]
    rows = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in st[:8]:
        rows.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |")
    mr = bench.get("roofline_memread") or {}
    open(os.path.join(DST, f"{tag}_memread_kernel_stats.md"), "w").write(
        f"# rocprofv3 --kernel-trace --stats -- python3 tools/memread_bench.py --k 5 ({tag})\n\n"
        "Whole memory reads (pass 1 sampled, threshold, pass 2, merge + gather) on random N(0, 0.8) keys, k = 5 objects, HIP events around 10 reads per shape:\n\n```\n"
        + "\n".join(log) + "\n```\n\n" + "\n".join(rows) + "\n\n"
        f"bench.py `roofline_memread` of the same build (T=104, k=5): {mr.get('achieved', 0):.1f} TFLOP/s on 2*N*Q*64 = {mr.get('frac', 0):.3f} of the fp32 MFMA peak, "
        f"{mr.get('algorithmic_gbytes_per_s', 0):.0f} GB/s on the algorithmic bytes.\n")


# ---- PMC traffic (FETCH_SIZE x2 + WRITE_SIZE, KB units, separate passes)
def per_kernel(sub, counter):
    f = glob.glob(os.path.join(SRC, sub, "**", "p_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


F, W = per_kernel("pmcF", "FETCH_SIZE"), per_kernel("pmcW", "WRITE_SIZE")
kernels = {}
for k in F:
    f = sum(F[k]) / len(F[k]) * 1024.0
    w = sum(W[k]) / len(W[k]) * 1024.0 if k in W else 0.0
    kernels[k] = dict(calls=len(F[k]), fetch_bytes_per_launch_raw=f, write_bytes_per_launch=w, traffic_bytes_per_launch=2 * f + w)
cg = [k for k in kernels if any(g in k for g in GEMM)]
n = sum(kernels[k]["calls"] for k in cg)
traffic = sum(kernels[k]["traffic_bytes_per_launch"] * kernels[k]["calls"] for k in cg) / n
wk = [k for k in kernels if "wino_input_kernel" in k]
out = dict(command="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --streams 1 "
                   "--no-profile --frames 30 --cpu-frames 0 --no-r2 --no-config3 --no-memread-roofline (STCN_LOOKAHEAD=0)",
           units="counter values are KB per the rocprofv3 derived metric; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of the "
                 "bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE exact; Infinity-Cache hits are included (fabric-side counters)",
           conv_gemm_traffic_bytes_per_launch=traffic, conv_gemm_launches=n,
           wino_input_traffic_bytes_per_launch=(sum(kernels[k]["traffic_bytes_per_launch"] * kernels[k]["calls"] for k in wk) / max(1, sum(kernels[k]["calls"] for k in wk))),
           kernels=kernels)
json.dump(out, open(os.path.join(DST, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(f"conv GEMM traffic {traffic / 1e6:.1f} MB per launch over {n} launches; algorithmic {roof['algorithmic_bytes_per_launch'] / 1e6:.1f} MB")
print(open(os.path.join(DST, f"{tag}_bench_streams1_kernel_stats.md")).read()[-1200:])
