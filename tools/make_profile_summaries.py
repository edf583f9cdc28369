"""Turn gpurun_out/prof_final (written by tools/refresh_profiles.sh on the GPU box) into the tracked profiles/ files.
Usage: python tools/make_profile_summaries.py [round-tag, default r01]"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_final")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def last_json(path):
    """The bench line: the last line of the file that is a JSON object (rocprofv3 appends its own log lines)."""
    for line in reversed(open(path).read().strip().split("\n")):
        if line.startswith("{"):
            return json.loads(line)
    raise ValueError(f"no JSON line in {path}")


bench = last_json(os.path.join(SRC, "bench_default.json"))
shutil.copy(os.path.join(SRC, "bench_default.json"), os.path.join(DST, f"{tag}_bench_default.json"))
shutil.copy(os.path.join(SRC, "bench_config3.json"), os.path.join(DST, f"{tag}_bench_config3_k3_memfreq1.json"))
s1 = last_json(os.path.join(SRC, "bench_streams1.json"))

# ---- kernel trace summary
stats = list(csv.DictReader(open(os.path.join(SRC, "trace", "r_kernel_stats.csv"))))
shutil.copy(os.path.join(SRC, "trace", "r_kernel_stats.csv"), os.path.join(DST, f"{tag}_bench_streams1_kernel_stats.csv"))
tot = sum(float(r["TotalDurationNs"]) for r in stats)
trace_bench = last_json(os.path.join(SRC, "trace.log"))
frames = trace_bench["config"]["frames_per_step"] * (trace_bench["steps"] + trace_bench["warmup"])
conv = [r for r in stats if "conv_gemm_kernel" in r["Name"]]
conv_calls = sum(int(r["Calls"]) for r in conv)
conv_ns = sum(float(r["TotalDurationNs"]) for r in conv)
roof = bench["roofline"]
lines = [f"# rocprofv3 --kernel-trace --stats - {tag}, final engine of the round (solo launches)", "",
         "Command (GPU box): `cd /tmp && STCN_LOOKAHEAD=0 rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py "
         "--streams 1 --steps 2 --warmup 1 --cpu-frames 0 --no-profile --no-f16x3-leg --no-r2`",
         f"{trace_bench['steps'] + trace_bench['warmup']} videos x {trace_bench['config']['frames_per_step']} propagated frames (480x854, k=1, "
         "mem_freq=5), one video in flight, no side stream: the same solo launches bench.py's roofline leg times with HIP events.", "",
         f"Total kernel time {tot / 1e6:.1f} ms = {tot / 1e6 / frames:.2f} ms per propagated frame.", "",
         "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in stats[:22]:
    name = r["Name"].replace("void ", "").split("(")[0]
    lines.append(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |")
lines += ["", f"conv_gemm_kernel (all variants): {conv_calls} launches, {conv_ns / 1e6:.1f} ms, average {conv_ns / conv_calls / 1e3:.2f} us per launch; "
          f"with {roof['flop_per_launch_avg'] / 1e9:.3f} GFLOP average per launch (bench roofline leg) = "
          f"{roof['flop_per_launch_avg'] / (conv_ns / conv_calls * 1e-9) / 1e12:.1f} TFLOP/s.",
          f"bench.py default run of the same build: value {bench['value']:.1f} frames/s ({bench['config']['streams_per_gpu']} videos in flight), "
          f"roofline leg avg_launch_ms {roof['avg_launch_ms'] * 1e3:.2f} us, achieved {roof['achieved']:.1f} TFLOP/s = {roof['frac']:.3f} of the fp32 MFMA peak; "
          f"one video in flight: {s1['value']:.1f} frames/s."]
open(os.path.join(DST, f"{tag}_bench_streams1_kernel_stats.md"), "w").write("\n".join(lines) + "\n")

# ---- trace of the default command (3 videos in flight + the solo roofline leg in one process)
td = os.path.join(SRC, "trace_default", "r_kernel_stats.csv")
if os.path.exists(td):
    st = list(csv.DictReader(open(td)))
    shutil.copy(td, os.path.join(DST, f"{tag}_bench_default_kernel_stats.csv"))
    cv = [r for r in st if "conv_gemm_kernel" in r["Name"]]
    calls = sum(int(r["Calls"]) for r in cv)
    ns = sum(float(r["TotalDurationNs"]) for r in cv)
    bd = last_json(os.path.join(SRC, "trace_default.log"))
    open(os.path.join(DST, f"{tag}_bench_default_kernel_stats.md"), "w").write(
        f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-frames 0 --no-f16x3-leg --no-r2 ({tag})\n\n"
        f"The default command: {bd['config']['streams_per_gpu']} videos in flight in the timed region ({bd['value']:.1f} frames/s under the "
        f"profiler) plus the solo roofline leg, in one process.\nconv_gemm_kernel, all variants: {calls} launches, {ns / 1e6:.1f} ms, "
        f"average {ns / calls / 1e3:.2f} us per launch.  Kernels of concurrent videos overlap here, so this average is NOT the "
        f"kernel's solo duration: while three conv kernels share the chip each one takes longer.  The roofline uses solo launches "
        f"(roofline leg avg {bd['roofline']['avg_launch_ms'] * 1e3:.2f} us = the `--streams 1` trace in "
        f"`{tag}_bench_streams1_kernel_stats.md`).\n\nFull table: `{tag}_bench_default_kernel_stats.csv`.\n")

# ---- PMC traffic (FETCH_SIZE x2 + WRITE_SIZE, KB units, separate passes)
def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].replace("void ", "").split("(")[0]].append(float(r["Counter_Value"]))
    return agg


F = per_kernel(os.path.join(SRC, "pmcF", "p_counter_collection.csv"), "FETCH_SIZE")
W = per_kernel(os.path.join(SRC, "pmcW", "p_counter_collection.csv"), "WRITE_SIZE")
kernels = {}
for k in F:
    f = sum(F[k]) / len(F[k]) * 1024.0
    w = sum(W[k]) / len(W[k]) * 1024.0 if k in W else 0.0
    kernels[k] = dict(calls=len(F[k]), fetch_bytes_per_launch_raw=f, write_bytes_per_launch=w, traffic_bytes_per_launch=2 * f + w)
cg = [k for k in kernels if "conv_gemm_kernel" in k]
n = sum(kernels[k]["calls"] for k in cg)
traffic = sum(kernels[k]["traffic_bytes_per_launch"] * kernels[k]["calls"] for k in cg) / n
out = dict(command="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --streams 1 "
                   "--cpu-frames 0 --no-profile --no-f16x3-leg --no-r2 --frames 30 (STCN_LOOKAHEAD=0)",
           units="counter values are KB per the rocprofv3 derived metric; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of the "
                 "bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE exact; Infinity-Cache hits are included (fabric-side counters)",
           conv_gemm_traffic_bytes_per_launch=traffic, conv_gemm_launches=n, kernels=kernels)
json.dump(out, open(os.path.join(DST, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(f"conv traffic {traffic / 1e6:.1f} MB per launch over {n} launches; algorithmic {roof['algorithmic_bytes_per_launch'] / 1e6:.1f} MB")
print(open(os.path.join(DST, f"{tag}_bench_streams1_kernel_stats.md")).read()[-900:])
