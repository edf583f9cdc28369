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
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
commit = sys.argv[2] if len(sys.argv) > 2 else "unknown"      # the commit the GPU box ran (the snapshot has no .git)
GEMM = ("conv_gemm_kernel", "pw_chain_kernel", "wino_gemm_kernel", "wino4_gemm_kernel", "fusion_conv_kernel")   # the fp32-MFMA conv GEMMs (the roofline's dominant kernels)


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
wi = [r for r in stats if "_input_kernel" in r["Name"] and "wino" in r["Name"]]
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
lines += ["", f"Conv GEMMs (conv_gemm_kernel all variants + pw_chain_kernel + wino_gemm_kernel + wino4_gemm_kernel): {calls} launches, {ns / 1e6:.1f} ms, average {ns / calls / 1e3:.2f} us per launch; "
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

# ---- rounds 2..8 of an annotation session: trace of eight rounds minus trace of the first round alone
t2, t1 = stats_csv("trace_r2"), stats_csv("trace_r1")
if t2 and t1:
    a = {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(t2))}
    b = {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(t1))}
    rows = sorted(((t - b.get(k, (0, 0.0))[1], c - b.get(k, (0, 0.0))[0], k) for k, (c, t) in a.items()), reverse=True)
    rows = [r for r in rows if r[1] > 0]
    tot2 = sum(r[0] for r in rows)
    with open(os.path.join(DST, f"{tag}_r2_kernel_stats.csv"), "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for t, c, k in rows:
            f.write(f"\"{k}\",{c},{t:.0f},{t / c:.1f},{100 * t / tot2:.4f}\n")
    txt = [l for l in open(os.path.join(SRC, "r2_profile.txt")).read().split("\n") if l.startswith("round") or l.startswith("rounds") or l.startswith("kernel time") or l.startswith("  ")]
    frames2 = 184
    md = [f"# Rounds 2..8 of an annotation session on one 66-frame 480x854 clip ({tag})", "",
          "The regime the reference's loops spend their time in (7 of the 8 rounds of interactions/mask.py:113-146, 59 of 60 of eval_annotation_method.py:30): "
          "cached key features, memory read + decoder on every frame, FusionNet + attention read on the frames between interacted frames, a value encode "
          "every 5th frame.", "",
          "Commands (GPU box): `rocprofv3 --kernel-trace --stats -- python3 tools/r2_profile.py --no-class-profile` (8 rounds) and `... --rounds 1`; "
          "the table is the difference of the two kernel-stat tables = rounds 2..8 alone (184 propagated frames).", "",
          f"Total kernel time {tot2 / 1e6:.1f} ms = {tot2 / 1e6 / frames2:.3f} ms per propagated frame.", "",
          "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for t, c, k in rows[:26]:
        md.append(f"| {short(k)} | {c} | {t / 1e6:.2f} | {t / c / 1e3:.1f} | {100 * t / tot2:.2f} |")
    md += ["", "Per launch class, HIP events (`python tools/r2_profile.py`, second pass):", "", "```"] + txt + ["```"]
    r2 = bench.get("roofline_r2") or {}
    if r2:
        md += ["", f"bench.py `roofline_r2` of the same build (interact(mask, T//2) after interact(mask, 0)): {r2.get('frames_per_s_one_video', r2.get('frames_per_s_solo', 0)):.0f} frames/s one video in flight ({r2.get('frames_per_s_solo', 0):.0f} on one stream only), "
               f"{r2.get('frames_per_s_videos_in_flight', 0):.0f} with videos in flight; conv GEMMs (decoder + value encoder + FusionNet) {r2.get('achieved', 0):.1f} TFLOP/s executed = "
               f"{r2.get('frac', 0):.3f} of the fp32 MFMA peak; FusionNet convs {r2.get('fusion_conv_tflops', 0):.1f} TFLOP/s, {r2.get('fusion_conv_ms_per_fused_frame', 0) * 1e3:.0f} us per fused frame."]
    open(os.path.join(DST, f"{tag}_r2_kernel_stats.md"), "w").write("\n".join(md) + "\n")

# ---- counters of the F(4x4) GEMM (tools/pmc_f4.sh)
pf = os.path.join(SRC, "pmc_wino4.txt")
if os.path.exists(pf):
    open(os.path.join(DST, f"{tag}_pmc_wino4.txt"), "w").write(
        "rocprofv3 --pmc passes (one counter set per pass, --kernel-trace only) of `STCN_BENCH_CONV_F4=1 python3 tools/conv_shapes.py --only \"256->256 @4\" --batch 5 --iters 3`\n"
        "(the decoder's 256 -> 256 3x3 conv at 120x216 over a 5-frame decode group as Winograd F(4x4,3x3)); means per launch.  FETCH_SIZE / WRITE_SIZE in KB\n"
        "(FETCH_SIZE x2 on gfx950 for 16 B/lane reads); GRBM_GUI_ACTIVE sums the 8 XCDs: / 8 / launch time = the clock the chip held.\n\n" + open(pf).read())

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
wk = [k for k in kernels if "wino" in k and "_input_kernel" in k]
out = dict(command="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --streams 1 "
                   "--no-profile --frames 30 --cpu-frames 0 --no-r2 --no-config3 --no-memread-roofline --no-davis-val (STCN_LOOKAHEAD=0)",
           commit=commit, csrc_hash=(open(os.path.join(SRC, "csrc_hash.txt")).read().strip() if os.path.exists(os.path.join(SRC, "csrc_hash.txt")) else None),
           frames=30, videos_in_capture=2, captured=f"{tag}, tools/refresh_profiles.sh",
           units="counter values are KB per the rocprofv3 derived metric; gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of the "
                 "bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE exact; Infinity-Cache hits are included (fabric-side counters)",
           conv_gemm_traffic_bytes_per_launch=traffic, conv_gemm_launches=n,
           wino_input_traffic_bytes_per_launch=(sum(kernels[k]["traffic_bytes_per_launch"] * kernels[k]["calls"] for k in wk) / max(1, sum(kernels[k]["calls"] for k in wk))),
           kernels=kernels)
json.dump(out, open(os.path.join(DST, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(f"conv GEMM traffic {traffic / 1e6:.1f} MB per launch over {n} launches; algorithmic {roof['algorithmic_bytes_per_launch'] / 1e6:.1f} MB")
print(open(os.path.join(DST, f"{tag}_bench_streams1_kernel_stats.md")).read()[-1200:])
