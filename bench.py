#!/usr/bin/env python3
"""bench.py - frames/sec of STCN mask propagation (480p, 1 object) on N MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1: launched by torch.distributed.run, one
rank per GPU).  One STEP = one pass of the hot path over one synthetic video: the first
``InferenceCore.interact(mask, 0)`` on a fresh engine ("R1"): every one of the T-1
propagated frames pays key encoder + memory read + decoder + aggregate, every 5th a value encode; mask
H2D, final argmax and mask D2H are inside the timed region; clip decode/H2D and engine construction are
not (inputs resident in HBM).  Videos shard across ranks with no data-path collective (weak scaling:
every rank runs K videos); the only collectives are the timing barrier/max and one gather of per-video
J&F rows (RCCL over xGMI).

Output: ONE JSON line on rank 0 (see the task contract) with two extra objects:
  roofline     - dominant kernel (fp32 implicit-GEMM conv): algorithmic FLOP of all conv launches in the
                 timed region / their summed device time (HIP events on the engine stream) vs the
                 fp32 MFMA peak 157.3 TFLOP/s (MI355X_MICROARCH.md).
  cpu_baseline - the CPU oracle (oracle/stcn_oracle.py, a port validated against the reference) timed on
                 this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (matrix)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default = WORLD_SIZE or 1")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=66, help="frames per video (DAVIS-17 val mean length ~66)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=854)
    ap.add_argument("--mem-freq", type=int, default=5)
    ap.add_argument("--objects", type=int, default=1, help="k>1: multi-object engine via the scribble/(k+1)-channel path (config 3)")
    ap.add_argument("--cpu-frames", type=int, default=40, help="frames of the bounded CPU-oracle sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip the roofline leg (roofline = null)")
    ap.add_argument("--roof-steps", type=int, default=1, help="videos of the profiled single-stream roofline leg")
    ap.add_argument("--streams", type=int, default=3, help="videos in flight per GPU (one host thread + HIP stream each)")
    ap.add_argument("--no-f16x3-leg", dest="f16x3_leg", action="store_false",
                    help="skip the extra (non-headline) f16x3 split-precision leg")
    ap.add_argument("--no-r2", dest="r2", action="store_false",
                    help="skip the extra R2 number (a second interaction: cached keys + fusion; reported, not the headline)")
    return ap.parse_args()


def cpu_baseline(psd, fsd, H, W, frames, mem_freq):
    """Oracle (kind 'port') on the host cores: interact(mask, 0) on a `frames`-long clip of the same shape."""
    from eva_vos_amd import synth
    from oracle.stcn_oracle import OracleCore
    from oracle import stcn_oracle as O
    img, msk = synth.synthetic_clip(frames, H, W), synth.synthetic_mask(frames, H, W, 1)
    # pick the intra-op thread count that is fastest on this host (hundreds of threads thrash on the
    # small GEMMs of the path): one key-encoder pass per candidate
    fw = O.fold_bn(psd)
    x0, _ = O.pad16(img[:, 0])
    best_t, best = 1, float("inf")
    for nt in sorted({n for n in (8, 16, 32, 64, os.cpu_count() or 1) if n <= (os.cpu_count() or 1)}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        O.encode_key(fw, x0)
        el = time.perf_counter() - t0
        if el < best:
            best_t, best = nt, el
    torch.set_num_threads(best_t)
    core = OracleCore(psd, fsd, img, 1, mem_freq=mem_freq)
    t0 = time.perf_counter()
    ref_masks = core.interact(msk[:, 0], 0)
    dt = time.perf_counter() - t0
    base = dict(value=(frames - 1) / dt, unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle OracleCore.interact(mask,0) on a {frames}-frame {H}x{W} synthetic clip "
                       f"({frames - 1} propagated frames, {dt:.1f} s, torch {torch.__version__} CPU)")
    return base, (img, msk, ref_masks)


def parity_vs_oracle(prop, fuse, sample, mem_freq):
    """The HIP engine on the clip the CPU oracle just processed: mask IoU between the two and J&F of each against the
    synthetic ground truth (north_star: masks within 1e-3 IoU, J&F within 0.1 of the CPU reference)."""
    from eva_vos_amd import metrics
    from mivos.inference_core import InferenceCore
    img, msk, ref_masks = sample
    got = InferenceCore(prop, fuse, img.cuda(), 1, mem_freq=mem_freq).interact(msk[:, 0], 0)
    a, b = got > 0, ref_masks > 0
    union = (a | b).sum()
    gt = (msk[0, :, 0] > 0.5).cuda()
    jf_gpu = metrics.sequence_scores_gpu(gt, torch.from_numpy(a).cuda())[1:, 2].mean()
    jf_cpu = metrics.sequence_scores_gpu(gt, torch.from_numpy(b).cuda())[1:, 2].mean()
    return dict(clip=f"{img.shape[1]} frames {img.shape[-2]}x{img.shape[-1]} (the cpu_baseline sample)",
                mask_iou_hip_vs_cpu_oracle=float((a & b).sum() / union) if union else 1.0,
                mask_pixels_differing=int((a != b).sum()), mask_pixels_total=int(a.size),
                j_and_f_hip=float(jf_gpu), j_and_f_cpu_oracle=float(jf_cpu))


def launch_ranks(n):
    """`python bench.py --gpus N` outside torch.distributed.run: this process - which has not touched the GPU (no
    torch.cuda call, libstcn_hip.so not loaded) - starts N fresh rank processes through torch.distributed.run (one per
    GPU, rendezvous on 127.0.0.1), lets them print (rank 0 prints the JSON line) and returns their exit status.  The
    reference shards the same way by hand: one process per `--min-idx/--max-idx` slice
    (eval_annotation_method.py:34-35,118-119; datasets/annotation_dataset.py:56-59)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    a = parse()
    if a.gpus is not None and a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus is None:
        a.gpus = world
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node equal to --gpus"
    torch.set_grad_enabled(False)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU fallback of the product)"
    # STCN_BENCH_DEVICE / STCN_BENCH_BACKEND exist only to exercise the multi-rank path on a 1-GPU box
    # (all ranks on one device, gloo); the driver's runs use one GPU per rank and RCCL ("nccl").
    local = int(os.environ.get("STCN_BENCH_DEVICE", local))
    backend = os.environ.get("STCN_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from eva_vos_amd import metrics, shard, synth
    from eva_vos_amd.params import FusionNet, PropagationNetwork
    from mivos.inference_core import InferenceCore

    prop, fuse = PropagationNetwork(), FusionNet()
    psd, fsd = synth.recipe_state_dict(prop), synth.recipe_state_dict(fuse)
    prop.load_state_dict(psd)
    fuse.load_state_dict(fsd)

    T, H, W = a.frames, a.height, a.width
    img = synth.synthetic_clip(T, H, W).cuda()
    K_OBJ = a.objects
    gt = synth.synthetic_mask(T, H, W, K_OBJ)

    def as_input(m):         # k == 1: [1,1,H,W] without bg row; k > 1: bg row first + scribble=True (reference semantics)
        return m.clone() if K_OBJ == 1 else torch.cat([1 - m.sum(0, keepdim=True).clamp(0, 1), m], 0)

    mask0 = as_input(gt[:, 0])
    mask_mid = as_input(gt[:, T // 2])

    # one HIP stream per in-flight video; engines are bound to the stream they are created under
    S = max(1, min(a.streams, a.steps))
    # engine knob: key-encoder look-ahead on a side stream helps a single video in flight (+6 %) but only
    # adds contention when several videos already overlap
    os.environ.setdefault("STCN_LOOKAHEAD", "0" if S > 1 else "2")
    streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
    # A bounded pool of engines (each ~5.5 GB at T=66) serves any --steps: lane l owns engines pool[l]; a video
    # takes the lane's next engine and resets it first (reset = what a fresh InferenceCore would hold).  Every engine of
    # the pool holds a DIFFERENT clip (the synthetic scene under its own noise), so concurrent videos never share inputs; a repeated video
    # of one engine must reproduce its previous result bit for bit while other clips run beside it.
    per_lane = 2
    def variant(n):          # clip 0 = the recipe clip; clip n = the same scene under its own seeded sensor noise
        if n == 0:
            return img
        g = torch.Generator(device="cuda").manual_seed(1000 + n)
        return img + 0.15 * torch.randn(img.shape, generator=g, device="cuda")

    clips = [[variant(l * per_lane + j) for j in range(per_lane)] for l in range(S)]

    def make(lane, j=0):
        with torch.cuda.stream(streams[lane]):
            return InferenceCore(prop, fuse, clips[lane][j], K_OBJ, mem_freq=a.mem_freq)

    pool = [[make(l, j) for j in range(per_lane)] for l in range(S)]
    torch.cuda.synchronize()
    for i in range(a.warmup):
        l = i % S
        with torch.cuda.stream(streams[l]):
            e = pool[l][(i // S) % per_lane]
            e.reset()
            e.interact(mask0, 0, scribble=K_OBJ > 1)
            if a.r2:
                e.interact(mask_mid, T // 2, scribble=K_OBJ > 1)
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_lane(lane, mask, idx, fresh):
        """Host thread `lane`: its videos (j = lane, lane+S, ...) one after another on its own stream
        (ctypes releases the GIL)."""
        torch.cuda.set_device(local)
        fr, out, prev, same = 0, None, {}, True
        with torch.cuda.stream(streams[lane]):
            for n, j in enumerate(range(lane, a.steps, S)):
                e = pool[lane][n % per_lane]
                if fresh:
                    e.reset()
                out = e.interact(mask, idx, scribble=K_OBJ > 1)
                fr += e.stats()["frames"]
                if fresh:                                   # same engine = same clip: repeats must be bit-identical
                    if n % per_lane in prev:
                        same = same and np.array_equal(prev[n % per_lane], out)
                    prev[n % per_lane] = out
        return fr, out, same, (len(range(lane, a.steps, S)) - 1) % per_lane

    def run_all(mask, idx, fresh=True):
        if S == 1:
            return [run_lane(0, mask, idx, fresh)]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(S) as ex:
            return list(ex.map(lambda l: run_lane(l, mask, idx, fresh), range(S)))

    barrier()
    t0 = time.perf_counter()
    res = run_all(mask0, 0)
    torch.cuda.synchronize()
    dt_r1 = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    last = res[0][1]
    # determinism under concurrency: (i) repeated videos of one engine agreed bit for bit inside the timed region,
    # (ii) lane 0's last video, re-run now with nothing else in flight, reproduces its concurrent result
    with torch.cuda.stream(streams[0]):
        e = pool[0][res[0][3]]
        e.reset()
        solo = e.interact(mask0, 0, scribble=K_OBJ > 1)
    lanes_identical = all(r[2] for r in res) and np.array_equal(solo, last)

    # Roofline leg: the same step (fresh engine, interact(mask,0)) on ONE stream with per-launch HIP events on
    # that stream.  Kept apart from the timed region on purpose: (i) two events per launch cost ~13 % of
    # wall time, (ii) with several videos in flight kernels overlap and a per-launch duration no longer
    # measures the kernel.  `rocprofv3 --kernel-trace --stats -- python bench.py --streams 1 ...` sees the
    # same solo launches (profiles/).
    prof = None
    if not a.no_profile:
        prof = {}
        roof_frames, t_roof = 0, 0.0
        la_saved = os.environ.get("STCN_LOOKAHEAD")
        os.environ["STCN_LOOKAHEAD"] = "0"            # solo launches only: no side-stream overlap in this leg
        for _ in range(max(1, a.roof_steps)):
            e = InferenceCore(prop, fuse, img, K_OBJ, mem_freq=a.mem_freq)
            e.set_profiling(True)
            torch.cuda.synchronize()
            tr = time.perf_counter()
            e.interact(mask0, 0, scribble=K_OBJ > 1)
            torch.cuda.synchronize()
            t_roof += time.perf_counter() - tr
            roof_frames += e.stats()["frames"]
            for cls, v in e.kernel_profile().items():
                acc = prof.setdefault(cls, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
                for k_ in acc:
                    acc[k_] += v[k_]
            del e
        os.environ["STCN_LOOKAHEAD"] = la_saved

    # Extra leg (not the headline): the same timed region with the convs on the f16 MFMA pipe through the
    # 3-term fp16 hi/lo operand split (fp32 accumulate, fp32-grade products; DESIGN.md "f16x3"), and how many
    # mask pixels differ from the exact-fp32 run above.
    extra = None
    if a.f16x3_leg and world == 1 and os.environ.get("STCN_PRECISION") is None:
        pool_main = pool
        os.environ["STCN_PRECISION"] = "f16x3"
        prop_main, prop = prop, PropagationNetwork()
        prop.load_state_dict(psd)
        pool = [[make(l, j) for j in range(per_lane)] for l in range(S)]
        run_all(mask0, 0)                                  # warm-up (one video per lane)
        torch.cuda.synchronize()
        tx = time.perf_counter()
        resx = run_all(mask0, 0)
        torch.cuda.synchronize()
        dtx = time.perf_counter() - tx
        extra = {"precision": "f16x3: fp16 hi/lo split operands, 3 x v_mfma_f32_32x32x16_f16 per K step, fp32 accumulate",
                 "frames_per_s_rank0": sum(r[0] for r in resx) / dtx,
                 "mask_pixels_differing_from_fp32_run": int((resx[0][1] != last).sum()),
                 "mask_pixels_total": int(last.size)}
        os.environ.pop("STCN_PRECISION")
        pool, prop = pool_main, prop_main

    r2 = None
    if a.r2:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res2 = run_all(mask_mid, T // 2, fresh=False)     # second interaction on the engines' last videos
        torch.cuda.synchronize()
        r2 = sum(r[0] for r in res2) / (time.perf_counter() - t1)

    # whole-job numbers: max time over ranks, frames summed over ranks
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ff = torch.tensor([frames], dtype=torch.float64, device=red_dev)
        dist.all_reduce(ff, op=dist.ReduceOp.SUM)
        dt_all, frames_all = float(tt.item()), float(ff.item())
    else:
        dt_all, frames_all = dt, float(frames)

    # per-video J&F rows of the last video of each rank, gathered once (the path's only exchange step)
    sc = metrics.sequence_scores_gpu((gt[0, :, 0] > 0.5).cuda(), torch.from_numpy(last == 1).cuda())   # HIP J/F kernel
    row = np.array([[rank, sc[1:, 0].mean(), sc[1:, 1].mean(), sc[1:, 2].mean()]], np.float32)
    rows = shard.gather_rows(row, 4)

    if rank == 0:
        out = {
            "metric": "frames/sec STCN mask-propagate 480p 1-obj",
            "value": frames_all / dt_all, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt_all / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if os.environ.get("STCN_PRECISION") != "f16x3" else "f16x3 (fp16 hi/lo split operands, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"DAVIS-17-val-shaped {H}x{W} (padded {pool[0][0].nh}x{pool[0][0].nw}) {'single' if K_OBJ == 1 else K_OBJ}-object "
                                   f"STCN propagate: fresh engine, interact(mask,0), T={T} frames/video, "
                                   f"mem_freq={a.mem_freq}, top_k=50; one video per step per GPU",
                       "frames_per_step": T - 1, "videos_per_gpu": a.steps, "sharding": f"videos x{world}", "streams_per_gpu": S,
                       "key_lookahead": int(os.environ["STCN_LOOKAHEAD"]),
                       "clips": f"{S * per_lane} distinct synthetic clips (one per pooled engine)",
                       "weights": "synthetic recipe seed 0 (no checkpoints offline)"},
            "ms_per_frame": 1e3 * dt_all / (frames_all / world),
            "jf_rows_rank_J_F_JF": rows.round(4).tolist(),
            "concurrent_videos_bit_identical": bool(lanes_identical),
        }
        if r2 is not None:
            out["r2_frames_per_s_rank0"] = r2
        if extra is not None:
            out["extra_f16x3_leg"] = extra
        if prof is not None:
            conv = prof["conv"]
            ach = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                               "kernel": "conv_gemm_kernel (fp32 implicit-GEMM conv, v_mfma_f32_32x32x2_f32)",
                               "launches": conv["launches"], "avg_launch_ms": conv["ms"] / max(conv["launches"], 1),
                               "flop_per_launch_avg": conv["flops"] / max(conv["launches"], 1)}
            hb = prof.pop("conv_hbm_bound")                      # subset of "conv": launches below 19.7 FLOP/B
            tot_ms = sum(v["ms"] for v in prof.values())
            out["kernel_time_share"] = {c: round(v["ms"] / tot_ms, 4) for c, v in prof.items() if v["ms"] > 0}
            if hb["ms"] > 0 and conv["ms"] > hb["ms"]:
                # the same kernel in its two regimes (the headline `roofline` above is over ALL its launches)
                mf = (conv["flops"] - hb["flops"]) / ((conv["ms"] - hb["ms"]) * 1e-3) / 1e12
                gb = hb["bytes"] / (hb["ms"] * 1e-3) / 1e9
                out["roofline_by_regime"] = {
                    "mfma_bound_launches": {"launches": conv["launches"] - hb["launches"], "time_share_of_conv": round(1 - hb["ms"] / conv["ms"], 4),
                                            "achieved": mf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mf / FP32_MFMA_PEAK_TFLOPS},
                    "hbm_bound_launches": {"launches": hb["launches"], "time_share_of_conv": round(hb["ms"] / conv["ms"], 4),
                                           "achieved": gb, "peak": 8000.0, "unit": "GB/s", "frac": gb / 8000.0,
                                           "what": "conv launches under 19.7 FLOP/B of algorithmic intensity (1x1 channel expansions, stems)"}}
            # HBM-side bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate runs)
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
                out["roofline"]["traffic"] = pmc["conv_gemm_traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc passes of this workload at T=30)"
            except OSError:
                pass
            out["roofline"]["algorithmic_bytes_per_launch"] = conv["bytes"] / max(conv["launches"], 1)
            out["roofline"]["leg"] = f"{max(1, a.roof_steps)} video(s), 1 stream, HIP events per launch"
            out["device_busy_frac_roofline_leg"] = tot_ms * 1e-3 / t_roof
            out["algorithmic_gflop_per_frame"] = sum(v["flops"] for v in prof.values()) / roof_frames / 1e9
            # chip-level view of the timed region: all algorithmic FLOP of the path / wall time
            out["timed_region_tflops"] = out["algorithmic_gflop_per_frame"] * 1e-3 * frames / dt_r1
        else:
            out["roofline"] = None
        if world == 1 and a.cpu_frames > 1:
            out["cpu_baseline"], sample = cpu_baseline(psd, fsd, H, W, a.cpu_frames, a.mem_freq)
            out["parity_vs_cpu_oracle"] = parity_vs_oracle(prop, fuse, sample, a.mem_freq)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
