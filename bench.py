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

Output: ONE JSON line on rank 0 (see the task contract; `value_repeats` = the timed region run three times, `frame_kernel_ms` /
`kernel_ms_per_frame_by_class` = the solo leg's kernel time, `host_*` = what the lanes cost the host) with two extra objects:
  roofline     - dominant kernel (fp32 implicit-GEMM conv): algorithmic FLOP of all conv launches in the
                 timed region / their summed device time (HIP events on the engine stream) vs the
                 fp32 MFMA peak 157.3 TFLOP/s (MI355X_MICROARCH.md).
  cpu_baseline - the CPU oracle (oracle/stcn_oracle.py, a port validated against the reference) timed on
                 this box's host cores on a bounded sample of BASELINE config 1 (both rounds), rank 0 at N=1 only.
Parity legs in the same line (CPU oracle AND HIP engine on the same inputs): parity_vs_cpu_oracle (BASELINE config 1: T=82, two rounds), parity_long_clip
(T=104), parity_session (8 rounds of the reference's oracle mask policy at 480p), config3.parity_vs_cpu_oracle (k=5, all pixels, multi-object recipe).
Further objects (not the headline): roofline_memread (the space-time memory read at config-3 scale, MFMA fraction on
2*N*Q*64 and GB/s on SURVEY 8(d)'s algorithmic bytes), config3 (k=5, mem_freq=1, T=104: the full-bank multi-object
case, + a portrait leg), roofline_r2 (a second interaction: cached keys + fusion), davis_val (30 DAVIS-val lengths, every 6th clip portrait, LPT
over ranks), drivers (configs 4 / 5 end to end at N = 1: fq_driver / eval_driver rounds/s on a synthetic dataset tree).  Every parity leg carries
coded `bound` / `within_bound` (clip_bound / frame_bound below).  Real data: when ./model_weights/mivos/{stcn,fusion}.pth and
./data/DAVIS_17 exist the same line is measured on real DAVIS-17-val clips with "data": "real" (there is no network
here, so the default is the synthetic recipe and the line says so).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.bench_legs import (DAVIS_VAL_LENGTHS, FP32_MFMA_PEAK_TFLOPS, NOISE_X, clip_bound, config3_leg, config3_parity, cpu_baseline,  # noqa: E402,F401
                              davis_val_leg, drivers_leg, frame_bound, long_clip_parity, memread_roofline, parity_vs_oracle, power_leg,
                              r2_roofline, real_inputs, ref_self_noise, roofline_objects, session_leg, session_parity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default = WORLD_SIZE or 1")
    ap.add_argument("--steps", type=int, default=None, help="videos per rank (uniform workload, default 12) / samples of the whole job (davis-val, default 30)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=66, help="frames per video (DAVIS-17 val mean length ~66)")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=854)
    ap.add_argument("--mem-freq", type=int, default=5)
    ap.add_argument("--objects", type=int, default=1, help="k>1: multi-object engine via the scribble/(k+1)-channel path (config 3)")
    ap.add_argument("--cpu-frames", type=int, default=82,
                    help="frames of the bounded CPU-oracle sample: BASELINE config 1 is T=82, interact(0) then interact(41) = the "
                         "default (about 60 s of host time); smaller = the same two rounds on a shorter clip; 0 = skip")
    ap.add_argument("--workload", choices=("uniform", "davis-val"), default="uniform",
                    help="uniform (default, the headline): every rank runs --steps videos of --frames frames (weak scaling).  davis-val: "
                         "the whole JOB runs --steps samples (default 30) whose lengths are the 30 DAVIS-2017-val sequence lengths "
                         "(34..104 frames), assigned to the ranks by LPT (strong scaling: SURVEY 8(d) config 2 / 8(e))")
    ap.add_argument("--davis-max-frames", type=int, default=104, help="clamp the DAVIS-val lengths (tests run tiny clips)")
    ap.add_argument("--no-davis-val", dest="davis_val", action="store_false",
                    help="skip the extra davis_val object (the 30-length workload measured beside the uniform headline)")
    ap.add_argument("--config3-oracle-frames", type=int, default=24,
                    help="frames of the config-3 parity sample (k objects, mem_freq=1) run on the CPU oracle AND the HIP engine; 0 = skip")
    ap.add_argument("--parity-long-frames", type=int, default=104,
                    help="frames of the long-clip parity leg (k=1, CPU oracle AND HIP engine, ~35 s of host time at 104); 0 = skip")
    ap.add_argument("--parity-session-rounds", type=int, default=8,
                    help="rounds of the annotation-session parity leg (oracle mask policy, CPU oracle AND HIP engine); 0 = skip")
    ap.add_argument("--parity-session-frames", type=int, default=34, help="clip length of the session parity leg (shortest DAVIS-val clip)")
    ap.add_argument("--no-config3", dest="config3", action="store_false",
                    help="skip the extra config-3 leg (k=5 objects, mem_freq=1, T=104: full-length bank)")
    ap.add_argument("--config3-frames", type=int, default=104)
    ap.add_argument("--config3-objects", type=int, default=5)
    ap.add_argument("--no-memread-roofline", dest="memread_roofline", action="store_false",
                    help="skip the memory-read roofline leg (stcn_bench_memory_read at config-3 bank sizes)")
    ap.add_argument("--data", choices=("auto", "synthetic", "real"), default="auto",
                    help="auto: real DAVIS-17 val clips + checkpoints when present on the box, else synthetic")
    ap.add_argument("--no-profile", action="store_true", help="skip the roofline leg (roofline = null)")
    ap.add_argument("--roof-steps", type=int, default=1, help="videos of the profiled single-stream roofline leg")
    ap.add_argument("--streams", type=int, default=4, help="videos in flight per GPU (one host thread + HIP stream each)")
    ap.add_argument("--value-repeats", type=int, default=3, help="the timed region is run this many times; `value` is the first, value_repeats reports min / median / max")
    ap.add_argument("--no-power", dest="power", action="store_false", help="skip the power leg (one more timed region under a rocm-smi sampler)")
    ap.add_argument("--no-drivers", dest="drivers", action="store_false",
                    help="skip the `drivers` leg (eval_driver + fq_driver rounds/s on a synthetic 480p dataset tree: configs 4 / 5 at N = 1)")
    ap.add_argument("--driver-videos", type=int, default=8)
    ap.add_argument("--driver-frames", type=int, default=40)
    ap.add_argument("--no-session", dest="session", action="store_false",
                    help="skip the `session` leg (configs 4 / 5 as resident-clip annotation sessions: 8 and 60 oracle rounds, lane sweep)")
    ap.add_argument("--session-videos", type=int, default=4)
    ap.add_argument("--no-r2", dest="r2", action="store_false",
                    help="skip the extra R2 number (a second interaction: cached keys + fusion; reported, not the headline)")
    return ap.parse_args()

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
    if a.steps is None:
        a.steps = 12 if a.workload == "uniform" else len(DAVIS_VAL_LENGTHS)
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
    # Real weights / clips when the box has them (reference: eval_annotation_method.py:51-64 loads
    # ./model_weights/mivos/{stcn,fusion}.pth and ./data/DAVIS_17); there is no network here, so the default is synthetic.
    real = real_inputs(a) if a.data in ("auto", "real") else None
    if a.data == "real" and real is None:
        raise SystemExit("--data real: ./model_weights/mivos/{stcn,fusion}.pth and ./data/DAVIS_17/trainval are not on this box")
    if real is not None:
        psd, fsd = real["prop_sd"], real["fuse_sd"]
    else:
        psd, fsd = synth.recipe_state_dict(prop), synth.recipe_state_dict(fuse)
    prop.load_state_dict(psd)
    fuse.load_state_dict(fsd)

    T, H, W = a.frames, a.height, a.width
    K_OBJ = a.objects
    if real is not None:
        img, gt = real["rgb"].cuda(), real["gt"]                  # first val sample: [1,T,3,H,W], [1,T,1,H,W]
        T, H, W, K_OBJ = img.shape[1], img.shape[-2], img.shape[-1], 1
    else:
        img = synth.synthetic_clip(T, H, W).cuda()
        gt = synth.synthetic_mask(T, H, W, K_OBJ)

    def as_input(m):         # k == 1: [1,1,H,W] without bg row; k > 1: bg row first + scribble=True (reference semantics)
        return m.clone() if K_OBJ == 1 else torch.cat([1 - m.sum(0, keepdim=True).clamp(0, 1), m], 0)

    mask0 = as_input(gt[:, 0])
    mask_mid = as_input(gt[:, T // 2])

    # one HIP stream per in-flight video; engines are bound to the stream they are created under
    # davis-val as the headline: the uniform region below only warms the kernels up (one video per lane)
    n_uniform = a.steps if a.workload == "uniform" else max(1, min(a.steps, a.streams))
    S = max(1, min(a.streams, n_uniform))
    # engine knob: key-encoder look-ahead on a side stream helps a single video in flight (+6 %) but only
    # adds contention when several videos already overlap
    # (given to every engine explicitly - stcn_engine_create_ex - instead of through os.environ, which other lanes' host threads
    # would read while a later leg changes it; an STCN_LOOKAHEAD set by the user still decides the lanes' value)
    la_main = int(os.environ.get("STCN_LOOKAHEAD", "0" if S > 1 else "2"))
    eo_main = {"lookahead": la_main}
    streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
    # A bounded pool of engines (each ~5.5 GB at T=66) serves any --steps: lane l owns engines pool[l]; a video
    # takes the lane's next engine and resets it first (reset = what a fresh InferenceCore would hold).  Every engine of
    # the pool holds a DIFFERENT clip (the synthetic scene under its own noise), so concurrent videos never share inputs; a repeated video
    # of one engine must reproduce its previous result bit for bit while other clips run beside it.
    per_lane = 2
    def variant(n):          # clip 0 = the recipe clip; clip n = the same scene under its own seeded sensor noise
        if n == 0:                                  # (real data: clip n = the first val clip under that noise too)
            return img
        g = torch.Generator(device="cuda").manual_seed(1000 + n)
        return img + 0.15 * torch.randn(img.shape, generator=g, device="cuda")

    clips = [[variant(l * per_lane + j) for j in range(per_lane)] for l in range(S)]

    def make(lane, j=0):
        with torch.cuda.stream(streams[lane]):
            return InferenceCore(prop, fuse, clips[lane][j], K_OBJ, mem_freq=a.mem_freq, engine_options=eo_main)

    pool = [[make(l, j) for j in range(per_lane)] for l in range(S)]
    pad_hw = (pool[0][0].nh, pool[0][0].nw)
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

    import itertools
    import threading

    host_acct = {"enqueue_s": 0.0, "videos": 0, "lane_cpu_s": [0.0] * S}      # summed over the timed regions (R1 steps only)
    acct_lock = threading.Lock()

    def run_lane(lane, mask, idx, fresh, ticket, lock, keep=True):
        """Host thread `lane`: takes the next video of the step counter whenever its previous one is done (ctypes releases the
        GIL), runs it on its own stream with its own engines; no lane idles while videos are left."""
        torch.cuda.set_device(local)
        fr, out, outs, n = 0, None, [], 0
        cpu0, enq = time.thread_time(), 0.0
        with torch.cuda.stream(streams[lane]):
            while True:
                with lock:
                    j = next(ticket)
                if j >= n_uniform:
                    break
                e = pool[lane][n % per_lane]
                if fresh:
                    e.reset()
                out = e.interact(mask, idx, scribble=K_OBJ > 1)
                enq += e.last_enqueue_s
                fr += e.stats()["frames"]
                if fresh and keep:
                    outs.append((n % per_lane, out))        # compared AFTER the timed region (27 MB each: harness work, not the path's)
                n += 1
        if fresh:
            with acct_lock:
                host_acct["enqueue_s"] += enq
                host_acct["videos"] += n
                host_acct["lane_cpu_s"][lane] += time.thread_time() - cpu0
        return fr, out, outs, (n - 1) % per_lane if n else 0

    def repeats_identical(outs):
        """same engine = same clip: the repeated videos of a lane must be bit-identical"""
        prev, same = {}, True
        for slot, o in outs:
            if slot in prev:
                same = same and np.array_equal(prev[slot], o)
            prev[slot] = o
        return same

    def run_all(mask, idx, fresh=True, keep=True):
        ticket, lock = itertools.count(), threading.Lock()
        if S == 1:
            return [run_lane(0, mask, idx, fresh, ticket, lock, keep)]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(S) as ex:
            return list(ex.map(lambda l: run_lane(l, mask, idx, fresh, ticket, lock, keep), range(S)))

    barrier()
    t0 = time.perf_counter()
    res = run_all(mask0, 0)
    torch.cuda.synchronize()
    dt_r1 = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    l0 = next(l for l in range(S) if res[l][1] is not None)      # a lane that ran at least one video (lane 0 unless it lost every ticket)
    last = res[l0][1]
    # determinism under concurrency: (i) repeated videos of one engine agreed bit for bit inside the timed region,
    # (ii) that lane's last video, re-run now with nothing else in flight, reproduces its concurrent result
    with torch.cuda.stream(streams[l0]):
        e = pool[l0][res[l0][3]]
        e.reset()
        solo = e.interact(mask0, 0, scribble=K_OBJ > 1)
    lanes_identical = all(repeats_identical(r[2]) for r in res) and np.array_equal(solo, last)
    del res

    # The same timed region twice more (value_repeats): `value` stays the FIRST region - the contract's K steps - and the spread of
    # three identical regions on this box is printed beside it, so that a reader can tell a change from the box's own noise.
    rep_rates = [frames / dt]
    for _ in range(max(0, a.value_repeats - 1)):
        barrier()
        tq = time.perf_counter()
        rr = run_all(mask0, 0, keep=False)
        torch.cuda.synchronize()
        barrier()
        rep_rates.append(sum(r[0] for r in rr) / (time.perf_counter() - tq))

    # Power leg (rank 0): the timed region once more while a thread samples rocm-smi (read-only): the conv GEMMs of this path run at the
    # board's power cap (profiles/r05_power_by_kernel.txt), so socket power belongs beside frames/s.  Its own region: nothing perturbs
    # `value` / value_repeats; null when rocm-smi is missing or prints something else.
    power = None
    if a.power and rank == 0:
        power = power_leg(lambda: (run_all(mask0, 0, keep=False), torch.cuda.synchronize()))

    # Roofline leg: the same step (fresh engine, interact(mask,0)) on ONE stream with per-launch HIP events on
    # that stream.  Kept apart from the timed region on purpose: (i) two events per launch cost ~13 % of
    # wall time, (ii) with several videos in flight kernels overlap and a per-launch duration no longer
    # measures the kernel.  `rocprofv3 --kernel-trace --stats -- python bench.py --streams 1 ...` sees the
    # same solo launches (profiles/).
    prof = None
    if not a.no_profile:
        prof = {}
        roof_frames, t_roof = 0, 0.0
        for _ in range(max(1, a.roof_steps)):
            e = InferenceCore(prop, fuse, img, K_OBJ, mem_freq=a.mem_freq, engine_options={"lookahead": 0})   # solo launches only: no side-stream overlap in this leg
            e.set_profiling(True)
            torch.cuda.synchronize()
            tr = time.perf_counter()
            e.interact(mask0, 0, scribble=K_OBJ > 1)
            torch.cuda.synchronize()
            t_roof += time.perf_counter() - tr
            roof_frames += e.stats()["frames"]
            for cls, v in e.kernel_profile().items():
                acc = prof.setdefault(cls, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, exec_flops=0.0))
                for k_ in acc:
                    acc[k_] += v[k_]
                if cls == "conv_hbm_bound":
                    for k_ in ("wino2_flops", "wino4_flops"):
                        acc[k_] = acc.get(k_, 0.0) + v[k_]
            del e

    r2, r2_roof = None, None
    if a.r2:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res2 = run_all(mask_mid, T // 2, fresh=False)     # second interaction on the engines' last videos
        torch.cuda.synchronize()
        r2 = sum(r[0] for r in res2) / (time.perf_counter() - t1)
        if not a.no_profile and rank == 0:
            r2_roof = r2_roofline(prop, fuse, img, mask0, mask_mid, T, a.mem_freq, K_OBJ > 1)
            r2_roof["frames_per_s_videos_in_flight"] = r2

    # Extra leg: the DAVIS-val-shaped workload (30 samples, lengths 34..104, LPT over the ranks): what SURVEY 8(d) calls
    # config 2 and 8(e) names as the scaling risk (tail imbalance).  The pooled engines are released first (HBM).
    del pool, clips
    torch.cuda.empty_cache()
    dv = None
    if (a.davis_val or a.workload == "davis-val") and real is None and K_OBJ == 1:
        dv = davis_val_leg(prop, fuse, a, H, W, rank, world, local, streams, barrier, eo_main)

    # Extra leg: BASELINE config 3 at its stated size - one multi-object engine (k objects through the scribble /
    # (k+1)-channel path, the only multi-object path of the reference: inference_core.py:220-233), mem_freq = 1 (every
    # frame enters the bank: full-length memory, bank rows up to T * 1620), T = 104 (the longest clip, download_data.py:42).
    cfg3 = None
    if a.config3 and world == 1 and real is None:
        # the multi-object weight recipe (synth.RECIPES[2]): same architecture and kernel launches, weights under which the
        # decoder separates several objects - the parity statement then covers (almost) every pixel
        prop3, fuse3 = PropagationNetwork(), FusionNet()
        psd3, fsd3 = synth.recipe_state_dict(prop3, 2), synth.recipe_state_dict(fuse3, 2)
        prop3.load_state_dict(psd3)
        fuse3.load_state_dict(fsd3)
        cfg3 = config3_leg(prop3, fuse3, a.config3_frames, H, W, a.config3_objects)
        if a.config3_frames >= 16:         # "480 x {854 or clip width}" - the same workload on a portrait clip (W x H), half the length
            pl = config3_leg(prop3, fuse3, a.config3_frames // 2, W, H, a.config3_objects)
            cfg3["portrait"] = {k_: pl[k_] for k_ in ("workload", "frames_per_s", "ms_per_frame", "frames", "repeat_bit_identical", "kernel_time_share", "conv_frac_of_fp32_mfma_peak")}
        if a.config3_oracle_frames > 1:
            cfg3["parity_vs_cpu_oracle"] = config3_parity(prop3, fuse3, psd3, fsd3, a.config3_oracle_frames, H, W, a.config3_objects)
            cfg3["mask_iou_vs_cpu_oracle"] = cfg3["parity_vs_cpu_oracle"]["mask_iou_vs_cpu_oracle"]
        del prop3, fuse3
    mr_roof = memread_roofline(a.config3_objects) if (a.memread_roofline and world == 1) else None
    ses = None
    if a.session and world == 1 and real is None and rank == 0 and K_OBJ == 1:
        try:
            ses = {"config4_shape": session_leg(prop, fuse, H, W, 40, 8, "j", a.session_videos, tag="(generate_fq_dataset.py:69: 8 rounds, selection by J)"),
                   "config5_shape": session_leg(prop, fuse, H, W, 66, 60, "j_and_f", max(2, a.session_videos // 2), lanes_list=(1, 2),
                                                tag="(eval_annotation_method.py:30: 60 rounds, J&F)")}
        except Exception as ex:                                   # an extra leg: never fail the line for it
            ses = {"error": f"{type(ex).__name__}: {ex}"}
    drv = None
    if a.drivers and world == 1 and real is None and rank == 0:
        try:
            drv = drivers_leg(prop, fuse, H, W, a.driver_videos, a.driver_frames)
        except Exception as ex:                                   # an extra leg: never fail the line for it
            drv = {"error": f"{type(ex).__name__}: {ex}"}

    # whole-job numbers: max time over ranks, frames summed over ranks
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ff = torch.tensor([frames], dtype=torch.float64, device=red_dev)
        dist.all_reduce(ff, op=dist.ReduceOp.SUM)
        dt_all, frames_all = float(tt.item()), float(ff.item())
    else:
        dt_all, frames_all = dt, float(frames)

    if dist is not None:                                  # repeats: every rank ran the same regions; whole-job rate = sum over ranks
        rr_t = torch.tensor(rep_rates, dtype=torch.float64, device=red_dev)
        dist.all_reduce(rr_t, op=dist.ReduceOp.SUM)
        rep_rates = [float(v) for v in rr_t.tolist()]
        rep_rates[0] = frames_all / dt_all                # the headline itself: frames of all ranks / the slowest rank's time

    # per-video J&F rows of the last video of each rank, gathered once (the path's only exchange step)
    sc = metrics.sequence_scores_gpu((gt[0, :, 0] > 0.5).cuda(), torch.from_numpy(last == 1).cuda())   # HIP J/F kernel
    row = np.array([[rank, sc[1:, 0].mean(), sc[1:, 1].mean(), sc[1:, 2].mean()]], np.float32)
    rows = shard.gather_rows(row, 4)

    if rank == 0:
        out = {
            "metric": "frames/sec STCN mask-propagate 480p 1-obj",
            "value": frames_all / dt_all, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt_all / n_uniform,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "real" if real is not None else "synthetic",
            "config": {"workload": f"{'DAVIS-17 val clip ' + real['name'] if real is not None else 'DAVIS-17-val-shaped'} {H}x{W} (padded {pad_hw[0]}x{pad_hw[1]}) {'single' if K_OBJ == 1 else K_OBJ}-object "
                                   f"STCN propagate: fresh engine, interact(mask,0), T={T} frames/video, "
                                   f"mem_freq={a.mem_freq}, top_k=50; one video per step per GPU",
                       "frames_per_step": T - 1, "videos_per_gpu": n_uniform, "sharding": f"videos x{world}", "streams_per_gpu": S,
                       "key_lookahead": la_main,
                       "clips": f"{S * per_lane} distinct synthetic clips (one per pooled engine)",
                       "weights": "model_weights/mivos/stcn.pth + fusion.pth" if real is not None else "synthetic recipe seed 0 (no checkpoints offline)"},
            "ms_per_frame": 1e3 * dt_all / (frames_all / world),
            "jf_rows_rank_J_F_JF": rows.round(4).tolist(),
            # the path's one exchange step really ran over every rank: rows the gather returned, and the backend that carried it ("nccl" = RCCL)
            "ranks_seen": int(len(rows)), "backend": (dist.get_backend() if dist is not None else None),
            "concurrent_videos_bit_identical": bool(lanes_identical),
            "power": power,
            "value_repeats": {"frames_per_s": [round(v, 2) for v in rep_rates], "min": min(rep_rates), "median": sorted(rep_rates)[len(rep_rates) // 2],
                              "max": max(rep_rates), "what": "the timed region run back to back on this box; `value` is the first"},
            # rank 0's host side of the timed regions: what one lane's thread costs the host (8 ranks x (lanes + prefetch threads) share one host at N = 8)
            "host_enqueue_ms_per_video": 1e3 * host_acct["enqueue_s"] / max(host_acct["videos"], 1),
            "host_cpu_s_per_lane": [round(v, 3) for v in host_acct["lane_cpu_s"]],
            "host_cpu_s_per_video": sum(host_acct["lane_cpu_s"]) / max(host_acct["videos"], 1),
            "host_cores": os.cpu_count(),
            # interact() downloads of this process that did NOT get a pinned block (per-core budget of 4 blocks, 4 GB per process): 0 = every
            # download of every leg took the one-DMA path
            "pageable_downloads": __import__("eva_vos_amd.inference_core", fromlist=["x"]).pageable_downloads(),
        }
        if r2 is not None:
            out["r2_frames_per_s_rank0"] = r2
        if r2_roof is not None:
            out["roofline_r2"] = r2_roof
        if dv is not None:
            out["davis_val"] = dv
            if a.workload == "davis-val":      # this workload as the headline: fixed total work, strong scaling
                out.update(value=dv["frames_per_s"], scaling="strong", ms_per_step=1e3 * dv["slowest_rank_s"] / dv["samples"],
                           steps=dv["samples"], uniform_region_frames_per_s=frames_all / dt_all)
                out["config"]["workload"] = dv["workload"]
                out["config"]["sharding"] = f"LPT over {world} rank(s)"
        if prof is not None:
            out.update(roofline_objects(prof, roof_frames, t_roof, frames, dt_r1, a.roof_steps))
        else:
            out["roofline"] = None
        if cfg3 is not None:
            out["config3"] = cfg3
        if mr_roof is not None:
            out["roofline_memread"] = mr_roof
        if ses is not None:
            out["session"] = ses
        if drv is not None:
            out["drivers"] = drv
        if world == 1 and a.cpu_frames > 1:
            out["cpu_baseline"], sample = cpu_baseline(psd, fsd, H, W, a.cpu_frames, a.mem_freq)
            out["parity_vs_cpu_oracle"] = parity_vs_oracle(prop, fuse, sample, a.mem_freq, eo_main)
            if a.parity_long_frames > 1 and real is None:
                out["parity_long_clip"] = long_clip_parity(prop, fuse, psd, fsd, H, W, a.parity_long_frames, a.mem_freq, eo_main)
            if a.parity_session_rounds > 0 and real is None:
                out["parity_session"] = session_parity(prop, fuse, psd, fsd, H, W, a.parity_session_frames, a.parity_session_rounds, a.mem_freq, eo_main)
        else:
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = ("the CPU oracle is timed on rank 0 at N=1 only (task contract); see the N=1 line" if world > 1
                                        else "skipped (--cpu-frames 0)")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
