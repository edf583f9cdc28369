"""The legs of bench.py (the repo root's benchmark contract): each function measures ONE thing and returns a JSON-able dict; bench.py's
main() runs the timed region, calls the legs it is asked for and assembles the line.  Kept apart from bench.py so that the contract
(arguments, the timed region, the JSON keys the driver reads) is readable in one screen there.  The parity legs import the CPU oracle
(oracle/: test infrastructure) - they run after the timed region and are never the thing measured as `value`."""
from __future__ import annotations

import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (matrix)"


def ref_self_noise(*tags):
    """Rows of tests/golden/selfnoise.npz (committed DATA: the reference against itself at 1 / 2 / 4 / 8 intra-op threads, captured by
    oracle/gen_golden.py --selfnoise): element-wise maximum over the rounds of the given fixtures = (clip 1-IoU, max |dprob|,
    p99.9 |dprob|, differing pixels, worst per-frame 1-IoU)."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "selfnoise.npz"))
    return np.max(np.concatenate([d[t] for t in tags], 0), 0)


NOISE_X = 1.5      # allowance over the reference's own per-frame spread (round 6: 1.5, was 3 - see tests/conftest.py)


def clip_bound(noise=None):
    """Per-object mask bound on a whole clip (1 - IoU): the north_star's plain 1e-3, for every k (round 6: every leg of this bench measures
    <= 2.9e-4 at 480p; the reference's own clip-level spread there is <= 6e-4, so no allowance over it is needed)."""
    return 1e-3


def frame_bound(noise, union_px):
    """Per-(object, frame) mask bound (1 - IoU), the form of tests/conftest.py::frame_bound: the north_star 1e-3, or 1.5 x the reference's own
    worst per-frame difference between its thread counts, or - small objects - two pixels, whichever is larger."""
    return max(1e-3, NOISE_X * float(noise[4]), 2.0 / max(float(union_px), 1.0))


def cpu_baseline(psd, fsd, H, W, frames, mem_freq):
    """Oracle (kind 'port') on the host cores, BASELINE config 1's two rounds on a `frames`-long clip of the same shape:
    R1 = interact(mask, 0) on a fresh core, R2 = interact(mask, frames // 2) (cached keys, fusion between the two frames)."""
    from eva_vos_amd import synth
    from oracle.stcn_oracle import OracleCore
    from oracle import stcn_oracle as O
    img, msk = synth.synthetic_clip(frames, H, W), synth.synthetic_mask(frames, H, W, 1)
    # pick the intra-op thread count that is fastest on this host (hundreds of threads thrash on the
    # small GEMMs of the path): one key-encoder pass per candidate
    host_cores = os.cpu_count() or 1
    fw = O.fold_bn(psd)
    x0, _ = O.pad16(img[:, 0])
    best_t, best = 1, float("inf")
    for nt in sorted({n for n in (8, 16, 32, 64, host_cores) if n <= host_cores}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        O.encode_key(fw, x0)
        el = time.perf_counter() - t0
        if el < best:
            best_t, best = nt, el
    torch.set_num_threads(best_t)
    core = OracleCore(psd, fsd, img, 1, mem_freq=mem_freq)
    mid = frames // 2
    t0 = time.perf_counter()
    ref1 = core.interact(msk[:, 0], 0).copy()
    t1 = time.perf_counter()
    ref2 = core.interact(msk[:, mid], mid).copy()
    t2 = time.perf_counter()
    n1, n2 = frames - 1, frames - 1
    base = dict(value=n1 / (t1 - t0), unit="frames/s", cores=best_t, kind="port", host_cores=host_cores, threads=best_t,
                r1_frames_per_s=n1 / (t1 - t0), r2_frames_per_s=n2 / (t2 - t1), torch=torch.__version__,
                sample=f"oracle OracleCore on a {frames}-frame {H}x{W} synthetic clip, k=1, mem_freq={mem_freq} (BASELINE config 1 is "
                       f"T=82): R1 interact(mask,0) {n1} frames in {t1 - t0:.1f} s, R2 interact(mask,{mid}) {n2} frames in "
                       f"{t2 - t1:.1f} s; {best_t} intra-op threads (fastest of 8/16/32/64/all) on {host_cores} host cores; "
                       f"value = R1")
    return base, (img, msk, ref1, ref2)


def parity_vs_oracle(prop, fuse, sample, mem_freq, eo=None):
    """The HIP engine on the clip the CPU oracle just processed, both rounds: mask IoU between the two, frames/s of the same
    two interactions on the GPU (one video in flight), and J&F of each against the synthetic ground truth - the CPU masks
    scored by the CPU restatement of interactions/metrics.py, the HIP masks by the HIP J/F kernel (north_star: masks within
    1e-3 IoU, J&F within 0.1 of the CPU reference)."""
    from eva_vos_amd import metrics
    from mivos.inference_core import InferenceCore
    img, msk, ref1, ref2 = sample
    mid = img.shape[1] // 2
    core = InferenceCore(prop, fuse, img.cuda(), 1, mem_freq=mem_freq, engine_options=eo)
    core.interact(msk[:, 0], 0)                         # warm-up (allocations, first-launch costs), then a fresh state
    core.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got1 = core.interact(msk[:, 0], 0).copy()
    t1 = time.perf_counter()
    got2 = core.interact(msk[:, mid], mid).copy()
    t2 = time.perf_counter()
    gt_np = msk[0, :, 0].numpy() > 0.5
    gt = torch.from_numpy(gt_np).cuda()
    out = dict(clip=f"{img.shape[1]} frames {img.shape[-2]}x{img.shape[-1]} (the cpu_baseline sample)",
               hip_r1_frames_per_s=(img.shape[1] - 1) / (t1 - t0), hip_r2_frames_per_s=(img.shape[1] - 1) / (t2 - t1))
    for tag, got, ref in (("r1", got1, ref1), ("r2", got2, ref2)):
        a_, b_ = got > 0, ref > 0
        union = (a_ | b_).sum()
        out[f"mask_iou_hip_vs_cpu_oracle_{tag}"] = float((a_ & b_).sum() / union) if union else 1.0
        out[f"mask_pixels_differing_{tag}"] = int((a_ != b_).sum())
        # per FRAME (a clip-volume IoU hides one bad frame among many): the worst frame and where it is
        fu, fi = (a_ | b_).reshape(len(a_), -1).sum(1), (a_ & b_).reshape(len(a_), -1).sum(1)
        fiou = np.where(fu >= 64, fi / np.maximum(fu, 1), 1.0)
        out[f"min_frame_iou_hip_vs_cpu_oracle_{tag}"] = float(fiou.min())
        out[f"min_frame_iou_frame_{tag}"] = int(fiou.argmin())
        noise = ref_self_noise("seq480", "seq480L", "seq480P")
        fb = np.array([frame_bound(noise, u) for u in fu])
        out[f"within_bound_{tag}"] = bool(1 - out[f"mask_iou_hip_vs_cpu_oracle_{tag}"] <= clip_bound(noise) and (1 - fiou <= fb).all())
        out[f"measured_over_bound_{tag}"] = {"clip": (1 - out[f"mask_iou_hip_vs_cpu_oracle_{tag}"]) / clip_bound(noise), "worst_frame": float(((1 - fiou) / fb).max())}
    out["mask_pixels_total"] = int(got1.size)
    # interacted frames carry no propagated mask (the callers overwrite them): score the others
    keep = np.ones(img.shape[1], bool)
    keep[[0, mid]] = False
    out["j_and_f_hip"] = float(metrics.sequence_scores_gpu(gt, torch.from_numpy(got2 > 0).cuda())[keep, 2].mean())
    out["j_and_f_cpu_oracle"] = float(metrics.sequence_scores(gt_np, ref2 > 0)[keep, 3].mean())
    out["j_and_f_scorers"] = "hip: stcn_metrics_jf_counts kernel; cpu_oracle: eva_vos_amd.metrics (NumPy restatement of interactions/metrics.py)"
    return out


def memread_roofline(k, hw16=1620):
    """The space-time memory read alone at config-3 bank sizes (T = 52 and 104 frames in the bank, k objects, one frame of
    queries), timed with HIP events around whole reads (pass 1 + threshold + pass 2 + merge/gather) by the C-ABI hook
    stcn_bench_memory_read.  Units: algorithmic FLOP 2*N*Q*64 (the affinity; the 50-sparse readout adds
    2*k*Q*50*512), algorithmic bytes = keys N*65*4 once + queries + 50 gathered value rows of 2 KB per query and object
    + the readout."""
    import ctypes as C
    from eva_vos_amd import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(0)
    rows = []
    for T in (52, 104):
        N, Q = T * hw16, hw16
        mk = (torch.randn(N, 64, generator=g) * 0.8).cuda()
        qk = (torch.randn(Q, 64, generator=g) * 0.8).cuda()
        mv = torch.randn(k, N, 512, generator=g).cuda()
        ro = torch.empty(k, Q, 512, device="cuda")
        ms, plan = C.c_float(), (C.c_int32 * 7)()
        _lib.check(lib.stcn_bench_memory_read(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(mk.data_ptr()),
                                              C.c_void_p(mv.data_ptr()), C.c_void_p(qk.data_ptr()), N, Q, k, 10,
                                              C.c_void_p(ro.data_ptr()), C.byref(ms), plan))
        fl = 2.0 * N * Q * 64
        by = 4.0 * (N * 65 + Q * 64 + k * Q * 50 * 512 + k * Q * 512)
        tf = fl / (ms.value * 1e-3) / 1e12
        rows.append(dict(bank_frames=T, N=N, Q=Q, k=k, ms_per_read=ms.value, affinity_tflops=tf, mfma_frac=tf / FP32_MFMA_PEAK_TFLOPS,
                         algorithmic_gbytes_per_s=by / (ms.value * 1e-3) / 1e9, hbm_frac=by / (ms.value * 1e-3) / 8e12,
                         pass1_sample_stride=int(plan[1])))
        del mk, qk, mv, ro
    full = rows[-1]
    return {"bound": "mfma", "achieved": full["affinity_tflops"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": full["mfma_frac"], "traffic": None,
            "kernel": "affinity_tile_kernel x2 (sampled pass 1 + pass 2, v_mfma_f32_16x16x4_f32) + threshold + merge_readout (gather)",
            "what": f"whole read at T=104, k={k}: 2*N*Q*64 FLOP / time of all four kernels; HIP events around 10 reads; random N(0,0.8) keys",
            "algorithmic_gbytes_per_s": full["algorithmic_gbytes_per_s"], "hbm_frac_of_8TBps": full["hbm_frac"], "by_bank_size": rows}


def real_inputs(a):
    """Checkpoints + the first DAVIS-17 val sample, or None when the box does not hold them (the usual case: no network)."""
    # STCN_BENCH_WEIGHTS / STCN_BENCH_DAVIS relocate the two directories (tests point them at a synthetic tree)
    wdir = os.environ.get("STCN_BENCH_WEIGHTS", os.path.join(ROOT, "model_weights", "mivos"))
    root = os.environ.get("STCN_BENCH_DAVIS", os.path.join(ROOT, "data", "DAVIS_17", "trainval"))
    imset = os.path.join(root, "ImageSets", "2017", "val.txt")
    paths = [os.path.join(wdir, "stcn.pth"), os.path.join(wdir, "fusion.pth"), imset]
    if not all(os.path.exists(q) for q in paths):
        return None
    from eva_vos_amd import fq_driver
    ds = fq_driver.ClipDataset(root, imset)
    smp = ds[0]
    return dict(prop_sd=torch.load(paths[0], map_location="cpu"), fuse_sd=torch.load(paths[1], map_location="cpu"),
                rgb=smp["rgb"], gt=smp["gt"], name=smp["name"])


def power_leg(region):
    """Runs region() while a thread samples `rocm-smi --showpower --showclocks --json` (socket power, instantaneous shader clock)."""
    import re
    import subprocess
    samples, stop = [], threading.Event()

    def smi():
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        c = json.loads(out)["card0"]
        m = re.search(r"(\d+)Mhz", c.get("sclk clock speed:", ""))
        return float(c["Current Socket Graphics Package Power (W)"]), float(m.group(1)) if m else None, float(c.get("Max Graphics Package Power (W)", "nan"))

    def loop():
        while not stop.is_set():
            try:
                samples.append(smi())
            except Exception:
                return
            time.sleep(0.05)

    try:
        idle = smi()
    except Exception:
        return None
    th = threading.Thread(target=loop, daemon=True)
    t0 = time.perf_counter()
    th.start()
    region()
    dt = time.perf_counter() - t0
    stop.set()
    th.join(timeout=15)
    if len(samples) > 4:
        samples = samples[1:-1]                          # the first / last sample straddle the region's edges
    if not samples:
        return None
    w = sorted(x[0] for x in samples)
    clk = sorted(x[1] for x in samples if x[1])
    return {"what": "socket power while the timed region runs once more (rocm-smi sampled from a thread; `value` is not taken from this region)",
            "socket_w_median": w[len(w) // 2], "socket_w_max": w[-1], "cap_w": idle[2], "frac_of_cap_median": w[len(w) // 2] / idle[2] if idle[2] == idle[2] else None,
            "sclk_mhz_median": clk[len(clk) // 2] if clk else None, "before_w": idle[0], "samples": len(samples), "region_s": dt}


def config3_leg(prop, fuse, T, H, W, k):
    """One video of BASELINE config 3 on a fresh engine: interact(mask, 0) with k objects, mem_freq = 1; a second, profiled
    run (HIP events per launch class) gives the kernel-time shares and the conv / memory-read rates of this shape."""
    from eva_vos_amd import synth
    from mivos.inference_core import InferenceCore
    img = synth.synthetic_clip(T, H, W).cuda()
    gt = synth.synthetic_mask(T, H, W, k)
    m0 = torch.cat([1 - gt[:, 0].sum(0, keepdim=True).clamp(0, 1), gt[:, 0]], 0)
    # one video in flight: key encoder ahead on a side stream (engine options are passed explicitly: no os.environ traffic)
    e = InferenceCore(prop, fuse, img, k, mem_freq=1, engine_options={"lookahead": 2})
    e.interact(m0, 0, scribble=True)                       # warm-up
    e.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = e.interact(m0, 0, scribble=True)
    dt = time.perf_counter() - t0
    st = e.stats()
    del e
    e = InferenceCore(prop, fuse, img, k, mem_freq=1, engine_options={"lookahead": 0})
    e.set_profiling(True)
    out2 = e.interact(m0, 0, scribble=True)
    torch.cuda.synchronize()
    prof = e.kernel_profile()
    prof.pop("conv_hbm_bound")
    del e
    torch.cuda.empty_cache()
    tot = sum(v["ms"] for v in prof.values())
    conv, mr = prof["conv"], prof["memread"]
    return {"workload": f"{H}x{W} {k}-object engine (scribble / (k+1)-channel path), mem_freq=1, T={T}: interact(mask,0) on a fresh engine; "
                        f"bank grows to {st['bank_fwd']} frames = {st['bank_fwd'] * ((H + 15) // 16) * ((W + 15) // 16)} rows",
            "frames_per_s": st["frames"] / dt, "ms_per_frame": 1e3 * dt / st["frames"], "frames": st["frames"], "value_encodes": st["value_enc"],
            "repeat_bit_identical": bool(np.array_equal(out, out2)),
            "object_pixels_fraction": float((out > 0).mean()),
            "kernel_time_share": {c: round(v["ms"] / tot, 4) for c, v in prof.items() if v["ms"] > 0},
            "conv_executed_tflops": conv["exec_flops"] / (conv["ms"] * 1e-3) / 1e12,
            "conv_frac_of_fp32_mfma_peak": conv["exec_flops"] / (conv["ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "conv_algorithmic_tflops_incl_transforms": conv["flops"] / ((conv["ms"] + prof["wino_input"]["ms"] + prof["conv_reduce"]["ms"]) * 1e-3) / 1e12,
            "memread_ms_per_frame": mr["ms"] / st["frames"], "memread_affinity_tflops": mr["flops"] / (mr["ms"] * 1e-3) / 1e12,
            "memread_algorithmic_gbytes_per_s": mr["bytes"] / (mr["ms"] * 1e-3) / 1e9}


# the 30 sequences of DAVIS-2017 val (bike-packing ... soapbox): frames per sequence, 1999 in all (mean 66.6)
DAVIS_VAL_LENGTHS = [69, 50, 80, 84, 90, 75, 40, 104, 90, 60, 66, 52, 50, 90, 78, 50, 81, 34, 50, 47, 49, 50, 79, 40, 80, 100, 79, 43, 40, 99]


def config3_parity(prop, fuse, psd, fsd, T, H, W, k):
    """BASELINE config 3's shape on the CPU oracle AND the HIP engine (first T frames: k objects through the scribble path,
    mem_freq = 1) under the MULTI-OBJECT weight recipe (synth.RECIPES[2]: the decoder separates the objects, so the reference's
    own top-1 minus top-2 margin is >= 1e-2 on > 99 % of the pixels - tests/golden/seq480k5): per-object mask IoU over ALL pixels
    (bar 1 - 1e-3), the worst frame, the probability difference; the fraction of decisive pixels is reported beside it."""
    from eva_vos_amd import synth
    from mivos.inference_core import InferenceCore
    from oracle.stcn_oracle import OracleCore
    img, gt = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, k)
    m0 = torch.cat([1 - gt[:, 0].sum(0, keepdim=True).clamp(0, 1), gt[:, 0]], 0)
    t0 = time.perf_counter()
    orc = OracleCore(psd, fsd, img, k, mem_freq=1)
    ref = orc.interact(m0.clone(), 0, scribble=True)
    t_cpu = time.perf_counter() - t0
    core = InferenceCore(prop, fuse, img.cuda(), k, mem_freq=1)
    got = core.interact(m0, 0, scribble=True)
    lw, uw, lh, uh = orc.pad
    crop = lambda p_: p_[:, :, 0, lh:p_.shape[3] - uh if uh else None, lw:p_.shape[4] - uw if uw else None]      # noqa: E731
    po, pg = crop(orc.prob), crop(core.prob.cpu())
    top = torch.topk(po, 2, dim=0).values
    dec = ((top[0] - top[1]) >= 1e-2).numpy()
    d = (po - pg).abs()
    out = dict(sample=f"first {T} frames of the config-3 workload ({H}x{W}, k={k}, mem_freq=1), interact(mask,0): CPU oracle {t_cpu:.1f} s",
               weights="synthetic multi-object recipe (seed 2: Philox draws + decoder.pred fitted on reference features, oracle/fit_multi_pred.py)",
               decisive_pixel_fraction=float(dec[1:].mean()), mask_pixels_differing=int((got != ref).sum()),
               mask_pixels_differing_on_decisive=int(((got != ref) & dec).sum()), mask_pixels_total=int(got.size),
               object_pixels_per_frame_min=[int((ref[1:] == o).reshape(T - 1, -1).sum(1).min()) for o in range(1, k + 1)],
               prob_abs_diff_p999=float(np.quantile(d.flatten()[::max(7, d.numel() // 8000000 + 1)].numpy(), 0.999)), prob_abs_diff_max=float(d.max()))
    # the reference against itself under the multi-object recipe at 480p; for (nearly) full-length clips also its own drift over 103
    # propagated frames of the config-3 clip (selfnoise row "cfg3full": 4 vs 8 intra-op threads, oracle/gen_golden.py NOISE_ONLY_CASES)
    tags = ["seq480k5", "seq480k3", "seq640k3"]
    sn_files = np.load(os.path.join(ROOT, "tests", "golden", "selfnoise.npz")).files
    if T >= 52 and "cfg3full" in sn_files:
        tags.append("cfg3full")
    if (T, H, W, k) == (24, 480, 854, 5) and "cfg3_24" in sn_files:
        tags.append("cfg3_24")          # the reference against itself (1 vs 8 threads) on exactly THIS clip (oracle/gen_golden_long.py cfg3_24)
    noise = ref_self_noise(*tags)
    ious, fmin, fwhere, per_obj, ok = [], 1.0, None, [], True
    for o in range(1, k + 1):
        a_, b_ = got == o, ref == o
        ious.append(float((a_ & b_).sum() / max((a_ | b_).sum(), 1)))
        fu, fi = (a_ | b_).reshape(T, -1).sum(1), (a_ & b_).reshape(T, -1).sum(1)
        fiou = np.where(fu >= 64, fi / np.maximum(fu, 1), 1.0)
        fb = np.array([frame_bound(noise, u) for u in fu])
        wf = int(np.argmax((1 - fiou) / fb))                           # the frame closest to (or furthest beyond) ITS bound
        row = dict(object=o, clip_miss=1 - ious[-1], clip_bound=clip_bound(noise), worst_frame=wf, worst_frame_miss=float(1 - fiou[wf]),
                   worst_frame_bound=float(fb[wf]), worst_frame_union_px=int(fu[wf]),
                   measured_over_bound={"clip": (1 - ious[-1]) / clip_bound(noise), "worst_frame": float(((1 - fiou) / fb).max())},
                   within_bound=bool(1 - ious[-1] <= clip_bound(noise) and (1 - fiou <= fb).all()))
        per_obj.append(row)
        ok = ok and row["within_bound"]
        if float(fiou.min()) < fmin:
            fmin, fwhere = float(fiou.min()), (o, int(fiou.argmin()))
    out.update(mask_iou_vs_cpu_oracle_per_object=ious, mask_iou_vs_cpu_oracle=min(ious), min_frame_iou=fmin, min_frame_iou_object_frame=fwhere,
               per_object=per_obj, within_bound=ok,
               bound="clip: 1e-3 flat; (object, frame): max(1e-3, 1.5 x reference per-frame self-noise, 2 px / union px) - "
                     f"self-noise = tests/golden/selfnoise.npz rows {' / '.join(tags)} (the reference against itself at different thread counts)",
               reference_self_noise=dict(clip_miss=float(noise[0]), frame_miss=float(noise[4]), differing_px=float(noise[3])),
               what="mask_iou_vs_cpu_oracle = worst object over ALL pixels of the clip; min_frame_iou = worst (object, frame); within_bound = every object on the clip AND on every frame")
    # ... and against the REFERENCE itself where its answer is on the box: tests/golden/long_cfg3.npz holds the label map the reference produced for
    # all 104 frames of the FULL-LENGTH workload (oracle/gen_golden_long.py).  The synthetic clip depends on its length, so the engine runs the first
    # T frames of the 104-frame clip once more (a forward sweep is causal: the first T frames of the reference's answer are its answer to that clip)
    def against(gm, a_masks, what):
        rows = []
        for o in range(1, k + 1):
            a_, b_ = a_masks == o, gm == o
            fu, fi = (a_ | b_).reshape(T, -1).sum(1), (a_ & b_).reshape(T, -1).sum(1)
            fiou = np.where(fu >= 64, fi / np.maximum(fu, 1), 1.0)
            rows.append(dict(object=o, clip_miss=1 - float(fi.sum() / max(fu.sum(), 1)), worst_frame=int(fiou.argmin()), worst_frame_miss=float(1 - fiou.min())))
        return dict(what=what, mask_pixels_differing=int((a_masks != gm).sum()), mask_pixels_total=int(gm.size), per_object=rows)

    exact = os.path.join(ROOT, "tests", "golden", f"long_cfg3_{T}.npz")
    gpath = os.path.join(ROOT, "tests", "golden", "long_cfg3.npz")
    if os.path.exists(exact) and tuple(int(v) for v in np.load(exact)["shape"][:4]) == (T, H, W, k):
        gm = np.load(exact)["masks"]
        out["vs_reference_golden"] = against(gm, got, f"the HIP masks of this leg against the label map the REFERENCE itself produced for this very clip (tests/golden/long_cfg3_{T}.npz, 8 threads)")
        out["vs_reference_golden"]["cpu_oracle"] = against(gm, ref, "the CPU oracle's masks of this leg against the same label map")
        sn = np.load(os.path.join(ROOT, "tests", "golden", "selfnoise.npz"))
        if f"cfg3_{T}" in sn.files:
            r_ = sn[f"cfg3_{T}"][0]
            out["vs_reference_golden"]["reference_vs_itself"] = dict(what="the reference at 1 vs 8 threads on this clip", clip_miss_worst_object=float(r_[0]),
                                                                     differing_px=float(r_[3]), worst_frame_miss=float(r_[4]))
    elif os.path.exists(gpath):
        gl = np.load(gpath)
        Tg = int(gl["shape"][0])
        if tuple(int(v) for v in gl["shape"][1:4]) == (H, W, k) and T <= Tg:
            img_g, gt_g = synth.synthetic_clip(Tg, H, W)[:, :T].contiguous(), synth.synthetic_mask(Tg, H, W, k)
            mg = torch.cat([1 - gt_g[:, 0].sum(0, keepdim=True).clamp(0, 1), gt_g[:, 0]], 0)
            got_g = InferenceCore(prop, fuse, img_g.cuda(), k, mem_freq=1).interact(mg, 0, scribble=True)
            out["vs_reference_golden"] = against(gl["masks"][:T], got_g, f"HIP engine on the first {T} frames of the {Tg}-frame config-3 clip against the same frames of "
                                                 "tests/golden/long_cfg3.npz: the label map of the REFERENCE itself (8 threads)")
    del core
    torch.cuda.empty_cache()
    return out


def long_clip_parity(prop, fuse, psd, fsd, H, W, T, mem_freq, eo=None):
    """The longest DAVIS / MOSE clip length (T = 104, download_data.py:42) at 480p on the CPU oracle AND the HIP engine: one first
    interaction, k = 1.  Per frame IoU along the clip - does the difference grow towards the end of a long propagation?"""
    from eva_vos_amd import synth
    from mivos.inference_core import InferenceCore
    from oracle.stcn_oracle import OracleCore
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, 1)
    t0 = time.perf_counter()
    ref = OracleCore(psd, fsd, img, 1, mem_freq=mem_freq).interact(msk[:, 0], 0)
    t_cpu = time.perf_counter() - t0
    core = InferenceCore(prop, fuse, img.cuda(), 1, mem_freq=mem_freq, engine_options=eo)
    got = core.interact(msk[:, 0], 0)
    a_, b_ = got > 0, ref > 0
    fu, fi = (a_ | b_).reshape(T, -1).sum(1), (a_ & b_).reshape(T, -1).sum(1)
    fiou = np.where(fu >= 64, fi / np.maximum(fu, 1), 1.0)
    quarters = [float(fiou[q * T // 4:(q + 1) * T // 4].min()) for q in range(4)]
    noise = ref_self_noise("seq480", "seq480L", "seq480P")
    fb = np.array([frame_bound(noise, u) for u in fu])
    del core
    torch.cuda.empty_cache()
    return dict(clip=f"{T} frames {H}x{W}, k=1, mem_freq={mem_freq}, interact(mask,0): CPU oracle {t_cpu:.1f} s ({(T - 1) / t_cpu:.2f} frames/s)",
                mask_iou_hip_vs_cpu_oracle=float((a_ & b_).sum() / max((a_ | b_).sum(), 1)), mask_pixels_differing=int((a_ != b_).sum()),
                mask_pixels_total=int(got.size), min_frame_iou=float(fiou.min()), min_frame_iou_frame=int(fiou.argmin()),
                min_frame_iou_by_quarter_of_the_clip=quarters, clip_bound=clip_bound(noise), frame_bound=float(fb[int(fiou.argmin())]),
                within_bound=bool(1 - float((a_ & b_).sum() / max((a_ | b_).sum(), 1)) <= clip_bound(noise) and (1 - fiou <= fb).all()),
                measured_over_bound={"clip": (1 - float((a_ & b_).sum() / max((a_ | b_).sum(), 1))) / clip_bound(noise), "worst_frame": float(((1 - fiou) / fb).max())},
                bound="clip: 1e-3 flat; every frame: max(1e-3, 1.5 x reference per-frame self-noise, 2 px / union px)")


def session_parity(prop, fuse, psd, fsd, H, W, T, rounds, mem_freq, eo=None):
    """A whole ANNOTATION SESSION at 480p on the CPU oracle AND the HIP engine: the oracle policy of the reference
    (interactions/mask.py:113-146: annotate frame 0, then after every round the frame with the worst J against the ground truth;
    annotated frames count with their ground-truth mask, interactions/eval.py:57-60) for `rounds` rounds - growing certain
    memory, shrinking spans, fusion on both sides of earlier interactions.  Both follow the ORACLE's frame choices (so that the
    comparison is of propagation, not of a tie in the policy); whether the HIP engine's own J picks the same frame is reported."""
    from eva_vos_amd import synth
    from mivos.inference_core import InferenceCore
    from oracle.stcn_oracle import OracleCore
    img, msk = synth.synthetic_clip(T, H, W, seed=7), synth.synthetic_mask(T, H, W, 1, seed=7)
    gtb = msk[0, :, 0].numpy() > 0.5
    orc = OracleCore(psd, fsd, img, 1, mem_freq=mem_freq)
    core = InferenceCore(prop, fuse, img.cuda(), 1, mem_freq=mem_freq, engine_options=eo)

    def per_frame_j(masks, done):
        gen = masks > 0
        gen[done] = gtb[done]
        u, n = (gen | gtb).reshape(T, -1).sum(1), (gen & gtb).reshape(T, -1).sum(1)
        return np.where(u > 0, n / np.maximum(u, 1), 0.0)

    noise = ref_self_noise("seq480", "seq480L", "seq480P")              # the reference against itself at 480p, k = 1 (1 / 2 / 4 / 8 threads)
    frames, rows, t_cpu, t_gpu = [0], [], 0.0, 0.0
    for r in range(rounds):
        f = frames[r]
        t0 = time.perf_counter()
        ref = orc.interact(msk[:, f], f).copy()
        t1 = time.perf_counter()
        got = core.interact(msk[:, f], f).copy()
        t2 = time.perf_counter()
        t_cpu, t_gpu = t_cpu + t1 - t0, t_gpu + t2 - t1
        a_, b_ = got > 0, ref > 0
        fu, fi = (a_ | b_).reshape(T, -1).sum(1), (a_ & b_).reshape(T, -1).sum(1)
        fiou = np.where(fu >= 64, fi / np.maximum(fu, 1), 1.0)
        q_ref, q_got = per_frame_j(ref, frames[:r + 1]), per_frame_j(got, frames[:r + 1])
        nxt = int(np.argmin(q_ref))
        fb = np.array([frame_bound(noise, u) for u in fu])
        miou = float((a_ & b_).sum() / max((a_ | b_).sum(), 1))
        rows.append(dict(round=r + 1, frame=int(f), mask_iou=miou, mask_pixels_differing=int((a_ != b_).sum()),
                         min_frame_iou=float(fiou.min()), min_frame_iou_frame=int(fiou.argmin()), mean_j_oracle=float(q_ref.mean()), mean_j_hip=float(q_got.mean()),
                         next_frame_oracle=nxt, next_frame_hip=int(np.argmin(q_got)),
                         clip_bound=clip_bound(noise), frame_bound=float(fb[int(fiou.argmin())]),
                         measured_over_bound={"clip": (1 - miou) / clip_bound(noise), "worst_frame": float(((1 - fiou) / fb).max())},
                         within_bound=bool(1 - miou <= clip_bound(noise) and (1 - fiou <= fb).all())))
        frames.append(nxt)
    st = core.stats()
    del core
    torch.cuda.empty_cache()
    return dict(session=f"{rounds} rounds of the oracle mask policy (interactions/mask.py:113-146) on a {T}-frame {H}x{W} clip, k=1, mem_freq={mem_freq}; "
                        f"CPU oracle {t_cpu:.1f} s, HIP engine {t_gpu:.2f} s", frames_annotated=[int(v) for v in frames[:rounds]],
                worst_round_mask_iou=min(r_["mask_iou"] for r_ in rows), worst_round_min_frame_iou=min(r_["min_frame_iou"] for r_ in rows),
                same_frame_choice_every_round=all(r_["next_frame_oracle"] == r_["next_frame_hip"] for r_ in rows),
                within_bound=all(r_["within_bound"] for r_ in rows),
                worst_measured_over_bound={"clip": max(r_["measured_over_bound"]["clip"] for r_ in rows), "worst_frame": max(r_["measured_over_bound"]["worst_frame"] for r_ in rows)},
                last_round_stats=st, rounds=rows,
                bound="per round - clip: 1e-3 flat; every frame: max(1e-3, 1.5 x reference per-frame self-noise, 2 px / union px); "
                      "self-noise = tests/golden/selfnoise.npz rows seq480 / seq480L / seq480P")


def r2_roofline(prop, fuse, img, mask0, mask_mid, T, mem_freq, scribble):
    """The regime the reference's annotation loops spend their time in (59 of 60 rounds of eval_annotation_method.py:30, 7 of 8 of
    interactions/mask.py:113-146): a SECOND interaction - cached key features, memory read + decoder on every frame,
    FusionNet + attention read on the frames between the two interacted frames (inference_core.py:184-207).  One video, one
    stream, HIP events per launch (same method as `roofline`)."""
    from mivos.inference_core import InferenceCore
    res = {}
    # the shipped mode first (side stream on: FusionNet of a decoded group runs beside the next group), then one stream only;
    # the first pair of interactions of the process is a warm-up (first launches of the rounds >= 2 kernels)
    def one_r2(prof_on=False, la=2):
        e = InferenceCore(prop, fuse, img, 1 if not scribble else mask0.shape[0] - 1, mem_freq=mem_freq, engine_options={"lookahead": la})
        e.interact(mask0, 0, scribble=scribble)
        e.set_profiling(prof_on)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.interact(mask_mid, T // 2, scribble=scribble)
        torch.cuda.synchronize()
        return e, time.perf_counter() - t0

    rates = []
    for i in range(4):                                     # first pair = warm-up; median of three 50 ms measurements
        e, dt = one_r2(la=2)                               # the engine's default: side stream on (the lanes of this bench run without)
        if i:
            rates.append(e.stats()["frames"] / dt)
        del e
    res["frames_per_s_one_video"] = sorted(rates)[1]
    rates = []
    for i in range(3):
        e, dt = one_r2(la=0)
        rates.append(e.stats()["frames"] / dt)
        del e
    res["frames_per_s_solo"] = sorted(rates)[1]
    for prof_on in (True,):
        e = InferenceCore(prop, fuse, img, 1 if not scribble else mask0.shape[0] - 1, mem_freq=mem_freq, engine_options={"lookahead": 0})
        e.interact(mask0, 0, scribble=scribble)
        e.set_profiling(prof_on)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.interact(mask_mid, T // 2, scribble=scribble)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = e.stats()
        prof = e.kernel_profile()
        prof.pop("conv_hbm_bound")
        del e
    tot = sum(v["ms"] for v in prof.values())
    conv, fus = prof["conv"], prof["fusion_conv"]
    gemm_ms = conv["ms"] + fus["ms"]
    ach = (conv["exec_flops"] + fus["exec_flops"]) / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    all_ms = gemm_ms + prof["wino_input"]["ms"] + prof["conv_reduce"]["ms"]
    res.update({"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS,
                "traffic": None,
                "what": "R2 = interact(mask, T//2) after interact(mask, 0): executed MFMA FLOP of all conv GEMM launches (decoder / value "
                        "encoder + FusionNet) / their summed device time",
                "frames": st["frames"], "fused_frames": st["fused"], "value_encodes": st["value_enc"], "key_misses": st["key_miss"],
                "kernel_ms_per_frame": tot / max(st["frames"], 1),
                "kernel_time_share": {c: round(v["ms"] / tot, 4) for c, v in prof.items() if v["ms"] > 0},
                "decoder_value_conv_executed_tflops": conv["exec_flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0,
                "fusion_conv_tflops": fus["flops"] / (fus["ms"] * 1e-3) / 1e12 if fus["ms"] > 0 else 0.0,
                "fusion_conv_ms_per_fused_frame": fus["ms"] / max(st["fused"], 1),
                "fusion_conv_launches": fus["launches"],
                "algorithmic_tflops_incl_transforms": (conv["flops"] + fus["flops"]) / (all_ms * 1e-3) / 1e12 if all_ms > 0 else 0.0,
                "algorithmic_gflop_per_frame": sum(v["flops"] for v in prof.values()) / max(st["frames"], 1) / 1e9,
                "memread_share": round(prof["memread"]["ms"] / tot, 4), "attention_share": round(prof["attention"]["ms"] / tot, 4)})
    return res


def davis_val_leg(prop, fuse, a, H, W, rank, world, local, streams, barrier, eo=None):
    """samples with the 30 DAVIS-2017-val sequence lengths, assigned to the ranks by LPT on their
    frame counts (eva_vos_amd.shard.lpt_assign; the reference slices by --min-idx/--max-idx, eval_annotation_method.py:34-35,
    113-119), inside a rank to the in-flight lanes the same way.  Fixed total work -> strong scaling: frames of ALL samples /
    the slowest rank's time, with the per-rank busy fraction and the imbalance the tail lengths (34..104) cause."""
    import threading
    from eva_vos_amd import shard, synth
    from mivos.inference_core import InferenceCore
    n = a.steps if a.workload == "davis-val" else len(DAVIS_VAL_LENGTHS)
    lengths = [min(DAVIS_VAL_LENGTHS[i % len(DAVIS_VAL_LENGTHS)], a.davis_max_frames) for i in range(n)]
    assign = shard.lpt_assign([t - 1 for t in lengths], world)
    mine = assign[rank]
    S = len(streams)
    lanes = [[mine[j] for j in part] for part in shard.lpt_assign([lengths[i] - 1 for i in mine], S)]
    Tmax = max(lengths)
    base = synth.synthetic_clip(Tmax, H, W).cuda()
    mask0 = synth.synthetic_mask(Tmax, H, W, 1)[:, 0].clone()
    # the stored shapes of DAVIS / MOSE are not all landscape (scripts/resize.py:9-24 resizes to min(w, h) = 480): every 6th sample of
    # the workload is a PORTRAIT clip (W x H: the same scene transposed, 54 x 30 keys instead of 30 x 54)
    portrait = [i % 6 == 5 for i in range(n)]
    mask0_p = mask0.transpose(-1, -2).contiguous()
    engines = {}
    for l, part in enumerate(lanes):
        with torch.cuda.stream(streams[l]):
            for i in part:
                g = torch.Generator(device="cuda").manual_seed(5000 + i)
                clip = base[:, :lengths[i]] + 0.15 * torch.randn((1, lengths[i]) + tuple(base.shape[2:]), generator=g, device="cuda")
                if portrait[i]:
                    clip = clip.transpose(-1, -2).contiguous()
                engines[i] = InferenceCore(prop, fuse, clip, 1, mem_freq=a.mem_freq, engine_options=eo)
    del base
    torch.cuda.synchronize()
    frames = [0] * S

    def run(l):
        torch.cuda.set_device(local)
        with torch.cuda.stream(streams[l]):
            for i in lanes[l]:                               # longest first (LPT order)
                engines[i].interact(mask0_p if portrait[i] else mask0, 0)
                frames[l] += engines[i].stats()["frames"]

    barrier()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(l,)) for l in range(S)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    t_rank = time.perf_counter() - t0
    barrier()
    rows = shard.gather_rows(np.array([[rank, t_rank, sum(frames), len(mine)]], np.float64), 4)
    engines.clear()
    torch.cuda.empty_cache()
    t_max = float(rows[:, 1].max())
    total = float(rows[:, 2].sum())
    per_rank_frames = [sum(lengths[i] - 1 for i in part) for part in assign]
    return {"workload": f"{n} single-object samples with the DAVIS-2017-val sequence lengths ({min(lengths)}..{max(lengths)} frames, "
                        f"{sum(lengths)} in all), {n - sum(portrait)} landscape {H}x{W} + {sum(portrait)} portrait {W}x{H}, fresh engine + interact(mask,0) each, "
                        f"mem_freq={a.mem_freq}; LPT over {world} rank(s), {S} lane(s) per rank", "scaling": "strong", "portrait_samples": int(sum(portrait)),
            "samples": n, "frames_total": total, "frames_per_s": total / t_max, "slowest_rank_s": t_max,
            "rank_seconds": [float(v) for v in rows[:, 1]], "rank_busy_fraction": [float(v / t_max) for v in rows[:, 1]],
            "rank_frames": per_rank_frames, "rank_samples": [len(p_) for p_ in assign],
            "imbalance_max_over_mean_frames": max(per_rank_frames) / (sum(per_rank_frames) / world),
            "lengths_by_rank": [[lengths[i] for i in part] for part in assign]}


def drivers_leg(prop, fuse, H, W, videos, frames, lanes=2, rounds=8):
    """BASELINE configs 4 / 5 at N = 1, end to end as the reference's users would feel them: the own counterparts of generate_fq_dataset.py
    (eva_vos_amd.fq_driver: JPEG decode, upload, 8 oracle rounds per sample, GPU J, 224x224 PNG states + CSV) and of eval_annotation_method.py
    with the oracle mask policy (eva_vos_amd.eval_driver: GPU J&F per round) on a synthetic dataset tree in the DAVIS layout, `lanes` videos in
    flight.  One of the videos is a portrait clip.  Rounds per second = annotation rounds (one interact() + metrics + outputs each)."""
    import shutil
    import tempfile
    from eva_vos_amd import eval_driver, fq_driver
    tmp = tempfile.mkdtemp(prefix="stcn_drivers_")
    try:
        t0 = time.perf_counter()
        tree = {f"v{i}": ((frames, W, H, 1) if i == videos - 1 else (frames, H, W, 1)) for i in range(videos)}
        imset = fq_driver.make_synthetic_tree(os.path.join(tmp, "db"), tree)
        t_tree = time.perf_counter() - t0
        fq_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, "warm"), prop, fuse, rounds=2, lanes=lanes)      # warm-up: file cache, first launches
        out = {"dataset": f"{videos} synthetic single-object videos x {frames} frames ({videos - 1} x {H}x{W} + 1 portrait {W}x{H}) in the DAVIS layout "
                          f"(JPEG frames, palette PNG annotations; written in {t_tree:.1f} s, outside the timed regions)",
               "lanes": lanes, "rounds_per_sample": rounds}
        for name, fn in (("fq_driver", lambda st: fq_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, "fq"), prop, fuse, rounds=rounds, lanes=lanes, stats=st)),
                         ("eval_driver_oracle_mask", lambda st: eval_driver.run(os.path.join(tmp, "db"), imset, os.path.join(tmp, "e.csv"), prop, fuse, "oracle_mask",
                                                                                rounds=rounds, lanes=lanes, stats=st))):
            torch.cuda.synchronize()
            st = {}
            t0 = time.perf_counter()
            rows = fn(st)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            # propagated frames = what the engines' do_pass really visited (stcn_get_stats per interaction): rounds >= 2 walk only the spans
            # between the new annotation and its neighbours (round 5 multiplied rounds by T - 1: 4-5 x too many)
            out[name] = {"rounds": int(len(rows)), "seconds": dt, "rounds_per_s": len(rows) / dt, "propagated_frames": int(st.get("propagated_frames", 0)),
                         "propagated_frames_per_s": st.get("propagated_frames", 0) / dt, "frames_per_round_mean": st.get("propagated_frames", 0) / max(len(rows), 1),
                         # where the lanes' wall time went (host clock, summed over the samples of a lane): waiting for the loader, building
                         # InferenceCores, the annotation sessions themselves (GPU-bound: the `session` leg is their resident-clip limit)
                         "host_account": {"lanes": st.get("lanes"), "create_s": round(st.get("create_s", 0.0), 4), "session_s": round(st.get("session_s", 0.0), 4),
                                          "wait_writers_s": st.get("wait_writers_s")}}
        out["what"] = ("rounds/s incl. JPEG decode, H2D, propagation (1 first + 7 later interactions per sample), GPU J / J&F and all output files; "
                       "reference counterparts: generate_fq_dataset.py:60-86, eval_annotation_method.py:118-190 with interactions/mask.py:113-146")
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        torch.cuda.empty_cache()


def session_leg(prop, fuse, H, W, T, rounds, metric, videos=4, lanes_list=(1, 2, 4), tag=""):
    """A SESSION-shaped number (BASELINE configs 4 / 5 are annotation sessions, not single first interactions): `videos` resident 480p clips,
    each through `rounds` rounds of the reference's oracle mask policy (interactions/mask.py:113-146 via eva_vos_amd.eval_driver.run_policy:
    one first interaction, then short fused spans; J or J&F per frame on the device after every round, annotated frames counting with their
    ground truth) on a fresh InferenceCore per sample, as generate_fq_dataset.py:63-70 / interactions/mask.py:24-26 build them.  Clips, ground
    truth and weights are resident before the timed region (no JPEG decode, no output files: the `drivers` leg has those).  Reported per lane
    count: rounds/s, TRUE propagated frames/s (the engines' own visit counts) and `device_busy_frac` = the ENGINE kernel ms the same sessions
    need when every launch runs alone (HIP events per launch, one profiled pass on one stream) / wall time: 1.0 = the wall time is the sum of
    the solo kernel durations (above 1.0: concurrent launches filled each other's tails; the metric kernels are not in the numerator)."""
    import threading
    from eva_vos_amd import eval_driver, synth
    from mivos.inference_core import InferenceCore
    base = synth.synthetic_clip(T, H, W, seed=7).cuda()
    gt = synth.synthetic_mask(T, H, W, 1, seed=7)                        # [1,T,1,H,W]
    samples = []
    for v in range(videos):
        g = torch.Generator(device="cuda").manual_seed(9000 + v)
        clip = base if v == 0 else base + 0.15 * torch.randn(base.shape, generator=g, device="cuda")
        samples.append({"rgb": clip, "gt": gt.cuda(), "num_frames": T, "name": f"s{v}"})
    torch.cuda.synchronize()

    def one(sample, eo, prof=False):
        core = InferenceCore(prop, fuse, sample["rgb"], 1, engine_options=eo)
        kms = [0.0]
        if prof:
            core.set_profiling(True)
            inner = core.interact

            def interact(*a_, **k_):                                     # kernel ms of every interaction of the session (syncs: profiled pass only)
                r_ = inner(*a_, **k_)
                kms[0] += sum(v["ms"] for c, v in core.kernel_profile().items() if c != "conv_hbm_bound")
                return r_
            core.interact = interact
        res = eval_driver.run_policy("oracle_mask", core, sample, rounds, metric)
        return len(res["mu_metrics"]), res["propagated_frames"], kms[0], res["frames"]

    def region(lanes, prof=False):
        # one lane: the engine's own side streams on; several lanes: off (they fill each other's gaps).  The profiled pass runs WITHOUT side
        # streams: its per-launch durations are then solo durations (kernels that overlap on the chip each take longer; their sum would
        # exceed the wall time - round 6's first capture read 1.25 that way)
        eo = {"lookahead": 0} if lanes > 1 or prof else {"lookahead": 2}
        dev = torch.cuda.current_device()
        parts = [samples[l::lanes] for l in range(lanes)]
        acc = [[0, 0, 0.0] for _ in range(lanes)]
        picks = [None] * lanes

        def lane(l):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(torch.cuda.Stream()):
                for smp in parts[l]:
                    r_, f_, k_, fr_ = one(smp, eo, prof)
                    acc[l][0] += r_; acc[l][1] += f_; acc[l][2] += k_
                    picks[l] = fr_
                torch.cuda.current_stream().synchronize()

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=lane, args=(l,)) for l in range(lanes)]
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt, sum(a_[0] for a_ in acc), sum(a_[1] for a_ in acc), sum(a_[2] for a_ in acc), picks[0]

    region(1)                                                            # warm-up: first launches of the rounds >= 2 kernels, pool
    _, n_r, n_f, kms, picks = region(1, prof=True)
    out = {"workload": f"{videos} resident {H}x{W} clips x {T} frames, k=1, mem_freq=5: {rounds} rounds of the oracle mask policy per clip (metric {metric}), "
                       f"fresh InferenceCore per clip, per-frame {metric} on the device after every round {tag}".strip(),
           "rounds_total": n_r, "propagated_frames_total": n_f, "frames_per_round_mean": n_f / max(n_r, 1),
           "kernel_ms_total_profiled_pass": kms, "kernel_ms_per_propagated_frame": kms / max(n_f, 1),
           "annotated_frames_first_clip": [int(v) for v in picks[:rounds]], "lanes": {}}
    for lanes in lanes_list:
        if lanes > videos:
            continue
        best = None
        for _ in range(2):                                               # two passes, the faster one (50-500 ms regions)
            dt, r_, f_, _, _ = region(lanes)
            if best is None or dt < best[0]:
                best = (dt, r_, f_)
        dt, r_, f_ = best
        out["lanes"][str(lanes)] = {"seconds": dt, "rounds_per_s": r_ / dt, "propagated_frames_per_s": f_ / dt, "device_busy_frac": kms * 1e-3 / dt}
    one_lane = out["lanes"].get("1")
    if one_lane:
        out.update(rounds_per_s_one_lane=one_lane["rounds_per_s"], device_busy_frac_one_lane=one_lane["device_busy_frac"])
    del samples, base
    torch.cuda.empty_cache()
    return out


def roofline_objects(prof, roof_frames, t_roof, frames, dt_r1, roof_steps):
    """The `roofline` object (+ the per-class kernel times of the solo leg) of the bench line from the per-launch-class profile of the roofline
    leg: prof[class] = dict(ms, launches, flops, bytes, exec_flops) summed over the leg's videos (HIP events per launch on one stream)."""
    out = {}
    conv, wi, rd = prof["conv"], prof["wino_input"], prof["conv_reduce"]
    # Dominant kernels: the two fp32-MFMA conv GEMMs (conv_gemm_kernel: direct implicit GEMM; wino_gemm_kernel: the
    # stride-1 3x3 convs as Winograd F(2x2,3x3), 2.25x fewer multiplies for the same result).  `achieved` / `frac`
    # are the FLOP the matrix cores EXECUTED per second of GEMM kernel time (<= peak by construction: how busy the
    # MFMA pipes are); the ALGORITHMIC rate (2*M*N*K of every conv, over the GEMMs plus the Winograd input transforms
    # and split-K reduces they need) is given beside it and may exceed the peak - that is skipped arithmetic, not a
    # faster pipe, and it is labelled as such.
    ach = conv["exec_flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
    conv_all_ms = conv["ms"] + wi["ms"] + rd["ms"]
    alg = conv["flops"] / (conv_all_ms * 1e-3) / 1e12 if conv_all_ms > 0 else 0.0
    out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                       "kernel": "conv_gemm_kernel + wino_gemm_kernel + wino4_gemm_kernel (fp32 MFMA conv GEMMs, v_mfma_f32_32x32x2_f32: direct "
                                 "implicit GEMM, Winograd F(2x2,3x3), Winograd F(4x4,3x3) for the stride-1 3x3 convs from 64 channels - decoder side and key-encoder trunk)",
                       "what": "executed MFMA FLOP of all conv GEMM launches / their summed device time (HIP events per launch)",
                       "launches": conv["launches"], "avg_launch_ms": conv["ms"] / max(conv["launches"], 1),
                       "executed_flop_per_launch_avg": conv["exec_flops"] / max(conv["launches"], 1),
                       "algorithmic_flop_per_launch_avg": conv["flops"] / max(conv["launches"], 1),
                       "algorithmic_tflops_incl_transforms": alg,
                       "algorithmic_frac_of_peak": alg / FP32_MFMA_PEAK_TFLOPS,
                       "executed_over_algorithmic_flop": conv["exec_flops"] / conv["flops"] if conv["flops"] > 0 else 0.0,
                       "wino_input_transform_ms_share_of_conv": wi["ms"] / conv_all_ms if conv_all_ms > 0 else 0.0}
    # what the matrix pipes deliver on this chip under a pure fp32-MFMA load (register operands, ~30 ms): the datasheet peak
    # assumes 2.4 GHz, the chip holds ~2.0 GHz under matrix load
    try:
        import ctypes as C
        from eva_vos_amd import _lib
        tf, ms_ = C.c_float(), C.c_float()
        _lib.check(_lib.lib().stcn_bench_mfma_rate(C.c_void_p(torch.cuda.current_stream().cuda_stream), 30, C.byref(tf), C.byref(ms_)))
        out["roofline"]["sustained_mfma_tflops_measured"] = tf.value
        out["roofline"]["frac_of_sustained_mfma_rate"] = ach / tf.value if tf.value > 0 else None
        out["roofline"]["sustained_what"] = (f"mfma_probe_kernel: v_mfma_f32_32x32x2_f32 on register operands, no memory traffic, {ms_.value:.1f} ms on all CUs "
                                             f"= {tf.value / FP32_MFMA_PEAK_TFLOPS * 2.4:.2f} GHz-equivalent of the {FP32_MFMA_PEAK_TFLOPS} TFLOP/s @ 2.4 GHz datasheet peak")
    except Exception as ex:                              # the probe is an extra: never fail the line for it
        out["roofline"]["sustained_mfma_tflops_measured"] = None
        out["roofline"]["sustained_what"] = f"probe failed: {ex}"
    hb = prof.pop("conv_hbm_bound")                      # subset of "conv": launches below 19.7 FLOP/B
    if conv["flops"] > 0:
        out["roofline"]["winograd_f2x2_share_of_algorithmic_flop"] = hb.get("wino2_flops", 0.0) / conv["flops"]
        out["roofline"]["winograd_f4x4_share_of_algorithmic_flop"] = hb.get("wino4_flops", 0.0) / conv["flops"]
    tot_ms = sum(v["ms"] for v in prof.values())
    out["kernel_time_share"] = {c: round(v["ms"] / tot_ms, 4) for c, v in prof.items() if v["ms"] > 0}
    if hb["ms"] > 0 and conv["ms"] > hb["ms"]:
        # the same kernel in its two regimes (the headline `roofline` above is over ALL its launches)
        mf = (conv["exec_flops"] - hb["flops"]) / ((conv["ms"] - hb["ms"]) * 1e-3) / 1e12
        gb = hb["bytes"] / (hb["ms"] * 1e-3) / 1e9
        out["roofline_by_regime"] = {
            "mfma_bound_launches": {"launches": conv["launches"] - hb["launches"], "time_share_of_conv": round(1 - hb["ms"] / conv["ms"], 4),
                                    "achieved": mf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mf / FP32_MFMA_PEAK_TFLOPS},
            "hbm_bound_launches": {"launches": hb["launches"], "time_share_of_conv": round(hb["ms"] / conv["ms"], 4),
                                   "achieved": gb, "peak": 8000.0, "unit": "GB/s", "frac": gb / 8000.0,
                                   "what": "conv launches under 19.7 FLOP/B of algorithmic intensity (1x1 channel expansions, stems)"}}
    # HBM-side bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate runs)
    try:
        import glob
        pmc_file = os.path.basename(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))[-1])
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
        from eva_vos_amd import _lib as _l
        out["roofline"]["traffic"] = pmc["conv_gemm_traffic_bytes_per_launch"]
        # is the committed capture a capture of THIS library?  (csrc hash stamped into the .so by the Makefile and into the capture by
        # tools/refresh_profiles.sh; a capture from before round 6 has no hash and counts as stale)
        out["roofline"]["traffic_commit"] = pmc.get("commit", "unknown")
        out["roofline"]["traffic_csrc_hash"] = pmc.get("csrc_hash")
        out["roofline"]["library_csrc_hash"] = _l.src_hash()
        out["roofline"]["traffic_stale"] = pmc.get("csrc_hash") != _l.src_hash()
        out["roofline"]["traffic_source"] = (f"profiles/{pmc_file}: a committed capture, NOT measured by this run (rocprofv3 --pmc passes of this "
                                             f"workload at T={pmc.get('frames', 30)}, FETCH_SIZE x2 + WRITE_SIZE per conv GEMM launch; captured at commit "
                                             f"{pmc.get('commit', 'unknown')}: {pmc.get('captured', 'round 2')})")
    except (OSError, IndexError):
        pass
    # the whole FRAME against the matrix peak: FLOP the matrix cores executed in every class (conv GEMMs as executed -
    # Winograd counted with its reduced multiplies -, memory-read affinity + read-out, Cout = 1 convs) over ALL kernel time
    # of the leg (transforms, reduces, elementwise, gathers included).  `frac` above is pipe occupancy inside the GEMM
    # launches; this is what the frame as a whole makes of the chip
    exec_all = sum(prof[c]["exec_flops"] if c in ("conv", "fusion_conv") else prof[c]["flops"] for c in prof)
    out["roofline"]["frame_executed_frac"] = exec_all / (tot_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS
    out["roofline"]["frame_executed_tflops"] = exec_all / (tot_ms * 1e-3) / 1e12
    out["roofline"]["frame_kernel_ms"] = tot_ms / roof_frames
    out["frame_kernel_ms"] = tot_ms / roof_frames               # solo leg: all kernel time per propagated R1 frame (HIP events per launch)
    out["kernel_ms_per_frame_by_class"] = {c: round(v["ms"] / roof_frames, 5) for c, v in prof.items() if v["ms"] > 0}
    out["roofline"]["algorithmic_bytes_per_launch"] = conv["bytes"] / max(conv["launches"], 1)
    out["roofline"]["leg"] = f"{max(1, roof_steps)} video(s), 1 stream, HIP events per launch"
    out["device_busy_frac_roofline_leg"] = tot_ms * 1e-3 / t_roof
    out["algorithmic_gflop_per_frame"] = sum(v["flops"] for v in prof.values()) / roof_frames / 1e9
    # chip-level view of the timed region: all algorithmic FLOP of the path / wall time
    out["timed_region_tflops"] = out["algorithmic_gflop_per_frame"] * 1e-3 * frames / dt_r1
    return out
