"""Frame-quality dataset generation on the HIP engine, sharded over the GPUs of one node (BASELINE config 4).

Own counterpart of the reference driver ``generate_fq_dataset.py:60-86`` and what it calls - the clip loader
``datasets/annotation_dataset.py:80-132`` (one sample per (video, object), ImageNet normalisation
``datasets/range_transform.py:3-6``), the oracle policy ``interactions/mask.py:113-156`` (8 rounds: annotate
the worst frame by J with its ground-truth mask), the per-frame evaluation ``interactions/eval.py:27-81`` and
the writer ``util/fq_dataset.py:50-91`` (224x224 nearest-resized masks as PNG + a CSV of
``state_name, ious, selected_frame``).  Differences by design:

* samples are assigned to ranks with an LPT schedule over their frame counts instead of ``--min-idx/--max-idx``;
  propagation has no collective; ONE gather of fixed-width rows at the end (RCCL over xGMI), rank 0 writes the CSV;
* a decoded clip is cached per video and reused for all its objects (the reference re-decodes it per object);
* J is computed on the GPU from the engine's mask tensor (``stcn_metrics_jf_counts``), only counts cross PCIe;
* the 224x224 PNG states are encoded and written by host threads while the GPU propagates the next round.
* two samples are in flight per GPU (host thread + HIP stream each, ``--lanes``): +9..17 % rounds/s;

Usage:  python -m eva_vos_amd.fq_driver --root data/MOSE --imset data/MOSE/ImageSets/subset_train_4.txt --out FQ_DB
        (multi-GPU: python -m torch.distributed.run --nproc-per-node N -m eva_vos_amd.fq_driver ...)
"""
from __future__ import annotations

import argparse
import csv
import os
import time
from typing import Dict, List

import numpy as np
import torch
from PIL import Image

from . import metrics, shard

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)
NO_OBJECT = 20.0          # the reference's token for frames without the object (interactions/eval.py:67)


# ------------------------------------------------------------------------------------------------ data
_DECODE_POOL = None
_DECODE_LOCK = __import__("threading").Lock()


class _DecodeProcs:
    """A pool of image-decode worker PROCESSES (eva_vos_amd/_decode_worker.py: NumPy + Pillow only) fed over pipes; the pixels come back
    through shared memory.  Threads would do the decoding itself just as well (Pillow releases the interpreter lock inside its codecs) but
    they take that lock thousands of times per video around the codec calls - and the lanes that drive the GPU share it (round 6:
    building one InferenceCore, 0.9 ms of Python, took 30-60 ms beside 16 decode threads)."""

    def __init__(self, n: int):
        import atexit
        import queue
        import subprocess
        import sys
        from concurrent.futures import ThreadPoolExecutor
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        self.idle = queue.Queue()
        self.procs = [subprocess.Popen([sys.executable, "-m", "eva_vos_amd._decode_worker"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                       text=True, bufsize=1, cwd=root, env=env) for _ in range(n)]
        for p_ in self.procs:
            self.idle.put(p_)
        self.waiters = ThreadPoolExecutor(n, thread_name_prefix="stcn-decode-wait")      # threads that only block on a worker's pipe
        atexit.register(self.close)

    def close(self):
        for p_ in self.procs:
            try:
                p_.stdin.close()
            except OSError:
                pass
        self.procs = []

    def _call(self, req: dict):
        import json
        p_ = self.idle.get()
        try:
            p_.stdin.write(json.dumps(req) + "\n")
            p_.stdin.flush()
            ans = p_.stdout.readline()
        finally:
            self.idle.put(p_)
        if not ans.startswith("ok"):
            raise RuntimeError(f"image-decode worker: {ans.strip() or 'died'}")

    def decode(self, jpgs, pngs, H: int, W: int):
        """uint8 [T,H,W,3] frames and [T,H,W] label maps of one video (either list may be None), decoded by the workers in chunks."""
        from multiprocessing import shared_memory
        T = len(jpgs if jpgs is not None else pngs)
        rgb = shared_memory.SharedMemory(create=True, size=T * H * W * 3) if jpgs is not None else None
        lab = shared_memory.SharedMemory(create=True, size=T * H * W) if pngs is not None else None
        try:
            step = max(1, -(-T // (2 * len(self.procs))))
            futs = []
            for f0 in range(0, T, step):
                frames = [[t, jpgs[t] if jpgs is not None else None, pngs[t] if pngs is not None else None] for t in range(f0, min(T, f0 + step))]
                futs.append(self.waiters.submit(self._call, {"rgb": rgb.name if rgb else None, "lab": lab.name if lab else None, "shape": [T, H, W], "frames": frames}))
            for f in futs:
                f.result()
            u8 = np.ndarray((T, H, W, 3), np.uint8, buffer=rgb.buf).copy() if rgb else None
            lb = np.ndarray((T, H, W), np.uint8, buffer=lab.buf).copy() if lab else None
            return u8, lb
        finally:
            for m in (rgb, lab):
                if m is not None:
                    m.close()
                    m.unlink()


def decode_pool():
    """The process-wide image-decode pool shared by every lane's loader (what matters is that the NEXT sample of every lane is ready when
    the lane asks for it, not who decodes it).  STCN_DECODE_PROCS worker processes (default min(8, host cores / 4); see _DecodeProcs);
    STCN_DECODE_PROCS=0: a pool of STCN_DECODE_THREADS threads instead (default min(16, cores / 2); 0 = decode inline)."""
    global _DECODE_POOL
    cores = os.cpu_count() or 2
    n_proc = int(os.environ.get("STCN_DECODE_PROCS", min(8, max(1, cores // 4))))
    n_thr = int(os.environ.get("STCN_DECODE_THREADS", min(16, max(1, cores // 2))))
    if n_proc <= 0 and n_thr <= 0:
        return None
    with _DECODE_LOCK:
        if _DECODE_POOL is None:
            if n_proc > 0 and os.path.isdir("/dev/shm"):
                try:
                    _DECODE_POOL = _DecodeProcs(n_proc)
                except OSError:
                    _DECODE_POOL = None
            if _DECODE_POOL is None and n_thr > 0:
                from concurrent.futures import ThreadPoolExecutor
                _DECODE_POOL = ThreadPoolExecutor(n_thr, thread_name_prefix="stcn-decode")
        return _DECODE_POOL


class ClipDataset:
    """DAVIS/MOSE directory layout: JPEGImages/480p/<video>/%05d.jpg, Annotations/480p/<video>/%05d.png
    (palette index = object id).  One sample per (video, object), named ``<video>__<obj>``."""

    def __init__(self, root: str, imset: str, resolution: str = "480p"):
        self.image_dir = os.path.join(root, "JPEGImages", resolution)
        self.mask_dir = os.path.join(root, "Annotations", resolution)
        self.samples: List[tuple] = []          # (video, object id, frames)
        for line in open(imset):
            v = line.strip()
            if not v:
                continue
            first = np.array(Image.open(os.path.join(self.mask_dir, v, "00000.png")).convert("P"))
            n = len(os.listdir(os.path.join(self.image_dir, v)))
            for obj in range(1, int(first.max()) + 1):
                self.samples.append((v, obj, n))
        self._cache: Dict[str, tuple] = {}

    def __len__(self):
        return len(self.samples)

    def name(self, i):
        v, o, _ = self.samples[i]
        return f"{v}__{o}"

    def _clip(self, video: str, n: int):
        """Decoded once per video (samples of a video are adjacent): the frames as uint8 [T,H,W,3] - pinned when a GPU is there, so
        that the prefetcher can upload the BYTES (a quarter of the fp32 clip) and normalise on the device - and the label maps.
        Round 6: the frames of a video are decoded by a SHARED pool of host threads (PIL releases the GIL while it decodes): one thread
        per lane took ~150 ms per 40-frame 480p video, more than the GPU needs for its eight rounds - the lanes waited for their loaders."""
        if video not in self._cache:
            self._cache.clear()                  # keep one decoded clip

            def frame(f):
                return np.asarray(Image.open(os.path.join(self.image_dir, video, f"{f:05d}.jpg")).convert("RGB"))

            def label(f):
                return np.array(Image.open(os.path.join(self.mask_dir, video, f"{f:05d}.png")).convert("P"), dtype=np.uint8)

            pool = decode_pool()
            if isinstance(pool, _DecodeProcs):
                w_, h_ = Image.open(os.path.join(self.image_dir, video, "00000.jpg")).size        # header only
                u8, lab = pool.decode([os.path.join(self.image_dir, video, f"{f:05d}.jpg") for f in range(n)],
                                      [os.path.join(self.mask_dir, video, f"{f:05d}.png") for f in range(n)], h_, w_)
            elif pool is None:
                u8, lab = np.stack([frame(f) for f in range(n)]), np.stack([label(f) for f in range(n)])
            else:
                fr, lb = [pool.submit(frame, f) for f in range(n)], [pool.submit(label, f) for f in range(n)]
                u8, lab = np.stack([x.result() for x in fr]), np.stack([x.result() for x in lb])
            u8 = torch.from_numpy(np.ascontiguousarray(u8))
            if torch.cuda.is_available():
                u8 = u8.pin_memory()
            self._cache[video] = [u8, torch.from_numpy(lab), None]
        return self._cache[video]

    @staticmethod
    def normalize_host(u8: torch.Tensor) -> torch.Tensor:
        """ToTensor + ImageNet normalisation on the host, [T,H,W,3] uint8 -> [T,3,H,W] float32 (datasets/range_transform.py:3-6)."""
        rgb = u8.numpy().astype(np.float32) / 255.0
        return torch.from_numpy(np.ascontiguousarray(((rgb - MEAN) / STD).transpose(0, 3, 1, 2)))

    @staticmethod
    def normalize_lut() -> np.ndarray:
        """The 3 x 256 values normalize_host can produce: a byte has 256 values per channel, so the host arithmetic evaluated on them IS the
        whole function - a table look-up on the device is bit-identical to the host path by construction (no GPU division in between)."""
        b = np.arange(256, dtype=np.uint8)[:, None].repeat(3, 1)                    # [256, 3]: byte value in every channel
        return np.ascontiguousarray(((b.astype(np.float32) / 255.0 - MEAN) / STD).T)  # [3, 256] float32, the expression of normalize_host

    @staticmethod
    def normalize_device(u8: torch.Tensor) -> torch.Tensor:
        """normalize_host on the device the bytes were uploaded to: per channel a 256-entry table look-up (one gather pass instead of four
        host passes over 200 MB).  Bit-identical to normalize_host (tests/test_gpu_driver_golden.py asserts equality, not closeness)."""
        lut = torch.from_numpy(ClipDataset.normalize_lut()).to(u8.device)
        out = torch.empty((u8.shape[0], 3) + tuple(u8.shape[1:3]), dtype=torch.float32, device=u8.device)
        for c in range(3):
            out[:, c] = torch.nn.functional.embedding(u8[..., c].to(torch.int32), lut[c].view(256, 1)).squeeze(-1)
        return out

    def _meta(self, i):
        v, obj, n = self.samples[i]
        entry = self._clip(v, n)
        gt = (entry[1] == obj).float()[None, :, None]              # [1,T,1,H,W], no bg channel (reference layout)
        return entry, {"gt": gt, "name": self.name(i), "video": v, "num_frames": n}

    def __getitem__(self, i):
        entry, sample = self._meta(i)
        if entry[2] is None:
            entry[2] = self.normalize_host(entry[0])               # host users (CPU tests, reference-style loops): normalised once per video
        sample["rgb"] = entry[2][None]
        return sample

    def raw(self, i):
        """The sample with its frames as uint8 [T,H,W,3] (``rgb_u8``) instead of the normalised float clip: for the prefetcher."""
        entry, sample = self._meta(i)
        sample["rgb_u8"] = entry[0]
        return sample


def prefetched(ds: "ClipDataset", indices, device: str = "cuda", timing: dict = None):
    """Yield (i, sample) for i in indices with sample['rgb'] already on `device`.  The NEXT sample is decoded on a
    host thread (PIL releases the GIL) into pinned memory as BYTES, copied H2D and normalised on a side stream while the caller
    propagates the current one (round 4: the host used to normalise in four NumPy passes over 200 MB per clip and upload fp32) - the reference decodes and uploads synchronously between samples (generate_fq_dataset.py:60-63)."""
    from concurrent.futures import ThreadPoolExecutor
    on_gpu = torch.cuda.is_available() and str(device).startswith("cuda")
    side = torch.cuda.Stream() if on_gpu else None

    def load(i):
        if not on_gpu:
            return dict(ds[i])
        sample = dict(ds.raw(i))
        host = sample.pop("rgb_u8")                               # pinned uint8 [T,H,W,3]: a quarter of the fp32 clip crosses PCIe
        with torch.cuda.stream(side):
            sample["rgb"] = ds.normalize_device(host.to(device, non_blocking=True))[None]      # normalised on the device, on the side stream
            ev = torch.cuda.Event()
            ev.record(side)
        sample["_ready"], sample["_host"] = ev, host              # keep the pinned buffer alive until the copy is done
        return sample

    indices = list(indices)
    with ThreadPoolExecutor(1) as pool:
        fut = pool.submit(load, indices[0]) if indices else None
        for n, i in enumerate(indices):
            t_w = time.perf_counter()
            sample = fut.result()
            if timing is not None:                                # how long the lane stood waiting for its loader (decode + upload not hidden)
                timing["wait_input_s"] = timing.get("wait_input_s", 0.0) + time.perf_counter() - t_w
            fut = pool.submit(load, indices[n + 1]) if n + 1 < len(indices) else None
            if on_gpu:
                torch.cuda.current_stream().wait_event(sample.pop("_ready"))
                sample["rgb"].record_stream(torch.cuda.current_stream())
                sample.pop("_host")
            yield i, sample


_STATS_LOCK = __import__("threading").Lock()


def lane_engine_options(lanes: int):
    """Engine tunables for `lanes` videos in flight per GPU: the engines' own side streams (key-encoder look-ahead, FusionNet of
    rounds >= 2) help ONE video in flight (+6..8 %); with several lanes the videos already fill each other's gaps and eight
    streams only contend (second interactions -3 %).  Given to InferenceCore explicitly - the process environment is not touched
    (an STCN_LOOKAHEAD set by the user still wins)."""
    return {"lookahead": 0} if lanes > 1 and "STCN_LOOKAHEAD" not in os.environ else None


def run_lanes(root: str, imset: str, mine, lanes: int, work, device: str = "cuda", stats: dict = None):
    """Process the samples `mine` on `lanes` host threads, each with its own HIP stream, clip loader and prefetcher
    (samples are independent; two videos in flight fill each other's kernel tails, as bench.py's lanes do).
    work(i, sample) -> list of rows; returns all rows.  Chunks are contiguous so that the objects of one video stay
    with one loader (per-video decode cache)."""
    from concurrent.futures import ThreadPoolExecutor
    mine = list(mine)
    lanes = max(1, min(lanes, len(mine)))
    on_gpu = torch.cuda.is_available() and str(device).startswith("cuda")
    dev_index = torch.cuda.current_device() if on_gpu else None
    bounds = [len(mine) * l // lanes for l in range(lanes + 1)]

    def lane(l):
        ds = ClipDataset(root, imset)
        rows = []
        if on_gpu:
            torch.cuda.set_device(dev_index)
            ctx = torch.cuda.stream(torch.cuda.Stream())
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        timing = {} if stats is not None else None
        t_lane = time.perf_counter()
        with ctx:
            for i, sample in prefetched(ds, mine[bounds[l]:bounds[l + 1]], device, timing):
                t_s = time.perf_counter()
                rows += work(i, sample)
                if timing is not None:
                    timing["work_s"] = timing.get("work_s", 0.0) + time.perf_counter() - t_s
            if on_gpu:
                torch.cuda.current_stream().synchronize()
        if stats is not None:                                      # host-side account of the lane: where its wall time went
            timing["lane_s"] = time.perf_counter() - t_lane
            with _STATS_LOCK:
                stats.setdefault("lanes", []).append({k_: round(v_, 4) for k_, v_ in timing.items()})
        return rows

    if lanes == 1:
        return lane(0)
    with ThreadPoolExecutor(lanes) as ex:
        return [r for part in ex.map(lane, range(lanes)) for r in part]


def make_synthetic_tree(root: str, videos: Dict[str, tuple], seed: int = 0) -> str:
    """Write a tiny DAVIS-layout dataset (for tests / smoke runs).  videos: name -> (T, H, W, n_objects)."""
    from . import synth
    names = []
    for vi, (name, (T, H, W, k)) in enumerate(videos.items()):
        os.makedirs(os.path.join(root, "JPEGImages", "480p", name), exist_ok=True)
        os.makedirs(os.path.join(root, "Annotations", "480p", name), exist_ok=True)
        img = synth.synthetic_clip(T, H, W, seed=seed + vi)[0].numpy()
        msk = synth.synthetic_mask(T, H, W, k, seed=seed + vi)[:, :, 0].numpy()
        for t in range(T):
            rgb = np.clip((img[t].transpose(1, 2, 0) * STD + MEAN) * 255, 0, 255).astype(np.uint8)
            Image.fromarray(rgb).save(os.path.join(root, "JPEGImages", "480p", name, f"{t:05d}.jpg"), quality=95)
            lab = np.zeros((H, W), np.uint8)
            for o in range(k):
                lab[msk[o, t] > 0.5] = o + 1
            pim = Image.fromarray(lab, mode="P")
            pim.putpalette([0, 0, 0, 255, 0, 0, 0, 255, 0, 0, 0, 255] + [0] * (256 * 3 - 12))
            pim.save(os.path.join(root, "Annotations", "480p", name, f"{t:05d}.png"))
        names.append(name)
    os.makedirs(os.path.join(root, "ImageSets"), exist_ok=True)
    imset = os.path.join(root, "ImageSets", "synthetic.txt")
    open(imset, "w").write("\n".join(names) + "\n")
    return imset


# ------------------------------------------------------------------------------------------------ policy
def per_frame_j(processor, gt_dev: torch.Tensor, interacted: List[int]) -> tuple:
    """J per frame on the GPU.  Annotated frames count with their GT mask (interactions/eval.py:57-60);
    frames without the object get the NO_OBJECT token.  Returns (quality[T], generated masks uint8 [T,H,W])."""
    lw, uw, lh, uh = processor.pad
    seg = processor.masks[:, 0, lh:processor.nh - uh, lw:processor.nw - uw] > 0
    gtb = gt_dev > 0.5
    gen = seg.clone()
    if interacted:
        gen[interacted] = gtb[interacted]
    rows = metrics.sequence_scores_gpu(gtb, gen, j_only=True)               # the FQ policy selects by J (interactions/mask.py:113-146)
    q = rows[:, 0].copy()
    q[(gtb.flatten(1).sum(1) == 0).cpu().numpy()] = NO_OBJECT
    return q, gen.to(torch.uint8)


def oracle_rounds(processor, sample, rounds: int = 8, stats: dict = None):
    """The oracle annotation policy of the FQ dataset (interactions/mask.py:113-156).  `stats` (optional dict, one per caller thread):
    the frames the engine really visited (`propagated_frames`) and the `interactions` are added to it.
    Round 6: the evaluation of a round stays on the device (metrics.RoundScorer: compose + J counts + fp64 quality + arg-min in one enqueue);
    one int per round crosses PCIe, the per-frame J rows of all rounds are fetched once at the end.  Same values, same selection as
    per_frame_j + numpy.argmin (tests/test_gpu_driver_golden.py)."""
    T = sample["num_frames"]
    gt = sample["gt"][0].to(processor.prob.device)              # [T,1,H,W]
    scorer = metrics.RoundScorer(gt[:, 0], "j", max_rounds=max(rounds, 1), no_object=NO_OBJECT)
    valid = set(np.where(~scorer.empty_host)[0].tolist())       # frames that can be annotated at all (the others carry the NO_OBJECT token)
    frames, sels, gens = [0], [], []
    for r in range(1, rounds + 1):
        if r >= T:
            continue
        if sels and not (valid - set(frames)):                  # not_avail_frames (interactions/eval.py:84-89): nothing left to annotate
            continue
        f = frames[r - 1]
        processor.interact(gt[f][None], f, download=False)         # [1,1,H,W] mask of the annotated frame
        if stats is not None:
            stats["propagated_frames"] = stats.get("propagated_frames", 0) + processor.stats()["frames"]
            stats["interactions"] = stats.get("interactions", 0) + 1
        worst, gen = scorer.score(processor, frames[:r])
        frames.append(worst)
        sels.append(worst)
        gens.append(gen)
    q = scorer.qualities()
    return [(w, q[i]) for i, w in enumerate(sels)], gens


# ------------------------------------------------------------------------------------------------ output
def write_png(path: str, a: np.ndarray, level: int = 1) -> None:
    """8-bit grayscale [H,W] or RGB [H,W,3] PNG, written with zlib directly: the writer threads of a run encode ~3 000 small images,
    and PIL's encoder holds the GIL while it works - the lanes that drive the GPU then wait for the interpreter (fq_driver: 52 rounds/s
    without output, 28 with it through PIL).  ``zlib.compress`` / ``zlib.crc32`` release the GIL on buffers of this size, so the
    encoding really runs beside the lanes.  Filter type 0 on every scanline; any PNG reader decodes the same pixels."""
    import struct
    import zlib
    a = np.ascontiguousarray(a, dtype=np.uint8)
    h, w = a.shape[:2]
    ctype = 2 if a.ndim == 3 else 0
    raw = np.empty((h, 1 + w * (3 if ctype == 2 else 1)), np.uint8)
    raw[:, 0] = 0
    raw[:, 1:] = a.reshape(h, -1)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + chunk(b"IEND", b""))



def save_state_masks(gen: torch.Tensor, out_dir: str, pool=None):
    """224x224 nearest-neighbour PNGs like util/fq_dataset.py:64-84 (mask_to_224).  The resize runs on the device; with
    `pool` (a ThreadPoolExecutor) the PNG encoding + file writes of the state happen on host threads while the GPU
    propagates the next round (zlib and file I/O release the GIL); returns the future, or None when done inline.
    Round 6: with a pool the download does not stop the lane either - the 224x224 frames go to pinned memory asynchronously and the
    WRITER thread waits for the copy (an event), not the thread that drives the GPU."""
    small = torch.nn.functional.interpolate(gen[:, None].float(), size=(224, 224), mode="nearest")[:, 0]
    small = (small * 255).to(torch.uint8)
    if pool is None or not small.is_cuda:
        frames, ready = small.cpu().numpy(), None
    else:
        host = torch.empty(small.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(small, non_blocking=True)
        ready = torch.cuda.Event(blocking=True)
        ready.record()
        small.record_stream(torch.cuda.current_stream())
        frames = host.numpy()

    def write():
        if ready is not None:
            ready.synchronize()
        os.makedirs(out_dir, exist_ok=True)
        for t, m in enumerate(frames):
            write_png(os.path.join(out_dir, f"{t:05d}.png"), m)

    if pool is None:
        write()
        return None
    return pool.submit(write)


def save_rgb_frames(rgb: torch.Tensor, out_dir: str, pool=None):
    """RGBFrames/224/<video>/%05d.png like util/fq_dataset.py:26-47 (``save_frames``): each normalised frame is resized to
    224x224 (bicubic, antialias), min-max normalised PER FRAME (``im_normalize``) and written as 8-bit RGB - the inputs of
    the reference's QNet training set.  rgb: [T,3,H,W] float (ImageNet-normalised, host or device); the resize runs where
    the tensor lives, the PNG encoding on `pool` threads when given."""
    small = torch.nn.functional.interpolate(rgb.float(), size=(224, 224), mode="bicubic", align_corners=False, antialias=True)
    lo = small.amin((1, 2, 3), keepdim=True)
    hi = small.amax((1, 2, 3), keepdim=True)
    dev8 = ((small - lo) / (hi - lo).clamp_min(1e-8) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    if pool is None or not dev8.is_cuda:
        frames, ready = dev8.cpu().numpy(), None
    else:                                                           # as save_state_masks: the WRITER waits for the download, not the lane
        host = torch.empty(dev8.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(dev8, non_blocking=True)
        ready = torch.cuda.Event(blocking=True)
        ready.record()
        dev8.record_stream(torch.cuda.current_stream())
        frames = host.numpy()

    def write():
        if ready is not None:
            ready.synchronize()
        os.makedirs(out_dir, exist_ok=True)
        for t, m in enumerate(frames):
            write_png(os.path.join(out_dir, f"{t:05d}.png"), m)

    if pool is None:
        write()
        return None
    return pool.submit(write)


def run(root: str, imset: str, out: str, prop_net, fuse_net, rounds: int = 8, save_masks: bool = True,
        device: str = "cuda", lanes: int = 2, stats: dict = None):
    """Process this rank's share of the samples (`lanes` videos in flight); returns the gathered rows on every rank
    (rows: sample id, round, selected frame, T, then T per-frame J values padded with NaN).
    `stats` (optional dict): this rank's `propagated_frames` (frames the engines visited) and `interactions` are added to it."""
    import torch.distributed as dist
    from concurrent.futures import ThreadPoolExecutor

    from mivos.inference_core import InferenceCore
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    ds = ClipDataset(root, imset)
    t_max = max(s[2] for s in ds.samples)
    mine = sorted(shard.lpt_assign([s[2] for s in ds.samples], world)[rank])     # adjacent objects share a decode
    width = 4 + t_max
    writers = ThreadPoolExecutor(8) if save_masks else None           # zlib releases the GIL: the encoders really run beside the lanes
    pending = []

    saved_rgb = set()
    rgb_lock = __import__("threading").Lock()

    def work(i, sample):
        rows = []
        if save_masks:                                     # once per video (generate_fq_dataset.py:77-80)
            with rgb_lock:
                first = sample["video"] not in saved_rgb
                saved_rgb.add(sample["video"])
            if first:
                pending.append(save_rgb_frames(sample["rgb"][0], os.path.join(out, "RGBFrames", "224", sample["video"]), writers))
        t_c = time.perf_counter()
        proc = InferenceCore(prop_net, fuse_net, sample["rgb"], 1, engine_options=lane_engine_options(lanes))
        mine_stats = {"create_s": time.perf_counter() - t_c} if stats is not None else None
        t_c = time.perf_counter()
        states, gens = oracle_rounds(proc, sample, rounds, mine_stats)
        if stats is not None:
            mine_stats["session_s"] = time.perf_counter() - t_c
            with rgb_lock:
                for k_, v_ in mine_stats.items():
                    stats[k_] = stats.get(k_, 0) + v_
        sid = 1
        for r, ((worst, q), gen) in enumerate(zip(states, gens)):
            if float(q[worst]) == NO_OBJECT:
                continue                                   # util/fq_dataset.py:55: no-object states are not saved
            row = np.full(width, np.nan, np.float32)
            row[:4] = (i, sid, worst, len(q))
            row[4:4 + len(q)] = q
            rows.append(row)
            if save_masks:
                pending.append(save_state_masks(gen, os.path.join(out, "Annotations", "224", f"{sample['name']}_round_{sid}"), writers))
            sid += 1
        return rows

    rows = run_lanes(root, imset, mine, lanes, work, device, stats)
    t_p = time.perf_counter()
    for f in pending:
        f.result()                                         # surface write errors; all PNGs are on disk before the CSV
    if stats is not None:
        stats["wait_writers_s"] = round(time.perf_counter() - t_p, 4)
    if writers is not None:
        writers.shutdown()
    allrows = shard.gather_rows(np.stack(rows) if rows else np.zeros((0, width), np.float32), width)
    if rank == 0:
        os.makedirs(out, exist_ok=True)
        order = np.lexsort((allrows[:, 1], allrows[:, 0]))
        # res_<imset>.csv, columns as generate_fq_dataset.py:48-52,85-86 (the ious cell holds the per-frame list)
        imset_str = os.path.splitext(os.path.basename(imset))[0]
        with open(os.path.join(out, f"res_{imset_str}.csv"), "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["state_name", "ious", "selected_frame"])
            for row in allrows[order]:
                n = int(row[3])
                wr.writerow([f"{ds.name(int(row[0]))}_round_{int(row[1])}", [float(v) for v in row[4:4 + n]], int(row[2])])
    return allrows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    ap.add_argument("--imset", required=True)
    ap.add_argument("--out", default="FQ_DB")
    ap.add_argument("--prop-weights", default="./model_weights/mivos/stcn_yt_vos.pth")
    ap.add_argument("--fusion-weights", default="./model_weights/mivos/fusion_stcn_yt_vos.pth")
    ap.add_argument("--synthetic-weights", action="store_true", help="use the deterministic recipe (no checkpoints)")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--lanes", type=int, default=2, help="videos in flight per GPU")
    a = ap.parse_args()
    import torch.distributed as dist

    from . import synth
    from .params import FusionNet, PropagationNetwork
    torch.set_grad_enabled(False)
    shard.init_from_env()                                  # one process per GPU; RCCL unless STCN_DIST_BACKEND says otherwise
    prop, fuse = PropagationNetwork(), FusionNet()
    if a.synthetic_weights:
        prop.load_state_dict(synth.recipe_state_dict(prop))
        fuse.load_state_dict(synth.recipe_state_dict(fuse))
    else:
        prop.load_state_dict(torch.load(a.prop_weights, map_location="cpu"))
        fuse.load_state_dict(torch.load(a.fusion_weights, map_location="cpu"))
    rows = run(a.root, a.imset, a.out, prop.eval(), fuse.eval(), a.rounds, lanes=a.lanes)
    if not dist.is_initialized() or dist.get_rank() == 0:
        print(f"{len(rows)} states -> {os.path.join(a.out, 'res_' + os.path.splitext(os.path.basename(a.imset))[0] + '.csv')}")
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
