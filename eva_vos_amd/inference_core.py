"""Drop-in ``InferenceCore`` over the HIP engine (libstcn_hip.so).

Mirrors the public surface of the reference class ``mivos/inference_core.py:16-259``:
``InferenceCore(prop_net, fuse_net, images, num_objects, mem_profile=0, mem_freq=5, device='cuda')``,
``.interact(mask, idx, scribble=False) -> np.uint8[t,h,w]`` and the attributes its callers read
(``prob``, ``pad``, ``t``, ``masks``, ``np_masks``, ``k``, ``h``, ``w``, ``nh``, ``nw``, ``kh``,
``kw``; ``get_image_buffered``; ``copy.deepcopy``).  This file only marshals arguments: padding,
feature caching, the memory bank, both propagation passes, fusion and the argmax all run inside the
engine on the current HIP stream.  PyTorch is used for device memory and stream ordering only.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import time
import weakref

import numpy as np
import torch

from . import _lib

_MODEL_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_PINNED_DOWNLOAD = os.environ.get("STCN_PINNED_DOWNLOAD", "1") != "0"      # 0: the reference's plain .cpu() (measurement aid)
# interact() returns its masks in a PINNED host block that the returned array owns (one DMA instead of a staged copy).  A caller that keeps
# every round's result - the reference-style loops do - would pile up page-locked memory (27 MB per 66-frame 480p clip).  The budget is
# PER CORE (round 6; a process-wide count of 8 was smaller than the working set of bench.py's own 8 engines, whose later downloads silently
# went pageable): at most _PINNED_MAX_LIVE result blocks of one core are alive at a time, and at most STCN_PINNED_MAX_MB of page-locked result
# memory per process; beyond either a result comes back in pageable memory, as the reference's .cpu() does.  pageable_downloads() counts those.
_PINNED_MAX_LIVE = int(os.environ.get("STCN_PINNED_MAX_LIVE", "4"))
_PINNED_MAX_BYTES = int(float(os.environ.get("STCN_PINNED_MAX_MB", "4096")) * (1 << 20))
_PINNED_BYTES = [0]                  # page-locked result bytes alive in this process
_PAGEABLE_DOWNLOADS = [0]            # interact() downloads that went through pageable memory (budget exhausted or STCN_PINNED_DOWNLOAD=0)
_PINNED_LOCK = threading.RLock()          # re-entrant: a block's __del__ may run (GC) on a thread that holds it


def pageable_downloads() -> int:
    """How many interact() downloads of this process did NOT get a pinned block (a measurement aid: bench.py prints it)."""
    return _PAGEABLE_DOWNLOADS[0]


class _PinnedBlock:
    """Owner of one pinned result block.  The array handed to the caller is ``np.asarray(block)``: NumPy keeps the object that provides
    ``__array_interface__`` as the array's base, so the block - and the weak reference that counts it - lives exactly as long as the
    array or any view of it.  (A weak reference to the tensor itself dies as soon as the Python wrapper is dropped, whatever still
    holds the storage.)"""
    __slots__ = ("tensor", "nbytes", "__array_interface__", "__weakref__")

    def __init__(self, tensor):
        self.tensor = tensor
        self.nbytes = tensor.numel()
        self.__array_interface__ = {"shape": tuple(tensor.shape), "typestr": "|u1", "data": (tensor.data_ptr(), False), "version": 3}

    def __del__(self):
        with _PINNED_LOCK:
            _PINNED_BYTES[0] -= self.nbytes


def _pinned_result(shape, live: list):
    """A pinned uint8 result block, or None when the core's `live` list (weak references to its earlier blocks) still holds
    _PINNED_MAX_LIVE blocks kept alive by their arrays, or the process holds STCN_PINNED_MAX_MB of them."""
    nbytes = int(np.prod(shape))
    with _PINNED_LOCK:
        live[:] = [r for r in live if r() is not None]
        if len(live) >= _PINNED_MAX_LIVE or _PINNED_BYTES[0] + nbytes > _PINNED_MAX_BYTES:
            _PAGEABLE_DOWNLOADS[0] += 1
            return None
        _PINNED_BYTES[0] += nbytes
    try:
        host = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
    except RuntimeError:                                            # the host refuses more page-locked memory: pageable, as beyond the budget
        with _PINNED_LOCK:
            _PINNED_BYTES[0] -= nbytes
            _PAGEABLE_DOWNLOADS[0] += 1
        return None
    block = _PinnedBlock(host)
    with _PINNED_LOCK:
        live.append(weakref.ref(block))
    return block


class _Model:
    """Owns one ``stcn_model`` handle (BN-folded, repacked weights on one device)."""

    def __init__(self, prop_net, fuse_net, device_index: int):
        lib = _lib.lib()
        keep = []

        def descs(module):
            sd = {k: v for k, v in module.state_dict().items() if v.is_floating_point()}
            arr = (_lib.WeightDesc * len(sd))()
            for i, (name, t) in enumerate(sd.items()):
                t = t.detach().to("cpu", torch.float32).contiguous()
                keep.append(t)
                arr[i].name = name.encode()
                arr[i].data = t.data_ptr()
                arr[i].ndim = t.dim()
                for d in range(t.dim()):
                    arr[i].shape[d] = t.shape[d]
            return arr, len(sd)

        p, n_p = descs(prop_net)
        f, n_f = descs(fuse_net) if fuse_net is not None else (None, 0)
        h = C.c_void_p()
        _lib.check(lib.stcn_model_create(device_index, p, n_p, f, n_f, C.byref(h)), "stcn_model_create")
        self.handle = h
        self._finalizer = weakref.finalize(self, lib.stcn_model_destroy, h)


_MODEL_LOCK = threading.Lock()


_FP_TENSORS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()      # module -> (time of the last state_dict() walk, weak refs to its tensors)
_FP_RESCAN_S = float(os.environ.get("STCN_FINGERPRINT_RESCAN_S", "2.0"))


def _fingerprint(module) -> tuple:
    """(storage address, in-place version) of every tensor of the state_dict - changes when a checkpoint is loaded into the
    module (load_state_dict copies in place and bumps the versions) or a parameter is replaced - plus a content probe of a
    few tensors: writes through ``p.data`` (``p.data.copy_()``, ``p.data.mul_()``) do NOT bump ``_version``; the probe
    (sums over the first 4096 elements of 8 tensors spread over the state_dict) catches a whole-model update done that
    way.  Limitation, by design of a snapshot: a ``.data`` write confined to tensors outside the probe is not seen - call
    ``eva_vos_amd.inference_core.forget_models()`` after such surgery (the reference reads live parameters).
    Round 6: this runs once per InferenceCore, i.e. once per SAMPLE in the reference-style drivers, from lane threads that share the
    interpreter with image-decode and PNG-writer threads - the 0.6 ms of Python below took 30-60 ms there (GIL hand-offs; measured by
    tools/driver_lanes.py).  The walk over ``state_dict()`` (the bulk of it) is therefore repeated at most every STCN_FINGERPRINT_RESCAN_S
    seconds per module; in between the tensors found by the last walk are checked (weak references: a parameter that was replaced and
    freed forces a new walk at once), addresses, versions and content probe as before."""
    if module is None:
        return ()
    now = time.monotonic()
    hit = _FP_TENSORS.get(module)
    sd = None
    if hit is not None and now - hit[0] < _FP_RESCAN_S:
        sd = [r() for r in hit[1]]
        if any(t is None for t in sd):
            sd = None
    if sd is None:
        sd = list(module.state_dict(keep_vars=True).values())
        _FP_TENSORS[module] = (now, [weakref.ref(t) for t in sd])
    ids = tuple((v.data_ptr(), v._version) for v in sd)
    fl = [v for v in sd if v.is_floating_point() and v.numel() > 0]
    if not fl:
        return ids
    # the content probe as ONE concatenated slice (two reductions, one host transfer: the reference-style drivers build one core per
    # sample from several lane threads; 16 blocking float() calls per module once serialised them)
    x = torch.cat([fl[i].detach().reshape(-1)[:4096] for i in sorted({(len(fl) - 1) * j // 7 for j in range(8)})]).double()
    return ids + tuple(torch.stack([x.sum(), x.abs().sum(), (x * x).sum()]).tolist())


_SNAPSHOTS_PER_DEVICE = 4      # LRU: alternating a few (prop_net weights, fuse_net) pairs must not re-fold the model every time


def forget_models() -> None:
    """Drop every cached weight snapshot (engines alive keep theirs): the next InferenceCore re-reads the modules."""
    with _MODEL_LOCK:
        _MODEL_CACHE.clear()


def release_pooled_memory() -> None:
    """Return the device memory of destroyed engines (parked in the library's buffer pool for the next engine) to the driver."""
    _lib.check(_lib.lib().stcn_pool_release(), "stcn_pool_release")


def _model_for(prop_net, fuse_net, device_index: int) -> _Model:
    """The engine works on a BN-folded, repacked SNAPSHOT of the weights.  The reference reads the live parameters, so the
    snapshot is keyed on a fingerprint of both modules' tensors: loading another checkpoint into the same module objects
    (one script evaluating several checkpoints) yields a fresh model.  A small LRU per (prop_net, device) keeps the last
    few snapshots, so alternating two fusion networks (or fuse_net / None) does not re-upload 218 MB per construction;
    older ones die with their last engine."""
    key = (device_index, _fingerprint(prop_net), _fingerprint(fuse_net))      # outside the lock: it may sync the device
    with _MODEL_LOCK:                       # engines may be created from several host threads (one per video)
        per_net = _MODEL_CACHE.setdefault(prop_net, {})
        hit = per_net.get(key)
        # the fuse_net is held weakly and compared by identity: a new module that happens to reuse a freed one's id /
        # storage addresses must not alias its packed weights
        if hit is not None and (hit[1]() if hit[1] is not None else None) is fuse_net:
            per_net[key] = per_net.pop(key)                         # most recently used last
            return hit[0]
        per_net.pop(key, None)
        mine = [k for k in per_net if k[0] == device_index]
        for k in mine[:max(0, len(mine) - (_SNAPSHOTS_PER_DEVICE - 1))]:     # least recently used first
            del per_net[k]
        model = _Model(prop_net, fuse_net, device_index)
        per_net[key] = (model, weakref.ref(fuse_net) if fuse_net is not None else None)
        return model


def _pad16(n: int):
    d = (-n) % 16
    lo = d // 2
    return lo, d - lo


# measurement aid (tools/driver_lanes.py): when set to a dict, InferenceCore.__init__ adds the host seconds of its phases to it
CREATE_ACCOUNT = None


class InferenceCore:
    def __init__(self, prop_net, fuse_net, images, num_objects, mem_profile=0, mem_freq=5, device="cuda", engine_options=None):
        """``engine_options`` (not in the reference; results never depend on it): dict with any of ``lookahead``, ``decode_batch``,
        ``key_batch``, ``fuse_side`` - the engine's tunables given explicitly (include/stcn_hip.h: stcn_engine_opts) instead of
        through the STCN_* environment variables, e.g. ``{"lookahead": 0}`` for drivers that keep several videos in flight."""
        if not torch.cuda.is_available():
            raise RuntimeError("eva_vos_amd.InferenceCore needs a HIP device (there is no CPU fallback); "
                               "the CPU oracle lives in oracle/ and is test infrastructure only")
        self.device = torch.device(device if device != "cuda" else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise RuntimeError(f"device {device!r}: the HIP engine only runs on a GPU")
        self.prop_net, self.fuse_net = prop_net, fuse_net
        self.mem_profile, self.mem_freq = mem_profile, mem_freq   # mem_profile: results never depend on it
        self.data_dev = self.result_dev = self.device
        self.k = int(num_objects)
        t = images.shape[1]
        h, w = images.shape[-2:]
        self.t, self.h, self.w = int(t), int(h), int(w)
        (lh, uh), (lw, uw) = _pad16(self.h), _pad16(self.w)
        self.pad = (lw, uw, lh, uh)
        self.nh, self.nw = self.h + lh + uh, self.w + lw + uw
        self.kh, self.kw = self.nh // 16, self.nw // 16
        t_0 = time.perf_counter()
        self._model = _model_for(prop_net, fuse_net, self.device.index or 0)
        t_1 = time.perf_counter()
        with torch.cuda.device(self.device):
            self._stream = torch.cuda.current_stream()
            def alloc():
                return (images.detach().to(self.device, torch.float32).contiguous(),
                        torch.empty((self.k + 1, self.t, 1, self.nh, self.nw), dtype=torch.float32, device=self.device),
                        torch.empty((self.t, 1, self.nh, self.nw), dtype=torch.uint8, device=self.device))
            try:
                imgs, self.prob, self.masks = alloc()
            except torch.cuda.OutOfMemoryError:
                # the library parks the buffers of destroyed engines in its own pool (STCN_POOL_GB), outside PyTorch's caching
                # allocator: hand them back to the driver and try once more before giving up
                release_pooled_memory()
                torch.cuda.empty_cache()
                imgs, self.prob, self.masks = alloc()
            self.np_masks = np.zeros((self.t, self.h, self.w), dtype=np.uint8)
            t_2 = time.perf_counter()
            h_ = C.c_void_p()
            eo = dict(engine_options or {})
            opts = _lib.EngineOpts(*(int(eo.pop(n, -1)) for n, _ in _lib.EngineOpts._fields_))
            if eo:
                raise TypeError(f"unknown engine_options {sorted(eo)}")
            _lib.check(_lib.lib().stcn_engine_create_ex(
                self._model.handle, self.t, self.h, self.w, self.k, int(mem_freq), self._stream.cuda_stream,
                imgs.data_ptr(), self.prob.data_ptr(), self.masks.data_ptr(), C.byref(opts), C.byref(h_)), "stcn_engine_create")
        if CREATE_ACCOUNT is not None:
            t_3 = time.perf_counter()
            with _MODEL_LOCK:
                for k_, v_ in (("model_for_s", t_1 - t_0), ("torch_alloc_s", t_2 - t_1), ("engine_create_s", t_3 - t_2)):
                    CREATE_ACCOUNT[k_] = CREATE_ACCOUNT.get(k_, 0.0) + v_
        self._images_unpadded = imgs
        self._engine = h_
        self._finalizer = weakref.finalize(self, _lib.lib().stcn_engine_destroy, h_)
        self.interacted = set()
        self.last_enqueue_s = 0.0
        self._dl_event = None                                      # blocking event of the mask download (created on first use)
        self._pinned_live = []                                     # weak references to this core's pinned result blocks still held by arrays

    # ------------------------------------------------------------------------------------------
    def interact(self, mask, idx, scribble=False, download=True):
        """Interact -> propagate -> fuse; returns np.uint8 [t,h,w] (reference inference_core.py:209-259).

        ``download=False`` (not in the reference) skips the device-to-host copy of the masks and returns None: callers
        that evaluate on the device (``processor.masks`` / ``processor.prob``, e.g. eva_vos_amd.eval_driver) save the
        27 MB transfer + host sync per annotation round."""
        idx = int(idx)
        with torch.cuda.device(self.device):
            mask = mask.detach().to(self.device, torch.float32).contiguous()
            if mask.dim() != 4 or mask.shape[1] != 1 or tuple(mask.shape[-2:]) != (self.h, self.w):
                raise RuntimeError(f"mask must be [C,1,{self.h},{self.w}], got {tuple(mask.shape)}")
            cur = self._join_engine_stream(mask)
            t_enq = time.perf_counter()
            try:
                _lib.check(_lib.lib().stcn_interact(self._engine, mask.data_ptr(), int(mask.shape[0]), idx,
                                                    1 if scribble else 0), "stcn_interact")
            finally:
                self._leave_engine_stream(cur)
            self.last_enqueue_s = time.perf_counter() - t_enq      # host time of the enqueue-only C call (no sync inside): what a lane's thread costs the host
            self.interacted.add(idx)
            if not download:
                return None
            lw, uw, lh, uh = self.pad
            out = self.masks[:, 0, lh:self.nh - uh, lw:self.nw - uw]
            self.np_masks = None                                    # the previous result no longer counts against this core's budget (unless the caller kept it)
            block = _pinned_result(out.shape, self._pinned_live) if _PINNED_DOWNLOAD else None
            if block is not None:
                host = block.tensor
                # D2H into PINNED host memory (PyTorch's caching host allocator hands the 27 MB block of a 66-frame 480p clip back and
                # forth): one DMA at PCIe speed instead of a staged copy into pageable memory through blit kernels on the CUs; the
                # array is still a fresh one per call (it owns its pinned block - at most _PINNED_MAX_LIVE of them are alive at a time,
                # see _pinned_result), the host waits on this stream only
                host.copy_(out, non_blocking=True)
                # a BLOCKING event wait: the thread sleeps until the copy has landed (hipEventBlockingSync) instead of spinning a host core in
                # hipStreamSynchronize for the whole propagation - with 8 ranks x 4 lanes on one host the spinning lanes starve the decoders
                if self._dl_event is None:
                    self._dl_event = torch.cuda.Event(blocking=True)
                self._dl_event.record()
                self._dl_event.synchronize()
                self.np_masks = np.asarray(block)
            else:
                if not _PINNED_DOWNLOAD:
                    _PAGEABLE_DOWNLOADS[0] += 1
                self.np_masks = out.cpu().numpy().astype(np.uint8, copy=False)     # D2H sync, as the reference's .cpu(); a fresh array per call
        return self.np_masks

    # The engine enqueues on the HIP stream that was current when the core was constructed.  Normal PyTorch stream
    # semantics are kept for callers that use the core under ANOTHER current stream: the engine stream first waits for the
    # caller's stream (the mask was produced there), and the caller's stream then waits for the engine, so whatever the
    # caller enqueues next on prob / masks is ordered behind the propagation.
    def _join_engine_stream(self, *tensors):
        cur = torch.cuda.current_stream()
        if cur != self._stream:
            self._stream.wait_stream(cur)
            for t in tensors:
                t.record_stream(self._stream)          # the caching allocator must not recycle it under the engine
        return cur

    def _leave_engine_stream(self, cur):
        if cur != self._stream:
            cur.wait_stream(self._stream)

    def reset(self):
        """Forget all interactions / cached features (same clip): equivalent to constructing a new
        InferenceCore on the same images, without re-allocating device memory."""
        with torch.cuda.device(self.device):
            cur = self._join_engine_stream()
            try:
                _lib.check(_lib.lib().stcn_engine_reset(self._engine), "stcn_engine_reset")
            finally:
                self._leave_engine_stream(cur)
        self.interacted = set()
        self.np_masks = np.zeros((self.t, self.h, self.w), dtype=np.uint8)

    def get_image_buffered(self, idx):
        lw, uw, lh, uh = self.pad
        return torch.nn.functional.pad(self._images_unpadded[:, idx], (lw, uw, lh, uh))

    @property
    def images(self):
        lw, uw, lh, uh = self.pad
        return torch.nn.functional.pad(self._images_unpadded, (lw, uw, lh, uh))

    def stats(self) -> dict:
        s = _lib.Stats()
        _lib.check(_lib.lib().stcn_get_stats(self._engine, C.byref(s)))
        return {n: getattr(s, n) for n, _ in _lib.Stats._fields_}

    def engine_options(self) -> dict:
        """The tunables the engine runs with (explicit ``engine_options``, else STCN_* environment at construction, else defaults; clipped)."""
        o = _lib.EngineOpts()
        _lib.check(_lib.lib().stcn_engine_get_opts(self._engine, C.byref(o)))
        return {n: getattr(o, n) for n, _ in _lib.EngineOpts._fields_}

    def set_profiling(self, on: bool) -> None:
        _lib.check(_lib.lib().stcn_engine_set_profiling(self._engine, 1 if on else 0))

    def kernel_profile(self) -> dict:
        """Per-kernel-class device ms / launches / algorithmic FLOP of the last interact()."""
        n = len(_lib.K_CLASSES)
        ms, ln, fl, by, ex = (C.c_float * n)(), (C.c_int32 * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        _lib.check(_lib.lib().stcn_get_kernel_ms(self._engine, ms, ln))
        _lib.check(_lib.lib().stcn_get_kernel_flops(self._engine, fl))
        _lib.check(_lib.lib().stcn_get_kernel_bytes(self._engine, by))
        _lib.check(_lib.lib().stcn_get_kernel_exec_flops(self._engine, ex))
        out = {c: dict(ms=ms[i], launches=ln[i], flops=fl[i], bytes=by[i], exec_flops=ex[i]) for i, c in enumerate(_lib.K_CLASSES)}
        reg = (C.c_double * 6)()
        _lib.check(_lib.lib().stcn_get_conv_regimes(self._engine, reg))
        # conv launches below the machine balance (HBM-bound); a subset of the "conv" totals
        out["conv_hbm_bound"] = dict(ms=reg[2], launches=int(reg[3]), flops=reg[0], bytes=reg[1], exec_flops=reg[0],
                                      wino2_flops=reg[4], wino4_flops=reg[5])
        return out

    def __deepcopy__(self, memo):
        new = object.__new__(InferenceCore)
        for k, v in self.__dict__.items():
            if k in ("_engine", "_finalizer", "prob", "masks", "np_masks", "interacted", "_pinned_live", "_dl_event"):
                continue
            new.__dict__[k] = v                                   # nets, model handle, images: shared (read-only)
        with torch.cuda.device(self.device):
            self._leave_engine_stream(torch.cuda.current_stream())     # the clones below read what the engine stream wrote
            torch.cuda.current_stream().synchronize()
            new.prob, new.masks = self.prob.clone(), self.masks.clone()
            new.np_masks, new.interacted, new._pinned_live, new._dl_event = self.np_masks.copy(), set(self.interacted), [], None
            new._stream = torch.cuda.current_stream()
            h_ = C.c_void_p()
            _lib.check(_lib.lib().stcn_engine_clone(self._engine, new.prob.data_ptr(), new.masks.data_ptr(),
                                                    new._stream.cuda_stream, C.byref(h_)), "stcn_engine_clone")
        new._engine = h_
        new._finalizer = weakref.finalize(new, _lib.lib().stcn_engine_destroy, h_)
        return new
