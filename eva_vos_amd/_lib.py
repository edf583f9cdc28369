"""ctypes binding of libstcn_hip.so (C ABI declared in include/stcn_hip.h).

The HIP library is the product: there is no CPU / PyTorch fallback.  If the shared object is missing
the import fails loudly (build it with ``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C eva_vos_amd/csrc``)."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (first: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64; loading them
#               before libstcn_hip.so makes the process use ONE HIP runtime, whatever the import order)

_HERE = os.path.dirname(os.path.abspath(__file__))
# STCN_LIB: another build of the same library (measurement aid: tools/gpu_ab_trace.sh compares two builds on one GPU box)
LIB_PATH = os.environ.get("STCN_LIB") or os.path.join(_HERE, "libstcn_hip.so")

K_CLASSES = ("conv", "conv_reduce", "memread", "elementwise", "conv_n1", "other", "wino_input", "fusion_conv", "attention")


class WeightDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class EngineOpts(C.Structure):
    """stcn_engine_opts: a negative field = not given (environment variable / default)."""
    _fields_ = [(n, C.c_int32) for n in ("lookahead", "decode_batch", "key_batch", "fuse_side")]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("frames", "key_miss", "value_enc", "fused", "bank_fwd", "bank_bwd")]


# name -> (restype, argtypes); kept in one table so tests can check every declared symbol is exported
_P, _I, _F, _D = C.c_void_p, C.c_int, C.c_float, C.c_double
PROTOTYPES = {
    "stcn_last_error": (C.c_char_p, []),
    "stcn_version": (C.c_char_p, []),
    "stcn_model_create": (_I, [_I, C.POINTER(WeightDesc), _I, C.POINTER(WeightDesc), _I, C.POINTER(_P)]),
    "stcn_model_destroy": (_I, [_P]),
    "stcn_engine_create": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, C.POINTER(_P)]),
    "stcn_engine_create_ex": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, C.POINTER(EngineOpts), C.POINTER(_P)]),
    "stcn_engine_get_opts": (_I, [_P, C.POINTER(EngineOpts)]),
    "stcn_engine_destroy": (_I, [_P]),
    "stcn_engine_reset": (_I, [_P]),
    "stcn_engine_clone": (_I, [_P, _P, _P, _P, C.POINTER(_P)]),
    "stcn_interact": (_I, [_P, _P, _I, _I, _I]),
    "stcn_get_stats": (_I, [_P, C.POINTER(Stats)]),
    "stcn_get_flops": (_I, [_P, C.POINTER(_D)]),
    "stcn_last_conv_path": (C.c_char_p, []),
    "stcn_test_conv_trace": (_I, [_I]),
    "stcn_test_conv_trace_get": (C.c_char_p, []),
    "stcn_test_conv": (_I, [_P, _P, _P, _P, _P, _P] + [_I] * 11),
    "stcn_test_encode_key": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "stcn_test_encode_value": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "stcn_test_memory_read": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "stcn_bench_memory_read": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, C.POINTER(_F), C.POINTER(C.c_int32)]),
    "stcn_memread_plan": (_I, [_I, _I, C.POINTER(C.c_int32)]),
    "stcn_test_fail_at": (_I, [_I]),
    "stcn_test_side_delay_us": (_I, [_I]),
    "stcn_test_decode": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "stcn_test_attention": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "stcn_test_fusion": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _P]),
    "stcn_metrics_jf_counts": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "stcn_metrics_j_counts": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "stcn_metrics_round": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _D, _P, _P, _P, _P, _P]),
    "stcn_bench_conv": (_I, [_P] + [_I] * 11 + [C.POINTER(_F), C.POINTER(_D)]),
    "stcn_bench_mfma_rate": (_I, [_P, _I, C.POINTER(_F), C.POINTER(_F)]),
    "stcn_pool_release": (_I, []),
    "stcn_engine_set_profiling": (_I, [_P, _I]),
    "stcn_get_kernel_ms": (_I, [_P, C.POINTER(_F), C.POINTER(C.c_int32)]),
    "stcn_get_kernel_flops": (_I, [_P, C.POINTER(_D)]),
    "stcn_get_kernel_bytes": (_I, [_P, C.POINTER(_D)]),
    "stcn_get_kernel_exec_flops": (_I, [_P, C.POINTER(_D)]),
    "stcn_get_conv_regimes": (_I, [_P, C.POINTER(_D)]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP engine has not been built (run __graft_entry__.build() "
                "or `make -C eva_vos_amd/csrc`).  There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = res, args
    return _lib


def check(rc: int, what: str = "stcn") -> None:
    if rc != 0:
        msg = lib().stcn_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def src_hash() -> str:
    """The source hash the LOADED library was built from (csrc/Makefile: sha256 of the sorted *.hip / *.cpp / *.h of csrc + the C-ABI header +
    the extra flags, 12 hex digits) - captures under profiles/ are stamped with it, bench.py compares it with theirs."""
    import re
    m = re.search(r"src ([0-9a-f]{12})", lib().stcn_version().decode())
    return m.group(1) if m else "unknown"


def tree_hash(extra: str = "") -> str:
    """The same hash computed from the sources in the tree (what a fresh `make` would stamp): differs from src_hash() when the .so is stale."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(os.path.basename(f) for pat in ("*.hip", "*.cpp", "*.h") for f in glob.glob(os.path.join(csrc, pat)))
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(_HERE, "..", "include", "stcn_hip.h"), "rb").read())
    h.update((extra + "\n").encode())
    return h.hexdigest()[:12]
