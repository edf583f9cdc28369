"""GPU box: what does constructing / destroying an InferenceCore cost the drivers (one per sample: generate_fq_dataset.py:63-70)?  Phases of
__init__ timed separately (weight-snapshot lookup, torch allocations, the C call stcn_engine_create_ex), engine pool warm.
python tools/engine_create_cost.py [--frames 40] [--lookahead 0|2]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva_vos_amd import _lib, inference_core as IC, synth  # noqa: E402
from eva_vos_amd.params import FusionNet, PropagationNetwork  # noqa: E402

T = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 40
torch.set_grad_enabled(False)
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop))
fuse.load_state_dict(synth.recipe_state_dict(fuse))
img = synth.synthetic_clip(T, 480, 854).cuda()
acc = {}


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return wrapper


IC._model_for = timed("model_for (fingerprint + snapshot lookup)", IC._model_for)
lib = _lib.lib()
orig_create, orig_destroy = lib.stcn_engine_create_ex, lib.stcn_engine_destroy
for la in (0, 2):
    with torch.cuda.stream(torch.cuda.Stream()):
        for _ in range(2):                                            # warm the pool with this configuration
            e = IC.InferenceCore(prop, fuse, img, 1, engine_options={"lookahead": la})
            del e
        acc.clear()
        n, t_create, t_c_call, t_destroy = 8, 0.0, 0.0, 0.0
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e = IC.InferenceCore(prop, fuse, img, 1, engine_options={"lookahead": la})
            t1 = time.perf_counter()
            del e
            t2 = time.perf_counter()
            t_create += t1 - t0
            t_destroy += t2 - t1
        print(f"lookahead {la}, T={T}: InferenceCore() {1e3 * t_create / n:.2f} ms, del {1e3 * t_destroy / n:.2f} ms per engine (pool warm); "
              + "; ".join(f"{k} {1e3 * v / n:.2f} ms" for k, v in acc.items()), flush=True)
        # the C call alone
        import ctypes as C
        t_c = 0.0
        for _ in range(n):
            prob = torch.empty((2, T, 1, 480, 864), device="cuda")
            masks = torch.empty((T, 1, 480, 864), dtype=torch.uint8, device="cuda")
            h = C.c_void_p()
            opts = _lib.EngineOpts(la, -1, -1, -1)
            model = IC._model_for(prop, fuse, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(orig_create(model.handle, T, 480, 854, 1, 5, torch.cuda.current_stream().cuda_stream, img.data_ptr(), prob.data_ptr(), masks.data_ptr(),
                                   C.byref(opts), C.byref(h)))
            t_c += time.perf_counter() - t0
            orig_destroy(h)
        print(f"   stcn_engine_create_ex alone: {1e3 * t_c / n:.2f} ms", flush=True)
