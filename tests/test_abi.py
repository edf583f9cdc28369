"""CPU: the C-ABI library loads and exports every symbol include/stcn_hip.h declares; the parameter
containers reproduce the reference state_dict layout; host-side argument checks."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from eva_vos_amd import _lib, synth
from eva_vos_amd.params import FusionNet, PropagationNetwork


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "stcn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(stcn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libstcn_hip.so lacks {n}"
    assert set(names) == set(_lib.PROTOTYPES), "ctypes table and header disagree"
    assert b"gfx950" in lib.stcn_version()


def test_null_arguments_are_rejected_without_touching_the_gpu():
    lib = _lib.lib()
    assert lib.stcn_interact(None, None, 1, 0, 0) == -1
    assert b"null" in lib.stcn_last_error()
    assert lib.stcn_model_destroy(None) == 0 and lib.stcn_engine_destroy(None) == 0


def test_state_dict_layout():
    p, f = PropagationNetwork(), FusionNet()
    sd = p.state_dict()
    assert len(sd) == 405 and len(f.state_dict()) == 12
    assert sum(v.numel() for v in sd.values() if v.is_floating_point()) == 54469252
    assert tuple(sd["key_encoder.conv1.weight"].shape) == (64, 3, 7, 7) and "key_encoder.conv1.bias" not in sd
    assert tuple(sd["value_encoder.conv1.weight"].shape) == (64, 5, 7, 7) and "value_encoder.conv1.bias" in sd
    assert tuple(sd["key_proj.key_proj.weight"].shape) == (64, 1024, 3, 3)
    assert tuple(sd["key_comp.weight"].shape) == (512, 1024, 3, 3)
    assert tuple(sd["decoder.pred.weight"].shape) == (1, 256, 3, 3)
    assert tuple(sd["value_encoder.fuser.attention.ChannelGate.mlp.1.weight"].shape) == (32, 512)
    assert tuple(sd["value_encoder.fuser.attention.SpatialGate.spatial.conv.weight"].shape) == (1, 2, 7, 7)
    assert "key_encoder.res2.0.downsample.0.weight" in sd and "key_encoder.layer3.5.bn3.running_var" in sd
    assert tuple(f.state_dict()["conv1.0.weight"].shape) == (32, 9, 3, 3)


def test_recipe_is_deterministic_and_loads_strictly():
    p = PropagationNetwork()
    a, b = synth.recipe_state_dict(p, 0), synth.recipe_state_dict(p, 0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    p.load_state_dict(a, strict=True)
    c = synth.recipe_state_dict(p, 1)
    assert not torch.equal(a["key_comp.weight"], c["key_comp.weight"])
    # fingerprint: guards the goldens against silent recipe drift
    assert abs(float(a["decoder.pred.weight"].double().sum())) < 1e-4
    assert np.isclose(float(a["key_comp.weight"].double().abs().sum()), 55455.23, rtol=1e-5)


def test_containers_have_no_forward():
    with pytest.raises(RuntimeError):
        PropagationNetwork()(torch.zeros(1))


def test_engine_needs_gpu_and_never_falls_back(nets):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from eva_vos_amd.inference_core import InferenceCore
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        InferenceCore(nets[0], nets[1], torch.zeros(1, 2, 3, 64, 64), 1)


def test_fastdiv_magic_numbers_are_exact():
    """csrc/kernels.h replaces run-time integer divisions by q = umulhi(x, magic) >> shift with magic = floor(2^(31+s)/d) + 1,
    s = ceil(log2 d) (addresses depend on it).  The same formula restated here must equal x // d for every divisor that can
    occur (tile counts, OH*OW, OW up to a few million) on boundary and random numerators below 2^31."""
    rng = np.random.default_rng(0)

    def make(d):
        s = 0
        while (1 << s) < d:
            s += 1
        return ((1 << (31 + s)) // d + 1) & 0xFFFFFFFF, s - 1

    divisors = list(range(2, 3000)) + [2 ** k for k in range(1, 24)] + [2 ** k + 1 for k in range(1, 24)] + \
        [25920, 103680, 1620, 6480, 32448, 129600, 518400, 2073600] + rng.integers(2, 1 << 23, 500).tolist()
    for d in divisors:
        magic, shift = make(d)
        assert magic < (1 << 32)
        xs = np.concatenate([np.array([0, 1, d - 1, d, d + 1, 2 * d - 1, (1 << 31) - 1, (1 << 31) - d, ((1 << 31) // d) * d - 1]),
                             rng.integers(0, 1 << 31, 64)]).astype(np.uint64)
        xs = xs[xs < (1 << 31)]
        q = ((xs * np.uint64(magic)) >> np.uint64(32)) >> np.uint64(shift)
        assert np.array_equal(q, xs // np.uint64(d)), d
