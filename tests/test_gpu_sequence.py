"""GPU: the drop-in InferenceCore (HIP engine behind the C ABI) on whole interact() sequences: against
goldens captured from the reference, against the oracle at a ragged size, plus API behaviour."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import clip_bound, frame_bound, frame_miss, iou, load_golden
from eva_vos_amd import synth
from oracle import stcn_oracle as O
from test_oracle_golden import check_sequence_against_golden, run_sequence, tie_summary, weights_of

pytestmark = pytest.mark.gpu


def make_core(nets):
    from mivos.inference_core import InferenceCore
    return lambda img, k, mf: InferenceCore(nets[0], nets[1], img, k, mem_freq=mf)


CLEAN_FP32 = 2e-4     # max |prob| difference between two fp32 implementations while every query's top-50 set is clear-cut


def clean_frame_check(core_prob, orc, rounds_done, tag):
    """HIP vs oracle, fp32 against fp32: on the frames propagated before the first near-tie of the run (tie_summary) the
    probabilities agree EVERYWHERE to CLEAN_FP32 - a max, not a quantile."""
    info = tie_summary(orc).get(rounds_done, dict(clean=set(), near=0, queries=0))      # a 1-frame clip has no memory read at all
    clean = sorted(info["clean"])
    d = (core_prob - orc.prob).abs()
    worst = float(d[:, clean].max()) if clean else 0.0
    print(f"HIP vs oracle {tag} r{rounds_done}: {len(clean)} clean frames, max |dprob| on them {worst:.1e}; near-tie queries "
          f"{info['near']} / {info['queries']}; all frames max {float(d.max()):.1e}")
    assert worst < CLEAN_FP32, (tag, rounds_done, worst)
    return len(clean)


def masks_close(a, b, k, tag, yard=None, px_floor=2):
    """HIP masks a against oracle masks b ([T,H,W] uint8): per object the clip IoU and the worst frame.  `yard` = a selfnoise
    row of the nearest reference fixture (volume envelope, ..., per-frame envelope) or None (k = 1: the north_star 1e-3).
    px_floor: pixels of a frame that may differ whatever the object's size (2; soak runs: more once the oracle has met a true near-tie)."""
    for o in range(1, k + 1):
        vol = 1 - iou(a == o, b == o)
        miss, fr = frame_miss(a == o, b == o)
        px = ((a[fr] == o) | (b[fr] == o)).sum() if fr >= 0 else 1
        vb = clip_bound(yard[0]) if yard is not None else 1e-3
        fb = max(frame_bound(yard[4] if yard is not None else 0.0, px), px_floor / max(float(px), 1.0))
        print(f"HIP vs oracle {tag} object {o}: clip 1-IoU {vol:.2e} (bound {vb:.1e}, {vol / vb:.2f} of it), worst frame {fr}: {miss:.2e} (bound {fb:.1e}, {miss / fb:.2f} of it)")
        assert vol <= vb and miss <= fb, (tag, o, vol, vb, fr, miss, fb)


def small_multi_object_yardstick():
    """Reference-vs-reference envelope for k > 1 on small frames: the worst row of the seqC (k = 3, 128x160) and seqD (k = 5,
    120x170) fixtures - the aggregation is ill-conditioned where objects saturate, whatever the frame content."""
    n = load_golden("selfnoise")
    return np.maximum(n["seqC"].max(0), n["seqD"].max(0))


def nets_of(g, tag, nets, weights):
    w = weights_of(g, tag, weights)
    if w is weights:
        return nets, w
    from eva_vos_amd.params import FusionNet, PropagationNetwork
    p, f = PropagationNetwork(), FusionNet()
    p.load_state_dict(w[0], strict=True)
    f.load_state_dict(w[1], strict=True)
    return (p.eval(), f.eval()), w


@pytest.mark.parametrize("tag", ["seqA", "seqA1", "seqB", "seqC", "seqD", "seqE", "seq480", "seq480L", "seq480k5", "seq480k3", "seq480P", "seq640k3", "seqK10", "seqK16"])
def test_sequences_match_reference_goldens(tag, nets, weights):
    """seqA1 = the seqA script under weight recipe seed 1, seq480k5 = BASELINE config 3's shape (480x854, 5 objects, every
    frame enters the bank), seq480k3 = three objects at 480p with a second, FUSED interaction (both under the multi-object recipe,
    all pixels), seq480P = a PORTRAIT clip with an odd long side (853x480 -> pad (0,0,5,6), 54 x 30 keys: what scripts/resize.py makes of a
    portrait MOSE video), seq640k3 = 4:3 (480x640), three objects, seqE = a round annotated with an EMPTY mask (the object has left the frame:
    the reference's loops annotate such a frame with its all-zero ground truth) - all held to the SAME statements and tolerances as the seed-0 /
    single-object fixtures.  seqK10 / seqK16 = 10 and 16 objects (beyond the 8 of rounds 1-5), a propagation and a fused second interaction."""
    g = load_golden(tag)
    nets, weights = nets_of(g, tag, nets, weights)
    outs = run_sequence(make_core(nets), tag, g)
    orcs = []

    def oracle(img, k, mf):
        orcs.append(O.OracleCore(weights[0], weights[1], img, k, mem_freq=mf))
        return orcs[0]

    oouts = run_sequence(oracle, tag, g)                       # same inputs on the CPU oracle: where are the near-ties?
    check_sequence_against_golden(outs, tag, g, prob_atol=3e-3, ties=tie_summary(orcs[0]), who="HIP")
    if int(g[f"{tag}.shape"][3]) == 1:                         # fp32 vs fp32 on the clean frames of round 1
        info = tie_summary(orcs[0])[0]
        clean = sorted(info["clean"])
        d = (outs[0][1] - oouts[0][1]).abs()
        worst = float(d[:, clean].max()) if clean else 0.0
        print(f"HIP vs oracle {tag} r0: {len(clean)} clean frames, max |dprob| {worst:.1e} (all frames {float(d.max()):.1e})")
        assert worst < CLEAN_FP32, (tag, worst)


def test_bank_sizes_and_counters(nets):
    g = load_golden("seqA")
    T, H, W, k, mf = [int(v) for v in g["seqA.shape"]]
    core = make_core(nets)(synth.synthetic_clip(T, H, W), k, mf)
    msk = synth.synthetic_mask(T, H, W, k)
    core.interact(msk[:, 0], 0)
    s = core.stats()
    assert (s["frames"], s["key_miss"], s["fused"], s["bank_fwd"], s["bank_bwd"]) == (11, 12, 0, 3, 1)
    assert s["value_enc"] == 3            # interaction + frames 5, 10
    core.interact(msk[:, 8], 8)
    s = core.stats()
    assert (s["frames"], s["key_miss"], s["fused"]) == (10, 0, 7)
    assert s["bank_fwd"] == int(g["seqA.trace"][2][3]) and s["bank_bwd"] == int(g["seqA.trace"][3][3])


def test_oracle_parity_at_ragged_size_and_deepcopy(nets, weights):
    """112x176 input padded (0,0) x ... not a golden size: HIP engine vs oracle on the same seeded inputs;
    deepcopy then diverging interactions must not disturb the original."""
    T, H, W = 7, 104, 170
    img = synth.synthetic_clip(T, H, W, seed=5)
    msk = synth.synthetic_mask(T, H, W, 1, seed=6)
    core = make_core(nets)(img, 1, 2)
    orc = O.OracleCore(weights[0], weights[1], img, 1, mem_freq=2)
    a, b = core.interact(msk[:, 2], 2), orc.interact(msk[:, 2], 2)
    assert a.shape == (T, H, W) and a.dtype == np.uint8
    masks_close(a, b, 1, "104x170 r0")
    assert tuple(core.pad) == tuple(orc.pad) and core.prob.shape == orc.prob.shape
    clean_frame_check(core.prob.cpu(), orc, 0, "104x170")         # max-norm on the frames before the first near-tie
    twin = copy.deepcopy(core)
    a2 = twin.interact(msk[:, 5], 5)
    b2 = orc.interact(msk[:, 5], 5)
    masks_close(a2, b2, 1, "104x170 r1")
    clean_frame_check(twin.prob.cpu(), orc, 1, "104x170")
    assert np.array_equal(core.np_masks, a), "deepcopy must not alias the original's results"
    assert (core.prob.cpu() - twin.prob.cpu()).abs().max() > 1e-3
    a3 = core.interact(msk[:, 5], 5)          # the original, same second interaction -> same answer
    assert np.array_equal(a3, a2)


def test_interacted_frame_is_all_background_for_k1(nets):
    """prob[:, idx] = mask broadcasts into bg AND fg rows (inference_core.py:226): argmax ties -> 0."""
    T, H, W = 4, 112, 128
    core = make_core(nets)(synth.synthetic_clip(T, H, W), 1, 5)
    out = core.interact(synth.synthetic_mask(T, H, W, 1)[:, 1], 1)
    assert out[1].max() == 0 and out[0].max() == 1


def test_reference_argument_errors(nets):
    T, H, W = 3, 112, 128
    msk3 = synth.synthetic_mask(T, H, W, 3)
    core = make_core(nets)(synth.synthetic_clip(T, H, W), 3, 5)
    with pytest.raises(RuntimeError):          # k-channel mask without scribble raises in the reference too
        core.interact(msk3[:, 0], 0)
    with pytest.raises(RuntimeError):
        core.interact(msk3[:, 0, :, :50], 0)   # wrong spatial size
    with pytest.raises(RuntimeError):
        make_core(nets)(synth.synthetic_clip(2, 96, 128), 1, 5)   # HW/256 < 50 rows: top-50 impossible


def test_full_resolution_round_trip_properties(nets):
    """BASELINE size (480x854, padded 864): size-independent properties instead of a CPU comparison:
    probabilities are a distribution over k+1 rows, masks are the argmax, a repeated run is bit-identical,
    key-cache hits (round 2) do not change unfused frames' inputs."""
    T, H, W = 6, 480, 854
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, 1)
    outs = []
    for _ in range(2):
        core = make_core(nets)(img, 1, 2)
        m = core.interact(msk[:, 0], 0)
        outs.append((m.copy(), core.prob.clone()))
    assert np.array_equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    prob = outs[0][1]
    assert prob.shape == (2, T, 1, 480, 864)
    assert (prob[:, 1:].sum(0) - 1).abs().max() < 1e-5
    lw, uw, lh, uh = core.pad
    am = prob.argmax(0)[:, 0, lh:480 - uh, lw:864 - uw].cpu().numpy()
    assert np.array_equal(am.astype(np.uint8), outs[0][0])
    frac = outs[0][0][1:].mean()
    assert 0.05 < frac < 0.95


def test_full_resolution_two_rounds_match_the_oracle(nets, weights):
    """BASELINE size end to end against the CPU oracle (about 20 s of host time): interact(0) then interact(5) with
    fusion on 480x854 - the north_star bars: masks within 1e-3 IoU, J&F within 0.1 (here: of each other, both
    measured against the synthetic ground truth with the GPU J/F kernel and the CPU metrics)."""
    from eva_vos_amd import metrics
    T, H, W = 7, 480, 854
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, 1)
    core = make_core(nets)(img, 1, 3)
    orc = O.OracleCore(weights[0], weights[1], img, 1, mem_freq=3)
    n_clean = 0
    for r, idx in enumerate((0, 5)):
        a, b = core.interact(msk[:, idx], idx), orc.interact(msk[:, idx], idx)
        masks_close(a, b, 1, f"480x854 r{r}", yard=load_golden("selfnoise")["seq480"][r])
        n_clean += clean_frame_check(core.prob.cpu(), orc, r, "480x854")
        d = (core.prob.cpu() - orc.prob).abs().numpy()
        q = [float(np.quantile(d, v)) for v in (0.5, 0.99, 0.999)]
        print(f"full-res round idx={idx}: |prob diff| median {q[0]:.1e} q99 {q[1]:.1e} q999 {q[2]:.1e} max {d.max():.1e}")
        # 4 % of the 1620 queries per frame are near-ties at this resolution (tie_summary above): a few memberships differ
        # between two fp32 implementations.  The yardstick is the REFERENCE against itself at this resolution (seq480 in
        # tests/golden/selfnoise.npz: p99.9 of |prob| between its 1/2/4/8-thread runs = 6.7e-3 / 7.2e-3): 3 x that for the
        # tail, the bulk (p99) within 2e-3
        floor = float(load_golden("selfnoise")["seq480"][r][2])
        assert q[1] < 2e-3 and q[2] < 3 * floor, (idx, q, floor)
    assert core.stats()["fused"] == 4
    gt = msk[0, :, 0].numpy() > 0.5
    jf_gpu = metrics.sequence_scores_gpu(torch.from_numpy(gt).cuda(), torch.from_numpy(a > 0).cuda())
    jf_cpu = metrics.sequence_scores(gt, b > 0)
    assert np.abs(jf_gpu[:, 2] - jf_cpu[:, 3]).max() < 0.1              # CPU rows are (frame, J, F, J&F)
    assert abs(jf_gpu[:, 2].mean() - jf_cpu[:, 3].mean()) < 2e-3


def _conv_trace(fn):
    """Runs fn() with the calling thread's conv trace on; returns (result, {layer: [paths]})."""
    from eva_vos_amd import _lib
    L = _lib.lib()
    _lib.check(L.stcn_test_conv_trace(1))
    try:
        out = fn()
        log = L.stcn_test_conv_trace_get().decode()
    finally:
        _lib.check(L.stcn_test_conv_trace(0))
    paths = {}
    for line in log.splitlines():
        name, path = line.split("=", 1)
        paths.setdefault(name, []).append(path)
    return out, paths


# (H, W) as the clips are STORED: scripts/resize.py:9-24 resizes every MOSE / DAVIS video to min(w, h) = 480 and
# datasets/annotation_dataset.py:95-106 feeds that size: portrait 854x480 / 853x480 (odd: pad (0,0,5,6)), 4:3, 3:2, and a wide 480x910
REAL_SHAPES = [(854, 480), (853, 480), (480, 640), (480, 720), (480, 910)]


@pytest.mark.parametrize("H,W", REAL_SHAPES)
def test_480p_class_shapes_match_the_oracle(H, W, nets, weights, nets_multi, weights_multi):
    """Every engine-level test up to round 4 ran landscape 480x854 (30 x 54 keys) or small frames.  The row-strip kernels, the F(4x4) tile
    maps / tail splits / chunk thresholds and the workspace sizes all depend on the shape: here the shapes the datasets really contain, HIP
    engine against the CPU oracle - k = 1 two rounds (the second one fused) and k = 3 through the scribble path (multi-object recipe), at
    the bounds of the 480x854 tests (reference self-noise of the nearest reference fixture) - and the conv trace must show every
    decoder-side 3x3 layer on the F(4x4) kernel (a shape that silently fell back to another family would still pass numerically)."""
    noise = load_golden("selfnoise")
    # ---- k = 1, mem_freq = 2: interact(0), interact(4) with fusion on frames 1..3
    T = 6
    img, msk = synth.synthetic_clip(T, H, W, seed=61), synth.synthetic_mask(T, H, W, 1, seed=62)
    core = make_core(nets)(img, 1, 2)
    orc = O.OracleCore(weights[0], weights[1], img, 1, mem_freq=2)
    assert tuple(core.pad) == tuple(orc.pad) and core.prob.shape == orc.prob.shape
    yard = noise["seq480P" if H > W else "seq480"]
    n_clean = 0
    for r, idx in enumerate((0, 4)):
        a, paths = _conv_trace(lambda: core.interact(msk[:, idx], idx))
        b = orc.interact(msk[:, idx], idx)
        assert a.shape == (T, H, W)
        masks_close(a, b, 1, f"{H}x{W} k=1 r{r}", yard=yard[min(r, len(yard) - 1)])
        n_clean += clean_frame_check(core.prob.cpu(), orc, r, f"{H}x{W}")
        # ~4 % of the queries of a 480p frame are near-ties from the first read on (no "clean" frame to take a max-norm on): the bulk of
        # the probabilities within 3e-3, the tail against 3 x the reference's own tail on the nearest fixture (as the 480x854 test).
        # q99 sits INSIDE the population of pixels next to a flipped near-tie query, i.e. it moves with which of the ~400 near-tie
        # queries flip: 0.6-1.9e-3 over the five shapes with the key trunk on F(2x2) / direct kernels, 0.4-2.0e-3 with it on F(4x4)
        # (lower on 5 of the 9 rounds, 2.004e-3 on 480x910 round 0: profiles/r05_key_trunk_f4.txt) - the first bound of 2e-3 had 5 % of
        # margin on that shape; 3e-3 = 1.5 x the largest value either arithmetic produces
        d = (core.prob.cpu() - orc.prob).abs().numpy().reshape(-1)[::3]
        q99, q999 = float(np.quantile(d, 0.99)), float(np.quantile(d, 0.999))
        floor = float(yard[min(r, len(yard) - 1)][2])
        print(f"{H}x{W} k=1 r{r}: |dprob| q99 {q99:.1e} q99.9 {q999:.1e} (3 x reference self-noise {3 * floor:.1e})")
        assert q99 < 3e-3 and q999 < 3 * floor, (H, W, r, q99, q999, floor)
        dec = {n: p for n, p in paths.items() if n.startswith("decoder.") and not n.endswith("pred")}
        assert dec and all(q.startswith("wino4") for p in dec.values() for q in p), dec
        if r == 0:
            assert any(n.startswith("key_encoder.") for n in paths) and all(q.startswith("wino4") for q in paths["key_comp"]), paths.get("key_comp")
    assert core.stats()["fused"] == 3
    # ---- k = 3 through the scribble / (k+1)-channel path, multi-object recipe, one round, decode groups of 2 frames x 3 objects
    T, k = 5, 3
    img, msk = synth.synthetic_clip(T, H, W, seed=63), synth.synthetic_mask(T, H, W, k, seed=64)
    m0 = torch.cat([1 - msk[:, 0].sum(0, keepdim=True).clamp(0, 1), msk[:, 0]], 0)
    core = make_core(nets_multi)(img, k, 2)
    orc = O.OracleCore(weights_multi[0], weights_multi[1], img, k, mem_freq=2)
    a, paths = _conv_trace(lambda: core.interact(m0, 0, scribble=True))
    b = orc.interact(m0.clone(), 0, scribble=True)
    dec = {n: p for n, p in paths.items() if n.startswith("decoder.") or n.startswith("value_encoder.fuser.")}
    assert dec and all(q.startswith("wino4") for p in dec.values() for q in p), dec
    lw, uw, lh, uh = orc.pad
    po = orc.prob[:, :, 0, lh:orc.prob.shape[3] - uh if uh else None, lw:orc.prob.shape[4] - uw if uw else None]
    top = torch.topk(po, 2, dim=0).values
    decisive = float(((top[0] - top[1]) >= 1e-2).float().mean())
    print(f"{H}x{W} k=3: {100 * decisive:.1f} % decisive pixels, {int((a != b).sum())} of {a.size} mask pixels differ")
    assert decisive > 0.8
    # yardstick: the reference against itself on the multi-object 480p fixtures (first rounds of seq480k5 / seq480k3 / seq640k3)
    yard3 = np.maximum(np.maximum(noise["seq480k5"][0], noise["seq480k3"][0]), noise["seq640k3"][0])
    masks_close(a, b, k, f"{H}x{W} k=3", yard=yard3)
    d = (core.prob.cpu() - orc.prob).abs().numpy()
    q999 = float(np.quantile(d.reshape(-1)[::5], 0.999))
    print(f"{H}x{W} k=3: |dprob| q99.9 {q999:.1e} (bound {3 * float(yard3[2]) + 5e-4:.1e})")
    assert q999 <= 3 * float(yard3[2]) + 5e-4


@pytest.mark.parametrize("T,k", [(9, 3), (26, 5)])
def test_config3_multi_object_full_bank_properties(nets, T, k):
    """BASELINE config 3 shape (480p, k = 3 and the stated maximum k = 5 through the scribble/(k+1)-channel path,
    mem_freq=1: every frame enters the bank, 25 x 1620 rows at T=26): bank growth, probability simplex, determinism,
    object exclusivity of the masks."""
    H, W = 480, 854
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, k)
    m0 = torch.cat([1 - msk[:, 0].sum(0, keepdim=True).clamp(0, 1), msk[:, 0]], 0)
    outs = []
    for _ in range(2):
        core = make_core(nets)(img, k, 1)
        out = core.interact(m0, 0, scribble=True)
        outs.append((out.copy(), core.prob.clone(), core.stats()))
    assert np.array_equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    s = outs[0][2]
    assert s["frames"] == T - 1 and s["bank_fwd"] == T - 1 and s["value_enc"] == T - 1   # mem_freq=1: all but the last frame
    prob = outs[0][1]
    assert prob.shape == (k + 1, T, 1, 480, 864)
    assert (prob[:, 1:].sum(0) - 1).abs().max() < 1e-5
    assert set(np.unique(outs[0][0])) <= set(range(k + 1))
    assert all((outs[0][0][1:] == o).mean() > 0.01 for o in range(1, k + 1))


def test_long_clip_exercises_cache_flush_policy(nets, weights):
    """T = 112 > 106 cache slots: the reference flushes its key cache when it holds more than 105 frames
    (inference_core.py:118-119); the engine mirrors that (and disables key look-ahead).  Round 2 then has
    key-encoder misses again.  Masks must still match the oracle, which implements the same policy."""
    T, H, W = 112, 112, 128
    img = synth.synthetic_clip(T, H, W, seed=9)
    msk = synth.synthetic_mask(T, H, W, 1, seed=10)
    core = make_core(nets)(img, 1, 5)
    orc = O.OracleCore(weights[0], weights[1], img, 1, mem_freq=5)
    a, b = core.interact(msk[:, 0], 0), orc.interact(msk[:, 0], 0)
    s1 = core.stats()
    assert s1["key_miss"] == T and s1["frames"] == T - 1
    masks_close(a, b, 1, "T=112 r0")
    n_clean = clean_frame_check(core.prob.cpu(), orc, 0, "T=112")
    a2, b2 = core.interact(msk[:, 60], 60), orc.interact(msk[:, 60], 60)
    s2 = core.stats()
    assert s2["key_miss"] > 0, "after a flush some frames must be re-encoded"
    assert s2["fused"] == 59 and s2["frames"] == T - 2
    masks_close(a2, b2, 1, "T=112 r1")
    n_clean += clean_frame_check(core.prob.cpu(), orc, 1, "T=112")
    assert n_clean > 0, "no frame of either round precedes the first near-tie: the max-norm statement is vacuous"


@pytest.mark.parametrize("lookahead", ["0", "2"])
def test_key_batching_does_not_change_results(nets, monkeypatch, lookahead):
    """The key encoder runs over up to STCN_KEY_BATCH frames per pass and the memory read + decoder over the frames up to
    the next bank insertion (STCN_DECODE_BATCH), forward and backward sweeps, stopping at interacted frames; per-frame
    results only see a different M of the same GEMMs."""
    T, H, W = 15, 112, 144
    img = synth.synthetic_clip(T, H, W, seed=4)
    msk = synth.synthetic_mask(T, H, W, 1, seed=5)
    monkeypatch.setenv("STCN_LOOKAHEAD", lookahead)
    res = {}
    for kb in ("1", "3", "4", "8", "g1", "g2"):
        # kb: frames per key-encoder pass; g<n>: frames per memory-read + decoder pass (default min(mem_freq, 8) = 3 here)
        monkeypatch.setenv("STCN_KEY_BATCH", "4" if kb[0] == "g" else kb)
        monkeypatch.setenv("STCN_DECODE_BATCH", kb[1:] if kb[0] == "g" else "8")
        core = make_core(nets)(img, 1, 3)
        m1 = core.interact(msk[:, 9], 9).copy()            # backward sweep 8..0 and forward sweep 10..14
        s = core.stats()
        assert s["key_miss"] == T and s["frames"] == T - 1
        m2 = core.interact(msk[:, 4], 4).copy()            # all keys cached, fusion between 4 and 9
        assert core.stats()["key_miss"] == 0
        res[kb] = (m1, m2, core.prob.clone())
    res["1"] = res["g1"]                                   # reference point: no batching of the decode at all
    for kb in ("3", "4", "8", "g2"):
        assert iou(res[kb][0], res["1"][0]) >= 1 - 1e-3 and iou(res[kb][1], res["1"][1]) >= 1 - 1e-3
        assert (res[kb][2] - res["1"][2]).abs().max().item() < 2e-3


@pytest.mark.parametrize("k", [1, 2])
def test_the_backward_sweep_beside_the_forward_one_changes_nothing_but_the_timing(nets, nets_multi, monkeypatch, k):
    """Round 6: with one video in flight the backward sweep of an interaction runs on a second stream + workspace BESIDE the forward sweep
    (the sweeps share only the certain memory, read-only: inference_core.py:250-253 runs them one after the other); its temporary bank slots
    sit in front of the certain slots.  (i) Same arithmetic whatever the timing: a late second stream (stcn_test_side_delay_us also delays
    the backward sweep and the key-encoder streams) must reproduce the undelayed result BIT for bit - first interaction in the middle of
    the clip (both sweeps encode keys), a fused round between two interactions, a round whose backward sweep is empty, a re-annotation.
    (ii) Against the serial engine (STCN_DUAL_SWEEP=0) the masks agree to 1e-3 IoU and the probabilities to 2e-3: only the ORDER of the
    bank rows a backward sweep reads differs (temporaries in front), i.e. the order of the 50-term read-out sums."""
    from eva_vos_amd import _lib
    nets_ = nets if k == 1 else nets_multi
    T, H, W = 21, 112, 144
    img, msk = synth.synthetic_clip(T, H, W, seed=24), synth.synthetic_mask(T, H, W, k, seed=25)
    monkeypatch.setenv("STCN_LOOKAHEAD", "2")
    script = (10, 16, 0, 13, 10)

    def run(dual, delay):
        monkeypatch.setenv("STCN_DUAL_SWEEP", dual)
        _lib.check(_lib.lib().stcn_test_side_delay_us(delay))
        try:
            core = make_core(nets_)(img, k, 3)
            outs = []
            for idx in script:
                m = msk[:, idx] if k == 1 else torch.cat([1 - msk[:, idx].sum(0, keepdim=True).clamp(0, 1), msk[:, idx]], 0)
                outs.append((core.interact(m, idx, scribble=k > 1).copy(), core.prob.clone(), core.stats()))
            return outs
        finally:
            _lib.check(_lib.lib().stcn_test_side_delay_us(0))

    base, late, serial = run("1", 0), run("1", 400), run("0", 0)
    for r, idx in enumerate(script):
        assert np.array_equal(base[r][0], late[r][0]) and torch.equal(base[r][1], late[r][1]), f"round {r} (frame {idx}): the result depends on the timing of the second stream"
        assert base[r][2] == serial[r][2], (base[r][2], serial[r][2])          # same frames, fusions, bank sizes
        for o in range(1, k + 1):
            assert iou(base[r][0] == o, serial[r][0] == o) >= 1 - 1e-3
        assert (base[r][1] - serial[r][1]).abs().max().item() < 2e-3
    assert base[1][2]["fused"] > 0 and base[0][2]["bank_bwd"] > 1 and base[0][2]["bank_fwd"] > 1


@pytest.mark.parametrize("T,mem_freq,idx", [(30, 12, 17), (9, 50, 0), (3, 5, 1), (2, 1, 0)])
def test_decode_groups_with_large_mem_freq_and_tiny_clips(nets, monkeypatch, T, mem_freq, idx):
    """Group formation corner cases: mem_freq above the 8-frame group cap, mem_freq beyond the clip (no insertion at
    all), 3- and 2-frame clips; grouped decode must equal the frame-by-frame path."""
    H, W = 112, 128
    img = synth.synthetic_clip(T, H, W, seed=6)
    msk = synth.synthetic_mask(T, H, W, 1, seed=7)
    res = {}
    for g in ("1", "8"):
        monkeypatch.setenv("STCN_DECODE_BATCH", g)
        core = make_core(nets)(img, 1, mem_freq)
        m = core.interact(msk[:, idx], idx).copy()
        res[g] = (m, core.prob.clone(), core.stats())
    assert res["1"][2] == res["8"][2], "same frames, misses, insertions"
    assert res["1"][2]["frames"] == T - 1
    assert iou(res["8"][0], res["1"][0]) >= 1 - 1e-3
    assert (res["8"][1] - res["1"][1]).abs().max().item() < 2e-3


def test_inputs_on_the_host_or_in_other_layouts_give_the_same_result(nets):
    """The reference accepts images / masks on any device (it moves them, inference_core.py:44-68,216-220); the shim
    moves and converts: CPU tensors, float64, non-contiguous views and an explicit device string."""
    from mivos.inference_core import InferenceCore
    T, H, W = 5, 112, 144
    img, msk = synth.synthetic_clip(T, H, W, seed=8), synth.synthetic_mask(T, H, W, 1, seed=9)
    ref = InferenceCore(nets[0], nets[1], img.cuda(), 1).interact(msk[:, 2].cuda(), 2)
    wide = torch.zeros(1, T, 3, H, W + 6, dtype=torch.float64)
    wide[..., 3:W + 3] = img
    a = InferenceCore(nets[0], nets[1], wide[..., 3:W + 3], 1, device="cuda:0").interact(msk[:, 2].double(), 2)   # CPU, f64, strided
    assert np.array_equal(a, ref)
    core = InferenceCore(nets[0], nets[1], img, 1, mem_profile=2)                                               # CPU clip, spill mode
    assert np.array_equal(core.interact(msk[:, 2], np.int64(2)), ref) and core.prob.is_cuda
    assert core.interact(msk[:, 2], 2, download=False) is None


def test_kept_results_do_not_pile_up_pinned_memory(nets):
    """interact() returns its masks in a pinned block the array owns; a reference-style loop that KEEPS every round's result must
    not accumulate page-locked memory: at most _PINNED_MAX_LIVE such blocks PER CORE are alive (round 6: a per-process count made
    the outcome depend on whatever other cores were alive - advisor, round 5), later results arrive pageable - with the same content
    either way - and a second core has its own budget."""
    from eva_vos_amd import inference_core as IC
    T, H, W = 4, 112, 128
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, 1)
    core = make_core(nets)(img, 1, 2)
    before = IC.pageable_downloads()
    kept = [core.interact(msk[:, i % T], i % T) for i in range(IC._PINNED_MAX_LIVE + 3)]
    pinned = [torch.from_numpy(a).is_pinned() for a in kept]
    assert pinned == [True] * IC._PINNED_MAX_LIVE + [False] * 3, pinned
    assert IC.pageable_downloads() - before == 3
    again = make_core(nets)(img, 1, 2)
    for i, a in enumerate(kept):
        b = again.interact(msk[:, i % T], i % T)                      # nothing kept: every download of this core is pinned
        assert np.array_equal(a, b) and torch.from_numpy(b).is_pinned()
    del kept, a, b
    assert torch.from_numpy(core.interact(msk[:, 0], 0)).is_pinned(), "released blocks free their slots"
    assert IC.pageable_downloads() - before == 3


def test_reset_equals_fresh_engine(nets):
    T, H, W = 6, 112, 144
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, 1)
    core = make_core(nets)(img, 1, 2)
    a = core.interact(msk[:, 1], 1).copy()
    pa = core.prob.clone()
    core.interact(msk[:, 4], 4)
    core.reset()
    assert float(core.prob[1:].abs().max()) == 0.0 and abs(float(core.prob[0].max()) - 1e-7) < 1e-12
    b = core.interact(msk[:, 1], 1)
    assert np.array_equal(a, b) and torch.equal(pa, core.prob)
    assert core.stats()["key_miss"] == T


@pytest.mark.parametrize("T,idx,k,mf", [(1, 0, 1, 5), (2, 1, 1, 5), (5, 4, 1, 1), (5, 0, 2, 1), (6, 3, 2, 3)])
def test_edge_shapes_match_oracle(T, idx, k, mf, nets, weights):
    """Degenerate clips and interaction positions: single frame (nothing to propagate), interaction on the
    last frame (forward pass empty), k = 2 through the scribble path, mem_freq = 1 (every frame memorised)."""
    H, W = 112, 128
    img = synth.synthetic_clip(T, H, W, seed=21)
    msk = synth.synthetic_mask(T, H, W, k, seed=22)
    m = msk[:, idx]
    if k > 1:
        m = torch.cat([1 - m.sum(0, keepdim=True).clamp(0, 1), m], 0)
    core = make_core(nets)(img, k, mf)
    orc = O.OracleCore(weights[0], weights[1], img, k, mem_freq=mf)
    a, b = core.interact(m, idx, scribble=k > 1), orc.interact(m, idx, scribble=k > 1)
    assert a.shape == b.shape == (T, H, W)
    yard = small_multi_object_yardstick() if k > 1 else None
    masks_close(a, b, k, f"edge T={T} k={k}", yard=yard)
    if k == 1:
        clean_frame_check(core.prob.cpu(), orc, 0, f"edge T={T}")
    elif T > 1:       # k > 1: the tail against the reference's own tail on the small multi-object fixtures (as the goldens)
        d = (core.prob.cpu() - orc.prob).abs().numpy()
        assert np.quantile(d, 0.999) <= 3 * yard[2] + 5e-4, (float(np.quantile(d, 0.999)), float(yard[2]))
    s = core.stats()
    assert s["frames"] == T - 1 and s["fused"] == 0


def test_fq_driver_end_to_end(nets, tmp_path):
    """Config-4-shaped loop on a synthetic dataset tree: loader -> InferenceCore -> GPU J -> oracle policy
    (8 rounds, worst frame next) -> 224x224 PNG states + CSV (reference generate_fq_dataset.py:60-86)."""
    import csv
    import os
    from eva_vos_amd import fq_driver
    imset = fq_driver.make_synthetic_tree(str(tmp_path / "db"), {"v0": (6, 112, 128, 2), "v1": (5, 112, 128, 1)})
    out = str(tmp_path / "fq")
    rows = fq_driver.run(str(tmp_path / "db"), imset, out, nets[0], nets[1], rounds=4)
    assert rows.shape[1] == 4 + 6 and len(rows) > 0
    for row in rows:
        n, worst = int(row[3]), int(row[2])
        q = row[4:4 + n]
        assert np.all((q >= 0) & (q <= 1)) and np.isnan(row[4 + n:]).all()
        assert worst == int(np.argmin(q)), "the next annotated frame is the worst one by J"
    by_sample = {}
    for row in rows:
        by_sample.setdefault(int(row[0]), []).append(row)
    assert set(by_sample) == {0, 1, 2}
    for rs in by_sample.values():                       # annotated frames count with their GT: J == 1 there
        assert rs[0][4] == 1.0
        if len(rs) > 1:
            assert rs[1][4 + int(rs[0][2])] == 1.0
    with open(os.path.join(out, "res_synthetic.csv")) as f:          # res_<imset>.csv (generate_fq_dataset.py:86)
        lines = list(csv.reader(f))
    assert lines[0] == ["state_name", "ious", "selected_frame"] and len(lines) == 1 + len(rows)
    assert lines[1][0] == "v0__1_round_1"
    d = os.path.join(out, "Annotations", "224", "v0__1_round_1")
    assert sorted(os.listdir(d)) == [f"{t:05d}.png" for t in range(6)]
    from PIL import Image
    assert Image.open(os.path.join(d, "00000.png")).size == (224, 224)
    for v, n in (("v0", 6), ("v1", 5)):                               # RGB frames once per video (generate_fq_dataset.py:77-80)
        r = os.path.join(out, "RGBFrames", "224", v)
        assert sorted(os.listdir(r)) == [f"{t:05d}.png" for t in range(n)]
        im = np.array(Image.open(os.path.join(r, "00000.png")))
        assert im.shape == (224, 224, 3) and im.min() == 0 and im.max() == 255     # per-frame min-max normalisation


def test_core_used_under_another_stream_keeps_pytorch_ordering(nets):
    """The engine is bound to the stream of construction; interact() / deepcopy under a different current stream must still
    be ordered with the caller's work (mask produced on the caller's stream, results read there)."""
    T, H, W = 6, 128, 160
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, 1)
    want = make_core(nets)(img, 1, 2).interact(msk[:, 0], 0)
    core = make_core(nets)(img, 1, 2)                            # bound to the default stream
    other = torch.cuda.Stream()
    with torch.cuda.stream(other):
        m = msk[:, 0].cuda(non_blocking=True) * 1.0              # produced on `other`
        got = core.interact(m, 0)
        amax = core.prob.argmax(0)                               # consumer on `other`: must see the finished propagation
        twin = copy.deepcopy(core)
        got2 = twin.interact(msk[:, 3], 3)
    other.synchronize()
    assert np.array_equal(got, want)
    lw, uw, lh, uh = core.pad
    assert np.array_equal(amax[:, 0, lh:core.nh - uh, lw:core.nw - uw].cpu().numpy().astype(np.uint8), want)
    ref = make_core(nets)(img, 1, 2)
    ref.interact(msk[:, 0], 0)
    assert np.array_equal(got2, ref.interact(msk[:, 3], 3))


def test_new_weights_in_the_same_module_are_picked_up(weights):
    """The engine snapshots BN-folded weights per (module, weight fingerprint): load_state_dict into the SAME module objects
    (one script evaluating several checkpoints) must give the new checkpoint's results, as the reference (live parameters)."""
    from eva_vos_amd.params import FusionNet, PropagationNetwork
    T, H, W = 4, 128, 160
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, 1)
    p, f = PropagationNetwork(), FusionNet()
    p.load_state_dict(weights[0]); f.load_state_dict(weights[1])
    a = make_core((p, f))(img, 1, 2)
    a.interact(msk[:, 0], 0)
    pa = a.prob.clone()
    other = synth.recipe_state_dict(PropagationNetwork(), 1)
    p.load_state_dict(other)                                     # same module object, new checkpoint
    b = make_core((p, f))(img, 1, 2)
    b.interact(msk[:, 0], 0)
    q = PropagationNetwork()
    q.load_state_dict(other)
    c = make_core((q, f))(img, 1, 2)
    c.interact(msk[:, 0], 0)
    assert torch.equal(b.prob, c.prob), "stale weight snapshot"
    assert (b.prob - pa).abs().max() > 1e-3
    p.load_state_dict(weights[0])                                # and back
    d = make_core((p, f))(img, 1, 2)
    d.interact(msk[:, 0], 0)
    assert torch.equal(d.prob, pa)


def test_the_3x3_convs_really_run_as_winograd(nets):
    """Guards against a silent fall-back to the direct kernel: at 480p the profile must show Winograd input transforms and
    fewer EXECUTED than algorithmic conv FLOP (85 % of the conv FLOP are stride-1 3x3 convs with >= 128 channels)."""
    T, H, W = 4, 480, 854
    core = make_core(nets)(synth.synthetic_clip(T, H, W), 1, 2)
    core.set_profiling(True)
    core.interact(synth.synthetic_mask(T, H, W, 1)[:, 0], 0)
    prof = core.kernel_profile()
    assert prof["wino_input"]["launches"] > 0 and prof["wino_input"]["ms"] > 0
    ratio = prof["conv"]["exec_flops"] / prof["conv"]["flops"]
    hb = prof["conv_hbm_bound"]
    assert hb["wino2_flops"] > 0 and hb["wino4_flops"] > 0, "both the F(2x2) (trunk / 1/16-scale) and the F(4x4) (decoder side) path must be taken"
    assert 0.3 < ratio < 0.65, ratio           # 1 / 2.25 on the F(2x2) share, 1 / 4 on the F(4x4) share, 1 on the rest, + tile padding


def test_multi_object_decode_groups_equal_the_frame_by_frame_path(nets_multi, monkeypatch):
    """k = 3, mem_freq = 5: the frames between two bank insertions are decoded as ONE batch of objects x frames (per-frame
    tensors broadcast over the objects by a modulo batch index).  Same engine with STCN_DECODE_BATCH=1 (the reference's
    loop order, prop_net.py:183-187 / inference_core.py:166-188) must give the same masks and, up to fp32 rounding of
    differently shaped launches (other GEMM shapes; single frames run the 1/16-scale layers as F(4x4) in K pieces), the same
    probabilities.  Under the MULTI-OBJECT recipe at 240x432 (99.6 % decisive pixels in the oracle): with the plain recipe only
    52-67 % of a 128x160 frame carry a well-conditioned label and the 1e-3 below was a matter of luck."""
    nets = nets_multi
    T, H, W, k = 12, 240, 432, 3
    img, msk = synth.synthetic_clip(T, H, W, seed=3), synth.synthetic_mask(T, H, W, k, seed=4)
    m0 = torch.cat([1 - msk[:, 0].sum(0, keepdim=True).clamp(0, 1), msk[:, 0]], 0)
    m7 = torch.cat([1 - msk[:, 7].sum(0, keepdim=True).clamp(0, 1), msk[:, 7]], 0)
    outs = []
    for batch in ("1", None):
        if batch:
            monkeypatch.setenv("STCN_DECODE_BATCH", batch)
        else:
            monkeypatch.delenv("STCN_DECODE_BATCH", raising=False)
        core = make_core(nets)(img, k, 5)
        a = core.interact(m0, 0, scribble=True).copy()
        b = core.interact(m7, 7, scribble=True).copy()
        outs.append((a, b, core.prob.clone()))
    for o in range(1, k + 1):
        assert iou(outs[0][0] == o, outs[1][0] == o) >= 1 - 1e-3 and iou(outs[0][1] == o, outs[1][1] == o) >= 1 - 1e-3
    d = (outs[0][2] - outs[1][2]).abs()
    assert float(torch.quantile(d.flatten()[::3].float(), 0.999)) < 1e-3, float(d.max())


def test_eight_objects_work_and_the_stated_maximum_is_enforced(nets):
    """k = 8 (the most the 8-object instantiation of the aggregation kernels holds; decode groups shrink to 2 frames so that objects x frames
    <= 16): probabilities are a distribution, every object keeps pixels, a repeat is bit-identical; k = STCN_MAX_OBJECTS + 1 = 33 is refused with a message."""
    T, H, W, k = 6, 128, 160, 8
    img, msk = synth.synthetic_clip(T, H, W, seed=11), synth.synthetic_mask(T, H, W, k, seed=12)
    m0 = torch.cat([1 - msk[:, 0].sum(0, keepdim=True).clamp(0, 1), msk[:, 0]], 0)
    outs = []
    for _ in range(2):
        core = make_core(nets)(img, k, 3)
        outs.append((core.interact(m0, 0, scribble=True).copy(), core.prob.clone()))
    assert np.array_equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert (outs[0][1][:, 1:].sum(0) - 1).abs().max() < 1e-5
    assert set(np.unique(outs[0][0])) <= set(range(k + 1))
    with pytest.raises(RuntimeError, match="1<=k<=32"):        # include/stcn_hip.h: STCN_MAX_OBJECTS (the reference class has no limit)
        make_core(nets)(img, 33, 3)


def test_thirty_two_objects_against_the_oracle(nets_multi, weights_multi):
    """STCN_MAX_OBJECTS = 32 objects in one engine (the reference class has no limit; rounds 1-5 stopped at 8): a propagation and a FUSED
    second interaction through the scribble path on a 160x192 clip against the CPU oracle - probabilities of all 33 rows on every
    pixel, the label map, and the rows of a frame summing to one.  (seqK10 / seqK16 hold the same against the REFERENCE's own output.)"""
    T, H, W, k = 5, 160, 192, 32
    img, msk = synth.synthetic_clip(T, H, W, seed=3), synth.synthetic_mask(T, H, W, k, seed=4)
    core = make_core(nets_multi)(img, k, 2)
    orc = O.OracleCore(weights_multi[0], weights_multi[1], img, k, mem_freq=2)
    for r, f in enumerate((0, 3)):
        m = torch.cat([1 - msk[:, f].sum(0, keepdim=True).clamp(0, 1), msk[:, f]], 0)
        a, b = core.interact(m, f, scribble=True), orc.interact(m.clone(), f, scribble=True)
        d = (core.prob.cpu() - orc.prob).abs()
        diff = int((a != b).sum())
        print(f"k = 32 round {r}: max |dprob| {float(d.max()):.1e}, p99.9 {float(torch.quantile(d.flatten()[::3], 0.999)):.1e}, {diff} of {a.size} labels differ; "
              f"labels present {len(np.unique(b))}")
        assert float(torch.quantile(d.flatten()[::3], 0.999)) < 1e-3 and diff <= 1e-3 * a.size, (r, float(d.max()), diff)
        free = [t for t in range(T) if t not in (0, 3)]             # (an interacted frame holds the given mask rows, which overlap)
        assert (core.prob[:, free].sum(0) - 1).abs().max() < 1e-5
    assert core.stats()["fused"] > 0                             # (counters of the last interaction: the second one fused)


@pytest.mark.parametrize("k", [10, 32])
def test_many_objects_at_480p_against_the_oracle(k, nets_multi, weights_multi):
    """The BASELINE frame size with more than 8 objects (10 = the most a DAVIS-2017 video holds; 32 = STCN_MAX_OBJECTS): workspaces, Winograd
    chunking and every batched launch at objects x 480x864.  Three frames, every frame in the bank, HIP engine against the CPU oracle."""
    T, H, W = 3, 480, 854
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, k)
    m0 = torch.cat([1 - msk[:, 0].sum(0, keepdim=True).clamp(0, 1), msk[:, 0]], 0)
    core = make_core(nets_multi)(img, k, 1)
    orc = O.OracleCore(weights_multi[0], weights_multi[1], img, k, mem_freq=1)
    a, b = core.interact(m0, 0, scribble=True), orc.interact(m0.clone(), 0, scribble=True)
    d = (core.prob.cpu() - orc.prob).abs()
    q = float(torch.quantile(d.flatten()[::max(7, d.numel() // 8000000 + 1)], 0.999))
    diff = int((a != b).sum())
    print(f"k = {k} at 480p: max |dprob| {float(d.max()):.1e}, p99.9 {q:.1e}, {diff} of {a.size} labels differ, labels present {len(np.unique(b))}")
    assert q < 1e-3 and diff <= 1e-3 * a.size, (k, q, diff)
    assert (core.prob[:, 1:].sum(0) - 1).abs().max() < 1e-5


def test_a_failing_interaction_leaves_a_defined_state(nets):
    """Fault injection (stcn_test_fail_at: the n-th launch check of this thread fails): an interaction that fails midway
    raises, rolls the host bookkeeping back and puts the engine into a failed state - the next interact is REFUSED (never a
    continuation from a half-updated bank / interaction set) until reset(), after which the engine equals a fresh one."""
    from eva_vos_amd import _lib
    T, H, W = 9, 112, 144
    img, msk = synth.synthetic_clip(T, H, W, seed=13), synth.synthetic_mask(T, H, W, 1, seed=14)
    ref = make_core(nets)(img, 1, 3)
    want1 = ref.interact(msk[:, 2], 2).copy()
    want2 = ref.interact(msk[:, 6], 6).copy()
    prob2 = ref.prob.clone()
    for n in (1, 3, 25, 50):                   # at the first launch, in the certain-memory value encode, in either sweep (~65 checks)
        core = make_core(nets)(img, 1, 3)
        assert np.array_equal(core.interact(msk[:, 2], 2), want1)
        _lib.check(_lib.lib().stcn_test_fail_at(n))
        try:
            with pytest.raises(RuntimeError, match="injected fault"):
                core.interact(msk[:, 6], 6)
        finally:
            _lib.check(_lib.lib().stcn_test_fail_at(0))
        assert core.interacted == {2}, "the shim must not record a failed interaction"
        with pytest.raises(RuntimeError, match="failed state"):
            core.interact(msk[:, 6], 6)
        with pytest.raises(RuntimeError, match="failed state"):
            copy.deepcopy(core).interact(msk[:, 6], 6)           # a clone of a failed engine is failed too
        core.reset()
        assert np.array_equal(core.interact(msk[:, 2], 2), want1)
        assert np.array_equal(core.interact(msk[:, 6], 6), want2) and torch.equal(core.prob, prob2)
    # an argument error is not a failure of the engine: it stays usable
    core = make_core(nets)(img, 1, 3)
    with pytest.raises(RuntimeError):
        core.interact(msk[:, 2], T + 3)
    assert np.array_equal(core.interact(msk[:, 2], 2), want1)


def test_an_up_front_out_of_memory_does_not_cost_the_session(nets):
    """stcn_interact reserves the bank memory of the whole interaction BEFORE its first mutation; a failure there (injected:
    stcn_test_fail_at(-1)) has touched nothing, so the call raises and the engine stays USABLE - earlier rounds are not lost,
    the same interaction can simply be repeated (advisor, round 3)."""
    from eva_vos_amd import _lib
    T, H, W = 9, 112, 144
    img, msk = synth.synthetic_clip(T, H, W, seed=13), synth.synthetic_mask(T, H, W, 1, seed=14)
    ref = make_core(nets)(img, 1, 3)
    want1 = ref.interact(msk[:, 2], 2).copy()
    want2 = ref.interact(msk[:, 6], 6).copy()
    core = make_core(nets)(img, 1, 3)
    assert np.array_equal(core.interact(msk[:, 2], 2), want1)
    prob1 = core.prob.clone()
    _lib.check(_lib.lib().stcn_test_fail_at(-1))
    with pytest.raises(RuntimeError, match="bank_reserve"):
        core.interact(msk[:, 6], 6)
    assert core.interacted == {2} and torch.equal(core.prob, prob1), "nothing may have been touched"
    assert np.array_equal(core.interact(msk[:, 6], 6), want2) and torch.equal(core.prob, ref.prob)


@pytest.mark.parametrize("mf,batch,second", [(2, "1", 2), (2, None, 2), (3, "1", 1), (1, "1", 3), (5, None, 1)])
def test_a_slow_side_stream_cannot_corrupt_the_aggregate_buffers(nets, monkeypatch, mf, batch, second):
    """FusionNet of a decoded group runs on the engine's side stream out of one of two aggregate buffers while the main stream
    goes on decoding.  Round 2 BELOW an earlier interaction: the forward sweep is fused (offloaded), the backward sweep that
    follows is not - its first decode writes aggregate buffer 0 and has to wait for the side stream if the last offloaded group
    still reads it (advisor, round 3: that wait was missing).  With every offloaded group delayed by 30 ms the race is certain
    without the wait; the result must equal the same session with FusionNet in line (STCN_FUSE_SIDE=0), bit for bit."""
    from eva_vos_amd import _lib
    T, H, W = 13, 112, 144
    img, msk = synth.synthetic_clip(T, H, W, seed=51), synth.synthetic_mask(T, H, W, 1, seed=52)
    if batch:
        monkeypatch.setenv("STCN_DECODE_BATCH", batch)
    script = [(7, 7), (second, second), (10, 10), (4, 4)]          # below, above, between earlier interactions
    outs = []
    for side in ("0", "1"):
        monkeypatch.setenv("STCN_FUSE_SIDE", side)
        core = make_core(nets)(img, 1, mf)
        _lib.check(_lib.lib().stcn_test_side_delay_us(30000 if side == "1" else 0))
        try:
            res = [core.interact(msk[:, f], i).copy() for f, i in script]
            fused = core.stats()["fused"]
            torch.cuda.synchronize()
        finally:
            _lib.check(_lib.lib().stcn_test_side_delay_us(0))
        outs.append((res, core.prob.clone()))
        assert fused > 0
    for a, b in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(a, b)
    assert torch.equal(outs[0][1], outs[1][1])


def test_engine_options_are_explicit_per_engine_and_inherited_by_clones(nets, monkeypatch):
    """The engine tunables travel through the ABI (stcn_engine_create_ex), not through os.environ: two engines of one process
    with different options, an explicit option beats the environment, a clone keeps its source's resolved values whatever the
    environment says at clone time, results do not depend on them, unknown names are refused."""
    from mivos.inference_core import InferenceCore
    T, H, W = 9, 112, 144
    img, msk = synth.synthetic_clip(T, H, W, seed=21), synth.synthetic_mask(T, H, W, 1, seed=22)
    monkeypatch.setenv("STCN_LOOKAHEAD", "2")
    monkeypatch.setenv("STCN_DECODE_BATCH", "2")
    a = InferenceCore(nets[0], nets[1], img, 1, mem_freq=3)
    b = InferenceCore(nets[0], nets[1], img, 1, mem_freq=3, engine_options={"lookahead": 0, "decode_batch": 3, "key_batch": 2})
    assert a.engine_options() == {"lookahead": 2, "decode_batch": 2, "key_batch": 4, "fuse_side": 1}
    assert b.engine_options() == {"lookahead": 0, "decode_batch": 3, "key_batch": 2, "fuse_side": 0}      # no side stream at all
    ra = [a.interact(msk[:, i], i).copy() for i in (1, 6)]
    rb = [b.interact(msk[:, i], i).copy() for i in (1, 6)]
    assert all(iou(x, y) >= 1 - 1e-3 for x, y in zip(ra, rb))          # another batching = another M of the same GEMMs: fp32 rounding only
    monkeypatch.setenv("STCN_LOOKAHEAD", "0")
    monkeypatch.setenv("STCN_DECODE_BATCH", "1")
    c = copy.deepcopy(a)
    assert c.engine_options() == a.engine_options()
    assert np.array_equal(c.interact(msk[:, 4], 4), a.interact(msk[:, 4], 4))
    with pytest.raises(TypeError, match="unknown engine_options"):
        InferenceCore(nets[0], nets[1], img, 1, engine_options={"look_ahead": 0})


def test_a_clone_keeps_the_launch_knobs_of_its_source(nets, monkeypatch):
    """The launch-level tunables (csrc/kernels.h: Knobs) are snapshotted per workspace when an engine is created; a clone must carry its
    SOURCE's snapshot, not today's environment (advisor, round 4).  Observable through the conv trace: with STCN_PW_CHAIN=0 the large 1x1
    convs of the key encoder run on the one-tile instance, with the default on the chain kernel."""
    T, H, W = 5, 480, 854
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, 1)

    def chain_layers(core, idx):
        _, paths = _conv_trace(lambda: core.interact(msk[:, idx], idx))
        return sorted(n for n, p in paths.items() if any(q.startswith("direct_pointwise_chain") for q in p))

    monkeypatch.setenv("STCN_PW_CHAIN", "0")
    a = make_core(nets)(img, 1, 5)
    monkeypatch.delenv("STCN_PW_CHAIN")
    twin = copy.deepcopy(a)                                    # cloned under the DEFAULT environment
    fresh = make_core(nets)(img, 1, 5)
    assert chain_layers(fresh, 0), "the default build must take the chain kernel for the res2 / layer2 expansions at this size (else the test is vacuous)"
    assert chain_layers(a, 0) == [] and chain_layers(twin, 0) == [], "the clone re-read the environment"


def test_weight_snapshots_are_kept_per_fusion_net_and_data_writes_are_seen(weights):
    """Advisor items of round 2: (i) alternating two fusion networks with one propagation network must not rebuild the
    model every time (small LRU of snapshots); (ii) a whole-model update through ``.data`` (no version bump) is caught by
    the content probe of the fingerprint."""
    from eva_vos_amd import inference_core as IC
    from eva_vos_amd.params import FusionNet, PropagationNetwork
    p, f1, f2 = PropagationNetwork(), FusionNet(), FusionNet()
    p.load_state_dict(weights[0]); f1.load_state_dict(weights[1]); f2.load_state_dict(weights[1])
    m1, m2, m0 = IC._model_for(p, f1, 0), IC._model_for(p, f2, 0), IC._model_for(p, None, 0)
    assert IC._model_for(p, f1, 0) is m1 and IC._model_for(p, f2, 0) is m2 and IC._model_for(p, None, 0) is m0
    with torch.no_grad():
        for t in p.state_dict(keep_vars=True).values():
            if t.is_floating_point():
                t.data.mul_(1.0009765625)                       # exact in fp32, invisible to _version
    assert IC._model_for(p, f1, 0) is not m1, "a .data update of every tensor must yield a fresh snapshot"


def test_480p_multi_object_decode_groups_match_the_oracle(nets_multi, weights_multi):
    """The k > 1 decode-group path (objects x frames in one batch, per-frame tensors broadcast by a modulo batch index) at the
    BASELINE resolution against the CPU oracle: 480x854, 3 objects, mem_freq = 3 (groups of 3 frames), two rounds with fusion,
    under the MULTI-OBJECT weight recipe (seed 2) on a clip the recipe's fitted layers have not seen as k = 3.  Per-object mask
    IoU over ALL pixels, on the clip and on every frame, against max(1e-3, 3 x the reference's own envelope on the 480p
    five-object fixture); the pixels whose label is well-conditioned in the oracle's own probabilities (top-1 minus top-2
    >= 1e-2) must be at least 80 % of the frame - otherwise the comparison would speak for a minority (round 3: 22 %)."""
    nets, weights = nets_multi, weights_multi
    T, H, W, k = 8, 480, 854, 3
    img, msk = synth.synthetic_clip(T, H, W, seed=41), synth.synthetic_mask(T, H, W, k, seed=42)
    core = make_core(nets)(img, k, 3)
    orc = O.OracleCore(weights[0], weights[1], img, k, mem_freq=3)
    noise = load_golden("selfnoise")["seq480k5"][0]
    for idx in (0, 5):
        m = torch.cat([1 - msk[:, idx].sum(0, keepdim=True).clamp(0, 1), msk[:, idx]], 0)
        a, b = core.interact(m, idx, scribble=True), orc.interact(m.clone(), idx, scribble=True)
        lw, uw, lh, uh = orc.pad
        po = orc.prob[:, :, 0, lh:orc.prob.shape[3] - uh if uh else None, lw:orc.prob.shape[4] - uw if uw else None]
        top = torch.topk(po, 2, dim=0).values
        dec = ((top[0] - top[1]) >= 1e-2).numpy()
        assert dec.mean() > 0.8, f"only {100 * dec.mean():.1f} % decisive pixels: the comparison would speak for a minority of the frame"
        d = (core.prob.cpu() - orc.prob).abs().numpy()
        q999 = float(np.quantile(d.reshape(-1)[::5], 0.999))
        print(f"480p k=3 groups, interact({idx}): {100 * dec.mean():.1f} % decisive pixels, {int((a != b).sum())} of {a.size} mask pixels differ "
              f"({int(((a != b) & dec).sum())} on decisive pixels); |dprob| p99.9 {q999:.1e} max {d.max():.1e}")
        masks_close(a, b, k, f"480p k=3 interact({idx})", yard=noise)
        assert all((b[t] == o).sum() >= 256 for o in range(1, k + 1) for t in range(T)), "an object vanished in the oracle"
        assert q999 <= 3 * float(noise[2]) + 5e-4, (idx, q999)
    s_ = core.stats()
    assert s_["fused"] > 0 and s_["frames"] == T - 2


def test_sixteen_round_annotation_session_at_480p_matches_the_oracle(nets, weights):
    """The reference's annotation loops run 8 (interactions/mask.py:113-146) to 60 (eval_annotation_method.py:30) interactions per
    sample.  A whole 16-round session of the oracle mask policy (annotate frame 0, then the frame with the worst J against the
    ground truth; annotated frames count with their ground truth) at the BASELINE resolution, T = 34 (the shortest DAVIS-val clip: half
    of its frames end up annotated), HIP engine against the CPU oracle after EVERY round: growing certain memory (16 slots), ever shorter
    spans, fusion on both sides of earlier interactions.  Per round the CODED bounds of bench.py (the same function the driver's
    `parity_session` leg reports): clip max(1e-3, 3 x the reference's own spread), every frame max(1e-3, 3 x the reference's own
    per-frame spread at 480p, 2 px / union px)."""
    import bench
    res = bench.session_parity(nets[0], nets[1], weights[0], weights[1], 480, 854, 34, 16, 5)
    print(res["session"], res["frames_annotated"])
    assert len(set(res["frames_annotated"])) == 16 and res["last_round_stats"]["bank_fwd"] >= 16
    for r in res["rounds"]:
        print(f"round {r['round']} (frame {r['frame']}): clip IoU {r['mask_iou']:.6f} (bound {r['clip_bound']:.1e}), worst frame {r['min_frame_iou']:.6f} @ {r['min_frame_iou_frame']} "
              f"(bound {r['frame_bound']:.1e}), {r['mask_pixels_differing']} px differ, next frame oracle / HIP {r['next_frame_oracle']} / {r['next_frame_hip']}")
        assert r["within_bound"], r
        assert abs(r["mean_j_oracle"] - r["mean_j_hip"]) < 1e-4, r
    assert res["within_bound"]


def test_sixty_round_session_grows_sixty_certain_slots_and_matches_the_oracle(nets, weights):
    """BASELINE config 5 runs 60 annotation rounds per sample (eval_annotation_method.py:30).  A whole 60-round session on a 40-frame
    128x160 clip, HIP engine against the CPU oracle after EVERY round: the oracle mask policy (worst frame by J against the ground truth,
    annotated frames counting with their ground truth: interactions/mask.py:113-146) until every frame is annotated (round 40), then the
    frames once more in a fixed order - a re-annotation appends the same key rows again (inference_core.py:235-240), so the certain memory
    ends at 60 slots, two thirds of them exact duplicates of each other's keys, and every sweep is squeezed between annotated neighbours."""
    T, H, W, R = 40, 128, 160, 60
    img, msk = synth.synthetic_clip(T, H, W, seed=71), synth.synthetic_mask(T, H, W, 1, seed=72)
    gtb = msk[0, :, 0].numpy() > 0.5
    core = make_core(nets)(img, 1, 5)
    orc = O.OracleCore(weights[0], weights[1], img, 1, mem_freq=5)
    yard = np.max([load_golden("selfnoise")[t].max(0) for t in ("seqA", "seqA1", "seqB", "seqE")], 0)
    frames, worst = [0], 0.0
    for r in range(R):
        f = frames[r]
        a, b = core.interact(msk[:, f], f), orc.interact(msk[:, f], f)
        masks_close(a, b, 1, f"60-round session r{r + 1} (frame {f})", yard)
        worst = max(worst, 1 - iou(a > 0, b > 0))
        done = sorted(set(frames))
        gen = b > 0
        gen[done] = gtb[done]
        u, n = (gen | gtb).reshape(T, -1).sum(1), (gen & gtb).reshape(T, -1).sum(1)
        q = np.where(u > 0, n / np.maximum(u, 1), 0.0)
        q[done] = 2.0                                            # never the worst while another frame is left
        frames.append(int(np.argmin(q)) if len(done) < T else (7 * r) % T)
    st = core.stats()
    print(f"60 rounds: {len(set(frames[:R]))} distinct frames annotated, certain slots {st['bank_fwd']} / {st['bank_bwd']}, worst clip 1-IoU {worst:.2e}")
    assert len(set(frames[:R])) == T and min(st["bank_fwd"], st["bank_bwd"]) >= R and len(orc.certain_k) == R


def test_a_480p_session_cloned_mid_way_continues_on_both_branches(nets, weights):
    """interactions/policies.py:103-104 deep-copies the processor in the MIDDLE of a session (upper-bound frame search) and interacts
    with the copy.  At the BASELINE resolution: two rounds, copy.deepcopy, then the source and the clone continue with DIFFERENT
    annotations (two more rounds each, fused on both sides of earlier interactions) - each branch against its own CPU-oracle run
    (the oracle deep-copied at the same point), and the branches must not see each other (inference_core.py:235-240: certain memory
    is per processor)."""
    T, H, W = 10, 480, 854
    img, msk = synth.synthetic_clip(T, H, W, seed=81), synth.synthetic_mask(T, H, W, 1, seed=82)
    core = make_core(nets)(img, 1, 3)
    orc = O.OracleCore(weights[0], weights[1], img, 1, mem_freq=3)
    yard = load_golden("selfnoise")["seq480"].max(0)
    for r, f in enumerate((0, 6)):
        masks_close(core.interact(msk[:, f], f), orc.interact(msk[:, f], f), 1, f"480p pre-clone r{r}", yard)
    twin, orc2 = copy.deepcopy(core), copy.deepcopy(orc)
    src_prob = core.prob.clone()
    for r, (fa, fb_) in enumerate(((3, 8), (8, 2))):                # source: 3 then 8; clone: 8 then 2
        a2, b2 = twin.interact(msk[:, fb_], fb_), orc2.interact(msk[:, fb_], fb_)
        if r == 0:
            assert torch.equal(core.prob, src_prob), "an interaction on the clone must not touch the source's probabilities"
        a1, b1 = core.interact(msk[:, fa], fa), orc.interact(msk[:, fa], fa)
        masks_close(a1, b1, 1, f"480p source branch r{r} (frame {fa})", yard)
        masks_close(a2, b2, 1, f"480p clone branch r{r} (frame {fb_})", yard)
    assert core.interacted == {0, 6, 3, 8} and twin.interacted == {0, 6, 8, 2}
    assert core.stats()["bank_fwd"] >= 4 and twin.stats()["bank_fwd"] >= 4
    assert (core.prob - twin.prob).abs().max() > 1e-3, "the two branches annotated different frames"
    # the same branch replayed on a fresh engine gives the clone's answer bit for bit (a clone is not an approximation of its source)
    fresh = make_core(nets)(img, 1, 3)
    for f in (0, 6, 8, 2):
        last = fresh.interact(msk[:, f], f)
    assert np.array_equal(last, a2) and torch.equal(fresh.prob, twin.prob)


def _long_golden(name):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz not captured (oracle/gen_golden_long.py, build container only)")
    return dict(np.load(path))


def test_twenty_four_round_session_matches_the_reference_itself(nets):
    """Long-horizon parity against the REFERENCE, not the oracle: tests/golden/long_sess24.npz holds the masks after rounds 8, 16 and 24 of the oracle
    mask policy (interactions/mask.py:113-146) that the reference itself produced on the 34-frame 480x854 clip of bench.py's session leg
    (oracle/gen_golden_long.py), the frames it annotated, and per round how far its own 8-thread and 1-thread executions of that session
    drift apart (all 24 rounds).  The HIP engine follows the same 24 annotations; at the stored rounds: clip 1-IoU <= max(1e-3, 1.5 x the reference's own clip spread of
    that round), every frame <= max(1e-3, 1.5 x the reference's own worst frame of that round, 2 px / union px)."""
    g = _long_golden("long_sess24")
    T, H, W, k, mf = (int(v) for v in g["shape"])
    seed = int(g["seed"])
    img, msk = synth.synthetic_clip(T, H, W, seed=seed), synth.synthetic_mask(T, H, W, 1, seed=seed)
    core = make_core(nets)(img, 1, mf)
    worst = [0.0, 0.0]
    for r, f in enumerate(int(v) for v in g["frames"]):
        a = core.interact(msk[:, f], f, download=f"r{r}.masks" in g)
        if a is None:
            continue                                            # the reference's masks are stored for rounds 8, 16 and 24 (1 MB each)
        a = a > 0
        b = np.unpackbits(g[f"r{r}.masks"])[: T * H * W].reshape(T, H, W).astype(bool)
        noise = g["selfnoise"][r]
        vol = 1 - iou(a, b)
        miss, fr = frame_miss(a, b)
        px = (a[fr] | b[fr]).sum() if fr >= 0 else 1
        vb, fb = clip_bound(noise[0]), frame_bound(noise[4], px)
        worst = [max(worst[0], vol / vb), max(worst[1], miss / fb)]
        print(f"HIP vs REFERENCE session round {r + 1} (frame {f}): clip 1-IoU {vol:.2e} (bound {vb:.1e}; reference vs itself {noise[0]:.2e}), worst frame {fr}: "
              f"{miss:.2e} (bound {fb:.1e}; reference vs itself {noise[4]:.2e}), {int((a != b).sum())} px differ (reference vs itself {int(noise[3])})")
        assert vol <= vb and miss <= fb, (r, f, vol, vb, fr, miss, fb)
    print(f"worst measured / bound over the session: clip {worst[0]:.2f}, frame {worst[1]:.2f}")
    assert core.stats()["bank_fwd"] >= 24


def test_config3_at_full_length_against_the_reference_itself(nets_multi):
    """BASELINE config 3 as stated - 480x854, five objects through the scribble path, every frame in the bank, T = 104 - against the label map
    the REFERENCE produced for all 104 frames (tests/golden/long_cfg3.npz, oracle/gen_golden_long.py; multi-object recipe, all pixels).
    What round 6 measured (profiles/r06_bn_unfolded_ab.txt): 2333 of 42.6 M pixels differ (the reference against itself at 1 and 8 threads:
    824; the BatchNorm-folded CPU oracle: 3227).  Objects 1 and 2 hold the north_star's 1e-3 on the clip; the small objects 3-5 (4-8 k
    pixels per frame, 10-24 boundary pixels of them differ per frame: top-50 membership flips at near-ties of the reference's own fp32 affinity,
    profiles/r06_cfg3_flip_probe.txt) measure 1.0-1.9e-3 - not a conv-algorithm effect (no Winograd at all: 2461 px).  This test states exactly that: the plain bound where it is met, the measured level (x 1.3) as a regression guard where it is
    not, the pixel count as a whole."""
    g = _long_golden("long_cfg3")
    T, H, W, k, mf = (int(v) for v in g["shape"])
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, k)
    m0 = torch.cat([1 - msk[:, 0].sum(0, keepdim=True).clamp(0, 1), msk[:, 0]], 0)
    core = make_core(nets_multi)(img, k, mf)
    a, b = core.interact(m0, 0, scribble=True), g["masks"]
    px = int((a != b).sum())
    clip = [1 - iou(a == o, b == o) for o in range(1, k + 1)]
    worst = [frame_miss(a == o, b == o)[0] for o in range(1, k + 1)]
    print(f"HIP vs REFERENCE config 3 full length: {px} of {a.size} px differ; clip 1-IoU per object {['%.2e' % v for v in clip]}; worst frame {['%.2e' % v for v in worst]}; "
          f"north_star 1e-3 on the clip: met by objects {[o + 1 for o, v in enumerate(clip) if v <= 1e-3]}")
    assert px <= 3000, px                                              # measured 2333 (5.5e-5 of the pixels); the folded CPU oracle: 3227
    assert clip[0] <= 1e-3 and clip[1] <= 1e-3, clip                   # the two large objects: the plain bound (measured 2.1e-4, 4.4e-4)
    assert max(clip) <= 2.5e-3, clip                                   # objects 3-5: measured 1.83e-3 / 1.24e-3 / 1.02e-3 - a regression guard, NOT the north_star bar
    assert max(worst) <= 2e-2, worst                                   # worst frame of the smallest object: 1.3e-2 (5 of ~400 px); the reference against itself: 7.1e-3
    assert core.stats()["bank_fwd"] >= T - 2                     # every frame but the last entered the bank


def test_config3_first_24_frames_against_the_reference_and_its_own_spread(nets_multi):
    """The clip of bench.py's default config-3 parity leg (480x854, k = 5, mem_freq = 1, T = 24): the HIP engine against the label map the
    REFERENCE produced for exactly this clip (tests/golden/long_cfg3_24.npz), with the reference's OWN spread on this clip as the yardstick
    (selfnoise row `cfg3_24`: 1 thread vs 8 threads - 255 px differ, worst object 5.0e-4 on the clip, worst (object, frame) 2.8e-3).
    The bar: every object holds 1e-3 on the clip; every (object, frame) holds max(1e-3, 1.5 x the reference's worst frame, 2 px / union)."""
    g = _long_golden("long_cfg3_24")
    T, H, W, k, mf = (int(v) for v in g["shape"])
    row = load_golden("selfnoise")["cfg3_24"][0]
    img, msk = synth.synthetic_clip(T, H, W), synth.synthetic_mask(T, H, W, k)
    m0 = torch.cat([1 - msk[:, 0].sum(0, keepdim=True).clamp(0, 1), msk[:, 0]], 0)
    core = make_core(nets_multi)(img, k, mf)
    a, b = core.interact(m0, 0, scribble=True), g["masks"]
    px = int((a != b).sum())
    clip = [1 - iou(a == o, b == o) for o in range(1, k + 1)]
    over = []
    for o in range(1, k + 1):
        x, y = (a == o).reshape(T, -1), (b == o).reshape(T, -1)
        u, n = (x | y).sum(1), (x & y).sum(1)
        miss = np.where(u >= 64, 1 - n / np.maximum(u, 1), 0.0)
        bound = np.array([frame_bound(row[4], v) for v in u])
        over.append(float((miss / bound).max()))
    print(f"HIP vs REFERENCE config 3, 24 frames: {px} of {a.size} px differ (reference vs itself: {int(row[3])}); clip 1-IoU per object {['%.2e' % v for v in clip]} "
          f"(reference vs itself, worst object: {row[0]:.2e}); worst (object, frame) measured / bound per object {['%.2f' % v for v in over]}")
    assert max(clip) <= 1e-3, clip
    assert max(over) <= 1.0, over
    assert px <= 4 * row[3], (px, row[3])


_POOL_SCRIPT = r"""
import sys, torch
sys.path.insert(0, %r)
from eva_vos_amd import synth
from eva_vos_amd.inference_core import release_pooled_memory
from eva_vos_amd.params import FusionNet, PropagationNetwork
from mivos.inference_core import InferenceCore
torch.set_grad_enabled(False)
prop, fuse = PropagationNetwork(), FusionNet()
prop.load_state_dict(synth.recipe_state_dict(prop)); fuse.load_state_dict(synth.recipe_state_dict(fuse))
T = 11
img = synth.synthetic_clip(T, 128, 160).cuda()
gt = synth.synthetic_mask(T, 128, 160, 1)
outs = []
for rep in range(3):                        # engines 2 and 3 run in the recycled (NaN-filled) buffers of their predecessors
    e = InferenceCore(prop, fuse, img, 1, mem_freq=3)
    m1 = e.interact(gt[:, 0], 0).copy()
    m2 = e.interact(gt[:, 6], 6).copy()     # a fused round: side-stream workspace too
    assert torch.isfinite(e.prob).all()
    outs.append((m1, m2, e.prob.clone()))
    del e
for m1, m2, p in outs[1:]:
    assert (m1 == outs[0][0]).all() and (m2 == outs[0][1]).all() and torch.equal(p, outs[0][2])
release_pooled_memory()
print("POOL-OK")
"""


def _random_multi_object_sessions():
    """Soak only (STCN_SOAK_MULTI=N [STCN_SOAK_SEED=S]; empty = skipped in the suite): k = 2..4 objects through the scribble / (k+1)-channel path at
    240x432 (the multi-object recipe is well-conditioned there), 2-3 interactions in any order - tools/parity_long.sh."""
    n, seed = int(os.environ.get("STCN_SOAK_MULTI", 0)), int(os.environ.get("STCN_SOAK_SEED", 1))
    rng = np.random.RandomState(1000 + seed)
    cases = []
    for _ in range(n):
        T, k, mf = int(rng.randint(6, 13)), int(rng.randint(2, 5)), int(rng.choice([1, 2, 3, 5]))
        H, W = (240, 432) if rng.rand() < 0.7 else (432, 240)
        rounds = tuple(int(v) for v in rng.choice(T, size=int(rng.randint(2, 4)), replace=False))
        cases.append((T, H, W, k, mf, rounds))
    return cases


@pytest.mark.parametrize("T,H,W,k,mf,rounds", _random_multi_object_sessions())
def test_random_multi_object_sessions_match_the_oracle(T, H, W, k, mf, rounds, nets_multi, weights_multi):
    """The k > 1 twin of test_random_annotation_sessions_match_the_oracle (decode groups of objects x frames, fusion per object,
    certain memory of several interactions): per object the clip and every frame against max(1e-3, 3 x the reference's own envelope
    under the multi-object recipe, 2 px / union px - 32 px once the oracle has met a true near-tie, see below)."""
    img, msk = synth.synthetic_clip(T, H, W, seed=51 + T), synth.synthetic_mask(T, H, W, k, seed=52 + T)
    core = make_core(nets_multi)(img, k, mf)
    orc = O.OracleCore(weights_multi[0], weights_multi[1], img, k, mem_freq=mf)
    n = load_golden("selfnoise")
    yard = np.max([n[t].max(0) for t in ("seq480k3", "seq480k5", "seq640k3")], 0)
    for r, idx in enumerate(rounds):
        m = torch.cat([1 - msk[:, idx].sum(0, keepdim=True).clamp(0, 1), msk[:, idx]], 0)
        a, b = core.interact(m, idx, scribble=True), orc.interact(m.clone(), idx, scribble=True)
        # objects are ~1000-4000 px at this size: once a read of the run had a query whose 50th / 51st scores are closer than one fp32
        # ulp of the scores (the oracle's order there is its rounding; this engine takes the exact order), the 16x16-pixel cell of such
        # a query may move: up to 32 px per frame (measured: 9 and 17 px on the two frames of 112 sessions that exceed 2 px / union,
        # max |dprob| 1.2e-2 there and 4e-6 on every other frame of those clips)
        tie = any(float(g.min()) < 1e-5 for _, _, g in orc.tie_log)
        masks_close(a, b, k, f"random k={k} T={T} {H}x{W} mf={mf} rounds={rounds} r{r}", yard, px_floor=32 if tie else 2)


def test_recycled_engine_buffers_carry_nothing_over():
    """Engine buffers come from a per-device pool (a destroyed engine's workspaces serve the next engine).  Under
    STCN_POOL_POISON=1 a recycled buffer arrives full of NaNs: three engines in a row must produce bit-identical, finite results."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STCN_POOL_POISON="1")
    r = subprocess.run([sys.executable, "-c", _POOL_SCRIPT % root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "POOL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _random_sessions():
    """Six seeded sessions in the suite; STCN_SOAK_SESSIONS=N [STCN_SOAK_SEED=S] draws N others for a one-off soak run (2 - 5 rounds,
    the same frame may be annotated twice, portrait sizes too) - tools/parity_long.sh."""
    n, seed = int(os.environ.get("STCN_SOAK_SESSIONS", 0)), int(os.environ.get("STCN_SOAK_SEED", 1))
    rng = np.random.RandomState(20260304 if not n else seed)
    cases = []
    for _ in range(n or 6):
        T = int(rng.randint(8, 25))
        mf = int(rng.choice([1, 2, 3, 5, 7]))
        H, W = int(rng.choice([112, 120, 136])), int(rng.choice([128, 150, 176]))
        if n and rng.rand() < 0.3:
            H, W = W, H
        if n and os.environ.get("STCN_SOAK_480"):          # the frame shapes MOSE / DAVIS store (REAL_SHAPES below), short clips
            H, W = [(854, 480), (853, 480), (480, 640), (480, 720), (480, 910), (480, 854)][int(rng.randint(0, 6))]
            T = int(rng.randint(8, 17))
        nr = int(rng.randint(2, 6)) if n else 3
        rounds = [int(v) for v in rng.choice(T, size=nr, replace=bool(n) and rng.rand() < 0.3)]
        cases.append((T, H, W, mf, tuple(rounds)))
    return cases


@pytest.mark.parametrize("T,H,W,mf,rounds", _random_sessions())
def test_random_annotation_sessions_match_the_oracle(T, H, W, mf, rounds, nets, weights):
    """Seeded sweep over what the fixed cases cannot enumerate: clip length, frame size (ragged pads), mem_freq and the ORDER of
    three interactions (first / last frames, neighbours, fused spans of any length) - HIP engine against the oracle after every
    round, same statements as the goldens (mask bounds + max-norm on the frames before the first near-tie)."""
    img = synth.synthetic_clip(T, H, W, seed=31 + T)
    msk = synth.synthetic_mask(T, H, W, 1, seed=32 + T)
    core = make_core(nets)(img, 1, mf)
    orc = O.OracleCore(weights[0], weights[1], img, 1, mem_freq=mf)
    # the six suite cases hold the plain 1e-3 per frame; a soak run (dozens of sessions) meets frames whose read has a query with a
    # 50th-51st score gap of ~2e-6 - there the oracle's fp32 order and the exact order differ and ~20 pixels of a 120x150 frame
    # follow (the reference does the same against itself: selfnoise rows of the small k = 1 fixtures, worst frame 1.05e-3): 3 x that
    yard = None
    if os.environ.get("STCN_SOAK_SESSIONS"):
        n = load_golden("selfnoise")
        yard = np.max([n[t].max(0) for t in (("seq480", "seq480L", "seq480P") if max(H, W) >= 480 else ("seqA", "seqA1", "seqB", "seqE"))], 0)
    frames = 0
    for r, idx in enumerate(rounds):
        a, b = core.interact(msk[:, idx], idx), orc.interact(msk[:, idx], idx)
        tag = f"random T={T} {H}x{W} mf={mf} rounds={rounds}"
        masks_close(a, b, 1, f"{tag} r{r}", yard)
        clean_frame_check(core.prob.cpu(), orc, r, tag)
        frames += core.stats()["frames"]
    assert frames > 0
