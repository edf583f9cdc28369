"""CPU: the parity bounds coded in bench.py (`bound` / `within_bound` of every parity leg of the driver's line) are the suite's bounds
(tests/conftest.py::frame_bound) over the committed self-noise of the reference (tests/golden/selfnoise.npz)."""
import numpy as np

from conftest import frame_bound, load_golden


def test_bench_bounds_are_the_suite_bounds():
    import bench
    noise = load_golden("selfnoise")
    k1 = bench.ref_self_noise("seq480", "seq480L", "seq480P")
    assert np.array_equal(k1, np.max(np.concatenate([noise["seq480"], noise["seq480L"], noise["seq480P"]], 0), 0))
    for px in (10, 671, 5000, 200000):
        assert bench.frame_bound(k1, px) == frame_bound(k1[4], px)
    assert bench.frame_bound(k1, 200000) == max(1e-3, 1.5 * k1[4]) == 1e-3 and bench.frame_bound(k1, 100) == 2.0 / 100
    k5 = bench.ref_self_noise("seq480k5", "seq480k3", "seq640k3")
    assert bench.frame_bound(k5, 200000) == 1.5 * k5[4] > 1e-3, "k > 1 at 480p: the reference's own per-frame spread (2.0e-3) exceeds the plain bound"
    # the clip bound is the north_star's plain 1e-3 for every k (the suite's clip_bound() agrees wherever the reference's own spread allows it)
    from conftest import clip_bound
    assert bench.clip_bound(k1) == bench.clip_bound(k5) == 1e-3 == clip_bound(k1[0]) == clip_bound(k5[0])
