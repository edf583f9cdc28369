"""CPU: per-video sharding (world_size 2 over gloo) and the J / F metrics."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eva_vos_amd import metrics, shard


def test_lpt_is_balanced_and_deterministic():
    costs = [104, 34, 50, 82, 66, 40, 90, 71, 36, 59]
    a = shard.lpt_assign(costs, 4)
    assert a == shard.lpt_assign(costs, 4)
    assert sorted(i for r in a for i in r) == list(range(len(costs)))
    loads = [sum(costs[i] for i in r) for r in a]
    assert max(loads) - min(loads) <= max(costs) // 2
    assert shard.lpt_assign(costs, 1) == [sorted(range(len(costs)), key=lambda i: (-costs[i], i))]
    # the 30 DAVIS-2017-val sequence lengths (what bench.py --workload davis-val shards): within 1 % of the mean on 2, 4, 8 ranks
    davis = [69, 50, 80, 84, 90, 75, 40, 104, 90, 60, 66, 52, 50, 90, 78, 50, 81, 34, 50, 47, 49, 50, 79, 40, 80, 100, 79, 43, 40, 99]
    for world in (2, 4, 8):
        parts = shard.lpt_assign([t - 1 for t in davis], world)
        assert sorted(i for r in parts for i in r) == list(range(30))
        loads = [sum(davis[i] - 1 for i in r) for r in parts]
        assert max(loads) <= 1.01 * sum(loads) / world, (world, loads)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    costs = [10, 3, 7, 5, 9]
    mine = shard.lpt_assign(costs, world)[rank]
    rows = np.array([[i, costs[i] * 0.5, rank] for i in mine], np.float32)
    out = shard.gather_rows(rows, 3)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_rows_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for r in range(2):
        assert got[r][:, 0].tolist() == [0, 1, 2, 3, 4]          # ordered by sample id on every rank
        assert np.allclose(got[r][:, 1], [5, 1.5, 3.5, 2.5, 4.5])
    assert np.array_equal(got[0], got[1])
    assert set(got[0][:, 2].tolist()) == {0.0, 1.0}              # both ranks contributed


def test_gather_rows_single_process_sorts():
    out = shard.gather_rows(np.array([[3, 1], [1, 2], [2, 3]], np.float32), 2)
    assert out[:, 0].tolist() == [1, 2, 3]


def test_j_and_f_known_answers():
    gt = np.zeros((40, 60), bool)
    gt[10:30, 20:40] = True
    assert metrics.jaccard(gt, gt) == 1.0 and metrics.f_measure(gt, gt) == 1.0
    assert metrics.j_and_f(gt, gt) == 1.0
    empty = np.zeros_like(gt)
    assert metrics.jaccard(gt, empty) == 0.0 and metrics.f_measure(gt, empty) == 0.0     # n_fg == 0, n_gt > 0
    assert metrics.f_measure(empty, empty) == 1.0 and metrics.jaccard(empty, empty) == 0.0
    shifted = np.roll(gt, 1, axis=1)                                                     # within the 1-px tolerance
    assert metrics.f_measure(gt, shifted) == 1.0
    assert abs(metrics.jaccard(gt, shifted) - (19 * 20) / (21 * 20)) < 1e-12
    far = np.roll(gt, 12, axis=1)
    assert metrics.f_measure(gt, far) < 0.7
    b = metrics.boundary_map(gt)
    assert b.sum() == 2 * 20 + 2 * 20 and b[9, 19] and not b[10, 20]                     # half-pixel offset to origin
    rows = metrics.sequence_scores(np.stack([gt, gt]), np.stack([gt, empty]))
    assert rows.shape == (2, 4) and rows[0, 3] == 1.0 and rows[1, 3] == 0.0


def test_clip_dataset_layout_and_normalisation(tmp_path):
    """CPU: the loader side of the FQ driver (reference datasets/annotation_dataset.py:80-132): one sample per
    (video, object), ImageNet normalisation, GT without a background channel, decode cache per video."""
    from eva_vos_amd import fq_driver
    imset = fq_driver.make_synthetic_tree(str(tmp_path), {"vidA": (4, 48, 64, 2), "vidB": (3, 48, 64, 1)})
    ds = fq_driver.ClipDataset(str(tmp_path), imset)
    assert [ds.name(i) for i in range(len(ds))] == ["vidA__1", "vidA__2", "vidB__1"]
    s0, s1, s2 = ds[0], ds[1], ds[2]
    assert tuple(s0["rgb"].shape) == (1, 4, 3, 48, 64) and tuple(s0["gt"].shape) == (1, 4, 1, 48, 64)
    assert s0["rgb"].data_ptr() == s1["rgb"].data_ptr(), "objects of one video share the decoded clip"
    assert tuple(s2["rgb"].shape) == (1, 3, 3, 48, 64)
    assert set(np.unique(s0["gt"].numpy())) <= {0.0, 1.0} and s0["gt"].sum() > 0 and s1["gt"].sum() > 0
    assert float((s0["gt"] * s1["gt"]).sum()) == 0.0, "objects are disjoint"
    px = s0["rgb"][0, 0, :, 5, 7].numpy() * fq_driver.STD + fq_driver.MEAN          # undo the normalisation
    assert np.all(px >= -1e-6) and np.all(px <= 1 + 1e-6)


def test_driver_lanes_cover_every_sample_once(tmp_path):
    """run_lanes: contiguous chunks per lane (objects of one video stay with one loader), every sample exactly once,
    loader + prefetcher work without a GPU."""
    from eva_vos_amd import fq_driver
    imset = fq_driver.make_synthetic_tree(str(tmp_path / "db"), {"a": (3, 48, 64, 2), "b": (4, 48, 64, 1), "c": (2, 48, 64, 3)})
    ds = fq_driver.ClipDataset(str(tmp_path / "db"), imset)
    assert [ds.name(i) for i in range(len(ds))] == ["a__1", "a__2", "b__1", "c__1", "c__2", "c__3"]
    seen = []

    def work(i, sample):
        assert sample["rgb"].shape == (1, sample["num_frames"], 3, 48, 64) and sample["gt"].shape[1] == sample["num_frames"]
        assert set(np.unique(sample["gt"].numpy())) <= {0.0, 1.0}
        seen.append(i)
        return [np.array([i, sample["num_frames"]], np.float32)]

    for lanes in (1, 2, 4, 9):
        seen.clear()
        rows = fq_driver.run_lanes(str(tmp_path / "db"), imset, range(len(ds)), lanes, work, device="cpu")
        assert sorted(seen) == list(range(6)) and sorted(int(r[0]) for r in rows) == list(range(6))
        assert {int(r[0]): int(r[1]) for r in rows} == {0: 3, 1: 3, 2: 4, 3: 2, 4: 2, 5: 2}
