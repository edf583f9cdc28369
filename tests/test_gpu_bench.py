"""GPU: bench.py's launch contract.  `python bench.py --gpus N` must start N ranks by itself (the driver's 1-GPU form has no
torch.distributed.run around it); on this 1-GPU box both ranks share device 0 and the collectives run over gloo
(STCN_BENCH_DEVICE / STCN_BENCH_BACKEND exist for exactly this test).  Reference sharding being mirrored: one process
per --min-idx/--max-idx slice (eval_annotation_method.py:34-35,118-119)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SMALL = ["--steps", "2", "--warmup", "0", "--frames", "12", "--height", "240", "--width", "432", "--streams", "1",
         "--no-profile", "--no-r2", "--cpu-frames", "0", "--no-config3", "--no-memread-roofline", "--no-davis-val", "--no-drivers", "--no-session", "--no-power"]


def run_bench(args, env_extra=None):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_flag_launches_that_many_ranks():
    one = run_bench(["--gpus", "1"] + SMALL)
    assert one["n_gpus"] == 1 and one["value"] > 0 and len(one["jf_rows_rank_J_F_JF"]) == 1
    assert one["config"]["sharding"] == "videos x1"
    # the record the driver keeps: three back-to-back timed regions (value = the first), what the lanes cost the host
    rep = one["value_repeats"]
    assert len(rep["frames_per_s"]) == 3 and abs(rep["frames_per_s"][0] - one["value"]) < 0.01 and rep["min"] <= rep["median"] <= rep["max"]
    assert one["host_enqueue_ms_per_video"] > 0 and len(one["host_cpu_s_per_lane"]) == 1 and one["host_cores"] >= 1
    assert one["power"] is None                                   # --no-power in SMALL
    # the power leg: socket power of one more timed region (rocm-smi; null only where the tool is missing)
    import shutil
    pw = run_bench(["--gpus", "1"] + [a for a in SMALL if a != "--no-power"])["power"]
    if shutil.which("rocm-smi"):
        assert pw and pw["samples"] >= 1 and 50 < pw["socket_w_median"] <= pw["socket_w_max"] < 3000 and pw["cap_w"] > 100, pw
    two = run_bench(["--gpus", "2"] + SMALL, {"STCN_BENCH_DEVICE": "0", "STCN_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["steps"] == 2
    rows = two["jf_rows_rank_J_F_JF"]
    assert [int(r[0]) for r in rows] == [0, 1], "one gathered J&F row per rank"
    # same clip, same weights on both ranks -> identical J&F; and the same as the single-rank run
    assert rows[0][1:] == rows[1][1:] == one["jf_rows_rank_J_F_JF"][0][1:]
    # whole-job value = frames of ALL ranks / max time: 2 ranks x 2 videos x 11 frames
    frames = two["value"] * two["ms_per_step"] * 1e-3 * two["steps"]
    assert abs(frames - 2 * 2 * 11) < 1e-6 * frames + 1e-3
    assert len(two["value_repeats"]["frames_per_s"]) == 3 and abs(two["value_repeats"]["frames_per_s"][0] - two["value"]) < 0.01
    assert two["cpu_baseline"] is None and "rank 0 at N=1" in two["cpu_baseline_note"]
    assert two["concurrent_videos_bit_identical"]


def test_davis_val_workload_is_sharded_by_lpt_over_the_ranks():
    """--workload davis-val: the whole job runs --steps samples with the DAVIS-2017-val lengths (clamped here so that the
    test stays small), assigned to the ranks by LPT; fixed total work -> "strong"; every sample is processed exactly once."""
    args = [a for a in SMALL if a != "--no-davis-val"]
    args[args.index("--steps") + 1] = "7"
    args[args.index("--streams") + 1] = "2"
    args += ["--workload", "davis-val", "--davis-max-frames", "9"]
    lens = [min(t, 9) for t in [69, 50, 80, 84, 90, 75, 40]]
    one = run_bench(["--gpus", "1"] + args)
    dv = one["davis_val"]
    assert one["scaling"] == "strong" and one["steps"] == 7 and dv["samples"] == 7
    assert dv["frames_total"] == sum(t - 1 for t in lens) and dv["rank_samples"] == [7]
    assert abs(one["value"] - dv["frames_per_s"]) < 1e-9
    two = run_bench(["--gpus", "2"] + args, {"STCN_BENCH_DEVICE": "0", "STCN_BENCH_BACKEND": "gloo"})
    dv = two["davis_val"]
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert dv["frames_total"] == sum(t - 1 for t in lens), "every sample exactly once over the two ranks"
    assert sorted(dv["rank_samples"]) == [3, 4] and len(dv["rank_seconds"]) == 2
    assert sorted(sum(dv["lengths_by_rank"], [])) == sorted(lens)
    assert max(dv["rank_busy_fraction"]) == 1.0 and dv["imbalance_max_over_mean_frames"] < 1.2


TINY = ["--warmup", "0", "--frames", "5", "--height", "128", "--width", "160", "--streams", "1", "--no-profile", "--no-r2",
        "--cpu-frames", "0", "--no-config3", "--no-memread-roofline", "--no-drivers", "--no-session", "--no-power"]


def test_eight_rank_preflight_on_one_device():
    """The driver's first real `bench.py --gpus 8` must not fail on rendezvous, row width or an empty rank: both workloads with
    EIGHT ranks over gloo on this box's one device at a tiny size (eval_annotation_method.py:34-35,113-119 shards the same way
    by hand).  Uniform: 8 gathered J&F rows, ranks 0..7, 8 x the frames.  davis-val: the 30 samples once each, no rank empty,
    and the LPT assignment of the TRUE lengths (34..104 frames) balanced to 1 % over 8 ranks."""
    env = {"STCN_BENCH_DEVICE": "0", "STCN_BENCH_BACKEND": "gloo"}
    uni = run_bench(["--gpus", "8", "--steps", "1", "--no-davis-val"] + TINY, env)
    assert uni["n_gpus"] == 8 and uni["scaling"] == "weak"
    rows = uni["jf_rows_rank_J_F_JF"]
    assert [int(r[0]) for r in rows] == list(range(8)) and all(len(r) == 4 for r in rows)
    assert all(r[1:] == rows[0][1:] for r in rows), "same clip and weights on every rank"
    frames = uni["value"] * uni["ms_per_step"] * 1e-3 * uni["steps"]
    assert abs(frames - 8 * 1 * 4) < 1e-3
    dvl = run_bench(["--gpus", "8", "--steps", "30", "--workload", "davis-val", "--davis-max-frames", "4"] + TINY, env)
    dv = dvl["davis_val"]
    assert dvl["n_gpus"] == 8 and dvl["scaling"] == "strong" and dv["samples"] == 30
    assert dv["frames_total"] == 30 * 3 and sum(dv["rank_samples"]) == 30 and min(dv["rank_samples"]) >= 1
    assert len(dv["rank_seconds"]) == 8 and max(dv["rank_busy_fraction"]) == 1.0
    from eva_vos_amd import shard
    sys.path.insert(0, ROOT)
    import bench
    lens = [t - 1 for t in bench.DAVIS_VAL_LENGTHS]
    loads = [sum(lens[i] for i in part) for part in shard.lpt_assign(lens, 8)]
    assert sorted(i for part in shard.lpt_assign(lens, 8) for i in part) == list(range(30))
    assert max(loads) / (sum(loads) / 8) <= 1.01, loads


def _run_driver(module, args, world, cwd):
    import socket
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if world == 1:
        cmd = [sys.executable, "-m", module] + args
    else:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env.update(STCN_DIST_BACKEND="gloo", STCN_DIST_DEVICE="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "-m", module] + args
    p = subprocess.run(cmd, env=env, cwd=cwd, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    return p.stdout


def test_eight_rank_preflight_of_the_drivers(tmp_path):
    """Configs 4 / 5 are `fq_driver` / `eval_driver` sharded over the GPUs of a node (the reference: one process per --min-idx/--max-idx
    slice, eval_annotation_method.py:34-35,113-119, CSV written at the end :188-191).  Eight ranks over gloo on this box's one device on a
    tiny tree of 7 samples: one rank stays EMPTY, rows are gathered at a fixed width, rank 0 alone writes the CSV - and the files equal
    the single-process run's byte for byte (the engine is deterministic; which rank propagated a sample must not matter)."""
    from eva_vos_amd import fq_driver
    db = str(tmp_path / "db")
    imset = fq_driver.make_synthetic_tree(db, {"a": (6, 112, 128, 2), "b": (5, 112, 128, 1), "c": (7, 128, 112, 2), "d": (4, 112, 144, 1), "e": (5, 112, 128, 1)})
    outs = {}
    for world in (1, 8):
        cwd = tmp_path / f"w{world}"
        cwd.mkdir()
        so = _run_driver("eva_vos_amd.fq_driver", ["--root", db, "--imset", imset, "--out", str(cwd / "fq"), "--synthetic-weights", "--rounds", "3", "--lanes", "2"], world, str(cwd))
        assert so.count("states ->") == 1, "only rank 0 reports / writes"
        se = _run_driver("eva_vos_amd.eval_driver", ["--root", db, "--imset", imset, "--synthetic-weights", "--rounds", "3", "--lanes", "2", "--db", "T"], world, str(cwd))
        assert se.count("rounds ->") == 1
        outs[world] = (open(cwd / "fq" / "res_synthetic.csv").read(), open(cwd / "Experiments" / "T" / "oracle_mask.csv").read(),
                       sorted(os.listdir(cwd / "fq" / "Annotations" / "224")))
    assert outs[1][0].count("\n") > 7 and outs[1][1].count("\n") > 7
    assert outs[8] == outs[1], "8 ranks (one of them empty) must write what one process writes"


def test_real_data_hook_is_taken_when_checkpoints_and_clips_exist(tmp_path):
    """eval_annotation_method.py:51-64 loads ./model_weights/mivos/{stcn,fusion}.pth and ./data/DAVIS_17: when both exist
    bench.py measures on real clips and says so.  Here: the recipe weights saved as checkpoints + a synthetic tree in the
    DAVIS layout stand in for them (there is no network for the real ones)."""
    import torch
    from eva_vos_amd import fq_driver, synth
    from eva_vos_amd.params import FusionNet, PropagationNetwork
    wdir = tmp_path / "weights"
    wdir.mkdir()
    torch.save(synth.recipe_state_dict(PropagationNetwork()), wdir / "stcn.pth")
    torch.save(synth.recipe_state_dict(FusionNet()), wdir / "fusion.pth")
    root = tmp_path / "trainval"
    imset = fq_driver.make_synthetic_tree(str(root), {"bear": (9, 240, 432, 1)})
    os.makedirs(root / "ImageSets" / "2017", exist_ok=True)
    os.replace(imset, root / "ImageSets" / "2017" / "val.txt")
    env = {"STCN_BENCH_WEIGHTS": str(wdir), "STCN_BENCH_DAVIS": str(root)}
    flags = [a for a in SMALL if a not in ("--frames", "12", "--height", "240", "--width", "432")]
    real = run_bench(["--gpus", "1"] + flags, env)
    assert real["data"] == "real" and "bear__1" in real["config"]["workload"] and "stcn.pth" in real["config"]["weights"]
    assert real["config"]["frames_per_step"] == 8 and real["value"] > 0
    synth_line = run_bench(["--gpus", "1", "--data", "synthetic"] + SMALL, env)
    assert synth_line["data"] == "synthetic"
