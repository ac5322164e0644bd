"""CPU, world_size 2 over gloo: scene sharding covers every scene once and the single result gather restores
global scene order (the N>1 path of bench.py / SURVEY.md §8e)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from magicdrive_amd import distributed as DD


def test_shard_is_a_partition():
    for n, w in [(8, 2), (7, 2), (5, 4), (3, 8), (16, 8)]:
        got = sorted(i for r in range(w) for i in DD.shard_scenes(n, r, w))
        assert got == list(range(n))


def _worker(rank, world, port, n_scenes, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = DD.init_from_env(backend="gloo")
    mine = DD.shard_scenes(n_scenes, r, w)
    local = torch.stack([torch.full((6, 4, 2, 3), float(i)) for i in mine]) if mine else torch.zeros(0, 6, 4, 2, 3)
    out = DD.gather_scene_results(local, n_scenes, r, w)
    t = DD.max_over_ranks(float(rank + 1), torch.device("cpu"))
    DD.barrier()
    q.put((rank, out[:, 0, 0, 0, 0].tolist(), t))
    dist.destroy_process_group()


def test_gather_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_scenes = 5
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_scenes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, vals, t in res:
        assert vals == [float(i) for i in range(n_scenes)], (rank, vals)
        assert t == 2.0


def test_bench_self_launch_command():
    """`python bench.py --gpus 8` (no WORLD_SIZE) must re-execute itself as 8 ranks, not measure one GPU and call it eight."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sp = importlib.util.spec_from_file_location("mdx_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(sp); sp.loader.exec_module(bench)
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "3"], 12345)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "3"] and cmd[-5].endswith("bench.py")
    src = open(os.path.join(root, "bench.py")).read()
    assert "assert world == args.gpus," in src, "a WORLD_SIZE / --gpus mismatch must abort"


def _nccl_worker(rank, world, port, n_scenes, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    r, w, local = DD.init_from_env(backend="nccl")
    dev = torch.device("cuda", local)
    n_dev = DD.assert_distinct_devices(dev, r, w)
    mine = DD.shard_scenes(n_scenes, r, w)
    g = torch.Generator().manual_seed(5)
    full = torch.randn(n_scenes, 6, 4, 7, 13, generator=g)                 # what ONE rank would have produced for all scenes
    out = DD.gather_scene_results(full[mine].to(dev), n_scenes, r, w)
    t = DD.max_over_ranks(float(rank + 1), dev)
    DD.barrier(); torch.cuda.synchronize()
    q.put((rank, bool(torch.equal(out.cpu(), full)), t, n_dev))
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_gather_two_ranks_nccl():
    """The RCCL path itself (the only collective of the N > 1 bench): 2 ranks on 2 GPUs, result order and bit-equality with the
    1-rank result.  Skipped on a 1-GPU box."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, t, n_dev in res:
        assert same and t == 2.0 and n_dev == 2, (rank, same, t, n_dev)


def test_bench_two_ranks_gloo_prints_one_json_line(tmp_path):
    """`bench.py --gpus 2` end to end under torch.distributed.run with the gloo backend and a stubbed pipeline
    (tests/bench_stub.py through bench.py's --pipe-factory hook): both ranks exit 0, exactly ONE JSON line is printed (by rank 0, after
    every rank has left the process group), it claims 2 ranks and the whole-job rate, and the side measurements are skipped for N > 1."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--scenes-per-gpu", "3", "--pipe-factory", "bench_stub:make"]
    cmd = bench.launcher_command(2, argv, port)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, os.path.join(root, "tests"), os.environ.get("PYTHONPATH", "")]),
               CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["scenes_per_gpu"] == 3 and out["config"]["parallelism"] == "scene-sharded x2"
    assert out["value"] > 0 and abs(out["value"] - 6 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]      # whole-job scenes / max-over-ranks time
    assert out["config"]["full_cond_scenes_per_s"] is None and out["config"]["hires"] is None and "cpu_baseline" not in out


def _run_sample_driver(tmp_path, world, out_name, extra=()):
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, os.path.join(root, "tests"), os.environ.get("PYTHONPATH", "")]),
               CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="2")
    argv = [os.path.join(root, "tools", "sample.py"), "--ckpt", str(tmp_path / "ckpt"), "--sd15", str(tmp_path / "sd15"), "--data", str(tmp_path / "data"),
            "--out", str(tmp_path / out_name), "--prompt-embeds", "--pipe-factory", "sample_stub:make", "--dist-backend", "gloo", "--device", "cpu",
            "runner.validation_times=2", *extra]
    if world == 1:
        cmd = [sys.executable] + argv
    else:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + argv
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r


def test_sample_driver_two_ranks_gloo(tmp_path):
    """`torchrun tools/sample.py` = the reference's val_set_gen.py flow (perception/data_prepare/val_set_gen.py:71-161): batch j ->
    rank j mod 2, one all_gather_object per batch, every file written exactly once (by its rank, or — with --gather-images — by rank 0), rank 0's
    index names rank and seed of every scene.  Five scenes in batches of two: rank 0 takes batches 0 and 2, rank 1 batch 1 and an EMPTY third round
    (it still has to join the collective).
    Seeding: default = `manual_seed(cfg.seed)` per batch on EVERY rank (magicdrive/misc/test_utils.py:233-238): a scene's pictures must
    not depend on the world size or on the rank its batch lands on; runner.validation_seed_global=true = one generator per rank seeded seed + rank
    before the loop, a local seed drawn per batch (val_set_gen.py:83-87, test_utils.py:184-188, 224-232)."""
    import json
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "sample_preprocess.pt"), weights_only=False)
    os.makedirs(tmp_path / "data"); os.makedirs(tmp_path / "ckpt" / "hydra"); os.makedirs(tmp_path / "sd15")
    for i, case in enumerate(gold["cases"]):
        torch.save(case["sample"], tmp_path / "data" / f"tok{i}.pth")
    with open(tmp_path / "ckpt" / "hydra" / "overrides.yaml", "w") as f:
        f.write("- +exp=224x400\n- seed=7\n")
    _run_sample_driver(tmp_path, 1, "one", ["--batch-size", "2"])
    _run_sample_driver(tmp_path, 2, "two", ["--batch-size", "2"])
    _run_sample_driver(tmp_path, 2, "two_gathered", ["--batch-size", "2", "--gather-images"])
    _run_sample_driver(tmp_path, 1, "one_global", ["runner.validation_seed_global=true", "--batch-size", "2"])
    _run_sample_driver(tmp_path, 2, "two_global", ["runner.validation_seed_global=true", "--batch-size", "2"])
    dirs = ("one", "two", "two_gathered", "one_global", "two_global")
    names = {d: sorted(x for x in os.listdir(tmp_path / d) if x.endswith(".png")) for d in dirs}
    assert len(names["one"]) == 5 * 2 * 6 and all(names[d] == names["one"] for d in dirs)      # global scene index in the name: each file once
    idx = {d: json.load(open(tmp_path / d / "index.json")) for d in names}
    assert idx["one"]["world"] == 1 and idx["two"]["world"] == 2 and idx["two_gathered"]["gather_images"] is True
    for d in ("two", "two_gathered", "two_global"):
        gens = idx[d]["generations"]
        assert [(g["scene"], g["gen"]) for g in gens] == [(s, t) for s in range(5) for t in range(2)]
        for g in gens:
            want_rank = (g["scene"] // 2) % 2
            want_seed = 7 + want_rank if d == "two_global" else 7             # the rank offset exists only in the global-generator branch
            assert g["rank"] == want_rank and g["seed"] == want_seed and len(g["files"]) == 6, g
            assert all(os.path.exists(tmp_path / d / f) for f in g["files"])
    assert all(g["rank"] == 0 and g["seed"] == 7 for g in idx["one"]["generations"])
    from PIL import Image
    import numpy as np
    px = lambda d, n: np.asarray(Image.open(tmp_path / d / n))
    # default mode: world = 1 and world = 2 produce the same picture for EVERY scene and generation, in both exchange modes
    for n in names["one"]:
        assert np.array_equal(px("one", n), px("two", n)), n
        assert np.array_equal(px("two", n), px("two_gathered", n)), n
    # global-generator mode: rank 0's first batch draws the same local seed as the one-process run (same generator, same position); rank 1's
    # batch comes from generator(seed + 1) instead of the second draw of generator(seed): its pictures differ; and the mode differs from the default
    assert np.array_equal(px("one_global", "0_gen0_view3.png"), px("two_global", "0_gen0_view3.png"))
    assert not np.array_equal(px("one_global", "2_gen0_view3.png"), px("two_global", "2_gen0_view3.png"))
    assert not np.array_equal(px("one", "0_gen0_view3.png"), px("one_global", "0_gen0_view3.png"))
