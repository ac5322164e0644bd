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
