"""Multi-GPU sampling: one process per GPU, scenes sharded, no collective inside the DDIM loop.

Mirrors how the reference's FID generator distributes work (perception/data_prepare/val_set_gen.py:71-87,
130-161: `accelerator.prepare(dataloader)` shards batches; `all_gather_object` collects results once per
batch).  Cross-view attention couples only the 6 views of ONE scene, and the CFG halves of a scene stay
together, so scenes are independent units: rank r takes scenes r, r+W, r+2W, ... and the only exchange is
the final gather of the result latents (67 kB / scene fp16) over RCCL (xGMI: every rank has a direct link
to every other, so a single all_gather of <1 MB is latency-bound; no ring tuning needed).
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; single process if unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_scenes(n_scenes: int, rank: int, world: int) -> List[int]:
    """Scene indices of this rank: i -> rank i mod world (what accelerate's dataloader sharding does)."""
    return list(range(rank, n_scenes, world))


def gather_scene_results(local: torch.Tensor, n_scenes: int, rank: int, world: int) -> torch.Tensor:
    """local: [n_local, ...] results of shard_scenes(n_scenes, rank, world), in that order.
    Returns [n_scenes, ...] in global scene order on every rank (one all_gather)."""
    if world == 1:
        return local
    per = (n_scenes + world - 1) // world
    pad = torch.zeros(per, *local.shape[1:], dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty(world * per, *local.shape[1:], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad)
    buf = buf.view(world, per, *local.shape[1:])
    out = torch.empty(n_scenes, *local.shape[1:], dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_scenes(n_scenes, r, world)
        if idx:
            out[torch.tensor(idx, device=local.device)] = buf[r, : len(idx)]
    return out


def device_identity(device) -> str:
    """The PHYSICAL identity of a GPU: its PCI location (domain:bus:device), else its uuid; "" when the runtime reports neither.
    No launcher-assigned index in it: under per-rank HIP_VISIBLE_DEVICES isolation every rank sees index 0, and an index is not a
    physical identity anyway."""
    pr = torch.cuda.get_device_properties(device)
    bus = getattr(pr, "pci_bus_id", None)
    if bus is not None and str(bus) != "":
        return f"pci:{getattr(pr, 'pci_domain_id', 0)}:{bus}:{getattr(pr, 'pci_device_id', 0)}"
    uuid = str(getattr(pr, "uuid", "") or "")
    return f"uuid:{uuid}" if uuid.strip("0-") else ""


def assert_distinct_devices(device, rank: int, world: int) -> int:
    """Every rank must drive its own GPU: gathers device_identity() over the group and returns the number of distinct devices
    (== world, asserted).  A launcher that put two ranks on one GPU would otherwise print a whole-job number for hardware it did not use.
    When the runtime exposes no physical identity at all the check cannot be made: warn and trust the launcher."""
    if world == 1:
        return 1
    ids: List[str] = [None] * world           # type: ignore[list-item]
    dist.all_gather_object(ids, device_identity(device))
    if any(i == "" for i in ids):
        import warnings
        warnings.warn(f"no PCI id / uuid reported for some devices ({ids}): distinct-device check skipped")
        return world
    n = len(set(ids))
    assert n == world, f"{world} ranks on {n} distinct GPUs: {ids}"
    return n


def shutdown():
    """Leave the process group on EVERY rank at the same point: barrier, then destroy (val_set_gen.py:149-160 ends its distributed part
    the same way before rank 0 writes results).  No-op for a single process."""
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
