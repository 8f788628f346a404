"""Seed-parallel multi-GPU launch: one process per GPU, ONE broadcast of the packed weight arena, no per-step
collectives.  The reference has no distributed code at all (batch is hard-wired to 1,
models/region_diffusion_sdxl.py:698-701); independent (rich-text JSON, seed) requests are the shardable unit.

`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI); the same functions run on "gloo" for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run contract)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_round_robin(requests, rank, world):
    """Static round-robin of independent requests over ranks (request i -> rank i % world)."""
    return [r for i, r in enumerate(requests) if i % world == rank]


class _DevicePointer:
    """Zero-copy view of engine-owned device memory for torch (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def arena_tensor(engine):
    ptr, nbytes = engine.arena()
    return torch.as_tensor(_DevicePointer(ptr, nbytes), device=f"cuda:{engine.device}")


def broadcast_tensor(t, src=0, chunk_bytes=1 << 30):
    """Broadcast a flat byte tensor in <= 1 GiB pieces (xGMI is per-link bound; a few large messages are ideal)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    flat = t.view(-1)
    n = 0
    for off in range(0, flat.numel(), chunk_bytes):
        dist.broadcast(flat[off:off + chunk_bytes], src=src)
        n += 1
    return n


def broadcast_weights(engine, src=0):
    """Rank `src` has bound (packed) all weights; every other rank receives the packed bf16 arena."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0.0
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engine.synchronize()
    broadcast_tensor(arena_tensor(engine), src)
    torch.cuda.synchronize()
    if dist.get_rank() != src:
        engine.arena_mark_bound()
    return time.perf_counter() - t0


def max_over_ranks(value, device="cpu"):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
