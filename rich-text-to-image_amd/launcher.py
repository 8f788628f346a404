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
        import datetime
        # a finite rendezvous / collective timeout: a mis-set IPC mode shows up as an error after minutes, not as a hang until the
        # driver's own limit (RTDIFF_DIST_TIMEOUT_S overrides)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("RTDIFF_DIST_TIMEOUT_S", "300"))))
    return rank, local_rank, world


def collective_self_check(nbytes=64 << 20, device=None):
    """Runs at the top of every N > 1 launch (bench.py --gpus N, sample.py --gpus N), BEFORE the 5 GB arena moves: rank 0 broadcasts a
    position-dependent 64 MB pattern on the launch's real backend (RCCL on GPUs), every rank verifies every byte, the verdicts are
    all-reduced, and rank 0 gets a record for the JSON line (backend, RCCL version, the IPC / debug environment, broadcast seconds).
    The 2-rank RCCL path had never executed on hardware when this was written (1-GPU leases in every round): the first 8-GPU run must
    fail LOUDLY - with the environment that matters in the message - instead of hanging or sampling from a half-received arena."""
    import time
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return {"world": 1, "ok": True}
    rank, world, backend = dist.get_rank(), dist.get_world_size(), dist.get_backend()
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    n = max(8, nbytes // 8)
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    pattern = (idx * 2654435761 + 12345) ^ (idx >> 7)                       # position-dependent: a shifted or truncated payload cannot pass
    buf = pattern.clone() if rank == 0 else torch.zeros(n, dtype=torch.int64, device=dev)
    env = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_P2P_DISABLE", "RCCL_MSCCL_ENABLE", "HIP_VISIBLE_DEVICES")}
    t0 = time.perf_counter()
    try:
        dist.broadcast(buf, src=0)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
    except Exception as e:                                                  # noqa: BLE001 - re-raised with the context a 3 a.m. reader needs
        raise RuntimeError(f"collective self-check: broadcast of {n * 8} bytes failed on rank {rank}/{world} (backend {backend}, env {env}): {e}") from e
    dt = time.perf_counter() - t0
    ok_local = bool(torch.equal(buf, pattern))
    flag = torch.tensor([1 if ok_local else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ok = bool(flag.item() == 1)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                                   # noqa: BLE001
            ver = "unknown"
    rec = {"world": world, "backend": backend, "rccl_version": ver, "bytes": n * 8, "broadcast_s": dt, "ok": ok, "env": env}
    if not ok:
        raise RuntimeError(f"collective self-check: rank {rank}/{world} {'received a WRONG payload' if not ok_local else 'is fine but another rank is not'} "
                           f"({rec}); for RCCL on this host driver HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC) is required")
    return rec


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(script, argv, nproc, require_gpus=True, module=None):
    """`python bench.py --gpus N` without a wrapper: when no torch.distributed.run environment is present (WORLD_SIZE unset) and
    N > 1, replace this process by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P script argv...` - one rank per GPU, LOCAL_RANK selects the device.  Returns None when nothing has to be done
    (N == 1 or already inside a launch); returns an error string when the box has fewer than N GPUs (the caller reports it);
    otherwise does not return.  `module`: launch `python -m <module>` instead of a script path."""
    import sys
    if nproc <= 1 or "WORLD_SIZE" in os.environ:
        return None
    if require_gpus and "RTDIFF_FORCE_DEVICE" not in os.environ:      # (RTDIFF_FORCE_DEVICE: every rank on one device - tests only)
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < nproc:
            return f"need {nproc} GPUs on this node for --gpus {nproc}, found {have}"
    env = dict(os.environ)
    # dmabuf IPC: the image's documentation says RCCL / cross-process device-memory sharing needs it on this host driver.  NOT validated
    # here: every GPU lease of rounds 1-5 had ONE device, the 2-rank RCCL path has never executed.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    # module: `python -m pkg.mod` programs (sample.py uses relative imports) are launched as torchrun's --module form
    target = ["--module", module] if module else [script]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(env.get("MASTER_PORT") or free_port())] + target + list(argv)
    sys.stdout.flush(); sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def shard_round_robin(requests, rank, world):
    """Static round-robin of independent requests over ranks (request i -> rank i % world)."""
    return [r for i, r in enumerate(requests) if i % world == rank]


class _DevicePointer:
    """Zero-copy view of engine-owned device memory for torch (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def arena_tensor(engine):
    """Packed weight arena of an Engine or a VaeDecoder as a flat uint8 tensor (no copy).  Objects that already hold their arena as a
    tensor (`arena_as_tensor()`: the CPU stand-ins of tests/test_distributed_cpu.py) hand it over directly."""
    if hasattr(engine, "arena_as_tensor"):
        return engine.arena_as_tensor()
    ptr, nbytes = engine.arena()
    return torch.as_tensor(_DevicePointer(ptr, nbytes), device=f"cuda:{engine.device}")


def _cuda_sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


LAST_BROADCAST_CALLS = 0      # dist.broadcast calls issued by the last broadcast_weights / broadcast_pipeline (bench line)


def broadcast_tensor(t, src=0, chunk_bytes=None):
    """Broadcast a flat byte tensor.  Default: ONE collective - the bytes are viewed as int64 words when length and address
    allow, so a 5 GB arena is a 0.64 G-element message (xGMI is per-link bound: one large message is ideal).  `chunk_bytes`
    splits it into pieces instead (returns the number of collectives issued)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    flat = t.view(-1)
    if flat.is_cuda and dist.get_backend() == "gloo":
        # gloo carries host memory: stage device arenas through pinned-size host chunks (tests that run two ranks on ONE GPU; the
        # product backend is RCCL, which takes the device tensor as it is)
        step, n, is_src = 1 << 28, 0, dist.get_rank() == src
        for off in range(0, flat.numel(), step):
            piece = flat[off:off + step]
            host = piece.cpu() if is_src else torch.empty(piece.shape, dtype=piece.dtype)     # receivers: no device -> host copy
            dist.broadcast(host, src=src)
            n += 1
            if not is_src:
                piece.copy_(host)
        return n
    if chunk_bytes is None:
        if flat.dtype == torch.uint8 and flat.numel() % 8 == 0 and flat.data_ptr() % 8 == 0:
            flat = flat.view(torch.int64)
        dist.broadcast(flat, src=src)
        return 1
    n = 0
    for off in range(0, flat.numel(), chunk_bytes):
        dist.broadcast(flat[off:off + chunk_bytes], src=src)
        n += 1
    return n


def broadcast_tensors(tensors, src=0):
    """ONE collective for a list of tensors of mixed dtypes (text-encoder weights): rank `src` packs their bytes into a staging
    buffer, everyone receives it and the other ranks unpack in place.  Every rank passes same-shaped tensors."""
    if not dist.is_initialized() or dist.get_world_size() == 1 or not tensors:
        return 0
    sizes = [t.numel() * t.element_size() for t in tensors]
    offs, total = [], 0
    for sz in sizes:
        offs.append(total)
        total += (sz + 15) & ~15
    buf = torch.zeros(total, dtype=torch.uint8, device=tensors[0].device)
    is_src = dist.get_rank() == src
    if is_src:
        for t, o, sz in zip(tensors, offs, sizes):
            buf[o:o + sz].copy_(t.contiguous().view(-1).view(torch.uint8))
    n = broadcast_tensor(buf, src)
    if not is_src:
        for t, o, sz in zip(tensors, offs, sizes):
            t.view(-1).view(torch.uint8).copy_(buf[o:o + sz])
    return n


def broadcast_weights(engine, src=0):
    """Rank `src` has bound (packed) all weights; every other rank receives the packed bf16 arena (one collective)."""
    global LAST_BROADCAST_CALLS
    LAST_BROADCAST_CALLS = 0
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0.0
    import time
    _cuda_sync()
    t0 = time.perf_counter()
    engine.synchronize()
    LAST_BROADCAST_CALLS = broadcast_tensor(arena_tensor(engine), src)
    _cuda_sync()
    if dist.get_rank() != src:
        engine.arena_mark_bound()
    return time.perf_counter() - t0


def broadcast_pipeline(unet_engine, vae=None, text_encoders=(), src=0):
    """Everything a rank needs to sample without touching the checkpoint files: the UNet arena, the VAE decoder arena and the
    text-encoder weights (HipCLIPTextEncoder.parameter_tensors()), three collectives in total.  Ranks != src construct the same
    objects from the configs alone (any state dict of the right shapes, e.g. engine.random_state_dict(config), or the
    constructors' lazily zero-filled arenas): only the tensor SHAPES must agree, which the configs determine."""
    global LAST_BROADCAST_CALLS
    t = broadcast_weights(unet_engine, src)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    import time
    t0 = time.perf_counter()
    n = 0
    if vae is not None:
        vae.synchronize()
        n += broadcast_tensor(arena_tensor(vae), src)
        if dist.get_rank() != src:
            vae.arena_mark_bound()
    tens = [p for enc in text_encoders for p in enc.parameter_tensors()]
    n += broadcast_tensors(tens, src)
    _cuda_sync()
    LAST_BROADCAST_CALLS += n
    return t + time.perf_counter() - t0


def max_over_ranks(value, device="cpu"):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


# ---------------------------------------------------------------------------------------------- intra-image split
def split_ranges(n_streams, text_ref_stream, inject, nparts):
    """[(first, count)] per part: the rule of csrc/step_driver.inl::region_split_range (host-only query of the C ABI)."""
    import ctypes as C
    from .engine import load_library
    lib = load_library()
    out = []
    for part in range(nparts):
        first, count = C.c_int(), C.c_int()
        rc = lib.rt_op_split_range(n_streams, text_ref_stream, int(bool(inject)), part, nparts, C.byref(first), C.byref(count))
        if rc != 0:
            raise ValueError(lib.rt_op_last_error().decode())
        out.append((first.value, count.value))
    return out


def eps_tensor(engine):
    """The engine's noise-prediction buffer as a flat uint8 tensor (no copy) + bytes per stream."""
    ptr, per, nmax = engine.eps_info()
    return torch.as_tensor(_DevicePointer(ptr, per * nmax), device=f"cuda:{engine.device}"), per


def split_region_step(engine, i, guidance_scale, inject_selfattn, inject_background, xl, elide=False, defer_blend=False):
    """ONE rich-text step of ONE image on all ranks of the process group: every rank holds the same engine state, runs the UNet forwards
    of its contiguous range of the step's streams (text_ref and the region streams that inject from it always share the last range, so
    nothing crosses GPUs inside a forward), the ranks exchange their slices of the noise predictions - one broadcast per rank, 1.8 MB
    in total at SDXL - and every rank runs the step epilogue on the full set.  Bit-identical with engine.region_step on one GPU
    (a stream's forward does not depend on the other streams of its launch).  Returns the ranges [(first, count)] per rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        engine.region_step(i, guidance_scale, inject_selfattn, inject_background, xl=xl, elide=elide, defer_blend=defer_blend)
        return [(0, -1)]
    first, count, (n_streams, s_tref, inject) = engine.region_step_part(i, guidance_scale, inject_selfattn, inject_background, xl, rank, world,
                                                                          elide=elide, defer_blend=defer_blend)
    ranges = split_ranges(n_streams, s_tref, inject, world)          # a function of the step alone: no collective needed to agree on it
    assert ranges[rank] == (first, count)
    engine.synchronize()                                            # this rank's slice is complete before any rank reads it
    eps, per = eps_tensor(engine)
    for r, (f, c) in enumerate(ranges):
        if c > 0:
            broadcast_tensor(eps[f * per:(f + c) * per], src=r)
    # ORDERING CONTRACT: with RCCL dist.broadcast returns once the collective is ENQUEUED on the process group's stream; the engine's
    # own HIP stream (rt_set_stream may have installed a non-blocking one, e.g. under graph capture) is not ordered behind it.  The
    # epilogue below reads every rank's slice, so the received bytes must have landed: a device-wide synchronise (1.8 MB per step, the
    # epilogue is a few microseconds: nothing worth overlapping).  broadcast_weights does the same after its collective.
    _cuda_sync()
    engine.region_step_finish(i, guidance_scale, inject_selfattn, inject_background, xl, elide=elide, defer_blend=defer_blend)
    return ranges


def plain_capture_rank():
    """--split_image: the rank whose engine runs the TEXT stream of the plain pass and therefore records the token maps (rank 1 of >= 2)."""
    return 1 if dist.is_initialized() and dist.get_world_size() > 1 else 0


def split_plain_step(engine, i, guidance_scale):
    """ONE plain-text step (rd.py:200-214 / xl.py:880-905) of ONE image on all ranks: rank 0 runs the unconditional forward, rank 1 the
    text forward (and records the token maps: plain_capture_rank), further ranks none; two broadcasts of one noise prediction each
    (256 KB at SDXL), then CFG + scheduler step on every rank.  Bit-identical with engine.plain_step on one GPU."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        engine.plain_step(i, guidance_scale)
        return [(0, 2)]
    first, count = engine.plain_step_part(i, rank, world)
    ranges = [(r, 1) if r < 2 else (2, 0) for r in range(world)]     # the rule of csrc/step_driver.inl::plain_step_part
    assert ranges[rank] == (first, count), (ranges[rank], first, count)
    engine.synchronize()
    eps, per = eps_tensor(engine) if not hasattr(engine, "eps_as_tensor") else engine.eps_as_tensor()
    for r, (f, c) in enumerate(ranges):
        if c > 0:
            broadcast_tensor(eps[f * per:(f + c) * per], src=r)
    _cuda_sync()                                                     # ordering contract: see split_region_step
    engine.plain_step_finish(i, guidance_scale)
    return ranges


def broadcast_objects(objs, src):
    """Small Python objects (the region masks the capture rank derived from its token maps) from `src` to every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return objs
    box = [objs if dist.get_rank() == src else None]
    if dist.get_backend() == "nccl":
        dist.broadcast_object_list(box, src=src, device=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.broadcast_object_list(box, src=src)
    return box[0]


def guidance_from_rank0(engine, fn, h, w):
    """--split_image: the colour-guidance VAE pass (82 ms at SDXL) changes ONLY the latents; rank 0 runs `fn()` and the others receive
    the updated latents (64 KB at SDXL) instead of repeating the decoder forward + backward on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        fn()
        return
    if dist.get_rank() == 0:
        fn()
    engine.synchronize()
    _cuda_sync()
    if hasattr(engine, "latents_as_tensor"):                       # (CPU stand-ins of tests/test_distributed_cpu.py)
        lat = engine.latents_as_tensor()
    else:
        lat_ptr, _ = engine.state_ptrs()
        lat = torch.as_tensor(_DevicePointer(lat_ptr, 4 * h * w * 4), device=f"cuda:{engine.device}")     # latents [4, h, w] fp32
    broadcast_tensor(lat.view(-1).view(torch.uint8), src=0)
    _cuda_sync()                                                   # same ordering contract as split_region_step


def assert_ranks_agree(tensor, what="state", every_rank_raises=True):
    """--split_image relies on every rank holding bit-identical latents / masks (each rank computes the plain pass, the CPU clustering
    and the guidance update itself).  A cheap guard: the int64 view of the tensor's bytes is summed with position weights and
    compared across ranks (one 16-byte all_gather); any divergence - a different GPU SKU, library version or thread count per rank -
    fails loudly instead of silently mixing inconsistent noise predictions into one image."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True
    t = tensor.detach().contiguous().view(-1)
    raw = t.view(torch.uint8) if t.dtype != torch.uint8 else t
    pad = (-raw.numel()) % 8
    if pad:
        raw = torch.cat([raw, raw.new_zeros(pad)])
    words = raw.view(torch.int64)
    w = torch.arange(1, words.numel() + 1, dtype=torch.int64, device=words.device)
    digest = torch.stack([words.sum(), (words * w).sum()]).cpu()          # int64 wrap-around arithmetic: order-sensitive, exact
    got = [torch.zeros(2, dtype=torch.int64) for _ in range(dist.get_world_size())]
    if dist.get_backend() == "nccl":
        dev = tensor.device if tensor.is_cuda else torch.device("cuda", torch.cuda.current_device())
        gl = [g.to(dev) for g in got]
        dist.all_gather(gl, digest.to(dev))
        got = [g.cpu() for g in gl]
    else:
        dist.all_gather(got, digest)
    same = all(torch.equal(g, got[0]) for g in got)
    if not same and every_rank_raises:
        raise RuntimeError(f"--split_image: ranks disagree on {what} (digests {[g.tolist() for g in got]}): the intra-image split needs "
                           "homogeneous ranks (same GPU model, same library build, same host thread count)")
    return same
