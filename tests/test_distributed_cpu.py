"""world_size-2 gloo test of the seed-parallel launcher (the N>1 path of bench.py without GPUs)."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from rich_text_to_image_amd import launcher
    r, lr, w = launcher.init_distributed("gloo")
    assert (r, w) == (rank, world)
    # "arena": rank 0 holds the packed weights, the others receive them in several chunks
    arena = torch.arange(10_000, dtype=torch.uint8) if rank == 0 else torch.zeros(10_000, dtype=torch.uint8)
    nchunks = launcher.broadcast_tensor(arena, src=0, chunk_bytes=4096)
    ok_bcast = bool(torch.equal(arena, torch.arange(10_000, dtype=torch.uint8))) and nchunks == 3
    # default: ONE collective for the whole arena (int64 view), and one for a packed list of mixed-dtype tensors
    arena2 = torch.arange(4096, dtype=torch.uint8) if rank == 0 else torch.zeros(4096, dtype=torch.uint8)
    ok_bcast &= launcher.broadcast_tensor(arena2, src=0) == 1 and bool(torch.equal(arena2, torch.arange(4096, dtype=torch.uint8)))
    want = [torch.arange(7, dtype=torch.float32), torch.arange(5, dtype=torch.bfloat16), torch.arange(3, dtype=torch.int32).reshape(3, 1)]
    have = [w.clone() if rank == 0 else torch.zeros_like(w) for w in want]
    ok_bcast &= launcher.broadcast_tensors(have, src=0) == 1 and all(torch.equal(a, b) for a, b in zip(have, want))
    mine = launcher.shard_round_robin(list(range(7)), rank, world)
    tmax = launcher.max_over_ranks(1.0 + rank)
    launcher.barrier()
    q.put((rank, ok_bcast, mine, tmax))
    torch.distributed.destroy_process_group()


def test_seed_parallel_launcher_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]                                 # broadcast delivered the arena
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5]    # every request exactly once
    assert res[0][3] == res[1][3] == 2.0                           # max over ranks


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from rich_text_to_image_amd import launcher
    assert launcher.shard_round_robin([1, 2, 3], 0, 1) == [1, 2, 3]
    assert launcher.max_over_ranks(3.5) == 3.5
    assert launcher.broadcast_tensor(torch.zeros(4, dtype=torch.uint8)) == 0


def test_bench_gpus2_self_launches_under_torch_distributed_run():
    """`python bench.py --gpus 2` (no wrapper, no WORLD_SIZE) must re-execute itself under torch.distributed.run with one rank per
    device; --dry-launch stops before the first engine call, so the whole launch path runs on CPU over gloo."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["weight_broadcast_calls"] == 1 and line["arena_received"] is True
    assert line["max_over_ranks"] == 2.0 and line["requests_rank0"] == [0, 2]


def test_bench_reports_missing_gpus_clearly():
    """On a box with fewer GPUs than --gpus the bare command must not die on an assert: it says what is missing."""
    import json
    import subprocess
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["value"] is None and "need 8 GPUs" in line["error"] and "need 8 GPUs" in out.stderr
