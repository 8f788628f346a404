"""GPU side of the seed-parallel launcher: the packed weight arena as a zero-copy torch tensor and an RCCL ("nccl") broadcast of
it.  One GPU is available to the tests, so the process group has a single rank (two ranks cannot share a device under RCCL);
the 2-rank logic is covered by the gloo test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from rich_text_to_image_amd import launcher
from rich_text_to_image_amd.engine import Engine
from oracle.unet import TINY_XL_CONFIG, random_state_dict
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d", rank=0, world_size=1)
eng = Engine(TINY_XL_CONFIG, 32, 32, device=0)
eng.load_state_dict(random_state_dict(TINY_XL_CONFIG, seed=3))
t = launcher.arena_tensor(eng)
ptr, nbytes = eng.arena()
assert t.is_cuda and t.dtype == torch.uint8 and t.numel() == nbytes and t.data_ptr() == ptr
before = t.clone()
assert int(before.count_nonzero()) > nbytes // 4                     # packed weights are really there
flat = t.view(-1)
for off in range(0, flat.numel(), 1 << 20):                             # the chunked broadcast of launcher.broadcast_tensor
    dist.broadcast(flat[off:off + (1 << 20)], src=0)
torch.cuda.synchronize()
assert torch.equal(t, before)
assert launcher.max_over_ranks(2.5, device="cuda:0") == 2.5
launcher.barrier()
x = torch.randn(2, 4, 32, 32, device="cuda:0")
eng.set_prompts(torch.randn(2, 77, TINY_XL_CONFIG["cross_attention_dim"], device="cuda:0"), torch.randn(2, 32, device="cuda:0"),
                torch.tensor([[256., 256, 0, 0, 256, 256]]))
y = eng.unet_forward(x, 500.0, [0, 1])
assert torch.isfinite(y).all()
dist.destroy_process_group()
print("LAUNCHER_GPU_OK")
'''


def test_arena_view_and_rccl_broadcast_single_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from rich_text_to_image_amd import launcher
    port = launcher.free_port()
    r = subprocess.run([sys.executable, "-c", _SCRIPT % (ROOT, port)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "LAUNCHER_GPU_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_receiving_rank_path_arena_copy_then_mark_bound():
    """What a non-zero rank does: it never binds weights, receives the packed arena bytes (here: a device copy from a second engine
    standing in for the RCCL broadcast), marks it bound, and must then compute exactly what the sending engine computes."""
    import torch
    from oracle.unet import TINY_XL_CONFIG, random_state_dict
    from rich_text_to_image_amd import launcher
    from rich_text_to_image_amd.engine import Engine, RtError
    src = Engine(TINY_XL_CONFIG, 32, 32, device=0)
    src.load_state_dict(random_state_dict(TINY_XL_CONFIG, seed=9))
    dst = Engine(TINY_XL_CONFIG, 32, 32, device=0)
    assert dst.weights_missing()[0] > 0
    a, b = launcher.arena_tensor(src), launcher.arena_tensor(dst)
    assert a.numel() == b.numel()
    b.copy_(a)
    torch.cuda.synchronize()
    dst.arena_mark_bound()
    assert dst.weights_missing()[0] == 0
    g = torch.Generator().manual_seed(1)
    emb, pooled = torch.randn(2, 77, TINY_XL_CONFIG["cross_attention_dim"], generator=g).cuda(), torch.randn(2, 32, generator=g).cuda()
    tid = torch.tensor([[256., 256, 0, 0, 256, 256]])
    x = torch.randn(2, 4, 32, 32, generator=g).cuda()
    outs = []
    for e in (src, dst):
        e.set_prompts(emb, pooled, tid)
        outs.append(e.unet_forward(x, 300.0, [0, 1]))
    assert torch.equal(outs[0], outs[1])


def test_bench_two_ranks_on_one_gpu_runs_the_whole_n_gt_1_path(tmp_path):
    """`python bench.py --gpus 2` end to end with REAL engines: self-launch under torch.distributed.run, rendezvous, the collective self-check,
    rank 0 draws and packs the weights, ONE arena broadcast, per-rank independent requests, barrier + max-over-ranks timing, rank 0 prints the
    contract line with n_gpus = 2 and value = 2 x steps / max time, scaling "weak".  Both ranks sit on the ONE GPU of the box and the
    collectives run over gloo (RTDIFF_DIST_BACKEND / RTDIFF_FORCE_DEVICE: RCCL cannot place two ranks on a device) - everything but the
    transport is the path the driver's N = 2 / 4 / 8 runs take."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(RTDIFF_DIST_BACKEND="gloo", RTDIFF_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the line"
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["unit"] == "steps/s"
    assert d["value"] > 0 and abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 4 * 1e-3)) < 1e-6 * d["value"]      # whole-job aggregate: 2 ranks x steps / max-over-ranks time
    assert d["finite"] is True
    cc = d["collective_check"]
    assert cc["ok"] and cc["world"] == 2 and cc["backend"] == "gloo" and cc["bytes"] == 64 << 20
    assert d["weight_broadcast_calls"] >= 1
