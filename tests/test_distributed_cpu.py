"""world_size-2 gloo test of the seed-parallel launcher (the N>1 path of bench.py without GPUs)."""
import os
import sys

import pytest
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
    from rich_text_to_image_amd import launcher
    port = launcher.free_port()                                    # (a pid-derived port collided once with a socket still in TIME_WAIT)
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


def _json_objects(text):
    """Every JSON object in the ranks' shared stdout, also when two ranks' lines ran into each other (they share one pipe)."""
    import json
    dec, out, i = json.JSONDecoder(), [], 0
    while True:
        i = text.find("{", i)
        if i < 0:
            return out
        try:
            obj, end = dec.raw_decode(text, i)
            out.append(obj)
            i = end
        except json.JSONDecodeError:
            i += 1


def test_sample_cli_gpus2_shards_requests_and_broadcasts_the_whole_pipeline(tmp_path):
    """BASELINE configs 4 / 5 as ONE command: `python -m rich_text_to_image_amd.sample --gpus 2 --rich_text_json A B --seeds ...`
    re-executes itself as 2 ranks under torch.distributed.run (launcher.self_launch, module form), deals the (JSON, seed) requests
    round-robin and runs launcher.broadcast_pipeline - the UNet arena, the VAE arena AND the text-encoder weights, the call
    build_model makes - here on stand-in arenas over gloo (--dry_launch: no GPU, no checkpoint).  Rank 1 starts from zeros and must
    end with rank 0's bytes and arenas marked bound."""
    import json
    import subprocess
    a = tmp_path / "a.json"
    a.write_text(json.dumps({"ops": [{"insert": "a "}, {"attributes": {"font": "slabo"}, "insert": "night sky"}, {"insert": "\n"}]}))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-m", "rich_text_to_image_amd.sample", "--model", "SDXL", "--gpus", "2", "--dry_launch",
                          "--rich_text_json", str(a), "--seeds", "0", "1", "2", "3", "4"], env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = sorted(_json_objects(out.stdout), key=lambda d: d["rank"])
    assert [l["rank"] for l in lines] == [0, 1] and all(l["world"] == 2 and l["requests_total"] == 5 for l in lines)
    assert lines[0]["requests_mine"] == [0, 2, 4] and lines[1]["requests_mine"] == [1, 3] and lines[1]["seeds_mine"] == [1, 3]
    assert all(l["pipeline_received"] and l["broadcast_collectives"] == 3 for l in lines)


def test_sample_cli_request_list_forms(tmp_path):
    """build_requests: one JSON x several seeds, JSONs paired with seeds, and a JSON-lines request file; a single request keeps the
    reference's single-image behaviour (sample.py:17-114)."""
    import json
    import types
    from rich_text_to_image_amd.sample import build_requests
    j1, j2 = '{"ops":[{"insert":"a cat\\n"}]}', '{"ops":[{"insert":"a dog\\n"}]}'
    ns = lambda **k: types.SimpleNamespace(**dict(dict(requests=None, rich_text_json=None, seeds=None, seed=6, negative_prompt=""), **k))
    r = build_requests(ns(rich_text_json=[j1]))
    assert len(r) == 1 and r[0]["seed"] == 6 and r[0]["index"] == 0
    r = build_requests(ns(rich_text_json=[j1], seeds=[0, 1, 2]))
    assert [x["seed"] for x in r] == [0, 1, 2] and all(x["text_input"] == json.loads(j1) for x in r)
    r = build_requests(ns(rich_text_json=[j1, j2], seeds=[4, 5]))
    assert [(x["text_input"]["ops"][0]["insert"], x["seed"]) for x in r] == [("a cat\n", 4), ("a dog\n", 5)]
    f = tmp_path / "reqs.jsonl"
    f.write_text(json.dumps({"rich_text_json": json.loads(j2), "seed": 9}) + "\n" + json.dumps({"rich_text_json": json.loads(j1)}) + "\n")
    r = build_requests(ns(requests=str(f), rich_text_json=[j1], seeds=[1]))
    assert [x["seed"] for x in r] == [9, 6, 1] and [x["index"] for x in r] == [0, 1, 2]
    import pytest
    with pytest.raises(SystemExit):
        build_requests(ns())


def test_text_encoder_empty_state_dict_has_the_shapes_of_a_real_one():
    """Ranks != 0 construct their HipCLIPTextEncoder parameter list from config.json alone (clip_text_encoder.empty_state_dict) and
    launcher.broadcast_tensors requires equal shapes on every rank: compare with transformers' own CLIPTextModelWithProjection."""
    import transformers
    from rich_text_to_image_amd.clip_text_encoder import empty_state_dict
    cfg = transformers.CLIPTextConfig(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=128, vocab_size=300,
                                      max_position_embeddings=77, projection_dim=48)
    real = transformers.CLIPTextModelWithProjection(cfg).state_dict()
    mine = empty_state_dict(cfg, with_projection=True)
    norm = lambda k: k if k.startswith("text_model.") or k.startswith("text_projection") else "text_model." + k
    real = {norm(k): tuple(v.shape) for k, v in real.items() if "position_ids" not in k}
    assert {k: tuple(v.shape) for k, v in mine.items()} == real


def test_intra_image_split_keeps_injecting_streams_with_their_source():
    """The stream ranges of launcher.split_region_step (csrc/step_driver.inl::region_split_range through the host-only C-ABI query):
    contiguous, disjoint, covering every stream of the step; while the injection is on, text_ref and all region streams (which read its
    self-attention Q / K and ResNet feature layer by layer) share the LAST range, and the independent streams before it are dealt
    evenly to the other ranks."""
    from rich_text_to_image_amd.launcher import split_ranges
    for F, s_tref, inject in ((7, 3, True), (7, 3, False), (5, -1, False), (11, 3, True), (2, -1, False), (4, 3, True)):
        for nparts in (1, 2, 3, 4):
            r = split_ranges(F, s_tref, inject, nparts)
            assert len(r) == nparts and r[0][0] == 0 and sum(c for _, c in r) == F
            for (f0, c0), (f1, _) in zip(r, r[1:]):
                assert f0 + c0 == f1                                   # contiguous, in order
            if inject and nparts > 1:
                assert r[-1] == (s_tref, F - s_tref)                   # text_ref + regions together
                assert max(c for _, c in r[:-1]) - min(c for _, c in r[:-1]) <= 1
            elif nparts > 1:
                assert max(c for _, c in r) - min(c for _, c in r) <= 1
    assert split_ranges(7, 3, True, 2) == [(0, 3), (3, 4)]            # config 3: {uncond, base, uncond_ref} | {text_ref, 3 regions}


def _worker_round6(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from rich_text_to_image_amd import launcher
    launcher.init_distributed("gloo")
    rec = launcher.collective_self_check(nbytes=1 << 20)

    class StandIn:                                                 # what guidance_from_rank0 needs of an engine
        device = 0

        def __init__(self):
            self.lat = torch.full((4, 8, 8), float(rank))

        def synchronize(self):
            pass

        def latents_as_tensor(self):
            return self.lat
    eng = StandIn()
    calls = []

    def guide():
        calls.append(rank)
        eng.lat += 41.5                                            # the "VAE pass" changes the latents on the rank that runs it
    launcher.guidance_from_rank0(eng, guide, 8, 8)
    agree = launcher.assert_ranks_agree(eng.lat, "latents", every_rank_raises=False)
    disagree = launcher.assert_ranks_agree(torch.full((5,), float(rank)), "a rank-dependent tensor", every_rank_raises=False)
    q.put((rank, rec, calls, float(eng.lat[0, 0, 0]), agree, disagree))
    torch.distributed.destroy_process_group()


def test_collective_self_check_and_split_mode_guidance_handoff_world2_gloo():
    """Round 6: every N > 1 launch starts with a pattern broadcast that every rank verifies (here over gloo); in
    --split_image mode the colour-guidance pass runs on rank 0 only and the others receive the latents; the rank-agreement digest
    tells identical from diverging state."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from rich_text_to_image_amd import launcher
    port = launcher.free_port()
    procs = [ctx.Process(target=_worker_round6, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, rec, calls, lat00, agree, disagree in res:
        assert rec["ok"] and rec["world"] == 2 and rec["backend"] == "gloo" and rec["bytes"] == 1 << 20
        assert calls == ([0] if rank == 0 else [])                 # the guidance pass ran on rank 0 only ...
        assert lat00 == 41.5                                       # ... and every rank holds ITS result (rank 1 started from 1.0)
        assert agree is True and disagree is False


def _worker_plain_split(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from rich_text_to_image_amd import launcher
    launcher.init_distributed("gloo")

    class StandIn:
        """What split_plain_step needs of an engine: the part / finish calls and the eps buffer (two slots of 16 bytes)."""
        device = 0

        def __init__(self):
            self.eps = torch.zeros(3 * 16, dtype=torch.uint8)
            self.ran, self.finished = [], []

        def plain_step_part(self, i, part, nparts):
            first, count = (part, 1) if part < 2 else (2, 0)
            if nparts == 1:
                first, count = 0, 2
            for s_ in range(first, first + count):                    # "forward" of stream s_: its slot gets a stream- and step-dependent value
                self.eps[s_ * 16:(s_ + 1) * 16] = 10 * (i + 1) + s_ + 1
                self.ran.append(s_)
            return first, count

        def plain_step_finish(self, i, g):
            self.finished.append((i, g, self.eps[:32].clone()))

        def synchronize(self):
            pass

        def eps_as_tensor(self):
            return self.eps, 16
    eng = StandIn()
    ranges = [launcher.split_plain_step(eng, i, 7.5) for i in range(2)]
    masks = launcher.broadcast_objects([torch.full((2, 2), 3.0)] if rank == launcher.plain_capture_rank() else None, launcher.plain_capture_rank())
    q.put((rank, ranges, eng.ran, [(i, g, e.tolist()) for i, g, e in eng.finished], launcher.plain_capture_rank(), float(masks[0].sum())))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_split_mode_plain_pass_one_stream_per_rank_gloo(world):
    """Round 6 (VERDICT r5 next #6b): in --split_image mode the plain pass is shared too - rank 0 runs the unconditional forward, rank 1 the
    text forward (and is the rank that records the token maps and derives the masks), further ranks none; every rank finishes the step on
    BOTH predictions; the masks travel from the capture rank as one object broadcast."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from rich_text_to_image_amd import launcher
    port = launcher.free_port()
    procs = [ctx.Process(target=_worker_plain_split, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ranges, ran, finished, cap, msum in got:
        assert ranges == [[(r, 1) if r < 2 else (2, 0) for r in range(world)]] * 2
        assert ran == ([rank, rank] if rank < 2 else [])                  # its own stream, once per step
        assert cap == 1 and msum == 12.0
        for i, g, eps in finished:                                        # every rank saw BOTH predictions before its epilogue
            assert g == 7.5 and eps == [10 * (i + 1) + 1] * 16 + [10 * (i + 1) + 2] * 16
    assert len({tuple(map(str, f)) for _, _, _, f, _, _ in got}) == 1
