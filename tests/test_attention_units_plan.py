"""Host logic of the shared-probability units of the self-attention launches (csrc/attention.hip::attention_units_plan through
rt_op_attention_units_plan; no GPU).  While a rich-text step injects, the region streams attend with text_ref's (Q, K) - the reference feeds
them the stored probabilities and does not recompute softmax(QK^T) either (/root/reference/models/attention_processor.py:522-524) - so the
launch is re-expressed as units whose members share one softmax per key tile.  The partition must be a function of the launch's own source
indices and shape only, cover every stream exactly once, and never put streams with different sources into one unit."""
import ctypes as C

import pytest

from rich_text_to_image_amd.engine import load_library


@pytest.fixture(scope="module")
def plan():
    lib = load_library()

    def f(src, tokens, mode=0, DP=64, k_src=None):
        B = len(src)
        ia = lambda v: (C.c_int * B)(*v)
        lo, un, me = (C.c_int * B)(), (C.c_int * B)(), (C.c_int * B)()
        G = lib.rt_op_attention_units_plan(ia(src), ia(k_src or src), B, tokens, DP, mode, lo, un, me)
        return G, list(lo), list(un), list(me)
    return f


def _check_partition(src, G, lo, un, me, k_src=None):
    k_src = k_src or src
    units = {}
    for b in range(len(src)):
        units.setdefault((lo[b], un[b]), []).append(b)
    for (l, u), mem in units.items():
        assert len({(src[b], k_src[b]) for b in mem}) == 1, "a unit mixes sources"
        assert all(me[b] == len(mem) for b in mem)
        assert len(mem) in (1, G)
        if l == 1:
            assert len(mem) == 1, "the separate launch holds one-stream units only"


def test_config3_injected_step_by_shape():
    """[uncond, base, uncond_ref, text_ref, 3 regions]: q / k source [0, 1, 2, 3, 3, 3, 3] (csrc/step_driver.inl)."""
    lib = load_library()
    src = [0, 1, 2, 3, 3, 3, 3]
    ia = lambda v: (C.c_int * len(v))(*v)
    lo, un, me = (C.c_int * 7)(), (C.c_int * 7)(), (C.c_int * 7)()
    # 1024 tokens (the 1280-channel level): pairs beside the one-stream units, ONE launch
    assert lib.rt_op_attention_units_plan(ia(src), ia(src), 7, 1024, 64, 0, lo, un, me) == 2
    assert list(me) == [1, 1, 1, 2, 2, 2, 2] and list(lo) == [0] * 7
    assert un[3] == un[4] and un[5] == un[6] and un[3] != un[5]
    assert {un[3], un[5]} == {0, 1}, "the shared units come first in the grid (the heavy workgroups start first)"
    # 4096 tokens (the 640-channel level): one unit of four in its own launch, the other three on the one-stream kernel
    assert lib.rt_op_attention_units_plan(ia(src), ia(src), 7, 4096, 64, 0, lo, un, me) == 4
    assert list(me) == [1, 1, 1, 4, 4, 4, 4] and list(lo) == [1, 1, 1, 0, 0, 0, 0]


@pytest.mark.parametrize("mode", [0, 2, 3, 4, 5])
@pytest.mark.parametrize("src", [[0, 1, 2, 3, 3], [0, 1, 2, 3, 3, 3], [0, 1, 2, 3, 3, 3, 3, 3], [0, 1, 2, 3, 3, 3, 3, 3, 3], [0, 0, 2, 3, 3, 3], [3, 3, 3, 3],
                                 [0, 1, 3, 2, 3, 3]])
@pytest.mark.parametrize("tokens", [256, 1024, 4096])
def test_every_stream_exactly_once_and_never_across_sources(plan, src, tokens, mode):
    G, lo, un, me = plan(src, tokens, mode)
    assert G in (2, 3, 4)
    if mode in (3, 5):
        assert G == 2
    _check_partition(src, G, lo, un, me)
    if mode in (2, 3):
        assert set(lo) == {0}, "modes 2 / 3: one launch"
    # regions beyond a multiple of G fall back to one-stream units, they are not dropped
    assert sorted(b for b in range(len(src))) == list(range(len(src))) and all(l in (0, 1) for l in lo)


def test_nothing_to_share_and_outside_the_domain(plan):
    assert plan([0, 1, 2, 3], 1024)[0] == 0                          # a non-injected step: the launches of rounds 1 - 5
    assert plan([0, 1, 2, 3, 3, 3, 3], 1024, mode=1)[0] == 0         # switched off
    assert plan([0, 1, 2, 3, 3, 3, 3], 1024, DP=96)[0] == 0          # SD-v1.5's d = 80 heads (padded 96): the one-stream kernel
    assert plan([0, 1, 2, 3, 3, 3, 3], 1000)[0] == 0                 # ragged key count (NK % 64 != 0)
    # same Q source but a different K source is NOT the same softmax
    assert plan([0, 1, 2, 3, 3], 1024, k_src=[0, 1, 2, 3, 2])[0] == 0


def test_partition_does_not_depend_on_other_streams(plan):
    """Batch invariance: the unit of the injected group is the same whatever else rides in the launch (the intra-image split hands a rank
    only {text_ref, regions})."""
    full = plan([0, 1, 2, 3, 3, 3, 3], 1024)
    part = plan([0, 0, 0, 0], 1024)                                   # the second rank's range: text_ref + 3 regions, sources renumbered
    assert full[0] == part[0] == 2
    assert full[3][3:] == part[3]                                     # member counts of the four streams
    full4, part4 = plan([0, 1, 2, 3, 3, 3, 3], 4096), plan([0, 0, 0, 0], 4096)
    assert full4[0] == part4[0] == 4 and full4[3][3:] == part4[3]
