"""Host logic of the split rule for problems that cannot fill the chip (csrc/gemm.hip::splitk_slices / conv3p_split_slices through
rt_op_split_plan; no GPU): route and slice count are pure functions of ONE stream's shape - never of the batch - so a stream's k order
is the same alone and inside any batch, and the SD-v1.5 shapes take the routes LABNOTES R6.10 documents."""
import ctypes as C

import pytest

from rich_text_to_image_amd.engine import load_library

EPI_BF16, EPI_F32, EPI_TEMB, EPI_GEGLU, EPI_F16 = 0, 1, 2, 3, 4
ONE, KSLICES, PATCH_CHUNKS = 0, 1, 2


@pytest.fixture(scope="module")
def plan():
    lib = load_library()

    def f(conv, epi, streams, rps, N, K):
        r, s = C.c_int(-1), C.c_int(-1)
        assert lib.rt_op_split_plan(conv, epi, streams, rps, N, K, C.byref(r), C.byref(s)) == 0
        return r.value, s.value
    return f


def test_plan_never_depends_on_the_number_of_streams(plan):
    shapes = [(1, EPI_TEMB, 256, 1280, 1280), (1, EPI_F16, 256, 1280, 2560), (1, EPI_TEMB, 256, 1280, 640), (1, EPI_F16, 1024, 640, 640),
              (1, EPI_TEMB, 64, 1280, 1280), (1, EPI_F16, 64, 1280, 2560), (3, EPI_F16, 256, 1280, 1280), (1, EPI_TEMB, 4096, 320, 320),
              (0, EPI_F16, 64, 1280, 1280), (0, EPI_F16, 64, 1280, 5120), (0, EPI_F16, 256, 1280, 1280), (0, EPI_F16, 256, 1280, 5120),
              (1, EPI_TEMB, 1024, 1280, 1280), (1, EPI_F16, 16384, 320, 320), (0, EPI_F16, 1024, 1280, 1280)]
    for conv, epi, rps, N, K in shapes:
        got = {plan(conv, epi, s, rps, N, K) for s in (1, 2, 3, 4, 5, 7, 8)}
        assert len(got) == 1, (conv, epi, rps, N, K, got)


def test_sd15_shapes_take_the_documented_routes(plan):
    # 16x16 maps (one 16x16 patch per image): the patch kernel split over input-channel chunks, ~8 slices of equal length
    assert plan(1, EPI_TEMB, 3, 256, 1280, 1280) == (PATCH_CHUNKS, 7)          # 20 chunks: 3, 3, 3, 3, 3, 3, 2
    assert plan(1, EPI_F16, 3, 256, 1280, 2560) == (PATCH_CHUNKS, 8)           # 40 chunks: 8 x 5
    assert plan(1, EPI_F16, 3, 256, 1280, 1920) == (PATCH_CHUNKS, 8)           # 30 chunks
    assert plan(1, EPI_TEMB, 3, 256, 1280, 640) == (PATCH_CHUNKS, 5)           # 10 chunks: 5 x 2
    assert plan(3, EPI_F16, 3, 256, 1280, 1280) == (PATCH_CHUNKS, 7)           # the up-sample-fused convolution with a 16x16 OUTPUT
    # 32x32 x 640 maps: two halves
    assert plan(1, EPI_F16, 3, 1024, 640, 640) == (PATCH_CHUNKS, 2)
    assert plan(1, EPI_TEMB, 3, 1024, 640, 1280) == (PATCH_CHUNKS, 2)
    # 8x8 maps are not patch-eligible: K slices of the implicit GEMM - four 64-pixel streams are TWO 128-row tiles (twice round 5's slices)
    assert plan(1, EPI_TEMB, 3, 64, 1280, 1280) == (KSLICES, 12)
    assert plan(1, EPI_F16, 3, 64, 1280, 2560) == (KSLICES, 12)
    # 64x64 x 320 and everything of an SDXL step fill the chip: one launch
    assert plan(1, EPI_TEMB, 3, 4096, 320, 320) == (ONE, 1)
    for rps, N, Cin in ((1024, 1280, 1280), (1024, 1280, 2560), (4096, 640, 640), (16384, 320, 320)):
        assert plan(1, EPI_F16, 7, rps, N, Cin) == (ONE, 1)
    # dense: K <= 1280 projections at >= 256 tokens per stream stay unsplit; the 64-token level and K = 5120 are sliced
    assert plan(0, EPI_F16, 3, 256, 1280, 1280) == (ONE, 1)
    assert plan(0, EPI_F16, 3, 64, 1280, 1280)[0] == KSLICES
    assert plan(0, EPI_F16, 3, 64, 1280, 5120) == (KSLICES, 12)
    assert plan(0, EPI_F16, 7, 1024, 1280, 5120) == (ONE, 1)
