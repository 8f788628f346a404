"""Host logic of the GEMM tile rule (csrc/gemm16.hip::gemm16_pick through rt_op_gemm16_pick; no GPU): the summation class is a pure function
of ONE stream's shape - never of the batch - which is what makes a stream computed alone and inside a batch give the same bits
(LABNOTES 4.7), and the routing decisions it quotes for the SDXL / SD-v1.5 shapes."""
import ctypes as C

import pytest

from rich_text_to_image_amd.engine import load_library

EPI_BF16, EPI_F32, EPI_TEMB, EPI_GEGLU, EPI_F16 = 0, 1, 2, 3, 4


@pytest.fixture(scope="module")
def pick():
    lib = load_library()

    def f(conv, epi, streams, rps, N, K, vt=0):
        ws = C.c_int(0)
        v = lib.rt_op_gemm16_pick(conv, epi, streams, rps, N, K, vt, C.byref(ws))
        return v, ws.value
    return f


def test_variant_never_depends_on_the_number_of_streams(pick):
    shapes = [(0, EPI_F16, 1024, 1280, 1280), (0, EPI_F16, 1024, 1280, 5120), (0, EPI_BF16, 1024, 2560, 1280), (0, EPI_GEGLU, 1024, 10240, 1280),
              (0, EPI_GEGLU, 4096, 5120, 640), (0, EPI_F16, 4096, 640, 640), (0, EPI_BF16, 4096, 1280, 640), (0, EPI_BF16, 256, 1280, 1280),
              (1, EPI_TEMB, 1024, 1280, 1280), (1, EPI_F16, 4096, 640, 640), (1, EPI_F16, 16384, 320, 320), (1, EPI_TEMB, 1024, 640, 320)]
    # what must not move is the CLASS (summation order): A = one ascending sum over k (variants 2, 3, 4, 5, 8: bit-identical with each
    # other and with gemm.hip), B = K-split (0, 1), B^T (6, 7), or "not in the family"; inside a class the tile follows the actual M
    cls = {-1: "none", 0: "B", 1: "B", 9: "B", 2: "A", 3: "A", 4: "A", 5: "A", 8: "A", 10: "A", 11: "A", 6: "BT", 7: "BT", 12: "BT"}
    for conv, epi, rps, N, K in shapes:
        got = {cls[pick(conv, epi, s, rps, N, K)[0]] for s in (1, 2, 3, 5, 7, 8)}
        assert len(got) == 1, (conv, epi, rps, N, K, got)


def test_sdxl_step_shapes_take_the_documented_classes(pick):
    assert pick(0, EPI_F16, 7, 1024, 1280, 1280)[0] == 0            # to_out / to_q at 1280 channels: K-split class B, 224x160
    assert pick(0, EPI_F16, 7, 1024, 1280, 5120)[0] == 0            # ff.net.2
    assert pick(0, EPI_BF16, 7, 1024, 2560, 1280)[0] == 4           # stacked Q|K: class A, 224x320
    assert pick(0, EPI_GEGLU, 7, 1024, 10240, 1280) == (2, 1)       # GEGLU: 224x256, W-stationary tile order (40 column tiles)
    assert pick(0, EPI_GEGLU, 7, 4096, 5120, 640)[1] == 0           # 20 column tiles do not divide over 8 XCDs: grouped order
    assert pick(0, EPI_F16, 7, 4096, 640, 640)[0] == 4              # the 640-channel level fills the chip per stream on 320-wide tiles
    assert pick(0, EPI_BF16, 7, 1024, 1280, 1280, vt=1)[0] in (6, 7)   # V^T = W_v X^T: transposed K-split class
    assert pick(0, EPI_BF16, 7, 64, 1280, 1280)[0] == -1            # 8x8 maps stay on gemm.hip (128x128 tiles / split-K)
    assert pick(0, EPI_BF16, 7, 1024, 1280, 320)[0] == -1           # K % 128 != 0


def test_convolutions_only_where_one_image_contributes_enough_tiles(pick):
    assert pick(1, EPI_TEMB, 7, 1024, 1280, 1280)[0] == 0           # SDXL 32^2 x 1280 -> 1280: 5 x 8 = 40 tiles per image
    assert pick(1, EPI_F16, 7, 4096, 640, 640)[0] == 4              # SDXL 64^2: 19 x 2 = 38
    assert pick(1, EPI_F16, 7, 16384, 320, 320)[0] == 4             # SDXL 128^2: 74
    assert pick(1, EPI_TEMB, 3, 1024, 640, 640)[0] == -1            # SD-v1.5 32^2 x 640: 5 x 4 = 20 tiles per image -> patch kernel
    assert pick(1, EPI_TEMB, 3, 256, 1280, 1280)[0] == -1           # SD-v1.5 16^2
    assert pick(1, EPI_F32, 1, 4096, 512, 512)[0] == 2              # a 64^2 x 512 VAE layer passes the rule (38 tiles); the single-image VAE
    #                                                                 opts out on its own through GemmArgs.prefer_patch_conv (vae.hip)


def test_grouped_qk_vt_launch_rule():
    """attn1's Q|K + V^T projections go out as one grouped launch (gemm16_dual_kernel) exactly where both problems have their tile in
    the grouped instantiations: SDXL's two attention levels with 7 streams (Q|K on 224x320), with the 4 Q|K streams of an injected
    step (224x256) and with the 2 streams of the plain pass; everything else - SD-v1.5 (K = 320 is outside the family), other batch
    sizes, 16x16 maps - stays on two launches.  Results are bit-identical either way
    (tests/test_kernels_gpu.py), so this rule MAY look at the batch."""
    lib = load_library()
    f = lib.rt_op_gemm_pair_pick
    assert f(7, 7, 1024, 2560, 1280, 1280) == 0
    assert f(4, 7, 1024, 2560, 1280, 1280) == 1
    assert f(7, 7, 4096, 1280, 640, 640) == 0
    assert f(4, 7, 4096, 1280, 640, 640) == 1
    assert f(2, 2, 1024, 2560, 1280, 1280) == 2                      # plain pass: 128x256 + 160x64
    assert f(2, 2, 4096, 1280, 640, 640) == 3                        # plain pass, 640-channel level: 224x256 + 160x128
    assert f(3, 3, 1024, 2560, 1280, 1280) == -1
    assert f(3, 3, 4096, 640, 320, 320) == -1
    assert f(7, 7, 256, 2560, 1280, 1280) == -1


def test_small_batches_take_64_row_tiles_of_the_same_class(pick):
    """The 2-stream plain pass and SD-v1.5's 3-stream steps leave half of the chip idle on 128- / 224-row tiles: where 64-row (64-token)
    tiles of the SAME class fit one round of 256 workgroups they are taken (measured: profiles/r4_gemm16_probe_small_batch.txt); the
    7-stream shapes of a rich-text step are untouched."""
    assert pick(0, EPI_F16, 2, 1024, 1280, 1280)[0] == 9             # plain pass to_out 1280: 32 x 8 = 256 tiles
    assert pick(0, EPI_F16, 2, 1024, 1280, 5120)[0] == 9
    assert pick(0, EPI_F16, 3, 1024, 640, 640)[0] == 9               # SD-v1.5, 3 streams, 32^2 x 640
    assert pick(0, EPI_F16, 5, 1024, 640, 640)[0] == 1               # 5 streams: 80 x 4 = 320 tiles would start a second round
    assert pick(0, EPI_F16, 3, 256, 1280, 1280)[0] == 9              # SD-v1.5 16^2 x 1280
    assert pick(0, EPI_F16, 2, 4096, 640, 640)[0] == 11              # plain pass, 640-channel level (class A): 128 x 2 = 256 tiles of 64 x 320
    assert pick(0, EPI_F16, 2, 4096, 640, 2560)[0] == 11
    assert pick(0, EPI_BF16, 2, 4096, 1280, 640)[0] == 2             # its Q|K keeps 224 x 256 (185 tiles: more than half of the chip)
    assert pick(0, EPI_BF16, 2, 1024, 1280, 1280, vt=1)[0] == 12     # V^T: 8 x 32 = 256 tiles of 160 x 64
    assert pick(0, EPI_BF16, 2, 4096, 640, 640, vt=1)[0] == 7        # 4 x 128 = 512 of those: stays on 160 x 128
    for args in [(0, EPI_F16, 7, 1024, 1280, 1280), (0, EPI_F16, 7, 4096, 640, 640), (0, EPI_BF16, 7, 1024, 2560, 1280), (0, EPI_BF16, 4, 1024, 2560, 1280),
                 (0, EPI_BF16, 4, 4096, 1280, 640)]:
        assert pick(*args)[0] in (0, 2, 4), args
    assert pick(0, EPI_BF16, 7, 1024, 1280, 1280, vt=1)[0] == 6
