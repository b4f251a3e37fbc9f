"""Host-side geometry planning of the tiled decode GEMM (no GPU needed): the planner's choice
and the candidate list the tuner times must only name geometries the library builds, with K
splits the consumers can take."""
import pytest

from mi355_decode import ops

SHAPES = [  # N, K, split_k: the projections of the benchmark models and of their tensor-parallel slices
    (6144, 4096, True), (4096, 4096, True), (28672, 4096, False), (4096, 14336, True), (128256, 4096, False),
    (4608, 3584, True), (3584, 3584, True), (37888, 3584, False), (3584, 18944, True), (152064, 3584, False),
    (1280, 8192, True), (8192, 1024, True), (7168, 8192, False), (16032, 4096, False), (1536, 4096, True),
]
KW = {302, 304, 308, 402, 404}        # K split across the waves of a workgroup: up to 32 rows
BUILT = {16: {4, 8, 104, 108} | KW, 32: {4, 8, 104, 108} | KW,
         64: {8, 104, 108, 208, 210, 212, 214, 216, 302, 304, 402}, 128: {8, 210, 212, 214, 216, 302},
         256: {8, 210, 212, 214, 216, 302}}       # two row blocks of the 128-row geometries
# beyond 64 / 32 rows also row blocks of 64 (+ 1000: the 64-row geometries) / of 32 rows (+ 2000: the 32-row ones
# that exist for epilogues 0-2 on plain x)
ROW_BLOCKS = {1000 + t for t in BUILT[64]} | {2000 + t for t in (8, 104, 108, 302, 304, 402)}


@pytest.mark.parametrize("M", [1, 16, 17, 32, 33, 64, 65, 128, 129, 192, 256])
@pytest.mark.parametrize("N,K,split", SHAPES)
def test_candidates_are_built_geometries(M, N, K, split):
    cls = 16 if M <= 16 else 32 if M <= 32 else 64 if M <= 64 else 128 if M <= 128 else 256
    plan = ops.plan_linear_tiled(N, K, split, M)
    cands = ops.linear_tiled_candidates(N, K, split, M)
    if plan is None:
        assert cands == []
        return
    assert cands[0] == plan and len(set(cands)) == len(cands)
    for tiles, groups in cands:
        assert tiles in BUILT[cls] or (tiles in ROW_BLOCKS and M > ops.row_block_rows(tiles))
        ksl = ops.tiled_slice_width(M, tiles)
        assert ksl == ops.tiled_slice_width(min(M, ops.row_block_rows(tiles)), tiles % 1000)
        assert K % ksl == 0 and 1 <= groups <= min(K // ksl, 8)
        assert split or groups == 1
        assert ops.tiles_per_workgroup(tiles) in ((1, 2, 4) if tiles % 1000 in KW else (4, 8, 10, 12, 14, 16))
    if M <= 32:
        assert all(t < 1000 for t, _ in cands)


def test_narrow_slices_get_the_k_split_across_waves():
    """The qkv / o / gate_up slices of a Llama-3-70B rank on 8 GPUs at 16 rows: the forms with several
    waves per tile are among the candidates; a SwiGLU matrix never gets 8 waves per tile (its gate and
    up tiles must share a workgroup); wide matrices and more than 32 rows get none."""
    assert {t for t, _ in ops.linear_tiled_candidates(1280, 8192, True, 16)} >= {304, 308}
    assert (308, 1) in ops.linear_tiled_candidates(8192, 1024, False, 16)
    gu = {t for t, _ in ops.linear_tiled_candidates(7168, 8192, False, 16, swiglu=True)}
    assert gu & KW and 308 not in gu
    assert not {t for t, _ in ops.linear_tiled_candidates(28672, 4096, False, 32, swiglu=True)} & KW
    assert {t for t, _ in ops.linear_tiled_candidates(4096, 7168, False, 64)} & KW == {302, 304, 402}
    assert {t for t, _ in ops.linear_tiled_candidates(7168, 4096, False, 128, swiglu=True)} & KW == {302}
    # ... and in row blocks of 64 / 32 the forms of those row classes (a TP4 rank at 128 rows)
    rb = {t for t, _ in ops.linear_tiled_candidates(7168, 4096, False, 128, swiglu=True) if t >= 1000}
    assert {1302, 1304, 2302, 2304, 1008, 2008} <= rb and 2308 not in rb
    assert ops.row_block_rows(8) == 128 and ops.row_block_rows(1302) == 64 and ops.row_block_rows(2104) == 32
    assert ops.tiled_slice_width(128, 1302) == 512 and ops.tiled_slice_width(256, 2104) == 512
    assert ops.tiles_per_workgroup(1302) == 4 and ops.tiles_per_workgroup(2008) == 8
    assert ops.tiled_slice_width(16, 308) == 1024 and ops.tiled_slice_width(32, 404) == 512
    assert ops.tiled_slice_width(64, 304) == 512 and ops.tiled_slice_width(128, 302) == 256
    assert ops.tiles_per_workgroup(302) == 4 and ops.tiles_per_workgroup(308) == 1


def test_rows_beyond_the_kernels_have_no_plan():
    assert ops.plan_linear_tiled(4096, 4096, True, 257) is None
    assert ops.linear_tiled_candidates(4096, 4096, True, 257) == []
    assert ops.plan_linear_tiled(4096, 448, True, 32) is None          # K not a multiple of any slice width


def test_slice_widths_and_workgroup_tiles():
    assert [ops.tiled_slice_width(m, 8) for m in (16, 32, 64, 128)] == [1024, 1024, 512, 256]
    assert [ops.tiled_slice_width(m, 104) for m in (16, 32, 64, 128)] == [512, 512, 256, 128]
    assert ops.tiled_slice_width(64, 208) == 256 and ops.tiled_slice_width(64, 216) == 512
    assert ops.tiled_slice_width(128, 216) == 256
    assert [ops.tiles_per_workgroup(t) for t in (4, 8, 104, 108, 208, 210, 216)] == [4, 8, 4, 8, 8, 10, 16]
    assert ops.tiled_slice_width(64, 210) == 512 and ops.tiled_slice_width(128, 214) == 256
    # 2368 tiles (gate_up of Qwen2-7B): 16 per workgroup fill 148 CUs, 10 per workgroup 237
    assert (210, 1) in ops.linear_tiled_candidates(37888, 3584, False, 64, swiglu=True)


def test_row_blocks_count_as_workgroups():
    """129 .. 256 rows run as two row blocks: the same column blocks fill the chip with half the K split."""
    t128, g128 = ops.plan_linear_tiled(4096, 3584, True, 128)
    t256, g256 = ops.plan_linear_tiled(4096, 3584, True, 256)
    assert t128 == t256 == 8 and g256 < g128
    assert ops.tiled_slice_width(256, 8) == ops.tiled_slice_width(128, 8) == 256
    # a tensor-parallel rank's gate_up slice at 128 rows may split K (a SwiGLU-from-slabs kernel follows)
    assert any(g > 1 for _, g in ops.linear_tiled_candidates(7168, 4096, True, 128, swiglu=True))
