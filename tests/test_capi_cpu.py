"""The C-ABI library loads without a GPU and exports every symbol include/lora_amd.h declares."""
import ctypes as C
import os
import re

import pytest

from lora_amd import _C
from tests.helpers import REPO


def _header_symbols():
    src = open(os.path.join(REPO, "include", "lora_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lora_amd_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_header_symbols():
    lib = _C.require()
    declared = _header_symbols()
    assert declared, "no declarations parsed from the header"
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"declared in include/lora_amd.h but not exported: {missing}"
    assert sorted(_C.SYMBOLS) == declared, "lora_amd/_C.py SYMBOLS out of sync with the header"
    assert lib.lora_amd_abi_version() == _C.ABI_VERSION == 7 and lib.lora_amd_target_arch() == b"gfx950"


def test_struct_layout_matches_header():
    assert C.sizeof(_C.MergeSite) == 80 and _C.MergeSite.tile_begin.offset == 56 and _C.MergeSite.N.offset == 32
    assert C.sizeof(_C.AdamWGroup) == 24
    assert C.sizeof(_C.WsSite) == 112 and _C.WsSite.dropout_p.offset == 80 and _C.WsSite.offset_dev.offset == 104


def test_merge_planner_is_pure_host_arithmetic():
    lib = _C.require()
    sites = (_C.MergeSite * 3)()
    for s, (N, K, r) in zip(sites, [(320, 320, 4), (2560, 320, 4), (1280, 2880, 16)]):
        s.N, s.K, s.r = N, K, r
        s.w_in = s.w_out = 4096
    summ = _C.MergeSummary()
    assert lib.lora_amd_merge_plan(sites, 3, _C.BF16, C.byref(summ)) == 0
    acc = 0
    for s in sites:
        assert s.cols_per_tile % 8 == 0 and s.rows_per_tile * s.r <= 2048
        assert s.tiles_k == -(-s.K // s.cols_per_tile) and s.tile_begin == acc and s.flags == 3
        ct8 = s.cols_per_tile // 8
        assert ct8 & (ct8 - 1) == 0 and (s.K // 8) % ct8 == 0  # power-of-two column tiles dividing the row
        acc += s.tiles_k * -(-s.N // s.rows_per_tile)
    assert summ.total_tiles == acc and summ.n_fast_sites == 3 and summ.rank_tile_fast == 16
    sites[0].r = 65
    assert lib.lora_amd_merge_plan(sites, 3, _C.BF16, C.byref(summ)) == -2
    assert b"rank 65" in lib.lora_amd_last_error()
    sites[0].r, sites[0].K = 4, 321  # odd K -> scalar lanes of the LDS-slab kernel
    assert lib.lora_amd_merge_plan(sites, 3, _C.BF16, C.byref(summ)) == 0 and sites[0].flags == 0
    sites[0].K, sites[0].r = 328, 32  # rank > 16 or no power-of-two factor -> 16-byte lanes of the slab kernel
    assert lib.lora_amd_merge_plan(sites, 3, _C.BF16, C.byref(summ)) == 0 and sites[0].flags == 1
    assert summ.n_fast_sites == 2 and sites[0].cols_per_tile * sites[0].r <= 8192


def test_workspace_queries():
    lib = _C.require()
    assert lib.lora_amd_colreduce_workspace(16384, 320, 4) == 256 * 4 * 320 * 4
    assert lib.lora_amd_colreduce_workspace(0, 320, 4) == 0
    assert lib.lora_amd_sumsq_workspace(10) >= 4


def test_device_path_never_falls_back(monkeypatch):
    """A device tensor without the library must raise, not run an eager fallback."""
    monkeypatch.setattr(_C, "_lib", None)
    monkeypatch.setattr(_C, "_load_error", "simulated: library absent")
    with pytest.raises(_C.HipExtensionMissing):
        _C.require()


def test_tune_cache_roundtrip(tmp_path, monkeypatch):
    """Per-shape kernel choices persist through LORA_AMD_TUNE_CACHE (so a profiled re-run does not re-time candidates)."""
    path = str(tmp_path / "tune.json")
    monkeypatch.setenv("LORA_AMD_TUNE_CACHE", path)
    a = _C._TuneCache("gemm_fwd")
    a["(1, 2, 3)"] = 22
    b = _C._TuneCache("sdpa")
    b["(4, 5)"] = ["EFFICIENT_ATTENTION", 64]
    assert _C._TuneCache("gemm_fwd")["(1, 2, 3)"] == 22 and _C._TuneCache("sdpa")["(4, 5)"] == ["EFFICIENT_ATTENTION", 64]
    assert "(1, 2, 3)" not in _C._TuneCache("gemm_bwd")


def test_host_pass_geometry_is_pure_host_arithmetic():
    """GroupNorm / LayerNorm planners of csrc/hostops.hip: every normalisation of the SD1.5 UNet at 512^2 and 768^2 is
    supported, the slice statistics stay a small fraction of the tensor, unsupported geometries say so."""
    lib = _C.require()
    sd15 = [(320, 64), (640, 64), (640, 32), (960, 64), (1280, 32), (1920, 32), (1280, 16), (2560, 16), (1280, 8),
            (2560, 8), (320, 96), (960, 96), (1920, 48), (2560, 24), (2560, 12)]
    for C_, hw in sd15:
        for B in (1, 4, 8):
            ws = lib.lora_amd_groupnorm_workspace(B, C_, hw * hw, 32)
            assert ws > 0 and lib.lora_amd_groupnorm_supported(B, C_, hw * hw, 32) == 1
            slices = ws // (B * 32 * 2 * 4)
            assert 1 <= slices <= 256
            assert ws <= max(B * C_ * hw * hw * 2 // 64, 4096), (B, C_, hw, ws)  # << one bf16 pass over the tensor
    assert lib.lora_amd_groupnorm_workspace(1, 32, 36, 8) == 0  # HW % 8 != 0
    assert lib.lora_amd_groupnorm_workspace(1, 30, 64, 8) == 0  # C % groups != 0
    assert lib.lora_amd_groupnorm_supported(0, 32, 64, 8) == 0
    for K in (8, 320, 640, 768, 1024, 1280, 2560):
        assert lib.lora_amd_layernorm_supported(K) == 1
    for K in (0, 4, 100, 2568, 4096):
        assert lib.lora_amd_layernorm_supported(K) == 0
    # argument checks of the compute entry points run before any launch: no GPU needed to see them
    assert lib.lora_amd_geglu_fwd(None, 16, None, 8, 4, 10, _C.BF16, None) != 0
    assert b"inner" in lib.lora_amd_last_error()
    assert lib.lora_amd_layernorm_fwd(None, None, None, None, None, 4, 100, 1e-5, _C.BF16, None) != 0


def test_conv3_nhwc_plan_is_pure_host_arithmetic():
    """Geometry of the channels-last 3x3 kernels (csrc/conv_nhwc.hip): which sites qualify and how they are cut."""
    assert C.sizeof(_C.Conv3NhwcPlan) == 72 and _C.Conv3NhwcPlan.pf_elems.offset == 40   # ABI 6: + fwd_tiles
    pl = _C.conv3_nhwc_plan(4, 320, 64, 64, 16)  # SD1.5 ResnetBlock2D conv at 512^2, batch 4, extended LoRA rank 16
    assert pl.native == 1 and pl.pt == 4 and pl.ks == 5 and pl.rank_pad == 16
    assert pl.fwd_tiles == 4 * (64 // 4) * (64 // 16)   # 16-column x pt-row pixel tiles: the fused forward's counters
    assert pl.ksplit == 1 and pl.csplit == 1 and pl.t_part_floats == 0  # 256 / 512 pixel tiles fill the chip
    assert pl.pf_elems == 9 * 320 * 16 and pl.pd_elems == 320 * 5 * 32
    assert pl.pr == 4 and (pl.pr + 2) * 64 * 8 <= 12 * 256  # strip of 4 rows (+2 halo rows) = 12 chunks per thread
    assert 1 <= pl.nsplit <= 4 * 16 and pl.down_part_floats == pl.nsplit * 16 * 320 * 9
    # dDown partials stay under ~30 % of the X stream or 4 MB, and one workgroup per CU at most
    assert pl.nsplit * 16 * 320 * 9 * 4 <= max(0.31 * (4 * 64 * 64 * 320 * 2), 4 << 20) and pl.nsplit * (320 // 64) <= 256
    pl = _C.conv3_nhwc_plan(1, 1280, 12, 12, 4)  # the 12x12 maps of 768^2 images: native here (masked tile edges)
    assert pl.native == 1 and pl.pt == 1 and pl.ks == 2 and pl.rank_pad == 4 and pl.pr == 12 and pl.nsplit == 1
    assert pl.ksplit == 10 and pl.t_part_floats == 10 * 144 * 4  # 12 pixel tiles: the channels are split as well
    assert pl.fwd_tiles == 12
    assert 1 < pl.csplit <= 1280 // 64
    for bad in [(4, 320, 64, 64, 3), (4, 320, 64, 64, 20), (4, 100, 64, 64, 4), (64, 2560, 256, 256, 4),
                (1, 64, 8, 200, 4)]:
        assert _C.conv3_nhwc_plan(*bad).native == 0
    lib = _C.require()
    assert lib.lora_amd_conv3_nhwc_plan(1, 64, 8, 8, 65, C.byref(_C.Conv3NhwcPlan())) == -2
    # round 6: the batched pack's table is planned on the host (piece counts per site: pf + pd + pu), bad sites refused
    sites = (_C.Conv3PackSite * 2)()
    for q, (r, ci, co) in zip(sites, [(16, 320, 640), (4, 1280, 1280)]):
        q.down = q.up = q.pf = q.pd = q.pu = 4096
        q.r, q.C_in, q.C_out = r, ci, co
    total = C.c_int64(0)
    assert lib.lora_amd_conv3_nhwc_pack_plan(sites, 2, C.byref(total)) == 0
    n0 = 9 * (320 // 32) * 64 + (320 // 16) * 5 * 64 + (640 // 32) * 128
    assert (sites[0].KS, sites[1].KS, sites[0].begin, sites[1].begin) == (5, 2, 0, n0)
    assert total.value == n0 + 9 * (1280 // 32) * 64 + (1280 // 16) * 2 * 64 + (1280 // 32) * 128
    sites[1].C_out = 1290
    assert lib.lora_amd_conv3_nhwc_pack_plan(sites, 2, C.byref(total)) == -1 and b"C_out" in lib.lora_amd_last_error()
    nb = C.c_int64(0)
    assert lib.lora_amd_linear_bwd_g_blocks(9216, 320, 16, C.byref(nb)) == 0 and nb.value >= 9216 // 128


def test_round3_planners_are_pure_host_arithmetic():
    """lora_amd_ragged_plan, lora_amd_linear_factors_self_plan and the head layout of lora_amd_merge_plan: no device."""
    lib = _C.require()
    assert C.sizeof(_C.RaggedDesc) == 128 and _C.RaggedDesc.begin1.offset == 88 and C.sizeof(_C.SubDesc) == 40
    descs = (_C.RaggedDesc * 2)()
    for d, (B, M, K) in zip(descs, [(3, 320, 320), (2, 1280, 23040)]):
        d.x, d.f, d.out, d.partial = 4096, 8192, 12288, 16384
        d.ldx, d.stride_x, d.M, d.K, d.batch = K, M * K, M, K, B
        d.stride_f, d.stride_out = M * 16, 16 * K
    g1, g2 = C.c_int64(0), C.c_int64(0)
    assert lib.lora_amd_ragged_plan(_C.RAGGED_COLREDUCE, descs, 2, 16, C.byref(g1), C.byref(g2)) == 0
    # 256-row blocks x 512-column tiles; stage 2: 64 outputs per block of the [16, K] result
    assert descs[0].blocks1 == 2 * 1 and descs[1].blocks1 == 5 * 45 and descs[1].begin1 == 3 * 2
    assert g1.value == 3 * 2 + 2 * 225 and g2.value == 3 * (16 * 320 // 64) + 2 * (16 * 23040 // 64)
    assert lib.lora_amd_ragged_plan(_C.RAGGED_ROWDOT, descs, 2, 16, C.byref(g1), C.byref(g2)) == 0
    assert g2.value == 0 and descs[0].rows_per_block > 0 and descs[0].kt_cols == 320
    descs[1].K = 23041  # rows no longer 32-byte friendly
    assert lib.lora_amd_ragged_plan(_C.RAGGED_ROWDOT, descs, 2, 16, C.byref(g1), C.byref(g2)) != 0

    pl = _C.factors_self_plan(16384, 320, 320, 4)
    assert pl.supported == 1 and pl.rank_tile == 4 and pl.nparts == 512
    assert pl.up_part_floats == 512 * 4 * 320 and pl.down_part_floats == 512 * 4 * 320
    assert _C.factors_self_plan(1024, 1280, 10240, 16).supported == 1
    assert _C.factors_self_plan(64, 24, 320, 4).supported == 0 and _C.factors_self_plan(64, 320, 324, 4).supported == 0

    sites = (_C.MergeSite * 1)()
    s = sites[0]
    s.N, s.K, s.r, s.w_in, s.w_out, s.down = 320, 320, 4, 4096, 8192, 4096
    s.out_heads = 40 | (64 << 16)
    summ = _C.MergeSummary()
    assert lib.lora_amd_merge_plan(sites, 1, 2, C.byref(summ)) == 0 and summ.n_fast_sites == 1
    s.out_heads = 48 | (64 << 16)  # 320 % 48 != 0
    assert lib.lora_amd_merge_plan(sites, 1, 2, C.byref(summ)) != 0


def test_factor_pass_planner_invariants_over_random_shapes():
    """lora_amd_linear_factors_self_plan_rows / lora_amd_linear_factors_self_ragged_plan (host arithmetic): for random
    (M, K, N, r) the row blocks cover the rows, the column tiles cover the row, LDS limits hold, and the ragged table's block
    ranges are the running sum of the per-site block counts."""
    import random

    lib = _C.require()
    rng = random.Random(0)
    sites_spec = []
    for _ in range(200):
        M = rng.choice([1, 7, 64, 130, 308, 1024, 4096, 16384])
        K = 8 * rng.randint(4, 400)
        N = 8 * rng.randint(4, 1300)
        r = rng.choice([1, 3, 4, 8, 12, 16])
        rows = rng.choice([0, 16, 64, 128, 300])
        pl = _C.factors_self_plan(M, K, N, r, rows)
        assert pl.supported == 1, (M, K, N, r)
        rt = 4 if r <= 4 else 8 if r <= 8 else 16
        assert pl.rank_tile == rt
        rows_eff = -(-M // pl.nparts)
        assert 1 <= pl.nparts <= M and rows_eff <= min(128, 2048 // rt)      # r-vectors of a block fit the LDS arrays
        assert pl.up_part_floats == pl.nparts * rt * N and pl.down_part_floats == pl.nparts * rt * K
        sites_spec.append((M, K, N, r, rows, pl.nparts))
    # one ragged table per rank tile
    for rt in (4, 8, 16):
        spec = [s_ for s_ in sites_spec if (4 if s_[3] <= 4 else 8 if s_[3] <= 8 else 16) == rt][:40]
        arr = (_C.SelfSite * len(spec))()
        for q, (M, K, N, r, rows, _) in zip(arr, spec):
            q.g = q.x = q.down = q.up = q.up_part = q.down_part = 4096
            q.ldg, q.ldx, q.M, q.N, q.K, q.r, q.scale, q.rows_per_block = N, K, M, N, K, r, 1.0, rows
        grid = C.c_int64(0)
        assert lib.lora_amd_linear_factors_self_ragged_plan(arr, len(spec), 2, C.byref(grid)) == 0
        begin = 0
        for q, (M, K, N, r, rows, nparts) in zip(arr, spec):
            assert q.block_begin == begin and q.nsplit == 1
            nrb = -(-M // q.rows_per_block)
            assert nrb == nparts                                             # slabs sized by the plan == blocks launched
            assert q.tile_g * q.nct_g >= N // 8 and q.tile_g <= 256 and q.tile_x * q.nct_x >= K // 8 and q.tile_x <= 256
            assert 0 <= q.logL_g <= 6 and 0 <= q.logL_x <= 6 and q.kt_g % 8 == 0 and q.kt_g * rt <= 8192
            begin += nrb
        assert grid.value == begin


def test_round4_hooks_and_plan_values_are_host_state():
    """The tuning hooks of ABI 4 read / write host globals (no GPU): each returns the previous value, refuses values outside
    its range, and the in-step merge's plan value carries the tile geometry it was planned for."""
    lib = _C.require()
    assert _C.rank16_mfma(-1) == 1 and _C.rank16_mfma(0) == 1 and _C.rank16_mfma(-1) == 0 and _C.rank16_mfma(1) == 0
    # ABI 6 dropped the dead lora_amd_factors_mfma_set_form entry (round 5 had removed the kernel it switched to) and the
    # unrouted input-stationary GEMM's three entries (scripts/gemm_xs/ now)
    for gone in ("lora_amd_factors_mfma_set_form", "lora_amd_linear_xs", "lora_amd_xs_config", "lora_amd_xs_set_tuning"):
        assert not hasattr(lib, gone), gone
    assert lib.lora_amd_merge_step_set_tuning(9, -1) != 0 and b"tile 9" in lib.lora_amd_last_error()
    sites = (_C.MstepSite * 2)()
    for s, (N, K, r) in zip(sites, [(320, 320, 4), (2560, 328, 16)]):
        s.N, s.K, s.r = N, K, r
        s.w = s.up = s.down = s.out = 4096
        s.ld_out = K
    tiles = {}
    for tile, (tr, tc) in enumerate([(128, 64), (64, 128), (128, 128), (256, 64)]):
        _C.merge_step_set_tuning(tile, -1)
        try:
            val = C.c_int64(0)
            assert lib.lora_amd_merge_step_plan(sites, 2, _C.BF16, C.byref(val)) == 0
        finally:
            _C.merge_step_set_tuning(2, -1)
        want = sum(-(-s.N // tr) * -(-s.K // tc) for s in sites)
        assert val.value >> 40 == tile and val.value & ((1 << 40) - 1) == want
        assert sites[1].tile_begin == -(-320 // tr) * -(-320 // tc) and sites[1].tiles_k == -(-328 // tc)
        tiles[tile] = want
    assert tiles[2] < tiles[0]
    sites[0].K = 321   # K % 8 != 0: the plan refuses (f32 weights and odd shapes stay on lora_amd_merge_batched)
    assert lib.lora_amd_merge_step_plan(sites, 2, _C.BF16, C.byref(val)) != 0


def test_matrix_core_factor_plan_prefers_64_row_blocks_for_the_register_kernel():
    import torch

    for (M, K, N), rows in (((16384, 320, 320), 64), ((4096, 640, 640), 64), ((1024, 1280, 1280), 32), ((308, 768, 1280), 32),
                            ((4096, 640, 5120), 64), ((16384, 320, 2560), 64)):
        pl = _C.factors_mfma_plan(M, K, N, 4, torch.bfloat16)
        assert pl.supported and pl.rows_per_block == rows and pl.blocks_per_wg == 1 and pl.nparts == -(-M // rows)
        # 20 resident 16-byte pieces per lane over four waves: rows x narrower width / (256 lanes x 8 elements)
        assert rows * min(K, N) <= 20 * 256 * 8


def test_block_map_of_a_planned_factor_pass_table_is_host_arithmetic():
    """ABI 7: lora_amd_factors_mfma_block_map fills block -> site for a table lora_amd_factors_mfma_ragged_plan planned (no GPU
    involved); an unplanned table is refused; the bytes helper puts the map behind the table at a 16-byte boundary."""
    import torch

    shapes = [(300, 1280, 1280), (4096, 640, 640), (1000, 640, 5120)]
    site = (_C.FmSite * len(shapes))()
    for q, (M, K, N) in zip(site, shapes):
        pl = _C.factors_mfma_plan(M, K, N, 4, torch.bfloat16)
        assert pl.supported and pl.lds_class == 2
        q.g = q.x = q.pk_up = q.pk_down = q.up_part = q.down_part = 4096
        q.ldg, q.ldx, q.M, q.N, q.K, q.r, q.scale = N, K, M, N, K, 4, 1.0
        q.rows_per_block, q.blocks_per_wg = pl.rows_per_block, pl.blocks_per_wg
    grid = C.c_int64(0)
    assert _C.require().lora_amd_factors_mfma_ragged_plan(site, len(shapes), _C.BF16, 2, C.byref(grid)) == 0
    m = list(_C.factors_mfma_block_map(site, grid.value))
    assert len(m) == grid.value and m == sorted(m) and set(m) == {0, 1, 2}
    for i, q in enumerate(site):
        assert m.index(i) == q.block_begin
    raw, off = _C.factors_mfma_table_bytes(site, grid.value)
    assert off % 16 == 0 and off >= C.sizeof(_C.FmSite) * len(shapes) and len(raw) == off + 4 * grid.value
    with pytest.raises(Exception):
        _C.factors_mfma_block_map((_C.FmSite * 2)(), 8)


def test_ws_head_layout_rules():
    """ABI 7: which head layouts the weight-stationary kernel's dropout instantiations take (lora_amd_linear_ws_heads /
    lora_amd_ws_site.y_heads): input heads in 16-byte chunks, output pad no wider than the head and at least half of it, whole
    panels, never both."""
    assert _C.ws_heads_ok(320, 320, (8, 40, 64), None) and _C.ws_heads_ok(320, 320, None, (8, 40, 64))
    assert _C.ws_heads_ok(640, 640, None, (8, 80, 128)) and _C.ws_heads_ok(1280, 1280, None, (8, 160, 256))
    assert not _C.ws_heads_ok(320, 320, (8, 40, 64), (8, 40, 64))          # both in one launch
    assert not _C.ws_heads_ok(768, 320, None, (8, 40, 64))                 # 320 columns are not whole 128-column panels at K = 768
    assert not _C.ws_heads_ok(320, 320, None, (8, 40, 128))                # pad wider than the head
    assert not _C.ws_heads_ok(320, 320, None, (10, 32, 40))                # head more than twice the pad
    assert not _C.ws_heads_ok(320, 320, (8, 36, 64), None) and not _C.ws_heads_ok(320, 320, (4, 40, 64), None)
    assert C.sizeof(_C.WsSite) == 112 and _C.WsSite.y_heads.offset == 84
