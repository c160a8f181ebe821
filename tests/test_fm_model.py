"""CPU checks of the matrix-core factor pass (csrc/factor_mfma.hip): the lane-level index model (scripts/fm_model.py)
against the dense formula, the host planner (pure CPU arithmetic in the C-ABI library) against the model's geometry, its
LDS budget, and the bank-conflict freedom of the padded row pitch (scripts/lds_banks.py)."""
import ctypes as C
import os
import sys

import pytest
import torch

from lora_amd import _C
from tests import helpers as H

sys.path.insert(0, os.path.join(H.REPO, "scripts"))
import fm_model  # noqa: E402
import lds_banks  # noqa: E402


@pytest.mark.parametrize("args", [dict(M=70, K=64, N=64, r=4, R=64), dict(M=40, K=320, N=96, r=8, R=32),
                                  dict(M=50, K=320, N=320, r=4, R=32, gh=(8, 40, 64)),
                                  dict(M=33, K=320, N=640, r=4, R=64, xh=(8, 40, 64))])
def test_lane_level_model_of_the_pass_equals_the_dense_formula(args):
    fm_model.check(**args)


@pytest.mark.skipif(not _C.available(), reason="C-ABI library not built")
@pytest.mark.parametrize("M,K,N,r", [(16384, 320, 320, 4), (4096, 640, 640, 8), (1024, 1280, 1280, 16), (308, 768, 320, 4),
                                     (16384, 320, 2560, 4), (4096, 640, 5120, 4), (1024, 1280, 10240, 4), (100, 64, 96, 4)])
def test_planner_geometry_fits_the_lds_and_matches_the_model(M, K, N, r):
    pl = _C.factors_mfma_plan(M, K, N, r, torch.bfloat16)
    assert pl.supported and pl.rows_per_block in (32, 64) and pl.lds_class in (1, 2)
    cap = 81920 if pl.lds_class == 1 else 163840
    assert 0 < pl.lds_bytes <= cap
    nrb = -(-M // pl.rows_per_block)
    assert 1 <= pl.blocks_per_wg <= min(8, nrb) and pl.nparts == -(-nrb // pl.blocks_per_wg)
    assert pl.up_part_floats == pl.nparts * pl.rank_tile * N and pl.down_part_floats == pl.nparts * pl.rank_tile * K
    assert pl.pack_up_elems == 2 * pl.rank_tile * N + 8 and pl.pack_down_elems == 2 * pl.rank_tile * K + 8  # + the 16-byte scale tail (f16)
    assert pl.rows_per_block * min(K, N) <= (10 if pl.lds_class == 1 else 20) * 2048  # the next block waits in registers
    g = fm_model.geometry(M, K, N, pl.rows_per_block, cap)
    assert g is not None and g["lds"] == pl.lds_bytes
    # the ragged planner fills the same geometry into the site table
    site = (_C.FmSite * 1)()
    q = site[0]
    q.g = q.x = q.pk_up = q.pk_down = q.up_part = q.down_part = 4096  # any non-null, 16-byte aligned address
    q.ldg, q.ldx, q.M, q.N, q.K, q.r, q.scale, q.rows_per_block = N, K, M, N, K, r, 1.0, pl.rows_per_block
    q.blocks_per_wg = pl.blocks_per_wg
    grid = C.c_int64(0)
    rc = _C.require().lora_amd_factors_mfma_ragged_plan(site, 1, _C.BF16, pl.lds_class, C.byref(grid))
    assert rc == 0 and grid.value == pl.nparts
    assert (q.cw, q.nchunk, q.lds_bytes) == (g["cw"], g["nchunk"], g["lds"])
    assert bool(q.resident_is_x) == (K <= N)
    assert q.cw % 32 == 0 and q.rows_per_block * q.cw <= 16384
    assert (q.x_head_magic, q.g_head_magic) == (0, 0)   # dense rows: the column map is the identity


@pytest.mark.skipif(not _C.available(), reason="C-ABI library not built")
def test_planner_refuses_what_the_pass_does_not_take():
    assert not _C.factors_mfma_plan(1024, 320, 320, 4, torch.float32).supported      # f32 activations: VALU pass
    assert not _C.factors_mfma_plan(1024, 328, 320, 4, torch.bfloat16).supported     # K not a multiple of 32
    assert not _C.factors_mfma_plan(1024, 320, 320, 17, torch.bfloat16).supported    # rank > 16
    assert not _C.factors_mfma_plan(1024, 4096, 4096, 4, torch.bfloat16).supported   # no 32-row block of 4096 columns fits


@pytest.mark.parametrize("cols", [64, 128, 160, 192, 256, 320, 384, 512, 640, 768, 1280])
def test_padded_row_pitch_is_bank_conflict_free_for_both_operand_reads(cols):
    pitch, res = lds_banks.check(cols)
    assert pitch % 64 == 32 and pitch >= cols * 2
    assert res["phase1_b128"] == 1 and res["phase2_tr_b64"] == 1


def test_rank16_fragment_index_arithmetic_reproduces_matrix_products():
    """scripts/r16_model.py: rank_update16's interleaved transposed product, bwd_g16's register phase and transpose-read
    phase, rowdot16's k-step — lane by lane against numpy, ranks 16, 12 and 9."""
    import r16_model

    assert r16_model.check()


def test_rank16_wave_tile_pitch_is_conflict_free_for_the_transpose_reads():
    pitch = 96  # kR16Pitch
    tr = max(lds_banks.worst(lds_banks.HALVES, [(4 * (l >> 4) + ((l & 15) >> 2)) * pitch + (c0 + 4 * (l & 3)) * 2
                                                 for l in range(64)], 8, 64) for c0 in (0, 16))
    wr = lds_banks.worst(lds_banks.OCTETS, [(l & 15) * pitch + (l >> 4) * 16 for l in range(64)], 16, 64)
    assert tr == 1 and wr == 1
