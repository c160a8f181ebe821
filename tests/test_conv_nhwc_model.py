"""Index arithmetic of the channels-last 3x3 kernels (csrc/conv_nhwc.hip) against torch's conv2d and its autograd.

scripts/nhwc_model.py restates the three kernels lane by lane in numpy (fragment layouts, slot -> (tap, rank) mapping,
clamps and masks); a wrong shift or packing shows up here without a GPU.  The device kernels themselves are compared
with the oracle in tests/test_gpu_conv_nhwc.py."""
import importlib.util
import os

from tests.helpers import REPO


def _model():
    spec = importlib.util.spec_from_file_location("nhwc_model", os.path.join(REPO, "scripts", "nhwc_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_lane_level_model_matches_conv2d_autograd():
    m = _model()
    # ragged map (W not a multiple of 16, H not a multiple of the tile height), rank 8 (low-part rows), 2 dDown splits
    assert max(m.check(2, 3, 5, 64, 8, 2, 2, seed=1)) < 1e-9
    # rank 12: (tap, rank) slots straddle k-steps; one image row tile taller than the map
    assert max(m.check(1, 2, 18, 64, 12, 4, 1, seed=2)) < 1e-9
