"""CPU tests of the packer (tc-viml_amd/csrc/tcv_pack.cpp) on the window shapes the GPU parity tests solve: the 800-block window,
ragged windows, a window without point factors.  No device: tcv_problem_plan_stats runs the whole symbolic packing (chain
elimination, chunking, gather programs) and the data layout.  tests/test_sanitize_cpu.py re-runs this file against the
AddressSanitizer + UBSan build of the host side."""
import numpy as np
import pytest

import synth
from util import golden_windows, sub_window


@pytest.fixture(scope="module")
def tcv(built):
    import tcv
    return tcv


def test_800_block_window_is_chunked(tcv):
    """200 landmarks x 4 observations (tests/test_gpu_solve.py::test_many_landmarks_are_chunked_through_lds)"""
    w = synth.window_at(synth.make_windows(600, 1, n_landmarks=200), 0)
    st = tcv.Window(w).plan_stats()
    assert st["nland"] == 200 and st["n_vis_chunk"] >= 2 and st["nc"] == 171
    tcv.lib().tcv_set_solver_variant(1)
    try:
        sd = tcv.Window(w).plan_stats()
    finally:
        tcv.lib().tcv_set_solver_variant(0)
    assert sd["nland"] == 200 and sd["nt"] == 11


@pytest.mark.parametrize("frames", [3, 6, 9])
def test_ragged_windows(tcv, frames):
    w = sub_window(synth.window_at(synth.make_windows(400, 1), 0), frames)
    st = tcv.Window(w).plan_stats()
    assert st["nx"] == 16 * frames + 7 and st["nc"] == 15 * frames + 6 and st["n_iunit"] == frames


def test_window_without_point_factors(tcv):
    w = dict(synth.window_at(synth.make_windows(500, 1), 0))
    pr = w["proj"]
    w["proj"] = {k: (np.asarray(v)[:0] if isinstance(v, np.ndarray) and v.shape[:1] == (200,) else v) for k, v in pr.items()}
    w["lam"] = np.zeros(0)
    st = tcv.Window(w).plan_stats()
    assert st["nland"] == 0 and st["nc"] == 171


def test_replay_sized_window_with_prior(tcv):
    """a front-end-sized window (~550 point factors) with the golden prior attached: several chunks in half a CU's LDS"""
    pre, main, z = golden_windows()
    big = synth.window_at(synth.make_windows(77, 1, n_landmarks=140), 0)
    w = dict(main)
    for k in ("proj", "lam"):
        w[k] = big[k]
    st = tcv.Window(w).plan_stats()
    assert st["nland"] == 140 and st["n_vis_chunk"] >= 3 and st["window_doubles"] > 10000
