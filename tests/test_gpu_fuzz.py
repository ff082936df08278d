"""The structure / input fuzzers of tests/dev (seeded, deterministic) as gates of the -m gpu suite: a short seed range of each, every case
against the oracle.  The long ranges and what they found are in profiles/r05_fuzz_*.txt; a seed range here is one that is clean there
(fuzz_solve seed 9 -- 290 ragged landmarks, final inverse depths 1.6e-6 apart after eight iterations -- and the exactly-collinear detections
of fuzz_lines scenes 0, 1, 6 are outside them; DESIGN.md 2 says why those are not defects)."""
import importlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

DEV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev")


def run(module, *argv):
    sys.path.insert(0, DEV)
    try:
        m = importlib.import_module(module)
        old = sys.argv
        sys.argv = [module + ".py"] + [str(a) for a in argv]
        try:
            return m.main()
        finally:
            sys.argv = old
    finally:
        sys.path.remove(DEV)


def test_window_structures_solved_four_ways_and_in_mixed_batches(gpu):
    assert run("fuzz_solve", 120, 10) == 0


def test_marginalisation_of_the_fuzz_structures(gpu):
    assert run("fuzz_marg", 150, 0) == 0


def test_time_offset_windows_on_the_fuzz_structures(gpu):
    assert run("fuzz_td", 8, 0) == 0


def test_line_association_on_random_scenes(gpu):
    assert run("fuzz_lines", 20, 7) == 0


def test_preintegration_on_random_streams(gpu):
    assert run("fuzz_preint", 15, 0) == 0


def test_gauge_fix_on_random_windows(gpu):
    assert run("fuzz_gauge", 600, 0) == 0


def test_factor_evaluators_on_random_inputs(gpu):
    assert run("fuzz_factors", 500, 0) == 0


def test_native_estimator_on_stress_streams(gpu):
    assert run("fuzz_estimator", 6, 0, 40) == 0
