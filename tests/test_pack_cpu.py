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


def _plan_ints(tcv, W):
    import ctypes as C
    L = tcv.lib()
    L.tcv_problem_plan_ints.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    n = C.c_int()
    tcv.check(L.tcv_problem_plan_ints(W.h, None, 0, C.byref(n)))
    out = np.zeros(n.value, np.int32)
    tcv.check(L.tcv_problem_plan_ints(W.h, out.ctypes.data_as(C.POINTER(C.c_int)), n.value, C.byref(n)))
    return out


def _replay_like_windows():
    """window structures as a live estimator produces them: tracks of different lengths starting in different frames, a few line
    factors per frame, with and without a prior, ragged frame counts"""
    pre, main, z = golden_windows()
    rng = np.random.Generator(np.random.PCG64(5))
    wins = []
    for n_land in (3, 40, 90, 140):
        big = synth.window_at(synth.make_windows(900 + n_land, 1, n_landmarks=n_land), 0)
        w = dict(main)
        for k in ("proj", "lam"):
            w[k] = big[k]
        # ragged tracks: drop a random tail of every landmark's observations (keeps at least one factor per landmark)
        pr = {k: np.asarray(v) for k, v in w["proj"].items()}
        keep = np.ones(len(pr["landmark"]), bool)
        for l in range(n_land):
            idx = np.nonzero(pr["landmark"] == l)[0]
            cut = rng.integers(1, len(idx) + 1)
            keep[idx[cut:]] = False
        w["proj"] = {k: (v[keep] if isinstance(v, np.ndarray) and v.shape[:1] == keep.shape else v) for k, v in pr.items()}
        wins.append(w)
        wins.append({k: v for k, v in w.items() if k != "prior"})
    wins.append(synth.window_at(synth.make_windows(600, 1, n_landmarks=200), 0))
    wins += [sub_window(synth.window_at(synth.make_windows(400, 1), 0), f) for f in (3, 6)]
    return wins


def test_fast_packer_builds_the_reference_plans_int_by_int(tcv):
    """the fast gather-program builder with the cached camera half (tcv_pack.cpp fast_prog / get_cam) against the generic builder
    (DestList + emit_rows, nothing cached): identical plans -- header and int pool -- for the chain and the dense layout, single-workgroup and
    cooperative chunking, first build and cache hit"""
    import os
    L = tcv.lib()
    wins = _replay_like_windows()
    old = os.environ.get("TCV_PLAN_COOP")
    built = 0
    try:
        for coop in ("0", "4"):
            os.environ["TCV_PLAN_COOP"] = coop
            for variant in (0, 1):
                L.tcv_set_solver_variant(variant)
                for i, w in enumerate(wins):
                    W = tcv.Window(w)

                    def plan():      # (a window that does not fit a layout must fail the same way on both paths)
                        try:
                            return _plan_ints(tcv, W)
                        except tcv.TcvError as e:
                            return np.frombuffer(str(e).encode(), np.uint8).astype(np.int32)
                    L.tcv_set_packer_reference(1)
                    ref = plan()
                    L.tcv_set_packer_reference(0)
                    fast, again = plan(), plan()
                    built += ref.size > 1000
                    assert ref.shape == fast.shape and (ref == fast).all() and (fast == again).all(), (coop, variant, i)
    finally:
        L.tcv_set_packer_reference(0)
        L.tcv_set_solver_variant(0)
        if old is None:
            os.environ.pop("TCV_PLAN_COOP", None)
        else:
            os.environ["TCV_PLAN_COOP"] = old
    assert built >= 30


def test_camera_half_is_cached_across_visual_structures(tcv):
    """windows that differ in their landmark tracks only share the camera half of their plans (IMU tables, chain records, prior tables)"""
    import ctypes as C
    L = tcv.lib()
    pre, main, z = golden_windows()
    st0 = (C.c_longlong * 4)()
    tcv.check(L.tcv_plan_cache_stats(st0))
    for n_land in (31, 32, 33, 34):
        big = synth.window_at(synth.make_windows(1200 + n_land, 1, n_landmarks=n_land), 0)
        w = dict(main)
        for k in ("proj", "lam"):
            w[k] = big[k]
        tcv.Window(w).plan_stats()
    st1 = (C.c_longlong * 4)()
    tcv.check(L.tcv_plan_cache_stats(st1))
    assert st1[1] - st0[1] == 4          # four new whole-plan structures ...
    assert st1[3] - st0[3] <= 1          # ... on at most one new camera half
    assert st1[2] - st0[2] >= 3


def test_a_structure_enters_the_plan_cache_at_its_second_appearance(tcv):
    """one-off structures (a live estimator's windows) are built for their caller alone; a structure that comes back is kept from then on;
    the plan is the same ints whichever of the three ways it came about"""
    import ctypes as C
    L = tcv.lib()
    pre, main, z = golden_windows()
    big = synth.window_at(synth.make_windows(90417, 1, n_landmarks=37), 0)
    w = dict(main)
    for k in ("proj", "lam"):
        w[k] = big[k]
    st = [(C.c_longlong * 4)() for _ in range(4)]
    tcv.check(L.tcv_plan_cache_stats(st[0]))
    for i in range(3):
        tcv.Window(w).plan_stats()                    # (packs the window once)
        tcv.check(L.tcv_plan_cache_stats(st[i + 1]))
    hits = [st[i + 1][0] - st[i][0] for i in range(3)]
    misses = [st[i + 1][1] - st[i][1] for i in range(3)]
    assert (misses[0], hits[0]) == (1, 0)             # first sight: built, not kept
    assert (misses[1], hits[1]) == (1, 0)             # second sight: built again, kept
    assert (misses[2], hits[2]) == (0, 1)             # from the cache
    cached = _plan_ints(tcv, tcv.Window(w))
    L.tcv_set_packer_reference(1)                     # (the reference builder bypasses the cache)
    try:
        fresh = _plan_ints(tcv, tcv.Window(w))
    finally:
        L.tcv_set_packer_reference(0)
    assert np.array_equal(cached, fresh)


def test_items_of_a_parallel_section_are_each_run_once(tcv):
    """tcv_problems_pack_bench packs its windows through the worker pool's one-at-a-time claim (parallel_items) and through the fixed-share
    split: every window ends up packed (a return code per window), on 1 .. 8 threads, frames of 5"""
    import ctypes as C
    import os
    L = tcv.lib()
    L.tcv_problems_pack_bench.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    Ws = [tcv.Window(w) for w in _replay_like_windows()[:12]]
    arr = (C.c_void_p * len(Ws))(*[w.h for w in Ws])
    s = C.c_double()
    old = os.environ.get("TCV_PACK_BENCH_FRAME")
    os.environ["TCV_PACK_BENCH_FRAME"] = "5"
    try:
        for nt in (1, 3, 8):
            tcv.check(L.tcv_problems_pack_bench(arr, len(Ws), nt, 0, C.byref(s)))
            assert s.value > 0
    finally:
        if old is None:
            os.environ.pop("TCV_PACK_BENCH_FRAME", None)
        else:
            os.environ["TCV_PACK_BENCH_FRAME"] = old


def test_a_worker_threads_error_text_reaches_the_caller(tcv):
    """tcv_last_error() is per thread: an error raised inside a parallel section (on one of the library's worker threads) has to be carried
    over to the thread that made the call.  Twelve windows on eight threads, one of them too large for the LDS-resident solver."""
    import ctypes as C
    import threading
    L = tcv.lib()
    L.tcv_problems_pack_bench.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    Ws = [tcv.Window(w) for w in _replay_like_windows()[:12]]
    # 13 frames: camera tangent dim 201 > 175, refused at the plan stage with a message (tests/test_abi_cpu.py)
    bad = C.c_void_p(); L.tcv_problem_create(C.byref(bad))
    poses = np.zeros((13, 7)); poses[:, 6] = 1; sbs = np.zeros((13, 9))
    for i in range(13):
        L.tcv_problem_add_parameter_block(bad, tcv.dptr(poses[i]), 7, tcv.TCV_PARAM_POSE)
        L.tcv_problem_add_parameter_block(bad, tcv.dptr(sbs[i]), 9, tcv.TCV_PARAM_EUCLIDEAN)
    hs = [w.h for w in Ws]; hs[7] = bad
    arr = (C.c_void_p * len(hs))(*hs)
    s = C.c_double()
    res = {}

    def call():      # from a fresh thread: its own error text starts empty, so whatever it reads afterwards was carried over
        res["before"] = L.tcv_last_error()
        res["rc"] = L.tcv_problems_pack_bench(arr, len(hs), 8, 0, C.byref(s))
        res["msg"] = L.tcv_last_error()

    for _ in range(4):      # (which thread claims window 7 is a race: a few rounds)
        t = threading.Thread(target=call); t.start(); t.join()
        assert res["before"] == b""
        assert res["rc"] == tcv.TCV_ERR_TOO_LARGE
        assert res["msg"] != b"", res
    L.tcv_problem_destroy(bad)
