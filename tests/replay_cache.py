"""Test infrastructure: full-length EuRoC-trajectory replays driven by the CPU oracle back end, computed once per pytest session and
shared by the free-running gates (tests/test_gpu_replay.py: trajectory of the oracle replay) and the teacher-forced gates
(tests/test_gpu_teacher.py: every window the oracle replay solved, with its results).

The fifteen replays (five sequences x three line modes, ~12 s of one host core each) are pure CPU work: the first request starts all
of them on a pool of worker PROCESSES (`spawn`: fresh interpreters that never touch the GPU) and every request then waits for its own;
if the pool cannot be started the replay is computed in this process."""
import functools
import os
import sys

import replay

# what a configs[3] replay is in this suite: the whole 36 s ground-truth excerpt, 60 tracked features per frame, 8 line tracks per frame
# (none / every observation given its 3D partner / the 2D-3D association in the loop)
MODES = {"none": dict(max_lines=0, associate=False), "given": dict(max_lines=8, associate=False), "associate": dict(max_lines=8, associate=True)}
FRAMES, START_S, FEATURES = 355, 0.5, 60

_pool = None
_futures = {}


# a platform that stands still for 11.5 s in the middle of the replay (replay.simulate_stream(hold=...)): 115 non-keyframes in a row, the newest
# interval's pre-integration grows beyond 10 s and is left out of the window (estimator.cpp:1726), then sits in the MIDDLE of the window for
# ten more keyframes (a broken IMU chain next to a prior) and is left out of MARGIN_OLD when it reaches frame 0 (:1933)
STANDSTILL = "standstill"
STANDSTILL_ARGS = dict(seed=3, n_frames=186, max_features=40, max_lines=4, hold=(2.5, 11.5))


@functools.lru_cache(maxsize=None)
def stream_of(seq, mode):
    if seq == STANDSTILL:
        a = dict(STANDSTILL_ARGS)
        return replay.simulate_stream(a.pop("seed"), a.pop("n_frames"), **a)
    return replay.simulate_stream_euroc(seq, FRAMES, start_s=START_S, max_features=FEATURES, **MODES[mode])


def _compute(seq, mode):
    from test_gpu_teacher import Teacher
    teacher = Teacher()
    ref = replay.run(stream_of(seq, mode), teacher, num_iterations=8)
    return dict(ref=ref, rec=teacher.rec)


def _worker(paths, seq, mode):
    for p in paths:
        if p not in sys.path:
            sys.path.insert(0, p)
    return _compute(seq, mode)


def prefetch():
    """start the worker pool now (tests/conftest.py calls this at the start of a GPU session that holds full-length replay tests, before the
    process touches the device: the replays -- the association-in-the-loop ones take a minute of one core each -- then run beside the
    other GPU tests)"""
    _start_pool()


def _start_pool():
    """all fifteen replays at once on worker processes; returns False when that is not possible here"""
    global _pool
    if _pool is not None:
        return True
    if os.environ.get("TCV_TEST_NO_POOL"):
        return False
    try:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        _pool = ProcessPoolExecutor(max_workers=max(1, min(8, cores)), mp_context=mp.get_context("spawn"))
        paths = [p for p in sys.path if p]
        # one thread per worker (the workers are started by the submits below and inherit the environment): eight BLAS thread pools on
        # eight cores only get in each other's way
        keys = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")
        saved = {k: os.environ.get(k) for k in keys}
        try:
            for k in keys:
                os.environ[k] = "1"
            for mode in MODES:      # (the order the tests ask in: tests/test_gpu_replay.py by mode, then tests/test_gpu_teacher.py)
                for seq in replay.EUROC_SEQUENCES:
                    _futures[(seq, mode)] = _pool.submit(_worker, paths, seq, mode)
            _futures[(STANDSTILL, "given")] = _pool.submit(_worker, paths, STANDSTILL, "given")
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        return True
    except Exception as e:      # noqa: BLE001 -- no pool: compute in this process
        print("replay_cache: worker pool unavailable (%s), computing in-process" % e)
        _pool = None
        return False


@functools.lru_cache(maxsize=None)
def teacher_replay(seq, mode):
    """dict(stream, ref = the oracle replay's trajectory / log, rec = the per-window records of test_gpu_teacher.Teacher)"""
    out = None
    if _start_pool() and (seq, mode) in _futures:
        try:
            out = _futures.pop((seq, mode)).result(timeout=1800)
        except Exception as e:      # noqa: BLE001
            print("replay_cache: worker failed (%s), computing %s / %s in-process" % (e, seq, mode))
    if out is None:
        out = _compute(seq, mode)
    out["stream"] = stream_of(seq, mode)
    return out
