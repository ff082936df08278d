"""world_size-2 `gloo` tests (CPU) of the multi-process logic bench.py uses at N > 1: rank-offset window shards with no
overlap, a barrier, MAX-over-ranks time and SUM-over-ranks windows; the self-launch path of `python bench.py --gpus N` (the parent
starts N fresh ranks as a child process before any GPU call); stream sharding s mod G of the replay mode.  The data path itself has
no collective."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import bench
rank, world, local, dist = bench.dist_setup(2, backend="gloo")
first = bench.shard_ids(rank, 64)
dist.barrier()
t, w = bench.reduce_stats(dist, 1.0 + 0.5 * rank, 64 * 3)
t2, w2, it2 = bench.reduce_stats(dist, 1.0, 10, extra=(80,))
print(json.dumps(dict(rank=rank, world=world, first=first, t=t, w=w, w2=w2, it2=it2, seen=bench.ranks_seen(dist), streams=bench.shard_streams(8, rank, world))))
dist.destroy_process_group()
"""


def test_two_rank_gloo_reduction(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e.decode()[-2000:]
        outs.append(json.loads(o.decode().strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [d["world"] for d in outs] == [2, 2]
    assert outs[1]["first"] - outs[0]["first"] == 64                      # disjoint shards, fixed work per rank (weak scaling)
    assert all(abs(d["t"] - 1.5) < 1e-12 for d in outs)                   # MAX over ranks
    assert all(d["w"] == 2 * 64 * 3 for d in outs)                        # whole-job window count
    assert all(d["w2"] == 20 and d["it2"] == 160 and d["seen"] == 2 for d in outs)
    assert outs[0]["streams"] == [0, 2, 4, 6] and outs[1]["streams"] == [1, 3, 5, 7]      # sequence s -> GPU s mod G


def _bench(args, env_extra, timeout=600):
    env = dict(os.environ, TCV_BENCH_DRY="1", **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


@pytest.mark.parametrize("mode", ["solve", "replay"])
def test_bench_self_launches_one_rank_per_gpu(mode):
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns torch.distributed.run as a child, two ranks rendezvous
    (gloo here, RCCL on the GPU box), and rank 0 prints ONE line whose n_gpus is the number of ranks the collective saw.  TCV_BENCH_DRY
    replaces the device work by a sleep -- the launch, rendezvous and reduction code is the real one."""
    rc, out, err = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", mode, "--windows", "16"], {})
    assert rc == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["steps"] == 3
    if mode == "solve":
        assert d["config"]["windows_per_gpu"] == 16 and abs(d["value"] * d["ms_per_step"] * 1e-3 - 2 * 16) < 1e-6      # whole-job count / MAX time
        assert d["scaling"] == "weak"
    else:
        assert d["config"]["streams"] == 8 and abs(d["value"] * d["ms_per_step"] * 1e-3 - 8) < 1e-6                    # 8 streams over 2 ranks
        assert d["scaling"] == "strong"


def test_replay_shards_an_odd_stream_count_over_two_ranks():
    """7 streams over 2 ranks (stream s -> rank s mod G): rank 0 advances 4, rank 1 advances 3, the line counts all 7, names
    the ranks the collective saw and carries every rank's own rate (a straggler is visible)"""
    rc, out, err = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", "replay", "--streams", "7"], {})
    assert rc == 0, err[-3000:]
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["streams_per_rank"] == [4, 3]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 7) < 1e-6
    assert len(d["per_rank_solves_per_s"]) == 2 and all(v > 0 for v in d["per_rank_solves_per_s"])
    assert d["per_rank_solves_per_s"][0] > d["per_rank_solves_per_s"][1]          # 4 against 3 windows per (equally long) dry step


def test_default_line_key_set():
    """the keys the default line promises beyond the contract's: present in the source of the line and, for the ones the dry
    run can produce, on the dry line"""
    rc, out, err = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--windows", "16"], {})
    assert rc == 0, err[-3000:]
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "ranks_seen", "per_rank_solves_per_s"):
        assert k in d, k
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ("roofline_fp64", "batch_sweep", "run_to_convergence", "mfma_busy", "valu_active", "replay_cpu_baseline", "cpu_baseline", "traffic_commit",
              # round 6: the roofline on SURVEY 8(d)'s unit (iteration) with the per-linearisation figure beside it, executed flops from the counters,
              # the wait split, and the staleness guard of the static counter figures
              "unit_of_work", "linearisations_per_launch", "frac_per_linearisation", "executed", "wait_split", "mean_latency_cycles", "stale", "csrc_sha16",
              "deployed_budget"):
        assert f'"{k}"' in src, k
    # the static counters belong to the kernel sources they were measured on: the guard's hash moves with any file under csrc/
    sys.path.insert(0, ROOT)
    import bench
    h = bench.csrc_sha16()
    assert len(h) == 16 and h == bench.csrc_sha16()
    cj = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    assert "commit" in cj and "kernel" in cj


def test_cpu_share_cuts_the_affinity_mask():
    sys.path.insert(0, ROOT)
    import bench
    cores = sorted(os.sched_getaffinity(0))
    if len(cores) < 2:
        pytest.skip("one core")
    pid = os.fork()
    if pid == 0:      # (in a child: the mask of the test process stays)
        try:
            a = bench.cpu_share(1, 2)
            ok = a == cores[len(cores) // 2: 2 * (len(cores) // 2)] and sorted(os.sched_getaffinity(0)) == a
            os._exit(0 if ok else 1)
        except BaseException:
            os._exit(2)
    assert os.waitpid(pid, 0)[1] == 0


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    rc, out, err = _bench(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and "WORLD_SIZE=1" in err and not out.strip()


def test_algorithmic_bytes_match_the_survey():
    sys.path.insert(0, ROOT)
    import bench
    # cfg 2 (200 pt, 0 line, no prior, L=50) 39 728 B; cfg 3 (200 pt, 40 line, prior n=75, X0=86, L=50) 89 056 B
    assert bench.algorithmic_bytes_per_iteration(10, 200, 0, 50, 0, 0) == 39728
    assert bench.algorithmic_bytes_per_iteration(10, 200, 40, 50, 75, 86) == 89056
    assert bench.BYTES_PER_ITERATION_CFG3 == 89056
