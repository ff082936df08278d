"""world_size-2 `gloo` test (CPU) of the multi-process logic bench.py uses at N > 1: rank-offset window shards with no
overlap, a barrier, MAX-over-ranks time and SUM-over-ranks windows.  The data path itself has no collective."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import bench
rank, world, local, dist = bench.dist_setup(2, backend="gloo")
first = bench.shard_ids(rank, 64)
dist.barrier()
t, w = bench.reduce_stats(dist, 1.0 + 0.5 * rank, 64 * 3)
print(json.dumps(dict(rank=rank, world=world, first=first, t=t, w=w)))
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


def test_algorithmic_bytes_match_the_survey():
    sys.path.insert(0, ROOT)
    import bench
    # cfg 2 (200 pt, 0 line, no prior, L=50) 39 728 B; cfg 3 (200 pt, 40 line, prior n=75, X0=86, L=50) 89 056 B
    assert bench.algorithmic_bytes_per_iteration(10, 200, 0, 50, 0, 0) == 39728
    assert bench.algorithmic_bytes_per_iteration(10, 200, 40, 50, 75, 86) == 89056
    assert bench.BYTES_PER_ITERATION_CFG3 == 89056
