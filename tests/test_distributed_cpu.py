"""N > 1 path of bench.py on CPU: world_size 2 over gloo.  Checks the
measurement contract (barrier + max-over-ranks time, whole-job unit count =
sum over ranks) and that ranks run independent, different streams."""
import json
import os
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_weak_scaling_harness():
    port = free_port()
    warmup, steps, world = 1, 3, 2
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "result.json")
        procs = []
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), out, str(warmup), str(steps)],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        logs = [p.communicate(timeout=300)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(logs)
        res = json.load(open(out))
    assert res["world"] == 2
    assert res["total_frames"] == world * steps  # whole-job count: sum over ranks
    assert res["elapsed"] > 0 and abs(res["value"] - res["total_frames"] / res["elapsed"]) < 1e-9
    a, b = res["per_rank"]
    assert a[2] == b[2] == warmup + steps       # every rank ran its own frames
    assert a[1] > 10 and b[1] > 10              # both reconstructed something
    assert a[0] != b[0]                          # phase-shifted orbits: different scenes
