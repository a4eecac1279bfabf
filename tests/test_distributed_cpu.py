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


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) starts two ranks itself, before
    anything touches a GPU, and relays rank 0's line: n_gpus = 2 and the whole-job frame count.  --standin replaces
    the GPU workload by a CPU delay (this container has no GPU); the launch path, the rendezvous, the barriers and
    the reductions are bench.py's own."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--standin"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 5 and res["scaling"] == "weak"
    # two ranks x 5 frames of 2 ms each, timed as the slower rank: 10 frames in >= 10 ms
    assert 0 < res["value"] <= 2 * 5 / 0.010 + 1
    assert "STANDIN" in res["data"]
    # every rank's own rate is in the line (a slow rank must be visible, not only the maximum)
    assert len(res["per_rank_frames_per_s"]) == 2 and all(0 < v <= 1 / 0.002 + 1 for v in res["per_rank_frames_per_s"])


def test_a_failing_rank_fails_the_launch():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # without --standin a rank needs a GPU: on this machine every rank exits with an error, and so must the launcher
    import torch
    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0


import pytest  # noqa: E402
import time  # noqa: E402


@pytest.mark.parametrize("where", ["before-init", "before-barrier"])
def test_a_rank_that_dies_early_stops_the_launch_within_seconds(where):
    """rank 1 gives up before the rendezvous / before the first barrier: rank 0 would sit there until the rendezvous or
    the collective times out (minutes).  The launcher watches every child, stops the others and exits non-zero at
    once, with the failing rank's stderr."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--standin",
                        "--standin-fail", f"1:{where}"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    dt = time.time() - t0
    assert p.returncode != 0
    assert dt < 45, f"the launcher took {dt:.0f} s to notice"
    err = p.stderr.decode()
    assert "rank 1 of 2 exited" in err and "gives up" in err, err
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")], "no result line from a failed run"
