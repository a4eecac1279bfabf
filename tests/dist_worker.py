"""Worker of tests/test_distributed_cpu.py: one rank of bench.py's multi-stream
harness on the gloo backend, with the CPU oracle standing in for the GPU engine
(tests may use the oracle; bench.py's GPU path never does)."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402
from voxelhashing_amd import synth, vhtypes as T  # noqa: E402


class OracleWorkload:
    """same interface as bench.GpuWorkload: one independent scene per rank, orbit phase-shifted by rank"""

    def __init__(self, rank, n_frames):
        self.hp = T.make_hash_params(1 << 10, 1 << 9, **synth.PARAM_SETS["P4"])
        self.cp = T.make_depth_camera_params(48, 36)
        self.sc = O.OracleScene(self.hp, self.cp, options=T.make_scene_options(offline=True, gc=True))
        phase = 2.0 * math.pi * rank / 8.0
        self.poses = [synth.orbit_pose(k, 1000, synth.S1_ORBIT_RADIUS, phase) for k in range(n_frames)]
        self.frames_done = 0

    def run(self, k0, k1):
        for k in range(k0, k1):
            if k > 0:
                self.sc.render(self.poses[k - 1])
            d, c = O.synth_frame(synth.S1_SPHERES, 0, self.poses[k], self.cp)
            self.sc.integrate(self.poses[k], d, c)
            self.frames_done += 1


def main():
    out_path, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world, local_rank, dist = bench.dist_setup(world_hint := int(os.environ["WORLD_SIZE"]))
    assert dist is not None and dist.get_backend() == "gloo" and world == world_hint
    dev = torch.device("cpu")
    wl = OracleWorkload(rank, warmup + steps)
    elapsed, total = bench.run_timed(wl, warmup, steps, dist, dev)
    # every rank holds a different scene: gather a checksum of the block positions
    pos = wl.sc.state()["positions"]
    chk = torch.tensor([float(np.abs(pos).sum()), float(len(pos)), float(wl.frames_done)], dtype=torch.float64)
    gathered = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, chk)
    if rank == 0:
        json.dump(dict(world=world, elapsed=elapsed, total_frames=total, value=total / elapsed,
                       per_rank=[g.tolist() for g in gathered]), open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
