#!/usr/bin/env python3
"""Measures the marching-cubes extraction on the bench scene: cfg2 after `--frames` frames of the
S1 orbit (or cfg3's 1 cm voxels with --config cfg3), `--reps` extractions, HIP-event time of pass 1 + pass 2 on the
stream, and the CPU oracle on the same scene for comparison.  One JSON line.

    python tools/bench_mesh.py [--config cfg2] [--frames 100] [--reps 20] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    from voxelhashing_amd import engine as E, synth, vhtypes as T
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    cfg = dict(synth.CONFIGS[args.config])
    cfg["num_sdf_blocks"] = min(cfg["num_sdf_blocks"], 1 << 18)
    hp, cp, rp = synth.config_params(cfg)
    spheres, inside, radius = synth.scene(cfg["scene"])
    scene = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=True, gc=True))  # offline: a reproducible block set
    frame = E.DepthFrame(cp)
    poses = [synth.orbit_pose(k, 1000, radius) for k in range(args.frames)]
    for pose in poses:
        E.synth_frame(spheres, inside, pose, cp, out=frame)
        scene.integrate(pose, frame, cp, None)
    mp = T.make_marching_cubes_params(hp, 1 << 22)
    mc = E.CUDAMarchingCubesHashSDF(mp)
    hd, hpp = scene.getHashData(), scene.getHashParams()
    mc.extractIsoSurfaceWithoutCopy(hd, hpp)  # warm-up
    torch.cuda.synchronize()
    counts = mc.counts()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)  # default stream = the engine's
    ev0.record()
    for _ in range(args.reps):
        mc.extractIsoSurfaceWithoutCopy(hd, hpp)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / args.reps
    nblk, ntri = counts["occupied_blocks"], counts["triangles"]
    # algorithmic bytes: every voxel of every allocated block once (8 B) + its entry (20 B) + the triangles written (72 B)
    alg = nblk * (512 * 8 + 20) + ntri * 72
    out = dict(metric="marching cubes: allocated blocks/s (extractIsoSurface pass1+pass2, incl. the blocking block count)",
               value=round(nblk / (ms * 1e-3), 1), unit="blocks/s", ms_per_extraction=round(ms, 4), blocks=nblk, triangles=ntri,
               voxels_per_s=round(nblk * 512 / (ms * 1e-3)), algorithmic_bytes=alg,
               config=dict(workload=f"{args.config} after {args.frames} frames of the S1 orbit", voxel_size=hp.m_virtualVoxelSize))
    if not args.no_cpu:
        from oracle import oracle as O
        o = O.OracleScene(hp, cp, None, T.make_scene_options(offline=True, gc=True))
        n_cpu = min(args.frames, 12)  # a bounded sample of the same orbit: the oracle integrates at ~2.5 frames/s
        for pose in poses[:n_cpu]:
            d, c = O.synth_frame(spheres, inside, pose, cp)
            o.integrate(pose, d, c)
        t0 = time.perf_counter()
        tris, n = o.extract_iso_surface(mp)
        dt = time.perf_counter() - t0
        nb = len(o.state()["positions"])
        out["cpu_baseline"] = dict(value=round(nb / dt, 1), unit="blocks/s", cores=1, kind="port",
                                   sample=f"oracle scene after {n_cpu} frames: {nb} blocks, {n} triangles, {dt:.2f} s")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
