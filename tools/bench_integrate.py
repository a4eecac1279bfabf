#!/usr/bin/env python3
"""Times the fused integrate kernel alone on a dense scene (S2: camera inside a 3 m sphere, 1 cm voxels, cfg3 tables):
the scene is built with the frame loop, then vh_integrate_fused is launched back to back on the last frame's block list
(flags 0: nothing is freed, so every launch sees the same list).  With VH_LIB_PATH set, a measurement build of the
library is timed (voxelhashing_amd.build --out scratch/... -DVH_KNOCKOUT=n)."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--scene", default="S2")
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--gc", action="store_true")
    ap.add_argument("--in-loop", action="store_true", help="no repeated launches: the time stamps of the frame loop's last integrate launch (a -DVH_KNOCKOUT=9 build)")
    a = ap.parse_args()
    import torch
    from voxelhashing_amd import engine as E, lib, synth, vhtypes as T
    L = lib.load()
    cfg = dict(synth.CONFIGS[a.config], scene=a.scene)
    hp, cp, rp = synth.config_params(cfg)
    spheres, inside, radius = synth.scene(a.scene)
    scene = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=False, gc=True, starve=15))
    frame = E.DepthFrame(cp)
    for k in range(a.frames):
        pose = synth.orbit_pose(k, 1000, radius)
        E.synth_frame(spheres, inside, pose, cp, out=frame)
        scene.integrate(pose, frame, cp, None)
    n = scene.getNumOccupiedBlocks()
    hd, hpp = scene.getHashData(), scene.getHashParams()
    # the packed frame of the last integrate() is the scene's own buffer: reuse it through a job
    job = scene.integrateAhead(pose, frame, cp, None)
    packed = job.contents.d_packedFrame if job is not None else None
    check = lib.check
    check(L.vh_alloc_job(job, None), "alloc")
    check(L.vh_compactify_job(job, None), "compactify")
    torch.cuda.synchronize()
    flags = 1 if a.gc else 0
    out = {}
    for what, pk in (() if a.in_loop else (("unpacked", None), ("packed", packed))):
        for _ in range(10):
            check(L.vh_integrate_fused(C.byref(hd), C.byref(hpp), C.byref(frame.data), C.byref(cp), flags, 12345, None, 0, pk, None), "integrate")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream().cuda_stream
        e0.record()
        for i in range(a.launches):
            check(L.vh_integrate_fused(C.byref(hd), C.byref(hpp), C.byref(frame.data), C.byref(cp), flags, 20000 + i, None, 0, pk, st), "integrate")
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / a.launches
        b = n * 8212 + (8 if pk else 20) * cp.m_imageWidth * cp.m_imageHeight  # (bench.py stage_bytes: the packed frame is 8 B per pixel)
        out[what] = dict(us=round(us, 2), GBs=round(b / us / 1e3, 1), frac=round(b / us / 1e3 / 8000, 4))
    st = scene.getState()
    if st[9]:
        out["wave0"] = dict(cycles=int(st[8]), realtime_ticks=int(st[9]), MHz=round(100.0 * st[8] / st[9], 1), rounds=int(st[10]), active_waves=int(st[11]))
        # a -DVH_KNOCKOUT=9 build leaves every wave's {start, end, hw id, xcc} behind (the last launch's)
        import numpy as np
        ne = hpp.m_hashNumBuckets * T.HASH_BUCKET_SIZE
        na = int(st[11])
        raw = lib.download(hd.d_hashCompactified + 16 * (ne // 2), np.uint32, 12 * na).reshape(na, 12)
        t0 = int(raw[:, 0].min())
        start, end = (raw[:, 0] - t0) / 100.0, (raw[:, 1] - t0) / 100.0  # us
        xcc = raw[:, 3] & 0xf
        q = lambda a: [round(float(x), 2) for x in np.percentile(a, [0, 10, 50, 90, 100])]
        out["waves"] = dict(start_us=q(start), end_us=q(end), life_us=q(end - start))
        ph = (raw[:, 4:10].astype(np.int64) - t0) / 100.0
        two = raw[:, 9] != 0
        out["phases_two_block_waves"] = {k: q(v[two]) for k, v in dict(staged1=ph[:, 0] - start, voxels1=ph[:, 1] - ph[:, 0], compute1=ph[:, 2] - ph[:, 1],
                                                                        staged2=ph[:, 3] - ph[:, 2], voxels2=ph[:, 4] - ph[:, 3], compute2=ph[:, 5] - ph[:, 4], rest=end - ph[:, 5]).items()}
        out["waves_alive_at_us"] = {str(t): int(((start <= t) & (end > t)).sum()) for t in (2, 4, 6, 8, 10, 11, 12, 13, 14, 15, 16, 18)}
        np.save(os.path.join(ROOT, "gpurun_out", "wave_stamps.npy"), raw)
    print(json.dumps(dict(lib=os.path.basename(lib.LIB_PATH), blocks=n, **out)))
    scene.integrateFinish(frame, cp)


if __name__ == "__main__":
    main()
