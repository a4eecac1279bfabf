#!/usr/bin/env python3
"""Per-workgroup time stamps of the frame loop's second launch (normals + compactify + interval splat + the schedule
workgroup), from a -DVH_KNOCKOUT=42 build through VH_LIB_PATH: which rider ends the launch, and when."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def main():
    import torch
    from voxelhashing_amd import engine as E, lib, synth, vhtypes as T
    cfg_name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    cfg = dict(synth.CONFIGS[cfg_name])
    hp, cp, rp = synth.config_params(cfg)
    spheres, inside, radius = synth.scene(cfg["scene"])
    scene, ray = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=False, gc=True, starve=15)), E.CUDARayCastSDF(rp)
    n = 120
    poses = [synth.orbit_pose(k, 1000, radius) for k in range(n)]
    frames = [E.synth_frame(spheres, inside, p, cp) for p in poses]
    recon = E.Reconstruction(scene, ray, None, cp)
    seq = E.Reconstruction.makeFrames(poses, [f.depth_ptr for f in frames], [f.color_ptr for f in frames])
    recon.run(seq, 0, n)
    recon.synchronize()
    hd, hpp = scene.getHashData(), scene.getHashParams()
    ne = hpp.m_hashNumBuckets * T.HASH_BUCKET_SIZE
    raw = lib.download(hd.d_hashCompactified + 16 * (ne // 2), np.uint32, 4 * 12000).reshape(12000, 4)
    raw = raw[raw[:, 3] == 0x5742]
    t0 = int(raw[:, 0].min())
    st, en, kind = (raw[:, 0] - t0) / 100.0, (raw[:, 1] - t0) / 100.0, raw[:, 2]
    q = lambda a: [round(float(x), 2) for x in np.percentile(a, [0, 10, 50, 90, 100])]
    out = dict(groups=int(len(raw)))
    for k, nm in ((1, "schedule"), (2, "splat"), (3, "compactify"), (4, "normals"), (6, "pass_wait"), (5, "pass")):
        m = kind == k
        if m.any():
            out[nm] = dict(n=int(m.sum()), start_us=q(st[m]), end_us=q(en[m]), life_us=q((en - st)[m]))
    ph = lib.download(hd.d_hashCompactified + 16 * (ne // 2 + 8192), np.uint32, 4 * 2 * 512).reshape(512, 2, 4)
    ph = ph[ph[:, 0, 3] == 0x5743]
    if len(ph): # a compactify workgroup that kept something: bits + gather | slots | queue -> list atomic | boxes + stores | all stores out
        t = np.stack([ph[:, 0, 0], ph[:, 0, 1], ph[:, 0, 2], ph[:, 1, 0], ph[:, 1, 1], ph[:, 1, 2]], axis=1).astype(np.int64)
        d = np.diff(t, axis=1) / 100.0
        out["compactify_phases_us"] = dict(n=int(len(ph)), kept=q(ph[:, 1, 3] & 0xffff), buckets=q(ph[:, 1, 3] >> 16),
                                           gather=q(d[:, 0]), slots=q(d[:, 1]), atomic=q(d[:, 2]), boxes=q(d[:, 3]), drain=q(d[:, 4]))
    print(json.dumps(out))

main()
