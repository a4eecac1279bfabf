#!/usr/bin/env python3
"""Per-workgroup time stamps of the fused integrate launch in its workgroup-per-block shape (cfg2: a few hundred blocks),
from a -DVH_KNOCKOUT=43 build through VH_LIB_PATH: how long the chain is and which workgroups end the launch."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def main():
    import torch
    from voxelhashing_amd import engine as E, lib, synth, vhtypes as T
    cfg = dict(synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"])
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
    raw = lib.download(hd.d_hashCompactified + 16 * (ne // 2), np.uint32, 4 * 2048).reshape(2048, 4)
    raw = raw[(raw[:, 3] >> 16) == 0x5743]
    t0 = int(raw[:, 0].min())
    st, mid, en = (raw[:, 0] - t0) / 100.0, (raw[:, 1] - t0) / 100.0, (raw[:, 2] - t0) / 100.0
    freed = raw[:, 3] & 3
    q = lambda a: [round(float(x), 2) for x in np.percentile(a, [0, 10, 50, 90, 100])]
    out = dict(groups=int(len(raw)), start_us=q(st), blended_us=q(mid - st), rest_us=q(en - mid), end_us=q(en), decided=int((freed > 0).sum()), freed=int((freed == 2).sum()))
    for k, nm in ((0, "kept"), (1, "decided_not_freed"), (2, "freed")):
        m = freed == k
        if m.any():
            out[nm] = dict(n=int(m.sum()), end_us=q(en[m]), rest_us=q((en - mid)[m]))
    print(json.dumps(out))

main()
