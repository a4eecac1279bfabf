#!/usr/bin/env python3
"""Per-wave time stamps of the ray caster in the native frame loop (a -DVH_KNOCKOUT=41 build through VH_LIB_PATH): where
the launch's time goes -- the table build, the march, the waves that end last."""
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
    nw = 4 * 9000
    raw = lib.download(hd.d_hashCompactified + 16 * (ne // 2), np.uint32, 12 * nw).reshape(nw, 12)
    ok = (raw[:, 7] >> 16) == 0x5741
    raw = raw[ok]
    t0 = int(raw[:, 0].min())
    st, built, en = (raw[:, 0] - t0) / 100.0, (raw[:, 1] - t0) / 100.0, (raw[:, 2] - t0) / 100.0
    half = raw[:, 3] >> 24
    cost = raw[:, 4]
    hw, xcc = raw[:, 5], raw[:, 6] & 0xf
    simd = (xcc.astype(np.int64) * 100000 + ((hw >> 13) & 7) * 10000 + ((hw >> 12) & 1) * 1000 + ((hw >> 8) & 0xf) * 10 + ((hw >> 4) & 3))
    q = lambda a: [round(float(x), 2) for x in np.percentile(a, [0, 10, 50, 90, 99, 100])]
    out = dict(waves=int(ok.sum()), start_us=q(st), table_built_us=q(built - st), march_us=q(en - built), life_us=q(en - st), end_us=q(en),
               halves=dict(whole=int((half == 0).sum()), near=int((half == 1).sum()), far=int((half == 2).sum())))
    for nm, m in (("whole", half == 0), ("near", half == 1), ("far", half == 2)):
        if m.any():
            out[nm] = dict(life_us=q((en - st)[m]), end_us=q(en[m]), cost=q(cost[m]))
    u, inv = np.unique(simd, return_inverse=True)
    endmax = np.zeros(len(u)); np.maximum.at(endmax, inv, en)
    out["simd_end_us"] = q(endmax)
    out["simds"] = len(u)
    late = np.argsort(en)[-12:]
    out["latest_waves"] = [dict(end=round(float(en[i]), 2), start=round(float(st[i]), 2), built=round(float(built[i] - st[i]), 2), half=int(half[i]), cost=int(cost[i])) for i in late]
    listlen = (raw[:, 8] & 0xffff).astype(np.float64)
    incomplete = ((raw[:, 8] >> 16) & 1).astype(np.float64)
    out["list_length"] = q(listlen)
    out["incomplete_tables"] = int(incomplete.sum())
    # what predicts a SIMD's finishing time?  least squares over the SIMDs
    X = np.stack([np.bincount(inv, weights=cost.astype(np.float64)), np.bincount(inv, weights=listlen), np.bincount(inv, weights=incomplete),
                  np.bincount(inv, weights=(built - st)), np.ones(len(u))], axis=1)
    coef, res, rk, sv = np.linalg.lstsq(X, endmax, rcond=None)
    pred = X @ coef
    out["fit_end_us"] = dict(per_cost=round(float(coef[0]), 4), per_list_entry=round(float(coef[1]), 4), per_incomplete=round(float(coef[2]), 3),
                             per_us_of_table_build=round(float(coef[3]), 3), const=round(float(coef[4]), 2),
                             r=round(float(np.corrcoef(pred, endmax)[0, 1]), 3),
                             r_cost_only=round(float(np.corrcoef(X[:, 0], endmax)[0, 1]), 3), r_list_only=round(float(np.corrcoef(X[:, 1], endmax)[0, 1]), 3),
                             r_build_only=round(float(np.corrcoef(X[:, 3], endmax)[0, 1]), 3))
    nw_s = np.bincount(inv)
    sumcost = np.bincount(inv, weights=cost.astype(np.float64))
    out["waves_per_simd"] = np.bincount(nw_s).tolist()
    out["corr_end_sumcost"] = round(float(np.corrcoef(endmax, sumcost)[0, 1]), 3)
    out["corr_end_nwaves"] = round(float(np.corrcoef(endmax, nw_s)[0, 1]), 3)
    out["sumcost_per_simd"] = q(sumcost)
    for k in np.unique(nw_s):
        m = nw_s == k
        out[f"simds_with_{k}_waves"] = dict(n=int(m.sum()), end_us=q(endmax[m]), sumcost=q(sumcost[m]))
    np.save(os.path.join(ROOT, "gpurun_out", "render_stamps.npy"), raw)
    print(json.dumps(out))

main()
