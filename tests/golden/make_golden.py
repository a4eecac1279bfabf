#!/usr/bin/env python3
"""Generates the fixtures in this directory with the CPU oracle (oracle/).

They are REGRESSION vectors produced by this repository's own oracle -- the
reference ships no golden data for this path and cannot
be run in this image -- so they pin the oracle against accidental edits and
let the GPU path be checked without the oracle; they are not reference outputs.

    python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from voxelhashing_amd import synth, vhtypes as T  # noqa: E402

CASES = {
    # name: (width, height, params, buckets, sdf blocks, scene, frame indices of a 100-frame orbit, gc, starve)
    "s1_64x48_p4": (64, 48, "P4", 1 << 12, 1 << 10, "S1", [0, 1, 2, 3], True, 2),
    "s1_48x36_p2_gradients": (48, 36, "P2", 1 << 12, 1 << 11, "S1", [10, 11], False, 15),
}


def run_case(name):
    W, H, ps, nb, nblk, scene, frames, gc, starve = CASES[name]
    hp = T.make_hash_params(nb, nblk, **synth.PARAM_SETS[ps])
    cp = T.make_depth_camera_params(W, H)
    rp = T.make_raycast_params(hp, cp, use_gradients=name.endswith("gradients"))
    opt = T.make_scene_options(offline=True, gc=gc, starve=starve)
    spheres, inside, radius = synth.scene(scene)
    sc = O.OracleScene(hp, cp, rp, opt)
    out = dict(width=W, height=H, params=ps, num_buckets=nb, num_sdf_blocks=nblk, scene=scene,
               frames=np.array(frames), gc=gc, starve=starve, use_gradients=int(rp.m_useGradients))
    last = None
    for i, k in enumerate(frames):
        pose = synth.orbit_pose(k, 100, radius)
        depth, color = O.synth_frame(spheres, inside, pose, cp)
        if last is not None:
            r = sc.render(last)
            for m in ("depth", "depth4", "normals", "colors"):
                out[f"f{i}_ray_{m}"] = r[m]
        sc.integrate(pose, depth, color)
        s = sc.state()
        out[f"f{i}_depth_in"] = depth
        out[f"f{i}_positions"] = s["positions"]
        out[f"f{i}_voxels"] = s["voxels"]
        out[f"f{i}_heap_free"] = s["heap_free"]
        out[f"f{i}_num_in_frustum"] = sc.hp.m_numOccupiedBlocks
        last = pose
    return out


def sorted_triangles(tris):
    """canonical form of a triangle soup: the 72-byte records in byte order"""
    v = np.ascontiguousarray(tris).view(np.dtype((np.void, T.TRIANGLE_DTYPE.itemsize))).ravel()
    return np.ascontiguousarray(tris)[np.argsort(v, kind="stable")]


MC_CASES = {
    # name: (width, height, params, buckets, sdf blocks, scene, frames of a 100-frame orbit, thresh factor)
    "mc_s1_64x48_p4": (64, 48, "P4", 1 << 12, 1 << 10, "S1", [0, 1, 2, 3], 10.0),
}


def run_mc_case(name):
    W, H, ps, nb, nblk, scene, frames, tf = MC_CASES[name]
    hp = T.make_hash_params(nb, nblk, **synth.PARAM_SETS[ps])
    cp = T.make_depth_camera_params(W, H)
    spheres, inside, radius = synth.scene(scene)
    sc = O.OracleScene(hp, cp, None, T.make_scene_options(offline=True, gc=False))
    for k in frames:
        pose = synth.orbit_pose(k, 100, radius)
        depth, color = O.synth_frame(spheres, inside, pose, cp)
        sc.integrate(pose, depth, color)
    mp = T.make_marching_cubes_params(hp, 1 << 18, tf)
    tris, n = sc.extract_iso_surface(mp)
    assert n == len(tris) and n > 100
    return dict(width=W, height=H, params=ps, num_buckets=nb, num_sdf_blocks=nblk, scene=scene, frames=np.array(frames),
                thresh_factor=tf, triangles=sorted_triangles(tris).view(np.float32).reshape(n, 18))


def crc(a):
    """checksum of an array's bytes (the per-frame stand-in for payloads too large to commit for every frame)"""
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


SEQ_CASES = {
    # 20 frames at 1 cm voxels with garbage collection and a starve pass at frame 15: alloc/free churn
    "s1_64x48_p1_gc": dict(width=64, height=48, params="P1", num_buckets=1 << 13, num_sdf_blocks=1 << 13, frames=20,
                           orbit=200, starve=15),
}


def run_seq_case(name):
    c = SEQ_CASES[name]
    hp = T.make_hash_params(c["num_buckets"], c["num_sdf_blocks"], **synth.PARAM_SETS[c["params"]])
    cp = T.make_depth_camera_params(c["width"], c["height"])
    rp = T.make_raycast_params(hp, cp)
    sc = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=True, starve=c["starve"]))
    out = {k: v for k, v in c.items()}
    rows = []
    for k in range(c["frames"]):
        pose = synth.orbit_pose(k, c["orbit"])
        depth, color = O.synth_frame(synth.S1_SPHERES, 0, pose, cp)
        sc.integrate(pose, depth, color)
        s = sc.state()
        rows.append((s["num_occupied"], s["heap_free"], sc.hp.m_numOccupiedBlocks, crc(s["positions"]),
                     crc(s["voxels"]["sdf"].view(np.uint32)), crc(s["voxels"]["weight"]), crc(s["voxels"]["color"])))
    out["per_frame"] = np.array(rows, dtype=np.int64)  # occupied, heap free, in frustum, crc(pos, sdf, weight, colour)
    out["final_positions"] = s["positions"]
    r = sc.render(pose)
    for m in ("depth", "normals", "colors"):
        out["final_ray_" + m] = r[m]
    assert sum(r[2] > r[0] for r in rows) >= 10, "the case is meant to free blocks (in frustum before GC > occupied after)"
    return out


STREAM_CASE = dict(width=64, height=48, extents=(0.5, 0.5, 0.5), dims=(65, 65, 65), min_pos=(-32, -32, -32), parts=4,
                   stream_pos=(0.0, 0.0, 1.6), radius=1.2, frames=24, orbit=40)


def run_stream_case():
    """stream_s1_64x48: blocks leave for the host chunk grid and come back while the camera orbits"""
    from oracle.chunk_grid import OracleChunkGrid
    c = STREAM_CASE
    ps = dict(synth.PARAM_SETS["P4"])
    ps.update(streaming_extents=c["extents"], streaming_dims=c["dims"], streaming_min=c["min_pos"])
    hp = T.make_hash_params(1 << 14, 1 << 13, **ps)
    cp = T.make_depth_camera_params(c["width"], c["height"])
    rp = T.make_raycast_params(hp, cp)
    sc = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=True, starve=15, streaming_out_parts=c["parts"]))
    grid = OracleChunkGrid(sc, c["extents"], c["dims"], c["min_pos"], c["parts"])
    sp = np.array(list(c["stream_pos"]) + [1.0], dtype=np.float32)
    rows = []
    for k in range(c["frames"]):
        pose = synth.orbit_pose(k, c["orbit"])
        p = (pose.reshape(4, 4) @ sp)[:3]
        depth, color = O.synth_frame(synth.S1_SPHERES, 0, pose, cp)
        n_out, n_in = grid.stream_out_to_cpu(p, c["radius"], True), grid.stream_in_to_gpu(p, c["radius"], True)
        sc.integrate(pose, depth, color, grid.bitmask)
        s = sc.state()
        descs, blocks = grid.host_blocks()
        order = np.lexsort((descs["pos"][:, 2], descs["pos"][:, 1], descs["pos"][:, 0])) if len(descs) else np.zeros(0, np.int64)
        rows.append((n_out, n_in, s["num_occupied"], s["heap_free"], len(descs), grid.statistics()["bits"],
                     crc(s["positions"]), crc(s["voxels"]), crc(descs["pos"][order]), crc(blocks[order])))
    out = {k: np.array(v) for k, v in c.items()}
    # n_out, n_in, occupied, heap free, host blocks, mask bits, crc(gpu positions, gpu voxels, host positions, host voxels)
    out["per_frame"] = np.array(rows, dtype=np.int64)
    out["final_positions"] = s["positions"]
    out["final_host_positions"] = np.ascontiguousarray(descs["pos"][order])
    assert sum(r[0] for r in rows) > 20 and sum(r[1] for r in rows) > 5
    return out


def run_tracking_case():
    """icp_s3_160x120: the numpy ICP oracle's pose for one frame-to-model alignment, with its inputs' checksums"""
    from oracle import icp
    W, H = 160, 120
    hp = T.make_hash_params(1 << 14, 1 << 13, **synth.PARAM_SETS["P2"])
    cp = T.make_depth_camera_params(W, H)
    rp = T.make_raycast_params(hp, cp)
    poses = [synth.orbit_pose(k, n_frames=400) for k in range(4)]
    sc = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=False))
    for p in poses[:3]:
        d, c = O.synth_frame(synth.S3_SPHERES, 0, p, cp)
        sc.integrate(p, d, c)
    model = sc.render(poses[2])
    depth, _ = O.synth_frame(synth.S3_SPHERES, 0, poses[3], cp)
    cam = O.image_op("convert_depth_float_to_camera_space_float4", depth, W, H, cp, out_channels=4)
    nrm = O.compute_normals(cam)
    ts = T.make_tracking_state()
    got, info = icp.apply_ct(cam, nrm, model["depth4"], model["normals"], poses[2], ts, np.eye(4, dtype=np.float32), cp, 3)
    assert got is not None
    return dict(width=W, height=H, last_pose=poses[2], true_pose=poses[3], pose=np.asarray(got, np.float32).reshape(16),
                num_corr=info["numCorr"], crc_inputs=np.array([crc(cam), crc(nrm), crc(model["depth4"]), crc(model["normals"])], dtype=np.int64))


def main():
    for name in SEQ_CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **run_seq_case(name))
        print(name, os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")
    np.savez_compressed(os.path.join(HERE, "stream_s1_64x48.npz"), **run_stream_case())
    print("stream_s1_64x48", os.path.getsize(os.path.join(HERE, "stream_s1_64x48.npz")), "bytes")
    np.savez_compressed(os.path.join(HERE, "icp_s3_160x120.npz"), **run_tracking_case())
    print("icp_s3_160x120", os.path.getsize(os.path.join(HERE, "icp_s3_160x120.npz")), "bytes")
    for name in MC_CASES:
        out = run_mc_case(name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes,", len(out["triangles"]), "triangles")
    for name in CASES:
        data = run_case(name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print(name, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
