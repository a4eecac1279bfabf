"""Marching cubes: extractIsoSurfacePass1/Pass2 + extractIsoSurfaceAtPosition.

CPU: the case tables are self-consistent (and equal to the reference's where /root/reference is present), the oracle
restatement has the expected geometry on analytic scenes and reproduces the committed vectors.
GPU: the HIP path (through the C ABI) produces the oracle's triangle SET bit for bit -- the order in the buffer
comes from atomics on both sides -- with and without the box, per chunk through the streaming grid, and reports
overflow the way the reference's host code expects."""
import os
import re
import sys

import numpy as np
import pytest

from helpers import small_config
from voxelhashing_amd import synth, vhtypes as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def sorted_triangles(tris):
    v = np.ascontiguousarray(tris).view(np.dtype((np.void, T.TRIANGLE_DTYPE.itemsize))).ravel()
    return np.ascontiguousarray(tris)[np.argsort(v, kind="stable")]


def same_triangle_set(a, b):
    a, b = sorted_triangles(a), sorted_triangles(b)
    return len(a) == len(b) and a.tobytes() == b.tobytes()


def unique_triangles(tris):
    v = np.ascontiguousarray(tris).view(np.dtype((np.void, T.TRIANGLE_DTYPE.itemsize))).ravel()
    _, idx = np.unique(v, return_index=True)
    return np.ascontiguousarray(tris)[np.sort(idx)]


def integrate_oracle(O, hp, cp, poses, spheres, inside=0):
    sc = O.OracleScene(hp, cp, None, T.make_scene_options(offline=True, gc=False))
    for pose in poses:
        depth, color = O.synth_frame(spheres, inside, pose, cp)
        sc.integrate(pose, depth, color)
    return sc


# ---------------------------------------------------------------------------- CPU

def test_case_tables_are_consistent_and_match_the_header():
    import gen_mc_tables as G
    tri, edge = G.tables()  # asserts: edge sets of the triangles == sign changes of the corners; complement symmetry
    hdr = open(os.path.join(ROOT, "include", "vh_mc_tables.h")).read()
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{16})ull", hdr)]
    masks = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{3})[,\s]", hdr.split("VH_MC_EDGE[256]")[-1])]
    assert words == tri and masks == edge
    assert sum(1 for w in tri if w == 0xFFFFFFFFFFFFFFFF) == 2  # only the empty and the full cube have no triangle
    assert max(sum(1 for k in range(16) if (w >> (4 * k)) & 0xF != 0xF) for w in tri) == 15


def test_case_tables_equal_the_tables_the_reference_embeds():
    """DSC/Tables.h carries Paul Bourke's published tables; read as DATA where the reference tree is mounted."""
    path = "/root/reference/DepthSensingCUDA/Source/Tables.h"
    if not os.path.exists(path):
        pytest.skip("reference tree not present (GPU box)")
    import gen_mc_tables as G
    tri, edge = G.tables()
    text = open(path).read()
    nums = [int(x) for x in re.findall(r"-?\d+", re.search(r"triTable\[256\]\[16\]\s*=\s*\{(.*?)\};", text, re.S).group(1))]
    assert len(nums) == 4096
    for c in range(256):
        ref = [x for x in nums[16 * c:16 * c + 16] if x != -1]
        mine = [(tri[c] >> (4 * k)) & 0xF for k in range(16)]
        mine = mine[:mine.index(0xF)] if 0xF in mine else mine
        assert mine == ref, f"case {c}"
    ref_edges = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", re.search(r"edgeTable\[256\]\s*=\s*\{(.*?)\};", text, re.S).group(1))]
    assert ref_edges == edge


def test_oracle_plane_surface(oracle_lib):
    """fronto-parallel plane at 2 m, identity pose: every vertex of the extracted surface lies on z = 2 (the SDF is
    linear across the plane, so the zero crossing is exact up to rounding), triangles are not degenerate"""
    O = oracle_lib
    hp, cp, _ = small_config(80, 60)
    pose = np.eye(4, dtype=np.float32).reshape(16)
    depth = np.full((60, 80), 2.0, dtype=np.float32)
    color = np.zeros((60, 80, 4), dtype=np.float32)
    color[..., 0], color[..., 1], color[..., 2], color[..., 3] = 1.0, 0.5, 0.0, 1.0  # [0,1] floats, as the sensor adapter hands them over
    sc = O.OracleScene(hp, cp, None, T.make_scene_options(offline=True, gc=False))
    sc.integrate(pose, depth, color)
    tris, n = sc.extract_iso_surface(T.make_marching_cubes_params(hp, 1 << 18))
    assert n == len(tris) and n > 500
    p = tris["v"]["p"].reshape(-1, 3)
    assert np.abs(p[:, 2] - 2.0).max() < 1e-4
    a, b, c = tris["v"]["p"][:, 0], tris["v"]["p"][:, 1], tris["v"]["p"][:, 2]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    assert (area > 0).mean() > 0.99 and area.sum() > 0.5  # ~ the visible part of the plane, m^2
    col = tris["v"]["c"].reshape(-1, 3)
    # the first integration averages the observed colour with the empty voxel's black (combineVoxel :229-250)
    assert np.allclose(col, [0.5, 0.25, 0.0], atol=2.0 / 255.0)


def test_oracle_sphere_surface_and_thresholds(oracle_lib):
    O = oracle_lib
    hp, cp, _ = small_config(96, 72)
    poses = [synth.orbit_pose(k, n_frames=100) for k in range(3)]
    sc = integrate_oracle(O, hp, cp, poses, synth.SPHERE_A)
    tris, n = sc.extract_iso_surface(T.make_marching_cubes_params(hp, 1 << 18))
    assert n > 300
    cx, cy, cz, r = synth.SPHERE_A[0]
    d = np.linalg.norm(tris["v"]["p"].reshape(-1, 3).astype(np.float64) - [cx, cy, cz], axis=1)
    assert np.abs(d - r).max() < 0.5 * hp.m_virtualVoxelSize
    # a tighter threshold rejects cells, never adds any; a box keeps a subset; capacity is respected
    fewer, n2 = sc.extract_iso_surface(T.make_marching_cubes_params(hp, 1 << 18, thresh_factor=1.75))
    assert 0 < n2 < n and len(unique_triangles(np.concatenate([tris, fewer]))) == len(unique_triangles(tris))
    mp = T.make_marching_cubes_params(hp, 1 << 18)
    mp.m_boxEnabled = 1
    mp.m_minCorner[:] = [cx - 10, cy - 10, cz - 10]
    mp.m_maxCorner[:] = [cx, cy + 10, cz + 10]
    half, n3 = sc.extract_iso_surface(mp)
    assert 0 < n3 < n and len(unique_triangles(np.concatenate([tris, half]))) == len(unique_triangles(tris))
    capped, n4 = sc.extract_iso_surface(T.make_marching_cubes_params(hp, 100))
    assert n4 == n and len(capped) == 100


def test_oracle_reproduces_golden_mesh(oracle_lib):
    O = oracle_lib
    g = np.load(os.path.join(GOLDEN, "mc_s1_64x48_p4.npz"))
    hp = T.make_hash_params(int(g["num_buckets"]), int(g["num_sdf_blocks"]), **synth.PARAM_SETS[str(g["params"])])
    cp = T.make_depth_camera_params(int(g["width"]), int(g["height"]))
    spheres, inside, radius = synth.scene(str(g["scene"]))
    sc = integrate_oracle(O, hp, cp, [synth.orbit_pose(int(k), 100, radius) for k in g["frames"]], spheres, inside)
    tris, n = sc.extract_iso_surface(T.make_marching_cubes_params(hp, 1 << 18, float(g["thresh_factor"])))
    want = np.ascontiguousarray(g["triangles"]).view(T.TRIANGLE_DTYPE).ravel()
    assert n == len(want) and same_triangle_set(tris, want)


# ---------------------------------------------------------------------------- GPU

def gpu_scene(E, hp, cp, poses, spheres, inside=0):
    scene = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=True, gc=False))
    frame = E.DepthFrame(cp)
    for pose in poses:
        E.synth_frame(spheres, inside, pose, cp, out=frame)
        scene.integrate(pose, frame, cp, None)
    return scene


@pytest.mark.gpu
@pytest.mark.parametrize("width,height,params,scene_name", [(96, 72, "P4", "S1"), (64, 48, "P2", "S1"), (80, 60, "P4", "S2")])
def test_gpu_triangle_set_equals_oracle(vh, oracle_lib, width, height, params, scene_name):
    from voxelhashing_amd import engine as E
    hp, cp, _ = small_config(width, height, params=params)
    spheres, inside, radius = synth.scene(scene_name)
    poses = [synth.orbit_pose(k, 100, radius) for k in range(3)]
    scene = gpu_scene(E, hp, cp, poses, spheres, inside)
    o = integrate_oracle(oracle_lib, hp, cp, poses, spheres, inside)
    mp = T.make_marching_cubes_params(hp, 1 << 19)
    mc = E.CUDAMarchingCubesHashSDF(mp)
    mc.extractIsoSurface(scene.getHashData(), scene.getHashParams())
    want, n = o.extract_iso_surface(mp)
    got = mc.triangles()
    assert mc.counts()["triangles"] == n and n > 200
    assert mc.counts()["occupied_blocks"] == len(scene.state()["positions"])
    assert same_triangle_set(got, want)
    # the host mesh is the soup: 3 vertices per triangle, colours (r, g, b, 1)
    m = mc.mesh()
    assert m["vertices"].shape == (3 * n, 3) and np.all(m["colors"][:, 3] == 1.0)
    assert m["vertices"].tobytes() == got["v"]["p"].reshape(-1, 3).tobytes()
    # with the box: same subset as the oracle's
    cx = float(np.median(got["v"]["p"][..., 0]))
    mpb = T.make_marching_cubes_params(hp, 1 << 19)
    mpb.m_boxEnabled = 1
    mpb.m_minCorner[:] = [cx, -10.0, -10.0]
    mpb.m_maxCorner[:] = [10.0, 10.0, 10.0]
    mc.extractIsoSurfaceWithoutCopy(scene.getHashData(), scene.getHashParams(), (cx, -10, -10), (10, 10, 10), True)
    want_b, nb = o.extract_iso_surface(mpb)
    assert 0 < nb < n and mc.counts()["triangles"] == nb and same_triangle_set(mc.triangles(), want_b)


@pytest.mark.gpu
def test_gpu_reproduces_golden_mesh(vh):
    from voxelhashing_amd import engine as E
    g = np.load(os.path.join(GOLDEN, "mc_s1_64x48_p4.npz"))
    hp = T.make_hash_params(int(g["num_buckets"]), int(g["num_sdf_blocks"]), **synth.PARAM_SETS[str(g["params"])])
    cp = T.make_depth_camera_params(int(g["width"]), int(g["height"]))
    spheres, inside, radius = synth.scene(str(g["scene"]))
    scene = gpu_scene(E, hp, cp, [synth.orbit_pose(int(k), 100, radius) for k in g["frames"]], spheres, inside)
    mc = E.CUDAMarchingCubesHashSDF(T.make_marching_cubes_params(hp, 1 << 18, float(g["thresh_factor"])))
    mc.extractIsoSurfaceWithoutCopy(scene.getHashData(), scene.getHashParams())
    want = np.ascontiguousarray(g["triangles"]).view(T.TRIANGLE_DTYPE).ravel()
    assert same_triangle_set(mc.triangles(), want)


@pytest.mark.gpu
def test_gpu_overflow_merge_and_ply(vh, tmp_path):
    from voxelhashing_amd import engine as E, lib
    hp, cp, _ = small_config(96, 72)
    poses = [synth.orbit_pose(k, n_frames=100) for k in range(3)]
    scene = gpu_scene(E, hp, cp, poses, synth.S1_SPHERES)
    big = E.CUDAMarchingCubesHashSDF(T.make_marching_cubes_params(hp, 1 << 19))
    big.extractIsoSurface(scene.getHashData(), scene.getHashParams())
    n = big.counts()["triangles"]
    # capacity too small: the count still says how many there were, the copy refuses like the reference's host code
    small = E.CUDAMarchingCubesHashSDF(T.make_marching_cubes_params(hp, n // 2))
    small.extractIsoSurfaceWithoutCopy(scene.getHashData(), scene.getHashParams())
    assert small.counts()["triangles"] == n
    kept = small.triangles()
    assert len(kept) == n // 2 and len(unique_triangles(np.concatenate([big.triangles(), kept]))) == len(unique_triangles(big.triangles()))
    with pytest.raises(lib.VhError):
        small.copyTrianglesToCPU()
    # saveMesh: merged vertices, no duplicate faces, a PLY a reader can parse
    path = str(tmp_path / "scan.ply")
    big.saveMesh(path, None, True)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    nv = int(re.search(rb"element vertex (\d+)", head).group(1))
    nf = int(re.search(rb"element face (\d+)", head).group(1))
    assert b"binary_little_endian" in head and 0 < nv < 3 * n and 0 < nf <= n
    assert len(body) == nv * 16 + nf * 13
    faces = np.frombuffer(body[nv * 16:], dtype=np.dtype([("n", "u1"), ("i", "<i4", 3)]))
    assert np.all(faces["n"] == 3) and faces["i"].min() >= 0 and faces["i"].max() < nv
    assert big.mesh()["vertices"].shape[0] == 0  # saveMesh clears the buffer (.cpp:143)
    # offline mode merges per extraction
    off = E.CUDAMarchingCubesHashSDF(T.make_marching_cubes_params(hp, 1 << 19))
    off.setOfflineProcessing(True)
    off.extractIsoSurface(scene.getHashData(), scene.getHashParams())
    m = off.mesh()
    assert m["vertices"].shape[0] == nv and m["faces"].shape[0] == nf


@pytest.mark.gpu
def test_gpu_chunkwise_extraction_covers_the_direct_one(vh):
    """extractIsoSurface(chunkGrid, ...): stream everything out, then per chunk stream its neighbourhood in and
    extract inside the chunk's box.  Boxes overlap by a block, so triangles repeat; as a set they are the direct
    extraction's, and the scene is back on the GPU afterwards."""
    from voxelhashing_amd import engine as E
    hp, cp, _ = small_config(96, 72, streaming_extents=(1.0, 1.0, 1.0), streaming_dims=(9, 9, 9), streaming_min=(-4, -4, -4))
    poses = [synth.orbit_pose(k, n_frames=100) for k in range(3)]
    scene = gpu_scene(E, hp, cp, poses, synth.S1_SPHERES)
    mp = T.make_marching_cubes_params(hp, 1 << 19)
    direct = E.CUDAMarchingCubesHashSDF(mp)
    direct.extractIsoSurfaceWithoutCopy(scene.getHashData(), scene.getHashParams())
    want = direct.triangles()
    before = scene.state()
    grid = E.CUDASceneRepChunkGrid(scene, (1.0, 1.0, 1.0), (9, 9, 9), (-4, -4, -4), 64, True, 4)
    mc = E.CUDAMarchingCubesHashSDF(mp)
    mc.extractIsoSurfaceChunkGrid(grid, (0.0, 0.0, 0.0), 100.0)
    m = mc.mesh()
    soup = np.zeros(len(m["vertices"]) // 3, dtype=T.TRIANGLE_DTYPE)
    soup["v"]["p"] = m["vertices"].reshape(-1, 3, 3)
    soup["v"]["c"] = m["colors"][:, :3].reshape(-1, 3, 3)
    assert len(soup) >= len(want) and same_triangle_set(unique_triangles(soup), unique_triangles(want))
    after = scene.state()
    from voxelhashing_amd import canonical
    canonical.assert_same_scene(before, after, "scene after chunk-wise extraction")
    grid.close()
