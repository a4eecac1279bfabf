"""Size-independent properties at BASELINE.json's full sizes (the oracle would
take minutes there): two different code paths of the engine must produce the
same canonical scene, structural invariants must hold, rendering must be
deterministic, and the async (online) block count must agree with a re-count."""
import numpy as np
import pytest

from helpers import assert_maps_equal
from voxelhashing_amd import canonical, synth, vhtypes as T

pytestmark = pytest.mark.gpu


def run_sequence(E, hp, cp, rp, opt, poses, frame, spheres, inside):
    scene, ray = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp)
    maps = None
    for k, pose in enumerate(poses):
        E.synth_frame(spheres, inside, pose, cp, out=frame)
        if k > 0:
            ray.render(scene.getHashData(), scene.getHashParams(), cp, poses[k - 1])
        scene.integrate(pose, frame, cp, None)
    ray.render(scene.getHashData(), scene.getHashParams(), cp, poses[-1])
    maps = ray.download()
    return scene, ray, maps


@pytest.mark.parametrize("cfg,scene_name,n_frames", [("cfg2", "S1", 24), ("cfg4", "S1", 6), ("cfg2", "S2", 8)])
def test_fused_equals_reference_launch_sequence_at_full_size(vh, cfg, scene_name, n_frames):
    """cfg2 = 640x480 / 4 cm / 5 M hash entries; cfg4 = 1920x1080 / 2 cm; S2 = every pixel valid (dense).
    Fused kernel + lock epochs + device-side counts  ==  the reference's separate launches with host counts."""
    from voxelhashing_amd import engine as E
    c = dict(synth.CONFIGS[cfg])
    c.update(scene=scene_name, num_sdf_blocks=1 << 16)  # full-size table; voxel pool sized to be downloadable
    hp, cp, rp = synth.config_params(c)
    spheres, inside, radius = synth.scene(scene_name)
    poses = [synth.orbit_pose(k, 200, radius) for k in range(n_frames)]
    frame = E.DepthFrame(cp)
    a, ra, ma = run_sequence(E, hp, cp, rp, T.make_scene_options(offline=True, gc=True, starve=5), poses, frame, spheres, inside)
    b, rb, mb = run_sequence(E, hp, cp, rp, T.make_scene_options(offline=True, gc=True, starve=5, reference_launch_sequence=True),
                             poses, frame, spheres, inside)
    sa, sb = a.state(), b.state()  # state() also runs debugHash-style invariants and the bucket-summary check
    canonical.assert_same_scene(sa, sb, f"{cfg}/{scene_name}")
    assert np.array_equal(canonical.compactified_set(sa["compactified"]), canonical.compactified_set(sb["compactified"]))
    assert_maps_equal(ma, mb, "raycast of both paths")
    assert sa["num_occupied"] > 100 and (ma["depth"] != -np.inf).sum() > 10000
    assert a.debugHash()["duplicates"] == 0
    # determinism: rendering again gives the same bits
    ra.render(a.getHashData(), a.getHashParams(), cp, poses[-1])
    assert_maps_equal(ra.download(), ma, "second render")
    # every hit pixel has a unit normal pointing towards the camera where computeNormals could form one
    n = ma["normals"]
    ok = n[..., 3] == 1.0
    assert ok.sum() > 5000
    assert np.allclose(np.linalg.norm(n[ok][:, :3].astype(np.float64), axis=1), 1.0, atol=1e-5)


@pytest.mark.parametrize("cfg,scene_name", [("cfg2", "S1"), ("cfg2", "S2"), ("cfg4", "S1")])
def test_interval_splatting_does_not_change_a_bit(vh, cfg, scene_name):
    """render() with the conservative ray intervals == the full-range march of the reference fork, for views from the
    orbit, from far away, tilted so that blocks straddle the image border, and from inside the allocated band
    (blocks behind and around the camera)"""
    from voxelhashing_amd import engine as E
    c = dict(synth.CONFIGS[cfg])
    c.update(scene=scene_name, num_sdf_blocks=1 << 16)
    hp, cp, rp = synth.config_params(c)
    spheres, inside, radius = synth.scene(scene_name)
    poses = [synth.orbit_pose(k, 200, radius) for k in range(6)]
    frame = E.DepthFrame(cp)
    scene, ray, _ = run_sequence(E, hp, cp, rp, T.make_scene_options(offline=True, gc=False), poses, frame, spheres, inside)
    full = E.CUDARayCastSDF(rp)
    full.setIntervalSplatting(False)
    import ctypes as C
    from voxelhashing_amd import lib
    L = lib.load()
    n_tiles = ((cp.m_imageWidth + 7) // 8) * ((cp.m_imageHeight + 7) // 8)
    heads, small_lists = lib.DeviceBuffer(n_tiles * 16), lib.DeviceBuffer(n_tiles * 3 * 16)
    big_lists = lib.DeviceBuffer(n_tiles * 128 * 16)
    lib.check(L.vh_ray_interval_clear(heads.ptr, cp.m_imageWidth, cp.m_imageHeight, None))
    views = [poses[-1], synth.orbit_pose(40, 200, radius), synth.orbit_pose(3, 200, radius * 1.7)]
    tilt = np.array(poses[2], dtype=np.float32).reshape(4, 4).copy()
    a = 0.45
    rot = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]], dtype=np.float32)
    views.append(np.ascontiguousarray((tilt @ rot).astype(np.float32)).reshape(16))
    near = np.array(poses[1], dtype=np.float32).reshape(4, 4).copy()
    near[:3, 3] *= 0.35  # camera moved into / next to the surface band
    views.append(np.ascontiguousarray(near).reshape(16))
    hits = 0
    for i, v in enumerate(views):
        ray.render(scene.getHashData(), scene.getHashParams(), cp, v)
        full.render(scene.getHashData(), scene.getHashParams(), cp, v)
        ma, mb = ray.download(), full.download()
        assert_maps_equal(ma, mb, f"view {i}: intervals vs full range")
        hits += int((ma["depth"] != -np.inf).sum())
        # launcher level: lists that overflow (capacity 3: misses fall back to the hash table), no lists at all, and the
        # large tables (capacity 128 and an odd 100)
        hd, hpp, rpp, rd = scene.getHashData(), scene.getHashParams(), full.getRayCastParams(), full.getRayCastData()
        for cap, blocks in ((3, small_lists), (0, None), (128, big_lists), (100, big_lists)):
            lib.check(L.vh_ray_interval_splat(C.byref(hd), C.byref(hpp), C.byref(cp), C.byref(rpp), heads.ptr,
                                              blocks.ptr if blocks else None, cap, None, 0, None, None))
            lib.check(L.vh_render_intervals(C.byref(hd), C.byref(hpp), C.byref(rd), C.byref(cp), C.byref(rpp), heads.ptr,
                                            blocks.ptr if blocks else None, cap, None, 0, None))
            mc = full.download()
            for k in ("depth", "depth4", "colors"):  # normals are computeNormals' output of the class-level call
                assert np.array_equal(mc[k].view(np.uint32), mb[k].view(np.uint32)), f"view {i} capacity {cap}: {k}"
        assert np.all(heads.download(np.uint32).reshape(-1, 4) == np.array([0x7f800000, 0, 0, 0], dtype=np.uint32)), "heads re-armed"
    assert hits > 10000


def test_online_mode_is_consistent_and_eventually_complete(vh):
    """online alloc (one pass per frame, losers retry next frame): after a few frames of a static view the block
    set equals the offline fixed point; the mirrored block count equals a blocking re-count"""
    from voxelhashing_amd import engine as E
    hp, cp, rp = synth.config_params(dict(synth.CONFIGS["cfg2"], num_sdf_blocks=1 << 15))
    pose = synth.orbit_pose(0)
    frame = E.synth_frame(synth.S1_SPHERES, 0, pose, cp)
    off = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=True, gc=False))
    off.integrate(pose, frame, cp, None)
    on = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=False, gc=False))
    for _ in range(4):
        on.integrate(pose, frame, cp, None)
    so, sn = off.state(with_voxels=False), on.state(with_voxels=False)
    assert np.array_equal(so["positions"], sn["positions"])
    on.synchronize()
    assert on.getHashParams().m_numOccupiedBlocks == on.getNumOccupiedBlocks() == off.getNumOccupiedBlocks()
    st = on.getState()
    assert st[T.STATE_HEAP_UNDERFLOW] == 0


def test_heap_exhaustion_is_reported_not_fatal(vh):
    """more blocks requested than the pool holds: the reference indexes out of bounds (consumeHeap has no check);
    here allocation stops, the status word is raised and every invariant still holds"""
    from voxelhashing_amd import engine as E
    hp, cp, rp = synth.config_params(dict(synth.CONFIGS["cfg2"], num_sdf_blocks=64))
    pose = synth.orbit_pose(0)
    frame = E.synth_frame(synth.S1_SPHERES, 0, pose, cp)
    sc = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=False, gc=False))
    for _ in range(3):
        sc.integrate(pose, frame, cp, None)
    s = sc.state(with_voxels=False)
    assert s["num_occupied"] == 64 and s["heap_free"] == 0
    assert sc.getState()[T.STATE_HEAP_UNDERFLOW] > 0


def test_stream_round_trip_at_cfg3_size(vh, tmp_path):
    """cfg3's table (20 M entries, 1 cm voxels) with the reference's streaming grid (1 m chunks, 257^3, 80 parts):
    everything streamed out to the host chunk grid and back, directly and through a .hashgrid file, is the same scene
    bit for bit; nothing is lost or duplicated on the way."""
    from voxelhashing_amd import engine as E
    c = dict(synth.CONFIGS["cfg3"])
    c.update(num_sdf_blocks=1 << 16)
    hp, cp, rp = synth.config_params(c)
    ext, dims, minp, parts = (1.0, 1.0, 1.0), (257, 257, 257), (-128, -128, -128), 80
    hp.m_streamingVoxelExtents[:] = ext
    hp.m_streamingGridDimensions[:] = dims
    hp.m_streamingMinGridPos[:] = minp
    scene = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=True, gc=True, starve=15, streaming_out_parts=parts))
    grid = E.CUDASceneRepChunkGrid(scene, ext, dims, minp, 2000, False, parts)
    frame = E.DepthFrame(cp)
    for k in range(6):
        pose = synth.orbit_pose(k, 200)
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        scene.integrate(pose, frame, cp, grid.getBitMaskGPU())
    before = scene.state()
    assert before["num_occupied"] > 1500
    grid.streamOutToCPUAll()
    assert scene.state()["num_occupied"] == 0 and grid.getStatistics()["blocks"] == before["num_occupied"]
    grid.debugCheckForDuplicates()
    centre, big = np.zeros(3, np.float32), 1000.0
    assert grid.streamInToGPUAll(centre, big, True) == before["num_occupied"]
    canonical.assert_same_scene(before, scene.state(), "out and back")
    path = str(tmp_path / "cfg3.hashgrid")
    grid.saveToFile(path, centre, big)
    grid.loadFromFile(path, centre, big)
    assert scene.state()["num_occupied"] == 0
    assert grid.streamInToGPUAll(centre, big, True) == before["num_occupied"]
    canonical.assert_same_scene(before, scene.state(), "through the .hashgrid file")
    assert scene.debugHash()["duplicates"] == 0
