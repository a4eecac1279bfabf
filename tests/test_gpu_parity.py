"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
identical synthetic inputs.  Integer / index results must be bit-exact on the
canonical forms (voxelhashing_amd.canonical); float results (sdf, depth,
normals) are compared BIT-EXACT as well -- tolerance 0, tighter than the 1e-4
relative the north star allows -- because both sides are IEEE fp32 with the
same operation order.
"""
import numpy as np
import pytest

from helpers import assert_maps_equal, bits, small_config
from voxelhashing_amd import canonical, synth, vhtypes as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E(vh):
    from voxelhashing_amd import engine
    return engine


def run_pair(E, O, hp, cp, rp, opt, poses, spheres, inside=0, render=True, check_each=True):
    """drive the GPU engine and the oracle through the reference frame loop
    (render with the previous pose, then integrate) and compare after every frame"""
    g_scene = E.CUDASceneRepHashSDF(hp, opt)
    g_ray = E.CUDARayCastSDF(rp)
    o_scene = O.OracleScene(hp, cp, rp, opt)
    frame = E.DepthFrame(cp)
    last = None
    for k, pose in enumerate(poses):
        E.synth_frame(spheres, inside, pose, cp, out=frame)
        depth, color = O.synth_frame(spheres, inside, pose, cp)
        gd, gc = frame.download()
        assert np.array_equal(bits(gd), bits(depth)), f"frame {k}: synthetic depth differs"
        assert np.array_equal(bits(gc), bits(color)), f"frame {k}: synthetic colour differs"
        if render and last is not None:
            g_ray.render(g_scene.getHashData(), g_scene.getHashParams(), cp, last)
            want = o_scene.render(last)
            assert_maps_equal(g_ray.download(), want, f"frame {k} raycast")
        g_scene.integrate(pose, frame, cp, None)
        o_scene.integrate(pose, depth, color)
        last = pose
        if check_each or k == len(poses) - 1:
            gs, os_ = g_scene.state(), o_scene.state()
            canonical.assert_same_scene(gs, os_, f"frame {k}")
            assert np.array_equal(canonical.compactified_set(gs["compactified"]), canonical.compactified_set(o_scene.compactified())), f"frame {k}: compactified sets differ"
            assert g_scene.getNumOccupiedBlocks() == o_scene.hp.m_numOccupiedBlocks
    return g_scene, g_ray, o_scene


def test_synth_frame_matches_oracle(E, oracle_lib):
    hp, cp, rp = small_config(200, 150)
    for k in (0, 37, 250, 611):
        pose = synth.orbit_pose(k)
        fr = E.synth_frame(synth.S1_SPHERES, 0, pose, cp)
        gd, gc = fr.download()
        d, c = oracle_lib.synth_frame(synth.S1_SPHERES, 0, pose, cp)
        assert np.array_equal(bits(gd), bits(d)) and np.array_equal(bits(gc), bits(c))
        assert (d != -np.inf).sum() > 1000
    pose = synth.orbit_pose(3, radius=synth.S2_ORBIT_RADIUS)
    fr = E.synth_frame(synth.S2_SPHERES, 1, pose, cp)
    gd, gc = fr.download()
    d, c = oracle_lib.synth_frame(synth.S2_SPHERES, 1, pose, cp)
    assert np.array_equal(bits(gd), bits(d)) and np.array_equal(bits(gc), bits(c))
    assert (d != -np.inf).all()


def test_single_frame_sphere(E, oracle_lib):
    hp, cp, rp = small_config(160, 120)
    opt = T.make_scene_options(offline=True, gc=False)
    g, r, o = run_pair(E, oracle_lib, hp, cp, rp, opt, [synth.orbit_pose(0)], synth.SPHERE_A)
    assert g.state()["num_occupied"] > 50
    # raycast the integrated frame from the same pose
    pose = synth.orbit_pose(0)
    r.render(g.getHashData(), g.getHashParams(), cp, pose)
    want = o.render(pose)
    got = r.download()
    assert_maps_equal(got, want, "raycast")
    assert (got["depth"] != -np.inf).sum() > 1000
    assert g.debugHash()["duplicates"] == 0


@pytest.mark.parametrize("reference_sequence", [False, True])
def test_orbit_gc_sequence(E, oracle_lib, reference_sequence):
    """20 frames of S1 with GC + starve (starve=5 so it triggers): fused kernel
    and the reference launch sequence both equal the oracle frame by frame"""
    hp, cp, rp = small_config(96, 72)
    opt = T.make_scene_options(offline=True, gc=True, starve=5, reference_launch_sequence=reference_sequence)
    poses = [synth.orbit_pose(k, n_frames=200) for k in range(20)]
    g, r, o = run_pair(E, oracle_lib, hp, cp, rp, opt, poses, synth.S1_SPHERES)
    st = g.getState()
    assert st[T.STATE_HEAP_UNDERFLOW] == 0 and st[T.STATE_INSERT_FAILED] == 0


def test_raycast_with_gradients(E, oracle_lib):
    """m_useGradients = true: normals from gradientForPoint (6 trilinear samples
    whose partial sums on failure are observable) instead of computeNormals"""
    hp, cp, _ = small_config(128, 96)
    rp = T.make_raycast_params(hp, cp, use_gradients=True)
    opt = T.make_scene_options(offline=True, gc=False)
    poses = [synth.orbit_pose(k, n_frames=100) for k in range(3)]
    g, r, o = run_pair(E, oracle_lib, hp, cp, rp, opt, poses, synth.S1_SPHERES)
    r.render(g.getHashData(), g.getHashParams(), cp, poses[-1])
    got, want = r.download(), o.render(poses[-1])
    assert_maps_equal(got, want, "gradients")
    assert (got["normals"][..., 3] == 1.0).sum() > 500


@pytest.mark.parametrize("width,height,params", [(100, 76, "P4"), (9, 7, "P4"), (8, 8, "P4"), (203, 149, "P2"), (64, 48, "P1")])
def test_ray_intervals_on_ragged_image_sizes(E, oracle_lib, width, height, params):
    """tile lists / depth intervals / launch schedule with partial tiles and a partial last workgroup, fine voxels
    (long lists) and several renders in a row (the schedule is live from the second one): == oracle, == full range"""
    ps = synth.PARAM_SETS[params]
    hp = T.make_hash_params(1 << 14, 1 << 13, **ps)
    cp = T.make_depth_camera_params(width, height)
    rp = T.make_raycast_params(hp, cp)
    opt = T.make_scene_options(offline=True, gc=False)
    poses = [synth.orbit_pose(k, n_frames=60) for k in range(4)]
    g, r, o = run_pair(E, oracle_lib, hp, cp, rp, opt, poses, synth.S1_SPHERES, check_each=False)
    full = E.CUDARayCastSDF(rp)
    full.setIntervalSplatting(False)
    hits = 0
    for k in (3, 1, 2, 3, 0):
        r.render(g.getHashData(), g.getHashParams(), cp, poses[k])
        full.render(g.getHashData(), g.getHashParams(), cp, poses[k])
        got = r.download()
        assert_maps_equal(got, full.download(), f"{width}x{height} view {k}: intervals vs full range")
        assert_maps_equal(got, o.render(poses[k]), f"{width}x{height} view {k}: vs oracle")
        hits += int((got["depth"] != -np.inf).sum())
    assert hits > 0.2 * width * height


def test_fine_voxels_switch_to_large_tile_tables(E, oracle_lib):
    """1 cm voxels at 320x240: some tiles list more than 64 blocks.  The first renders use the small tables (overflow:
    hash fallback), the ray caster's feedback then selects the large ones; every render equals the full-range march."""
    hp = T.make_hash_params(1 << 16, 1 << 14, **synth.PARAM_SETS["P1"])
    cp = T.make_depth_camera_params(320, 240)
    rp = T.make_raycast_params(hp, cp)
    scene = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=True, gc=False))
    frame = E.DepthFrame(cp)
    poses = [synth.orbit_pose(k, n_frames=60) for k in range(3)]
    for pose in poses:
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        scene.integrate(pose, frame, cp, None)
    ray, full = E.CUDARayCastSDF(rp), E.CUDARayCastSDF(rp)
    full.setIntervalSplatting(False)
    full.render(scene.getHashData(), scene.getHashParams(), cp, poses[-1])
    want = full.download()
    assert (want["depth"] != -np.inf).sum() > 5000
    for i in range(6):
        ray.render(scene.getHashData(), scene.getHashParams(), cp, poses[-1])
        assert_maps_equal(ray.download(), want, f"render {i}")


@pytest.mark.parametrize("gradients", [False, True])
def test_large_tile_tables_against_the_oracle(E, oracle_lib, gradients):
    """the same scene as above against the ORACLE's render, without and with gradients: with the large tables the march asks
    for the next sample's voxels before it blends this one's (march_ray, PIPELINED) -- every ray's samples, and what is made
    of them, must still be the reference's"""
    O = oracle_lib
    hp = T.make_hash_params(1 << 16, 1 << 14, **synth.PARAM_SETS["P1"])
    cp = T.make_depth_camera_params(320, 240)
    rp = T.make_raycast_params(hp, cp, use_gradients=gradients)
    opt = T.make_scene_options(offline=True, gc=False)
    scene, ref = E.CUDASceneRepHashSDF(hp, opt), O.OracleScene(hp, cp, rp, opt)
    frame = E.DepthFrame(cp)
    poses = [synth.orbit_pose(k, n_frames=60) for k in range(3)]
    for pose in poses:
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        d, c = O.synth_frame(synth.S1_SPHERES, 0, pose, cp)
        scene.integrate(pose, frame, cp, None)
        ref.integrate(pose, d, c)
    want = ref.render(poses[-1])
    assert (want["depth"] != -np.inf).sum() > 5000
    ray = E.CUDARayCastSDF(rp)
    for i in range(5):  # (the first renders run on the small tables; the feedback of a list longer than 64 selects the large ones)
        ray.render(scene.getHashData(), scene.getHashParams(), cp, poses[-1])
        assert_maps_equal(ray.download(), want, f"render {i}, gradients {gradients}")
    ray.close()
    scene.close()


@pytest.mark.parametrize("voxel,buckets", [(0.04, 500000), (0.01, 2000000), (0.02, 1 << 18), (0.004, 1 << 14), (0.035, 7)])
def test_exact_shortcuts(vh, voxel, buckets):
    """div_exact == `/` and umod_fast == `%` bit for bit on 16 M pseudo-random operands"""
    import ctypes as C
    from voxelhashing_amd.lib import DeviceBuffer, check
    buf = DeviceBuffer(8)
    check(vh.vh_debug_check_fast_math(C.c_float(voxel), buckets, 1 << 24, 1234567, buf.ptr, None), "check")
    mism = buf.download(np.uint32)
    assert mism[0] == 0, f"div_exact differs from IEEE division on {mism[0]} operands"
    assert mism[1] == 0, f"umod_fast differs from % on {mism[1]} operands"


@pytest.mark.parametrize("seed", [1, 77, 990001])
def test_refined_division_equals_ieee_division_in_the_certified_ranges(vh, seed):
    """integrate_block_certified's division (one refined reciprocal, two residual corrections, packed) == `/` bit for
    bit on 64 M pseudo-random operand pairs per range: the projection's (divisor 2^-20..2^20, |numerator| 2^-100..2^60
    or zero) and the blend's (divisor 1..510, |numerator| 2^-100..2^90)"""
    import ctypes as C
    from voxelhashing_amd.lib import DeviceBuffer, check
    buf = DeviceBuffer(8)
    check(vh.vh_debug_check_refined_division(1 << 26, seed, buf.ptr, None), "check")
    mism = buf.download(np.uint32)
    assert mism[0] == 0, f"projection range: differs from IEEE division on {mism[0]} operand pairs"
    assert mism[1] == 0, f"blend range: differs from IEEE division on {mism[1]} operand pairs"


def general_poses(n, seed):
    """camera-to-world matrices with every rotation axis in play: the orbit pose, then roll / pitch / yaw of up to 25
    degrees about the camera's own axes and a shift of up to 20 cm (the orbit alone only ever turns about y)"""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        base = np.asarray(synth.orbit_pose(int(rng.integers(0, 200)), 200), np.float64).reshape(4, 4)
        a, b, c = np.radians(rng.uniform(-25, 25, size=3))
        rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        local = np.eye(4)
        local[:3, :3] = rz @ ry @ rx
        local[:3, 3] = rng.uniform(-0.2, 0.2, size=3)
        out.append((base @ local).astype(np.float32).reshape(16))
    return out


@pytest.mark.parametrize("params,gradients,seed", [("P4", False, 1), ("P2", False, 2), ("P2", True, 3), ("P1", False, 4)])
def test_general_camera_orientations(E, oracle_lib, params, gradients, seed):
    """rays and frusta with all three direction components of either sign, non-axis-aligned image planes: alloc's
    DDA, the frustum tests, the fused / exact voxel-index routes of the ray caster and the splat's projections all
    against the oracle, bit for bit, on the four-sphere scene"""
    O = oracle_lib
    hp, cp, rp = small_config(112, 80, params, 1 << 14, 1 << 14)
    rp = T.make_raycast_params(hp, cp, use_gradients=gradients)
    poses = general_poses(5, seed)
    g_scene, g_ray, o_scene = run_pair(E, O, hp, cp, rp, T.make_scene_options(offline=True, gc=True, starve=3), poses, synth.S3_SPHERES)
    assert g_scene.getNumOccupiedBlocks() > 40
    # and from a viewpoint nothing was integrated from
    view = general_poses(1, seed + 100)[0]
    g_ray.render(g_scene.getHashData(), g_scene.getHashParams(), cp, view)
    assert_maps_equal(g_ray.download(), o_scene.render(view), "novel view")


@pytest.mark.parametrize("params,seed", [("P4", 11), ("P2", 12)])
def test_noisy_depth_with_holes(E, oracle_lib, params, seed):
    """what a sensor delivers rather than an analytic surface: centimetre noise, pixels without a measurement (MINF and
    0), readings beyond the integration distance, colour missing where depth is present and the other way round"""
    O = oracle_lib
    hp, cp, rp = small_config(128, 96, params, 1 << 14, 1 << 14)
    opt = T.make_scene_options(offline=True, gc=True, starve=2)
    rng = np.random.default_rng(seed)
    g_scene, g_ray, o_scene = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp), O.OracleScene(hp, cp, rp, opt)
    last = None
    minf = np.float32(-np.inf)
    for k in range(5):
        pose = synth.orbit_pose(3 * k, 120)
        depth, color = O.synth_frame(synth.S3_SPHERES, 0, pose, cp)
        valid = np.isfinite(depth)
        depth = np.where(valid, depth + rng.normal(0.0, 0.01, depth.shape).astype(np.float32), depth).astype(np.float32)
        u = rng.random(depth.shape)
        depth[u < 0.06] = minf
        depth[(u >= 0.06) & (u < 0.09)] = 0.0
        depth[(u >= 0.09) & (u < 0.11)] = 4.5            # beyond m_maxIntegrationDistance
        depth[(u >= 0.11) & (u < 0.12)] = 3.9999         # just inside it
        color = color.copy()
        color[(u >= 0.5) & (u < 0.55)] = minf            # no colour here: such pixels are not integrated
        color[..., :3] = np.where(np.isfinite(color[..., :3]), rng.random(color[..., :3].shape).astype(np.float32), color[..., :3])
        frame = E.DepthFrame(cp, depth=depth, color=color)
        if last is not None:
            g_ray.render(g_scene.getHashData(), g_scene.getHashParams(), cp, last)
            assert_maps_equal(g_ray.download(), o_scene.render(last), f"frame {k} raycast")
        g_scene.integrate(pose, frame, cp, None)
        o_scene.integrate(pose, depth, color)
        canonical.assert_same_scene(g_scene.state(), o_scene.state(), f"frame {k}")
        last = pose
    assert g_scene.getNumOccupiedBlocks() > 60


def test_explicit_stream_gives_the_same_scene(E, oracle_lib):
    """every launch and copy of the classes goes to the stream they were given: a scene, ray caster, chunk grid, sensor
    and tracker built on a private non-blocking stream (nothing else synchronises it with the null stream) produce the
    oracle's results like the default-stream ones"""
    import ctypes as C
    from voxelhashing_amd import lib
    O = oracle_lib
    L = lib.load()
    st = C.c_void_p()
    lib.check(L.vh_stream_create(C.byref(st)), "vh_stream_create")
    try:
        hp, cp, rp = small_config(96, 72, "P2", 1 << 13, 1 << 13)
        opt = T.make_scene_options(offline=True, gc=True, starve=2)
        g_scene, g_ray = E.CUDASceneRepHashSDF(hp, opt, stream=st), E.CUDARayCastSDF(rp, stream=st)
        o_scene = O.OracleScene(hp, cp, rp, opt)
        frame = E.DepthFrame(cp, stream=st)
        last = None
        for k in range(6):
            pose = synth.orbit_pose(k, 60)
            E.synth_frame(synth.S3_SPHERES, 0, pose, cp, out=frame, stream=st)
            depth, color = O.synth_frame(synth.S3_SPHERES, 0, pose, cp)
            if last is not None:
                g_ray.render(g_scene.getHashData(), g_scene.getHashParams(), cp, last)
                assert_maps_equal(g_ray.download(), o_scene.render(last), f"frame {k} raycast")
            g_scene.integrate(pose, frame, cp, None)
            o_scene.integrate(pose, depth, color)
            last = pose
        canonical.assert_same_scene(g_scene.state(), o_scene.state(), "explicit stream")
    finally:
        for obj in ("g_scene", "g_ray"):
            if obj in locals():
                locals()[obj].close()
        lib.check(L.vh_stream_destroy(st), "vh_stream_destroy")


def test_two_scenes_on_two_streams_interleaved(E, oracle_lib):
    """one instance = one scene: two scenes with different parameters, each on its own stream, advanced in turns
    without any synchronisation between them, each equal to its own oracle (no state is shared between instances)"""
    import ctypes as C
    from voxelhashing_amd import lib
    O = oracle_lib
    L = lib.load()
    streams, rigs = [], []
    try:
        for params, spheres, size in (("P4", synth.S1_SPHERES, (96, 72)), ("P2", synth.S3_SPHERES, (80, 64))):
            st = C.c_void_p()
            lib.check(L.vh_stream_create(C.byref(st)), "vh_stream_create")
            streams.append(st)
            hp, cp, rp = small_config(size[0], size[1], params, 1 << 13, 1 << 13)
            opt = T.make_scene_options(offline=True, gc=True, starve=2)
            rigs.append(dict(cp=cp, spheres=spheres, st=st, scene=E.CUDASceneRepHashSDF(hp, opt, stream=st), ray=E.CUDARayCastSDF(rp, stream=st),
                             oracle=O.OracleScene(hp, cp, rp, opt), frame=E.DepthFrame(cp, stream=st), last=None))
        for k in range(5):
            for r in rigs:  # enqueue both before looking at either
                pose = synth.orbit_pose(2 * k, 60)
                E.synth_frame(r["spheres"], 0, pose, r["cp"], out=r["frame"], stream=r["st"])
                if r["last"] is not None:
                    r["ray"].render(r["scene"].getHashData(), r["scene"].getHashParams(), r["cp"], r["last"])
                r["scene"].integrate(pose, r["frame"], r["cp"], None)
            for r in rigs:
                pose = synth.orbit_pose(2 * k, 60)
                depth, color = O.synth_frame(r["spheres"], 0, pose, r["cp"])
                if r["last"] is not None:
                    assert_maps_equal(r["ray"].download(), r["oracle"].render(r["last"]), f"frame {k} raycast")
                r["oracle"].integrate(pose, depth, color)
                canonical.assert_same_scene(r["scene"].state(), r["oracle"].state(), f"frame {k}")
                r["last"] = pose
    finally:
        for r in rigs:
            r["scene"].close()
            r["ray"].close()
        for st in streams:
            lib.check(L.vh_stream_destroy(st), "vh_stream_destroy")
