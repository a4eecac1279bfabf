"""The native frame loop (vh_reconstruction_*: reconstruction(), DSC/DepthSensing.cpp:720-924) against the oracle,
frame by frame: with alloc + compactify of a frame running beside the ray cast of the previous pose
(CUDASceneRepHashSDF::integrateAhead), without, fed from device memory and fed from the host; and BASELINE.json's
full-size configurations (cfg1, cfg2, one 1080p frame of cfg4) against the oracle through the ray caster's
scheduled, split-tile path."""
import numpy as np
import pytest

from helpers import assert_maps_equal, small_config
from voxelhashing_amd import canonical, synth, vhtypes as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E(vh):
    from voxelhashing_amd import engine
    return engine


# The reference's hash sends (x, y, z) and (-x, -y, z) to the same bucket whenever x and y have equally many trailing
# zero bits (-a and -b differ from a and b in the same bit positions then, and the XOR cancels): in a scene that is
# point-symmetric about the origin (S1) a tenth of a frame's new blocks lose the bucket lock in an online pass and wait
# for the next frame -- which of the two wins is a matter of scheduling.  Away from the origin all coordinates are
# positive and an online pass is deterministic (asserted below through the oracle), so the scene is moved there.
OFFSET = np.array([7.3, 5.1, 3.7])
SHIFTED_S1 = synth.S1_SPHERES.copy()
SHIFTED_S1[:, :3] += OFFSET


def shifted_pose(k, n_frames):
    q = np.array(synth.orbit_pose(k, n_frames=n_frames), dtype=np.float32).copy()
    q[3] += np.float32(OFFSET[0])
    q[7] += np.float32(OFFSET[1])
    q[11] += np.float32(OFFSET[2])
    return q


def make_inputs(E, O, cp, poses, spheres, inside):
    frames, host = [], []
    for p in poses:
        frames.append(E.synth_frame(spheres, inside, p, cp))
        host.append(O.synth_frame(spheres, inside, p, cp))
    return frames, host


def oracle_online_is_deterministic(O, hp, cp, rp, poses, host, gc, starve):
    """online alloc gives the same table on every schedule iff no two NEW blocks of one pass share a bucket (the one
    that comes second loses the bucket lock until the next frame): then one pass allocates what the fixed point does"""
    on = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=False, gc=gc, starve=starve))
    off = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=gc, starve=starve))
    same = True
    for p, (d, c) in zip(poses, host):
        on.integrate(p, d, c)
        off.integrate(p, d, c)
        same = same and np.array_equal(canonical.block_positions(on.hash_table()), canonical.block_positions(off.hash_table()))
    return same


@pytest.mark.parametrize("ahead", [True, False])
def test_native_loop_equals_oracle_frame_by_frame(E, oracle_lib, ahead):
    O = oracle_lib
    hp, cp, rp = small_config(160, 120, num_buckets=1 << 17, num_sdf_blocks=1 << 12)
    n = 14
    poses = [shifted_pose(k, 120) for k in range(n)]
    frames, host = make_inputs(E, O, cp, poses, SHIFTED_S1, 0)
    assert oracle_online_is_deterministic(O, hp, cp, rp, poses, host, True, 5), "pick a scene without same-pass bucket sharing"
    opt = T.make_scene_options(offline=False, gc=True, starve=5)
    scene, ray, ref = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp), O.OracleScene(hp, cp, rp, opt)
    recon = E.Reconstruction(scene, ray, None, cp, E.Reconstruction.defaultOptions(s_allocAhead=1 if ahead else 0, s_maxFramesInFlight=4))
    seq = E.Reconstruction.makeFrames(poses, [f.depth_ptr for f in frames], [f.color_ptr for f in frames])
    hits = 0
    for k in range(n):
        recon.run(seq, k, 1)
        recon.synchronize()
        if k > 0:  # the maps of render(pose k-1), which ran beside alloc + compactify of frame k
            want = ref.render(poses[k - 1])
            assert_maps_equal(ray.download(), want, f"frame {k}")
            hits += int((want["depth"] != -np.inf).sum())
        ref.integrate(poses[k], host[k][0], host[k][1])
        canonical.assert_same_scene(scene.state(), ref.state(), f"frame {k}")
    assert hits > 10000
    st = recon.getStats()
    assert st["frames"] == n and st["invalidFrames"] == 0
    stw = scene.getState()
    assert stw[T.STATE_HEAP_UNDERFLOW] == 0 and stw[T.STATE_INSERT_FAILED] == 0


def test_native_loop_in_one_call_equals_frame_by_frame(E, oracle_lib):
    """all frames enqueued by ONE call (the host far ahead of the device, alloc beside the ray cast) == the oracle"""
    O = oracle_lib
    hp, cp, rp = small_config(160, 120, num_buckets=1 << 17, num_sdf_blocks=1 << 12)
    n = 40
    poses = [shifted_pose(k, 150) for k in range(n)]
    frames, host = make_inputs(E, O, cp, poses, SHIFTED_S1, 0)
    assert oracle_online_is_deterministic(O, hp, cp, rp, poses, host, True, 7)
    opt = T.make_scene_options(offline=False, gc=True, starve=7)
    ref = O.OracleScene(hp, cp, rp, opt)
    for k in range(n):
        ref.integrate(poses[k], host[k][0], host[k][1])
    want = ref.render(poses[n - 1])
    seq = E.Reconstruction.makeFrames(poses + [poses[n - 1]], [f.depth_ptr for f in frames] + [frames[-1].depth_ptr],
                                      [f.color_ptr for f in frames] + [frames[-1].color_ptr])
    for in_flight in (0, 3):
        scene, ray = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp)
        recon = E.Reconstruction(scene, ray, None, cp, E.Reconstruction.defaultOptions(s_allocAhead=1, s_maxFramesInFlight=in_flight))
        recon.run(seq, 0, n)
        recon.synchronize()
        canonical.assert_same_scene(scene.state(), ref.state(), f"{n} frames, {in_flight} in flight")
        ray.render(scene.getHashData(), scene.getHashParams(), cp, poses[n - 1])
        assert_maps_equal(ray.download(), want, "render after the sequence")
        recon.close()


def test_co_launch_through_the_host_classes(E, oracle_lib):
    """integrateAhead -> render(..., coLaunch=job) -> integrateFinish by hand: the alloc pass rides in the ray caster's
    launch and the compactify pass in computeNormals'; a job nobody launched is launched by integrateFinish; offline mode
    hands out no job.  Every variant leaves the oracle's scene and maps."""
    O = oracle_lib
    hp, cp, rp = small_config(160, 120, num_buckets=1 << 17, num_sdf_blocks=1 << 12)
    n = 8
    poses = [shifted_pose(k, 90) for k in range(n)]
    frames, host = make_inputs(E, O, cp, poses, SHIFTED_S1, 0)
    for offline, use_job in ((False, True), (False, False), (True, True)):
        opt = T.make_scene_options(offline=offline, gc=True, starve=3)
        scene, ray, ref = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp), O.OracleScene(hp, cp, rp, opt)
        for k in range(n):
            job = scene.integrateAhead(poses[k], frames[k], cp, None)
            assert (job is None) == offline
            if k > 0:
                ray.render(scene.getHashData(), scene.getHashParams(), cp, poses[k - 1], coLaunch=job if use_job else None)
                assert_maps_equal(ray.download(), ref.render(poses[k - 1]), f"offline={offline} job={use_job} frame {k}")
            if job is not None and use_job and k > 0:
                assert job.contents.allocLaunched == 1 and job.contents.compactifyLaunched == 1
            scene.integrateFinish(frames[k], cp)
            ref.integrate(poses[k], host[k][0], host[k][1])
            canonical.assert_same_scene(scene.state(), ref.state(), f"offline={offline} job={use_job} frame {k}")
        with pytest.raises(Exception):
            scene.integrateFinish(frames[0], cp)  # nothing pending


def test_splat_made_ahead_is_dropped_when_the_table_changes(E, oracle_lib):
    """render(..., job) also makes the NEXT render's interval splat (inside its computeNormals launch).  That splat is
    used only by a render for the very pose, table and frame it was made for: a reset, an extra integrate, another
    pose or a render without a job in between must fall back to a fresh splat, and the maps must be the oracle's."""
    O = oracle_lib
    hp, cp, rp = small_config(160, 120, num_buckets=1 << 17, num_sdf_blocks=1 << 12)
    poses = [shifted_pose(k, 90) for k in range(8)]
    frames, host = make_inputs(E, O, cp, poses, SHIFTED_S1, 0)
    opt = T.make_scene_options(offline=False, gc=True, starve=3)
    scene, ray, ref = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp), O.OracleScene(hp, cp, rp, opt)

    def frame(k, render_pose=None, use_job=True):
        job = scene.integrateAhead(poses[k], frames[k], cp, None)
        if render_pose is not None:
            ray.render(scene.getHashData(), scene.getHashParams(), cp, render_pose, coLaunch=job if use_job else None)
            assert_maps_equal(ray.download(), ref.render(render_pose), f"frame {k}")
        scene.integrateFinish(frames[k], cp)
        ref.integrate(poses[k], host[k][0], host[k][1])
        canonical.assert_same_scene(scene.state(), ref.state(), f"frame {k}")

    frame(0)
    frame(1, poses[0])               # makes the splat for pose 1 ahead
    frame(2, poses[1])               # uses it, makes the one for pose 2
    frame(3, poses[0])               # another pose than the splat was made for: dropped
    frame(4, poses[3], use_job=False)  # no job: a fresh splat, nothing made ahead
    frame(5, poses[4])               # makes the one for pose 5
    scene.integrate(poses[6], frames[6], cp, None)  # an extra frame in between: the frame number no longer fits
    ref.integrate(poses[6], host[6][0], host[6][1])
    frame(7, poses[6])
    # a reset in between
    job = scene.integrateAhead(poses[0], frames[0], cp, None)
    ray.render(scene.getHashData(), scene.getHashParams(), cp, poses[7], coLaunch=job)
    scene.integrateFinish(frames[0], cp)
    scene.reset()
    ref.reset()
    frame(0)
    frame(1, poses[0])
    assert (ray.download()["depth"] != -np.inf).sum() > 500


@pytest.mark.parametrize("width,height", [(203, 149), (64, 48), (9, 7)])
def test_ragged_image_sizes_through_the_co_launches(E, oracle_lib, width, height):
    """image sizes that are no multiple of the 8x8 tiles (and one smaller than two tiles), no colour on some frames, zero
    frames in a call: the riders' workgroup arithmetic (alloc tiles behind the ray caster's, compactify and splat behind
    computeNormals') must not depend on round numbers"""
    O = oracle_lib
    hp, cp, rp = small_config(width, height, num_buckets=1 << 16, num_sdf_blocks=1 << 11)
    n = 6
    poses = [shifted_pose(k, 80) for k in range(n)]
    frames, host = make_inputs(E, O, cp, poses, SHIFTED_S1, 0)
    assert oracle_online_is_deterministic(O, hp, cp, rp, poses, host, True, 2)
    opt = T.make_scene_options(offline=False, gc=True, starve=2)
    scene, ray, ref = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp), O.OracleScene(hp, cp, rp, opt)
    recon = E.Reconstruction(scene, ray, None, cp)
    seq = E.Reconstruction.makeFrames(poses, [f.depth_ptr for f in frames], [f.color_ptr for f in frames])
    recon.run(seq, 0, 0)  # nothing to do
    for k in range(n):
        recon.run(seq, k, 1)
        recon.synchronize()
        if k > 0:
            assert_maps_equal(ray.download(), ref.render(poses[k - 1]), f"{width}x{height} frame {k}")
        ref.integrate(poses[k], host[k][0], host[k][1])
        canonical.assert_same_scene(scene.state(), ref.state(), f"{width}x{height} frame {k}")


def test_frames_without_colour_integrate_nothing(E, oracle_lib):
    """integrateDepthMapKernel reads the colour first and integrates only where it is valid
    (DSC/CUDASceneRepHashSDF.cu:443): a frame without a colour map allocates its blocks and integrates nothing"""
    O = oracle_lib
    hp, cp, rp = small_config(96, 72)
    poses = [shifted_pose(k, 60) for k in range(3)]
    opt = T.make_scene_options(offline=True, gc=False)
    scene, ray, ref = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp), O.OracleScene(hp, cp, rp, opt)
    recon = E.Reconstruction(scene, ray, None, cp)
    frames = [E.synth_frame(SHIFTED_S1, 0, p, cp) for p in poses]
    seq = E.Reconstruction.makeFrames(poses, [f.depth_ptr for f in frames], [None, frames[1].color_ptr, None])
    recon.run(seq)
    recon.synchronize()
    for k, p in enumerate(poses):
        d, c = O.synth_frame(SHIFTED_S1, 0, p, cp)
        ref.integrate(p, d, c if k == 1 else None)
    canonical.assert_same_scene(scene.state(), ref.state(), "colour on the middle frame only")
    assert (scene.state()["voxels"]["weight"] > 0).sum() > 100


def test_an_empty_voxel_pool_is_reported_by_the_loop(E):
    """a pool of 64 blocks for a frame that wants 150: the device raises VH_STATE_HEAP_UNDERFLOW, hands out no block it does
    not have, and the loop's statistics show the count (nothing else tells a caller of vh_reconstruction_run)"""
    hp, cp, rp = small_config(160, 120, num_sdf_blocks=64)
    poses = [shifted_pose(k, 60) for k in range(3)]
    scene, ray = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=False, gc=False)), E.CUDARayCastSDF(rp)
    recon = E.Reconstruction(scene, ray, None, cp)
    frames = [E.synth_frame(SHIFTED_S1, 0, p, cp) for p in poses]
    seq = E.Reconstruction.makeFrames(poses, [f.depth_ptr for f in frames], [f.color_ptr for f in frames])
    recon.run(seq)
    recon.synchronize()
    st = recon.getStats()
    assert st["frames"] == 3 and st["heapUnderflows"] > 0 and st["failedInserts"] == 0, st
    s = scene.state()  # (the table's invariants hold: no block is both free and allocated)
    assert s["num_occupied"] <= 64 and s["heap_free"] == 64 - s["num_occupied"]


def test_invalid_pose_is_skipped_and_loop_can_restart(E, oracle_lib):
    """DSC/DepthSensing.cpp:738-741: a frame whose recorded pose starts with -inf / NaN is not processed"""
    O = oracle_lib
    hp, cp, rp = small_config(96, 72)
    poses = [synth.orbit_pose(k, n_frames=100) for k in range(5)]
    frames, host = make_inputs(E, O, cp, poses, synth.S1_SPHERES, 0)
    bad = np.array(poses[2], dtype=np.float32).copy()
    bad[0] = -np.inf
    nan = np.array(poses[3], dtype=np.float32).copy()
    nan[0] = np.nan
    opt = T.make_scene_options(offline=True, gc=False)
    scene, ray, ref = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp), O.OracleScene(hp, cp, rp, opt)
    recon = E.Reconstruction(scene, ray, None, cp)
    seq = E.Reconstruction.makeFrames([poses[0], poses[1], bad, nan, poses[4]], [f.depth_ptr for f in frames], [f.color_ptr for f in frames])
    recon.run(seq)
    recon.synchronize()
    for k in (0, 1, 4):
        ref.integrate(poses[k], host[k][0], host[k][1])
    canonical.assert_same_scene(scene.state(), ref.state(), "two invalid frames skipped")
    st = recon.getStats()
    assert st["frames"] == 3 and st["invalidFrames"] == 2
    recon.reset()
    scene.reset()
    ref.reset()
    recon.run(seq, 0, 2)
    recon.synchronize()
    for k in (0, 1):
        ref.integrate(poses[k], host[k][0], host[k][1])
    canonical.assert_same_scene(scene.state(), ref.state(), "after a restart")
    assert recon.getStats()["frames"] == 2


def test_host_fed_loop_uploads_what_the_sensor_delivers(E, oracle_lib):
    """s_framesOnHost: float depth + RGBX bytes in host memory, uploaded on the loop's copy stream into two staging
    slots and converted as CUDARGBDAdapter::process does (DSC/CUDARGBDAdapter.cpp:107-131, convertColorRawToFloat4:
    c / 255, black = no colour); the scene equals the oracle's on the same converted inputs"""
    O = oracle_lib
    hp, cp, rp = small_config(160, 120, num_buckets=1 << 17, num_sdf_blocks=1 << 12)
    n = 9
    poses = [shifted_pose(k, 100) for k in range(n)]
    host = [O.synth_frame(SHIFTED_S1, 0, p, cp) for p in poses]
    h_depth = [np.ascontiguousarray(d, dtype=np.float32) for d, _ in host]
    h_rgbx = []
    conv = []
    for d, c in host:
        b = np.zeros(c.shape, dtype=np.uint8)
        valid = c[..., 0] != -np.inf
        b[valid] = np.clip(np.float32(255.0) * c[valid], 0, 255).astype(np.uint8)
        b[..., 3] = 255
        b[~valid] = 0
        h_rgbx.append(np.ascontiguousarray(b))
        black = (b[..., 0] == 0) & (b[..., 1] == 0) & (b[..., 2] == 0)
        f = b.astype(np.float32) / np.float32(255.0)
        f[..., 3] = (b[..., 3] // 255).astype(np.float32)
        f[black] = -np.inf
        conv.append(f)
    opt = T.make_scene_options(offline=False, gc=True, starve=4)
    ref = O.OracleScene(hp, cp, rp, opt)
    on = O.OracleScene(hp, cp, rp, opt)
    off = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=True, starve=4))
    for k in range(n):
        on.integrate(poses[k], h_depth[k], conv[k])
        off.integrate(poses[k], h_depth[k], conv[k])
        assert np.array_equal(canonical.block_positions(on.hash_table()), canonical.block_positions(off.hash_table()))
    scene, ray = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp)
    recon = E.Reconstruction(scene, ray, None, cp, E.Reconstruction.defaultOptions(s_framesOnHost=1, s_allocAhead=1, s_maxFramesInFlight=2))
    seq = E.Reconstruction.makeFrames(poses, [a.ctypes.data for a in h_depth], [a.ctypes.data for a in h_rgbx])
    recon.run(seq)
    recon.synchronize()
    for k in range(n):
        ref.integrate(poses[k], h_depth[k], conv[k])
    canonical.assert_same_scene(scene.state(), ref.state(), "host-fed frames")
    st = recon.getStats()
    assert st["uploadsTimed"] == (n + 7) // 8 and st["uploadMs"] > 0 and st["uploadBytes"] == 8 * cp.m_imageWidth * cp.m_imageHeight  # (every 8th upload is timed)
    ray.render(scene.getHashData(), scene.getHashParams(), cp, poses[-1])
    assert_maps_equal(ray.download(), ref.render(poses[-1]), "render of the host-fed scene")
    # the same frames in PINNED memory: read by the upload kernel straight over the link (numpy memory above went
    # through hipMemcpyAsync), more frames than staging slots in flight
    from voxelhashing_amd.lib import PinnedArray
    p_depth = [PinnedArray.from_numpy(a) for a in h_depth]
    p_rgbx = [PinnedArray.from_numpy(a) for a in h_rgbx]
    scene2, ray2 = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp)
    recon2 = E.Reconstruction(scene2, ray2, None, cp, E.Reconstruction.defaultOptions(s_framesOnHost=1, s_allocAhead=1, s_maxFramesInFlight=16))
    seq2 = E.Reconstruction.makeFrames(poses, [t.ptr for t in p_depth], [t.ptr for t in p_rgbx])
    recon2.run(seq2)
    recon2.synchronize()
    canonical.assert_same_scene(scene2.state(), ref.state(), "host-fed frames, pinned")


# ---- full size, against the oracle --------------------------------------------------------------------------------

def full_size_against_oracle(E, O, cfg, n_frames, pose_step, vh, n_blocks=1 << 14):
    c = dict(synth.CONFIGS[cfg])
    c.update(num_sdf_blocks=n_blocks)  # the table at full size; the voxel pool sized to be downloadable
    hp, cp, rp = synth.config_params(c)
    n_tiles = ((cp.m_imageWidth + 7) // 8) * ((cp.m_imageHeight + 7) // 8)
    split = vh.vh_render_split_tiles(cp.m_imageWidth, cp.m_imageHeight)
    assert n_tiles >= 1024 and split > 0, "the image must be large enough for the split-tile schedule"
    spheres, inside, radius = synth.scene(c["scene"])
    opt = T.make_scene_options(offline=True, gc=True, starve=2)
    scene, ray, ref = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp), O.OracleScene(hp, cp, rp, opt)
    poses = [synth.orbit_pose(k * pose_step, 1000, radius) for k in range(n_frames)]
    frame = E.DepthFrame(cp)
    hits = 0
    for k, pose in enumerate(poses):
        E.synth_frame(spheres, inside, pose, cp, out=frame)
        d, col = O.synth_frame(spheres, inside, pose, cp)
        if k > 0:
            # from the second render on the launch order is the cost-sorted one and the dearest tiles are split
            ray.render(scene.getHashData(), scene.getHashParams(), cp, poses[k - 1])
            want = ref.render(poses[k - 1])
            assert_maps_equal(ray.download(), want, f"{cfg} render {k}")
            hits += int((want["depth"] != -np.inf).sum())
        scene.integrate(pose, frame, cp, None)
        ref.integrate(pose, d, col)
        canonical.assert_same_scene(scene.state(), ref.state(), f"{cfg} frame {k}")
    # twice more from the last pose: the schedule now holds this view's costs
    for i in range(2):
        ray.render(scene.getHashData(), scene.getHashParams(), cp, poses[-1])
    want = ref.render(poses[-1])
    assert_maps_equal(ray.download(), want, f"{cfg} last render")
    hits += int((want["depth"] != -np.inf).sum())
    return hits, scene.state()["num_occupied"]


def test_cfg1_full_size_equals_oracle(E, oracle_lib, vh):
    """BASELINE.json configs[0]: 640x480, 4 cm voxels, 2^18 buckets -- on the HIP path and on the oracle, bit for bit:
    all four maps through the scheduled split-tile ray caster, and the canonical scene after every frame"""
    hits, blocks = full_size_against_oracle(E, oracle_lib, "cfg1", 3, 7, vh)
    assert hits > 100000 and blocks > 100


def test_cfg2_full_size_equals_oracle(E, oracle_lib, vh):
    """BASELINE.json configs[1]: the same at 500 k buckets / 5 M entries"""
    hits, blocks = full_size_against_oracle(E, oracle_lib, "cfg2", 3, 11, vh)
    assert hits > 100000 and blocks > 100


@pytest.mark.slow
def test_cfg4_one_1080p_frame_equals_oracle(E, oracle_lib, vh):
    """BASELINE.json configs[3]: 1920x1080, 2 cm voxels"""
    hits, blocks = full_size_against_oracle(E, oracle_lib, "cfg4", 2, 9, vh, n_blocks=1 << 15)
    assert hits > 400000 and blocks > 300
