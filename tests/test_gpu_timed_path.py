"""The code path bench.py times, at the size it times, against the oracle.

bench.py's headline number is the NATIVE frame loop (vh_reconstruction_run) with online alloc and the riders on:
two launches per frame -- the alloc pass of frame k inside k_render's launch (ray cast of pose k-1); compactify, the
next pose's interval splat and (up to 2048 blocks in view) the pass over the voxels inside k_compute_normals'; with
more blocks k_integrate_fused is a third launch.  At 640x480 that k_render launch has a
shape no small image reaches: all 4 800 tiles resident at six waves per SIMD, the dearest tiles split between two
waves, the alloc rider filling the tail.  The tests here put the oracle (oracle/libvh_oracle.so, bit for bit: the
canonical scene and all four ray-cast maps) behind exactly that path:

  * cfg2's tables (500 k buckets / 5 M entries), 640x480, frame by frame and in ONE run() call;
  * one 1080p sequence with cfg4's tables (five rounds of ray-caster workgroups);
  * cfg3's tables (2 M buckets / 20 M entries, 1 cm voxels) with streaming on and the streaming step decided a frame
    ahead, against oracle/chunk_grid.py after every frame (a small sphere, so that blocks do leave).

Order of calls per frame: DSC/DepthSensing.cpp:750-763 (render with the previous pose), :881-900 (stream out / in),
:903 (integrate).  The scene is S1 moved away from the origin, where an ONLINE alloc pass is deterministic (asserted
through the oracle: tests/test_gpu_frame_loop.py explains the hash's symmetry)."""
import numpy as np
import pytest

from helpers import assert_maps_equal
from voxelhashing_amd import canonical, synth, vhtypes as T

pytestmark = pytest.mark.gpu

OFFSET = np.array([7.3, 5.1, 3.7])
SHIFTED_S1 = synth.S1_SPHERES.copy()
SHIFTED_S1[:, :3] += OFFSET


def shifted_pose(k, n_frames=1000):
    q = np.array(synth.orbit_pose(k, n_frames=n_frames), dtype=np.float32).copy()
    q[3] += np.float32(OFFSET[0])
    q[7] += np.float32(OFFSET[1])
    q[11] += np.float32(OFFSET[2])
    return q


@pytest.fixture(scope="module")
def E(vh):
    from voxelhashing_amd import engine
    return engine


def online_is_deterministic(O, hp, cp, rp, poses, host, starve):
    on = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=False, gc=True, starve=starve))
    off = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=True, starve=starve))
    for p, (d, c) in zip(poses, host):
        on.integrate(p, d, c)
        off.integrate(p, d, c)
        if not np.array_equal(canonical.block_positions(on.hash_table()), canonical.block_positions(off.hash_table())):
            return False
    return True


def timed_path_against_oracle(E, O, vh, cfg, n, pose_step, starve, n_blocks, two_launches=True):
    """-> statistics of the frame-by-frame run.  The loop's options are bench.py's: online alloc, s_allocAhead = 1,
    s_maxFramesInFlight = 16, garbage collection on.  two_launches: the frame's pass over the voxels rides in
    computeNormals' launch (the default for scenes of up to 2048 blocks in view), or has its own launch."""
    c = dict(synth.CONFIGS[cfg])
    c.update(num_sdf_blocks=n_blocks)  # the table at full size; the voxel pool sized to be downloadable
    hp, cp, rp = synth.config_params(c)
    n_tiles = ((cp.m_imageWidth + 7) // 8) * ((cp.m_imageHeight + 7) // 8)
    assert n_tiles >= 1024 and vh.vh_render_split_tiles(cp.m_imageWidth, cp.m_imageHeight) > 0, "the launch must have the scheduled split-tile shape"
    poses = [shifted_pose(k * pose_step) for k in range(n)]
    frames = [E.synth_frame(SHIFTED_S1, 0, p, cp) for p in poses]
    host = [O.synth_frame(SHIFTED_S1, 0, p, cp) for p in poses]
    assert online_is_deterministic(O, hp, cp, rp, poses, host, starve), "pick poses without same-pass bucket sharing"
    opt = T.make_scene_options(offline=False, gc=True, starve=starve)
    ropt = dict(s_allocAhead=1, s_maxFramesInFlight=16)
    seq = E.Reconstruction.makeFrames(poses, [f.depth_ptr for f in frames], [f.color_ptr for f in frames])

    # ---- frame by frame: every map and every scene
    scene, ray, ref = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp), O.OracleScene(hp, cp, rp, opt)
    recon = E.Reconstruction(scene, ray, None, cp, E.Reconstruction.defaultOptions(**ropt))
    hits, freed, want_last, states = 0, 0, None, []
    for k in range(n):
        recon.run(seq, k, 1)
        recon.synchronize()
        if k > 0:
            want_last = ref.render(poses[k - 1])
            assert_maps_equal(ray.download(), want_last, f"{cfg} frame {k}: ray cast of pose {k - 1}")
            hits += int((want_last["depth"] != -np.inf).sum())
        ref.integrate(poses[k], host[k][0], host[k][1])
        want = ref.state()
        canonical.assert_same_scene(scene.state(), want, f"{cfg} frame {k}")
        states.append(want["num_occupied"])
        freed += int(np.count_nonzero(ref.decisions()[: int(ref.hp.m_numOccupiedBlocks)]))  # blocks the GC pass flagged
    st = recon.getStats()
    assert st["frames"] == n and st["invalidFrames"] == 0
    # the launches really had the timed shape: riders in every frame that has a ray cast, the splat made ahead used by
    # every ray cast but the first (two launches per frame, three with the pass in its own)
    assert st["framesWithRiders"] == n - 1, st
    assert st["splatsMadeAheadUsed"] == n - 2, st
    # (the pass rides when the scene's last known count of blocks in view is at most 2048: frame by frame that is the
    # count of the frame before)
    riding = sum(1 for k in range(1, n) if states[k - 1] <= 2048) if two_launches else 0
    assert st["framesInTwoLaunches"] == riding, (st, states)
    sw = scene.getState()
    assert sw[T.STATE_HEAP_UNDERFLOW] == 0 and sw[T.STATE_INSERT_FAILED] == 0 and sw[T.STATE_RIDER_GAVE_UP] == 0
    final = ref.state()
    recon.close()
    ray.close()
    scene.close()

    # ---- the same frames in ONE call: the host runs ahead of the device, nothing synchronises between the frames
    scene, ray = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp)
    recon = E.Reconstruction(scene, ray, None, cp, E.Reconstruction.defaultOptions(**ropt))
    recon.run(seq, 0, n)
    recon.synchronize()
    assert_maps_equal(ray.download(), want_last, f"{cfg}, {n} frames in one call: the last ray cast")
    canonical.assert_same_scene(scene.state(), final, f"{cfg}, {n} frames in one call")
    st1 = recon.getStats()
    assert st1["framesWithRiders"] == n - 1 and st1["splatsMadeAheadUsed"] == n - 2, st1
    if not two_launches or max(states) <= 2048:
        assert st1["framesInTwoLaunches"] == riding, st1
    assert scene.getState()[T.STATE_RIDER_GAVE_UP] == 0
    ray.render(scene.getHashData(), scene.getHashParams(), cp, poses[-1])
    assert_maps_equal(ray.download(), ref.render(poses[-1]), f"{cfg}: render after the sequence")
    recon.close()
    ray.close()
    scene.close()
    return dict(hits=hits, freed=freed, blocks=states)


def test_cfg2_native_loop_with_riders_at_the_timed_size(E, oracle_lib, vh):
    """BASELINE.json configs[1] as bench.py runs it: 640x480, 4 cm voxels, 500 k buckets, native loop, online alloc,
    riders on; ten frames, starve every third (so that the fused pass starves and frees)"""
    r = timed_path_against_oracle(E, oracle_lib, vh, "cfg2", 10, 9, 3, 1 << 14)
    assert r["hits"] > 500000 and min(r["blocks"]) > 100 and r["freed"] > 0, r


def test_cfg2_native_loop_with_the_pass_in_its_own_launch(E, oracle_lib, vh, monkeypatch):
    """the same with three launches a frame (what scenes of more than 2048 blocks in view get): the scene reads the
    switch when it is made"""
    monkeypatch.setenv("VH_INTEGRATE_RIDER_MAX_BLOCKS", "0")
    r = timed_path_against_oracle(E, oracle_lib, vh, "cfg2", 6, 9, 3, 1 << 14, two_launches=False)
    assert r["hits"] > 250000 and r["freed"] > 0, r


def test_rider_and_own_launch_agree_over_a_long_run(E, vh, monkeypatch):
    """150 frames of the cfg2 loop in ONE call, once with the pass over the voxels riding in computeNormals' launch (its
    workgroups poll their list entries while the compactify workgroups are still writing others; blocks are freed while
    the splat reads the table) and once with the pass in its own launch: the same scene and the same maps, bit for bit.
    HIP against HIP -- the oracle is behind both paths in the tests above, for ten frames; this is about the many
    interleavings of a long run."""
    c = dict(synth.CONFIGS["cfg2"])
    c.update(num_sdf_blocks=1 << 14)
    hp, cp, rp = synth.config_params(c)
    n = 150
    poses = [shifted_pose(3 * k) for k in range(n)]
    frames = [E.synth_frame(SHIFTED_S1, 0, p, cp) for p in poses]
    opt = T.make_scene_options(offline=False, gc=True, starve=5)
    seq = E.Reconstruction.makeFrames(poses, [f.depth_ptr for f in frames], [f.color_ptr for f in frames])
    results = []
    for most in ("2048", "0"):
        monkeypatch.setenv("VH_INTEGRATE_RIDER_MAX_BLOCKS", most)
        scene, ray = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp)
        recon = E.Reconstruction(scene, ray, None, cp, E.Reconstruction.defaultOptions(s_allocAhead=1, s_maxFramesInFlight=16))
        recon.run(seq, 0, n)
        recon.synchronize()
        st = recon.getStats()
        sw = scene.getState()
        assert sw[T.STATE_RIDER_GAVE_UP] == 0 and sw[T.STATE_HEAP_UNDERFLOW] == 0, sw
        results.append((scene.state(), ray.download(), st["framesInTwoLaunches"]))
        recon.close()
        ray.close()
        scene.close()
    (a, maps_a, two_a), (b, maps_b, two_b) = results
    assert two_a == n - 1 and two_b == 0, (two_a, two_b)
    assert a["num_occupied"] > 100
    canonical.assert_same_scene(a, b, "rider against own launch, 150 frames")
    assert_maps_equal(maps_a, maps_b, "rider against own launch: the last ray cast")


@pytest.mark.slow
def test_cfg4_native_loop_with_riders_at_1080p(E, oracle_lib, vh):
    """BASELINE.json configs[3]: 1920x1080, 2 cm voxels -- 32 400 tiles, five rounds of ray-caster workgroups with the
    alloc rider behind them"""
    r = timed_path_against_oracle(E, oracle_lib, vh, "cfg4", 4, 13, 2, 1 << 15)
    assert r["hits"] > 1000000 and min(r["blocks"]) > 300, r


# ---- cfg3's tables with streaming, the step decided a frame ahead -------------------------------------------------

EXT, DIMS, MINP, PARTS = (0.5, 0.5, 0.5), (65, 65, 65), (-32, -32, -32), 8
STREAM_POS = np.array([0.0, 0.0, 1.6, 1.0], dtype=np.float32)
RADIUS = 1.2


def sorted_blocks(descs, blocks):
    order = canonical.lexsort_pos(np.ascontiguousarray(descs["pos"]))
    return np.ascontiguousarray(descs["pos"][order]), np.ascontiguousarray(blocks[order])


@pytest.mark.slow
def test_cfg3_native_loop_with_streaming_decided_ahead(E, oracle_lib, vh):
    """BASELINE.json configs[2]: 1 cm voxels, 2 M buckets / 20 M entries, streaming on with the worker thread, through the
    native loop with the streaming step pipelined (CUDASceneRepChunkGrid's pipeline: counts on the device, the device's own
    bit mask, the chunk that comes in chosen a frame ahead by the worker): frames with nothing to move take three launches
    (alloc rides), frames with traffic the reference's order of launches without a host wait.  After every call the table,
    the voxels, the host chunk grid and the bit mask's population equal the oracle pair's (oracle/vh_oracle.c +
    oracle/chunk_grid.py, single-threaded: streamOutToCPU, streamInToGPU, integrate), and the ray-cast maps too."""
    from oracle.chunk_grid import OracleChunkGrid
    O = oracle_lib
    c = dict(synth.CONFIGS["cfg3"])
    c.update(num_sdf_blocks=1 << 14)
    hp, cp, rp = synth.config_params(c)
    hp.m_streamingVoxelExtents[:] = EXT
    hp.m_streamingGridDimensions[:] = DIMS
    hp.m_streamingMinGridPos[:] = MINP
    rp = T.make_raycast_params(hp, cp)
    n = 12
    poses = [shifted_pose(k * 5) for k in range(n)]
    frames = [E.synth_frame(SHIFTED_S1, 0, p, cp) for p in poses]
    host = [O.synth_frame(SHIFTED_S1, 0, p, cp) for p in poses]
    opt = T.make_scene_options(offline=False, gc=True, starve=4, streaming_out_parts=PARTS)
    scene, ray = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp)
    grid = E.CUDASceneRepChunkGrid(scene, EXT, DIMS, MINP, 2000, True, PARTS)  # worker thread running
    ref = O.OracleScene(hp, cp, rp, opt)
    og = OracleChunkGrid(ref, EXT, DIMS, MINP, PARTS)
    recon = E.Reconstruction(scene, ray, grid, cp, E.Reconstruction.defaultOptions(
        s_streamingEnabled=1, s_streamingPos=STREAM_POS[:3], s_streamingRadius=RADIUS, s_allocAhead=1, s_maxFramesInFlight=16))
    seq = E.Reconstruction.makeFrames(poses, [f.depth_ptr for f in frames], [f.color_ptr for f in frames])
    out = inn = 0
    deterministic = True
    shadow = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=True, starve=4, streaming_out_parts=PARTS))
    sg = OracleChunkGrid(shadow, EXT, DIMS, MINP, PARTS)
    batch = 3
    for k0 in range(0, n, batch):
        # Three frames per call, the loop told the pose that follows (vh_reconstruction_run_ahead).  Looking at the host grid
        # between the calls makes the grid take back the choice its worker had made for the coming frame (an observer must
        # find the reference's state), so the first frame of every call takes the reference's order of calls and the others
        # run pipelined: both kinds of frame, and the change-over between them, are held to the oracle.
        k1 = min(k0 + batch, n)
        recon.run(seq, k0, k1 - k0, lookahead=True)
        recon.synchronize()
        for k in range(k0, k1):  # the oracle pair, in the reference's order
            if k > 0:
                want = ref.render(poses[k - 1])
            p = (poses[k].reshape(4, 4) @ STREAM_POS)[:3]
            out += og.stream_out_to_cpu(p, RADIUS, True)
            inn += og.stream_in_to_gpu(p, RADIUS, True)
            ref.integrate(poses[k], host[k][0], host[k][1], og.bitmask)
            sg.stream_out_to_cpu(p, RADIUS, True)
            sg.stream_in_to_gpu(p, RADIUS, True)
            shadow.integrate(poses[k], host[k][0], host[k][1], sg.bitmask)
            deterministic = deterministic and np.array_equal(canonical.block_positions(ref.hash_table()), canonical.block_positions(shadow.hash_table()))
        k = k1 - 1
        assert_maps_equal(ray.download(), want, f"cfg3 streaming frame {k}: ray cast of pose {k - 1}")
        canonical.assert_same_scene(scene.state(), ref.state(), f"cfg3 streaming frame {k}")
        gd, gb = sorted_blocks(*grid.downloadHostBlocks())
        od, ob = sorted_blocks(*og.host_blocks())
        assert np.array_equal(gd, od), f"frame {k}: host chunk grid holds different blocks"
        assert gb.tobytes() == ob.tobytes(), f"frame {k}: host voxel payloads differ"
        assert grid.getStatistics()["bits"] == og.statistics()["bits"]
        grid.debugCheckForDuplicates()
    assert deterministic, "pick poses without same-pass bucket sharing"
    st = recon.getStats()
    assert st["frames"] == n
    assert (st["blocksStreamedOut"], st["blocksStreamedIn"]) == (out, inn), (st, out, inn)
    assert out > 50, "the sphere must be small enough for blocks to leave"
    # (streamingStepsSkipped counts the pipelined frames in which nothing moved: three launches)
    assert st["streamingStepsSkipped"] < n, st
    assert st["streamingFramesPipelined"] == n - n // batch, st  # all but the first frame of every call
    sw = scene.getState()
    assert sw[T.STATE_HEAP_UNDERFLOW] == 0 and sw[T.STATE_INSERT_FAILED] == 0
    recon.close()
    grid.close()
