"""World-chunk streaming (CUDASceneRepChunkGrid): blocks outside a sphere
around the camera move to the host chunk grid, chunks entirely inside come
back; alloc is suppressed for streamed-out chunks through the bit mask.
GPU engine vs the oracle twin (oracle/chunk_grid.py), single-threaded passes
for determinism, plus a smoke run of the worker-thread protocol."""
import os
import tempfile

import numpy as np
import pytest

from helpers import small_config
from voxelhashing_amd import canonical, synth, vhtypes as T

pytestmark = pytest.mark.gpu

EXT, DIMS, MINP, PARTS = (0.5, 0.5, 0.5), (65, 65, 65), (-32, -32, -32), 4
STREAM_POS = np.array([0.0, 0.0, 1.6, 1.0], dtype=np.float32)
RADIUS = 1.2


def sorted_blocks(descs, blocks):
    order = canonical.lexsort_pos(np.ascontiguousarray(descs["pos"]))
    return np.ascontiguousarray(descs["pos"][order]), np.ascontiguousarray(blocks[order])


def make_pair(E, O, offline=True):
    from oracle.chunk_grid import OracleChunkGrid
    hp, cp, rp = small_config(64, 48, streaming_extents=EXT, streaming_dims=DIMS, streaming_min=MINP)
    opt = T.make_scene_options(offline=offline, gc=True, starve=15, streaming_out_parts=PARTS)
    gs = E.CUDASceneRepHashSDF(hp, opt)
    gg = E.CUDASceneRepChunkGrid(gs, EXT, DIMS, MINP, 16, False, PARTS)
    os_ = O.OracleScene(hp, cp, rp, opt)
    og = OracleChunkGrid(os_, EXT, DIMS, MINP, PARTS)
    return hp, cp, rp, gs, gg, os_, og


def compare(gs, gg, os_, og, what):
    canonical.assert_same_scene(gs.state(), os_.state(), what)
    gd, gb = sorted_blocks(*gg.downloadHostBlocks())
    od, ob = sorted_blocks(*og.host_blocks())
    assert np.array_equal(gd, od), f"{what}: host chunk grid holds different blocks"
    assert gb.tobytes() == ob.tobytes(), f"{what}: host voxel payloads differ"
    st = gg.getStatistics()
    assert st["blocks"] == len(od) and st["bits"] == og.statistics()["bits"]
    gg.debugCheckForDuplicates()


def test_stream_out_in_sequence(vh, oracle_lib):
    from voxelhashing_amd import engine as E
    O = oracle_lib
    hp, cp, rp, gs, gg, os_, og = make_pair(E, O)
    frame = E.DepthFrame(cp)
    moved_out = moved_in = 0
    for k in range(24):
        pose = synth.orbit_pose(k, 40)
        p = (pose.reshape(4, 4) @ STREAM_POS)[:3]  # DepthSensing.cpp:882-883
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        depth, color = O.synth_frame(synth.S1_SPHERES, 0, pose, cp)
        n_out, n_in = gg.streamOutToCPU(p, RADIUS, True), gg.streamInToGPU(p, RADIUS, True)
        assert (n_out, n_in) == (og.stream_out_to_cpu(p, RADIUS, True), og.stream_in_to_gpu(p, RADIUS, True)), f"frame {k}"
        moved_out += n_out
        moved_in += n_in
        mask = gg.getBitMaskGPU()
        gs.integrate(pose, frame, cp, mask)
        os_.integrate(pose, depth, color, og.bitmask)
        compare(gs, gg, os_, og, f"frame {k}")
    assert moved_out > 20 and moved_in > 5, (moved_out, moved_in)
    st = gs.getState()
    assert st[T.STATE_INSERT_FAILED] == 0 and st[T.STATE_HEAP_UNDERFLOW] == 0


def test_stream_everything_out_and_back_and_hashgrid_file(vh, oracle_lib):
    """streamOutToCPUAll empties the GPU hash; saveToFile/loadFromFile round-trip the .hashgrid;
    streaming everything back restores the scene bit for bit"""
    from voxelhashing_amd import engine as E
    O = oracle_lib
    hp, cp, rp, gs, gg, os_, og = make_pair(E, O)
    frame = E.DepthFrame(cp)
    for k in range(3):
        pose = synth.orbit_pose(k, 40)
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        gs.integrate(pose, frame, cp, gg.getBitMaskGPU())
    before = gs.state()
    assert before["num_occupied"] > 30
    gg.streamOutToCPUAll()
    empty = gs.state()
    assert empty["num_occupied"] == 0 and empty["heap_free"] == hp.m_numSDFBlocks
    assert gg.getStatistics()["blocks"] == before["num_occupied"]
    hd, hb = sorted_blocks(*gg.downloadHostBlocks())
    assert np.array_equal(hd, before["positions"]) and hb.tobytes() == before["voxels"].tobytes()

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scene.hashgrid")
        centre, big = np.zeros(3, np.float32), 1000.0
        gg.saveToFile(path, centre, big)  # streams the sphere around `centre` back in afterwards
        size = os.path.getsize(path)
        # header 4+4+12+12+12+12+4 + count 4, then per chunk 4 + 8 + n*4096 + 8 + n*16 (mLib BinaryDataStreamFile)
        n_blocks = before["num_occupied"]
        assert (size - 64 - n_blocks * (4096 + 16)) % 20 == 0
        after_save = gs.state()
        canonical.assert_same_scene(before, after_save, "after saveToFile")
        gg.loadFromFile(path, centre, big)  # streams everything out, then reads the file into the host grid
        assert gs.state()["num_occupied"] == 0 and gg.getStatistics()["blocks"] == n_blocks
        n = gg.streamInToGPUAll(centre, big, True)
        assert n == n_blocks
        canonical.assert_same_scene(before, gs.state(), "after loadFromFile + streamInToGPUAll")
        # a wrong version is refused
        raw = bytearray(open(path, "rb").read())
        raw[0] = 9
        open(path, "wb").write(raw)
        with pytest.raises(Exception, match="version"):
            gg.loadFromFile(path, centre, big)


def test_worker_thread_protocol_smoke(vh, oracle_lib):
    """the two-thread producer/consumer hand-off (streamOutToCPUPass0GPU / streamInToGPUPass1GPU on the caller,
    pass 1 / pass 0 on the worker): no deadlock, no block lost or duplicated"""
    from voxelhashing_amd import engine as E
    hp, cp, rp = small_config(64, 48, streaming_extents=EXT, streaming_dims=DIMS, streaming_min=MINP)
    opt = T.make_scene_options(offline=False, gc=False, streaming_out_parts=PARTS)
    gs = E.CUDASceneRepHashSDF(hp, opt)
    gg = E.CUDASceneRepChunkGrid(gs, EXT, DIMS, MINP, 16, True, PARTS)  # worker thread running
    frame = E.DepthFrame(cp)
    for k in range(30):
        pose = synth.orbit_pose(k, 40)
        p = (pose.reshape(4, 4) @ STREAM_POS)[:3]
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        gg.streamOutToCPUPass0GPU(p, RADIUS, True, True)
        gg.streamInToGPUPass1GPU(True)
        gs.integrate(pose, frame, cp, gg.getBitMaskGPU())
    gg.reset()  # stops the worker, clears the grid, restarts it
    gg.close()
    s = gs.state(with_voxels=False)
    assert s["num_occupied"] > 10
    assert gs.debugHash()["duplicates"] == 0


def test_per_frame_streaming_at_cfg3_size_keeps_every_invariant(vh):
    """BASELINE.json configs[2] as DepthSensing.cpp:881-903 runs it: cfg3's tables (20 M entries, 1 cm voxels, 640x480),
    80 parts, the worker thread on, streamOutToCPUPass0GPU / streamInToGPUPass1GPU / integrate(..., getBitMaskGPU()) every
    frame, through the native frame loop, 120 frames of the orbit.  (With the reference's own sphere -- radius 3.9 m --
    nothing of scene S1 ever leaves; here the sphere is 1.2 m around a point 1.6 m in front of the camera and the chunks
    are 0.5 m, so blocks leave and come back all the time.)  Size-independent properties: no block is in the table and in
    the host grid at once (debugCheckForDuplicates), heap and table partition the block pool and every free block is zero
    (canonical.check_invariants via state()), the occupancy summary is in step with the table, no status word is raised."""
    from voxelhashing_amd import engine as E
    c = dict(synth.CONFIGS["cfg3"])
    c.update(num_sdf_blocks=1 << 16)  # the table at full size; the voxel pool sized to be downloadable
    hp, cp, rp = synth.config_params(c)
    hp.m_streamingVoxelExtents[:] = EXT
    hp.m_streamingGridDimensions[:] = DIMS
    hp.m_streamingMinGridPos[:] = MINP
    opt = T.make_scene_options(offline=False, gc=True, starve=15, streaming_out_parts=80)
    scene, ray = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp)
    grid = E.CUDASceneRepChunkGrid(scene, EXT, DIMS, MINP, 2000, True, 80)
    n = 120
    poses = [synth.orbit_pose(k, 240) for k in range(n)]
    frames = [E.synth_frame(synth.S1_SPHERES, 0, p, cp) for p in poses]
    recon = E.Reconstruction(scene, ray, grid, cp, E.Reconstruction.defaultOptions(s_streamingEnabled=1, s_streamingPos=STREAM_POS[:3], s_streamingRadius=RADIUS,
                                                                                  s_maxFramesInFlight=8))
    seq = E.Reconstruction.makeFrames(poses, [f.depth_ptr for f in frames], [f.color_ptr for f in frames])
    for k0 in range(0, n, 40):
        recon.run(seq, k0, 40)
        recon.synchronize()
        grid.debugCheckForDuplicates()
        st = scene.getState()
        assert st[T.STATE_HEAP_UNDERFLOW] == 0 and st[T.STATE_INSERT_FAILED] == 0
    stats = recon.getStats()
    assert stats["frames"] == n
    assert stats["blocksStreamedOut"] > 100 and stats["blocksStreamedIn"] > 10, stats  # (one chunk per frame comes back at most)
    grid.reset()  # stops the worker (the host grid is dropped), so that the table can be read in peace
    s = scene.state()  # invariants + occupancy summary
    assert s["num_occupied"] > 500
    assert scene.debugHash()["duplicates"] == 0
    hits = (ray.download()["depth"] != -np.inf).sum()
    assert hits > 20000
    recon.close()
    grid.close()


@pytest.mark.parametrize("radius", [RADIUS, 5.0])
def test_streaming_step_decided_ahead_changes_nothing(vh, radius):
    """The native loop asks the device a frame early whether the next stream-out pass will find anything
    (vh_stream_out_probe) and, when neither that nor the worker has anything to move, runs the frame with its three
    launches as if streaming were off.  The same 60 frames with the question asked (s_allocAhead on) and without
    (off: every frame takes the reference's order of calls): the table, the voxels, the host chunk grid, the ray-cast
    maps after every batch and the streaming counters are the same, and both kinds of frame occurred."""
    from voxelhashing_amd import engine as E
    hp, cp, rp = small_config(160, 120, num_buckets=1 << 15, num_sdf_blocks=1 << 13, streaming_extents=EXT, streaming_dims=DIMS, streaming_min=MINP)
    n = 60
    off = np.array([7.3, 5.1, 3.7], np.float32)  # away from the origin: online alloc is deterministic there (test_gpu_frame_loop.py)
    spheres = synth.S1_SPHERES.copy()
    spheres[:, :3] += off
    poses = []
    for k in range(n):
        q = np.array(synth.orbit_pose(k, 120), dtype=np.float32).copy()
        q[3] += off[0]; q[7] += off[1]; q[11] += off[2]
        poses.append(q)

    def run(ahead):
        opt = T.make_scene_options(offline=False, gc=True, starve=15, streaming_out_parts=PARTS)
        scene, ray = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp)
        grid = E.CUDASceneRepChunkGrid(scene, EXT, DIMS, MINP, 64, True, PARTS)
        frames = [E.synth_frame(spheres, 0, p, cp) for p in poses]
        recon = E.Reconstruction(scene, ray, grid, cp, E.Reconstruction.defaultOptions(s_streamingEnabled=1, s_streamingPos=STREAM_POS[:3], s_streamingRadius=radius,
                                                                                      s_allocAhead=1 if ahead else 0, s_maxFramesInFlight=4))
        seq = E.Reconstruction.makeFrames(poses, [f.depth_ptr for f in frames], [f.color_ptr for f in frames])
        maps = []
        for k0 in range(0, n, 20):
            recon.run(seq, k0, 20)
            recon.synchronize()
            maps.append(ray.download())
        stats = recon.getStats()
        host = sorted_blocks(*grid.downloadHostBlocks())
        grid.debugCheckForDuplicates()
        grid.reset()
        state = scene.state()
        recon.close()
        grid.close()
        return state, host, maps, stats

    sa, ha, ma, ta = run(True)
    sb, hb, mb, tb = run(False)
    canonical.assert_same_scene(sa, sb, "decided ahead vs the reference's order")
    assert np.array_equal(ha[0], hb[0]) and ha[1].tobytes() == hb[1].tobytes(), "host chunk grids differ"
    for i, (a, b) in enumerate(zip(ma, mb)):
        for m in ("depth", "depth4", "colors", "normals"):
            assert np.array_equal(a[m].view(np.uint32), b[m].view(np.uint32)), f"batch {i}: raycast map {m} differs"
    assert (ta["blocksStreamedOut"], ta["blocksStreamedIn"]) == (tb["blocksStreamedOut"], tb["blocksStreamedIn"])
    assert tb["streamingStepsSkipped"] == 0 and ta["frames"] == n
    if radius == RADIUS:  # blocks leave in most frames, a few frames have nothing to move
        assert ta["blocksStreamedOut"] > 20 and 1 <= ta["streamingStepsSkipped"] < n - 5, ta
    else:  # the sphere holds the whole scene: every frame but the first of each run() call (nobody asked about that one)
        assert ta["blocksStreamedOut"] == 0 and ta["streamingStepsSkipped"] == n - 3, ta


@pytest.mark.timeout(300)
@pytest.mark.parametrize("radius", [RADIUS, 5.0])
def test_a_failing_ray_cast_unwinds_the_loop(vh, radius):
    """frame() holds two hand-offs across the ray cast: the worker's stream-in mutex (streamInWait ... streamInFinish) and
    the scene's job (integrateAhead ... integrateFinish).  A ray cast that throws in between (injected:
    vh_reconstruction_debug_fail_render) must leave neither behind -- stopping the worker afterwards returns instead of
    deadlocking, the scene accepts integrate() again, no block is lost between the table and the host grid."""
    from voxelhashing_amd import engine as E
    hp, cp, rp = small_config(160, 120, num_buckets=1 << 15, num_sdf_blocks=1 << 13, streaming_extents=EXT, streaming_dims=DIMS, streaming_min=MINP)
    n = 24
    poses = [synth.orbit_pose(k, 120) for k in range(n)]
    for nth in (1, 2, 3, 5, 8, 13):
        opt = T.make_scene_options(offline=False, gc=True, starve=15, streaming_out_parts=PARTS)
        scene, ray = E.CUDASceneRepHashSDF(hp, opt), E.CUDARayCastSDF(rp)
        grid = E.CUDASceneRepChunkGrid(scene, EXT, DIMS, MINP, 64, True, PARTS)
        frames = [E.synth_frame(synth.S1_SPHERES, 0, p, cp) for p in poses]
        recon = E.Reconstruction(scene, ray, grid, cp, E.Reconstruction.defaultOptions(s_streamingEnabled=1, s_streamingPos=STREAM_POS[:3], s_streamingRadius=radius,
                                                                                      s_allocAhead=1, s_maxFramesInFlight=4))
        seq = E.Reconstruction.makeFrames(poses, [f.depth_ptr for f in frames], [f.color_ptr for f in frames])
        recon.run(seq, 0, 4)
        recon.debugFailRender(nth)
        with pytest.raises(Exception, match="injected"):
            recon.run(seq, 4, n - 4)
        recon.synchronize()
        done = recon.getStats()["frames"]
        assert done == 4 + nth - 1
        # the loop goes on where it stopped (the frame that failed is run again), and the worker is still there
        recon.run(seq, done, n - done)
        recon.synchronize()
        assert recon.getStats()["frames"] == n
        grid.debugCheckForDuplicates()
        st = scene.getState()
        assert st[T.STATE_HEAP_UNDERFLOW] == 0 and st[T.STATE_INSERT_FAILED] == 0
        # a second failure, then straight to tearing everything down: stopping the worker must not deadlock
        recon.debugFailRender(1)
        with pytest.raises(Exception, match="injected"):
            recon.run(seq, n - 2, 2)
        scene.integrate(poses[0], frames[0], cp, grid.getBitMaskGPU())  # not refused: no job is pending
        grid.reset()
        s = scene.state()  # invariants
        assert s["num_occupied"] > 10
        recon.close()
        grid.close()
        ray.close()
        scene.close()
