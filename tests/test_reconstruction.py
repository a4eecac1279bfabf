"""The reference's frame loop on a recorded sequence: parameter file + `.sens` file ->
voxelhashing_amd.reconstruction.Reconstruction (reconstruction(), DepthSensing.cpp:720-924).

A synthetic sequence is written as a `.sens` file (zlib depth in millimetres, raw colour, the true trajectory) and
played back: with the recorded trajectory the scene equals the oracle's fed with the very maps the sensor stage
produced, bit for bit; with tracking instead the poses stay on the true trajectory; the run can be recorded and
the recording replays to the same scene; the mesh comes out."""
import os

import numpy as np
import pytest

from voxelhashing_amd import canonical, synth, vhtypes as T

pytestmark = pytest.mark.gpu

W, H, N = 160, 120, 10

PARAMS = """
s_sensorIdx = 8;
s_adapterWidth = 160;
s_adapterHeight = 120;
s_sensorDepthMax = 5.0f;
s_sensorDepthMin = 0.5f;
s_hashNumBuckets = 16384;
s_hashNumSDFBlocks = 8192;
s_hashMaxCollisionLinkedListSize = 7;
s_SDFVoxelSize = 0.02f;
s_SDFMarchingCubeThreshFactor = 10.0f;
s_SDFTruncation = 0.10f;
s_SDFTruncationScale = 0.05f;
s_SDFMaxIntegrationDistance = 4.0f;
s_SDFIntegrationWeightSample = 10;
s_SDFIntegrationWeightMax = 255;
s_SDFRayIncrementFactor = 0.8f;
s_SDFRayThresSampleDistFactor = 50.5f;
s_SDFRayThresDistFactor = 50.0f;
s_SDFUseGradients = false;
s_integrationEnabled = true;
s_trackingEnabled = true;
s_garbageCollectionEnabled = false;
s_garbageCollectionStarve = 15;
s_marchingCubesMaxNumTriangles = 400000;
s_streamingEnabled = false;
s_offlineProcessing = true;
s_playData = true;
s_reconstructionEnabled = true;
"""


def make_sequence(path, O):
    from voxelhashing_amd import sensor_data as SD
    cp = T.make_depth_camera_params(W, H)
    intr = SD.make_intrinsic_matrix(cp.fx, cp.fy, cp.mx, cp.my)
    sd = SD.SensorData.create((W, H), (W, H), intr, depth_shift=1000.0, sensor_name="synthetic S3", depth_type=SD.TYPE_ZLIB_USHORT)
    poses = [synth.orbit_pose(k, n_frames=400) for k in range(N)]
    for k, p in enumerate(poses):
        d, c = O.synth_frame(synth.S3_SPHERES, 0, p, cp)
        mm = np.where(np.isfinite(d), np.floor(1000.0 * d.astype(np.float64) + 0.5), 0).astype(np.uint16)
        rgb = np.clip(np.where(np.isfinite(c[..., :3]), c[..., :3], 0) * 255.0, 0, 255).astype(np.uint8)
        sd.addFrame(rgb, mm, p, 100 + k, 200 + k)
    sd.saveToFile(path)
    return poses, cp


def app_state(extra=""):
    from voxelhashing_amd import reconstruction as R
    return R.read_app_state((PARAMS + extra).encode())


def test_recorded_trajectory_replay_equals_oracle(vh, oracle_lib, tmp_path):
    from voxelhashing_amd import reconstruction as R
    O = oracle_lib
    path = str(tmp_path / "s3.sens")
    poses, _ = make_sequence(path, O)
    g = app_state("s_binaryDumpSensorUseTrajectory = true;\ns_binaryDumpSensorUseTrajectoryOnlyInit = false;\n")
    rec = R.Reconstruction(g, sens_files=[path])
    cp = rec.cp
    assert (cp.m_imageWidth, cp.m_imageHeight) == (W, H)
    osc = O.OracleScene(rec.hp, cp, rec.rp, T.make_scene_options(offline=True, gc=False))
    for k in range(N):
        pose = rec.frame()
        assert np.array_equal(pose, np.asarray(poses[k]).reshape(4, 4)), k
        maps = rec.sensor.download()  # what the sensor stage handed to integrate
        if k > 0:
            want = osc.render(poses[k - 1])
            got = rec.ray.download()
            for m in ("depth", "depth4", "normals", "colors"):
                assert got[m].tobytes() == want[m].tobytes(), (k, m)
        osc.integrate(poses[k], maps["depth"], maps["color"])
        canonical.assert_same_scene(rec.scene.state(), osc.state(), f"frame {k}")
    assert rec.frame() is None and rec.frame_number == N
    assert rec.scene.state()["num_occupied"] > 300
    mesh = rec.extractIsoSurface(str(tmp_path / "scan.ply"))
    assert len(mesh["faces"]) > 10000 and len(mesh["vertices"]) < len(mesh["faces"])  # s_offlineProcessing: merged, indexed
    assert os.path.getsize(tmp_path / "scan.ply") > 100000


def test_tracked_replay_recording_and_second_generation(vh, oracle_lib, tmp_path):
    from voxelhashing_amd import reconstruction as R, sensor_data as SD
    O = oracle_lib
    path = str(tmp_path / "s3.sens")
    poses, _ = make_sequence(path, O)
    out = str(tmp_path / "dump" / "tracked.sens")
    g = app_state(f's_binaryDumpSensorUseTrajectory = false;\ns_recordData = true;\ns_recordDataFile = "{out}";\n')
    rec = R.Reconstruction(g, sens_files=[path])
    assert rec.run() == N and rec.lost_frames == 0
    # without a trajectory the world frame is the first camera frame: compare with inv(T0) * Tk
    t0_inv = np.linalg.inv(np.asarray(poses[0], np.float64).reshape(4, 4))
    for k in (1, N - 1):
        rel = np.linalg.inv(rec.trajectory[k].astype(np.float64)) @ (t0_inv @ np.asarray(poses[k], np.float64).reshape(4, 4))
        ang = np.degrees(np.arccos(np.clip(0.5 * (np.trace(rel[:3, :3]) - 1.0), -1, 1)))
        assert np.linalg.norm(rel[:3, 3]) < 0.006 and ang < 0.15, (k, np.linalg.norm(rel[:3, 3]), ang)
    assert np.array_equal(rec.trajectory[0], np.eye(4, dtype=np.float32))
    first = rec.scene.state()

    assert rec.saveRecordedFramesToFile() == out
    src, dump = SD.SensorData.loadFromFile(path), SD.SensorData.loadFromFile(out)
    assert dump.info().m_numFrames == N and dump.info().m_depthCompressionType == SD.TYPE_ZLIB_USHORT
    for k in range(N):
        a, b = src.frame(k), dump.frame(k)
        assert np.array_equal(a["depth"], b["depth"]) and np.array_equal(a["color"], b["color"]) and a["timeStamps"] == b["timeStamps"]
        assert np.array_equal(b["cameraToWorld"].reshape(4, 4), rec.trajectory[k])

    # second generation: the recording replayed with ITS trajectory rebuilds the same scene
    g2 = app_state("s_binaryDumpSensorUseTrajectory = true;\n")
    rec2 = R.Reconstruction(g2, sens_files=[out])
    assert rec2.run() == N
    canonical.assert_same_scene(rec2.scene.state(), first, "replay of the recording")


def test_replay_with_streaming(vh, oracle_lib, tmp_path):
    """the streaming branch of the loop (DepthSensing.cpp:877-898): blocks outside the sphere around the camera sit
    in the host chunk grid, nothing is lost or duplicated, and the mesh extraction walks the chunk grid"""
    from voxelhashing_amd import reconstruction as R
    O = oracle_lib
    path = str(tmp_path / "s3.sens")
    make_sequence(path, O)
    stream = """s_streamingEnabled = true;
s_streamingVoxelExtents = 0.5f 0.5f 0.5f;
s_streamingGridDimensions = 65 65 65;
s_streamingMinGridPos = -32 -32 -32;
s_streamingInitialChunkListSize = 16;
s_streamingRadius = 1.3f;
s_streamingPos = 0.0f 0.0f 1.8f;
s_streamingOutParts = 4;
s_binaryDumpSensorUseTrajectory = true;
"""
    plain = R.Reconstruction(app_state("s_binaryDumpSensorUseTrajectory = true;\n"), sens_files=[path])
    assert plain.run() == N
    rec = R.Reconstruction(app_state(stream), sens_files=[path])
    assert rec.run() == N
    on_gpu, on_host = rec.scene.state()["num_occupied"], rec.chunk_grid.getStatistics()["blocks"]
    assert on_host > 20 and on_gpu > 100
    rec.chunk_grid.debugCheckForDuplicates()
    assert rec.scene.debugHash()["duplicates"] == 0
    # a block is in exactly one place; alloc is suppressed for streamed-out chunks, so the union can only be smaller
    total = plain.scene.state()["num_occupied"]
    assert 0.8 * total < on_gpu + on_host <= total, (on_gpu, on_host, total)
    mesh = rec.extractIsoSurface()
    assert len(mesh["faces"]) > 8000


def test_recorded_trajectory_as_initial_guess(vh, oracle_lib, tmp_path):
    """s_binaryDumpSensorUseTrajectoryOnlyInit: the recorded motion between two frames moves the model view before ICP
    runs (DepthSensing.cpp:757-765); the tracked poses stay on the recorded trajectory (here the true one, so the
    initial guess is already the answer and ICP must not walk away from it)"""
    from voxelhashing_amd import reconstruction as R
    O = oracle_lib
    path = str(tmp_path / "s3.sens")
    poses, _ = make_sequence(path, O)
    g = app_state("s_binaryDumpSensorUseTrajectory = true;\ns_binaryDumpSensorUseTrajectoryOnlyInit = true;\n")
    rec = R.Reconstruction(g, sens_files=[path])
    assert rec.run() == N and rec.lost_frames == 0
    assert np.array_equal(rec.trajectory[0], np.asarray(poses[0]).reshape(4, 4))  # the first frame has nothing to track against
    for k in range(1, N):
        rel = np.linalg.inv(rec.trajectory[k].astype(np.float64)) @ np.asarray(poses[k], np.float64).reshape(4, 4)
        ang = np.degrees(np.arccos(np.clip(0.5 * (np.trace(rel[:3, :3]) - 1.0), -1, 1)))
        assert np.linalg.norm(rel[:3, 3]) < 0.004 and ang < 0.1, (k, np.linalg.norm(rel[:3, 3]), ang)
        assert not np.array_equal(rec.trajectory[k], np.asarray(poses[k]).reshape(4, 4))  # it is ICP's pose, not the file's
