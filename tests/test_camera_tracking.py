"""Projective ICP camera tracking: CUDACameraTrackingMultiRes.

CPU: the numpy oracle recovers a known camera motion from oracle-made maps and reports lost tracking; the tracking
parameter file is read with the reference's rules.
GPU: the kernels' correspondences and linear-system terms equal the oracle's (1e-5 relative: float32 sums in a
different order), applyCT returns the oracle's pose within 1e-4 absolute per matrix entry (the north star's
tolerance), finds the ground-truth pose, flags lost tracking, and a sequence tracked and fused WITHOUT given poses
stays on the true trajectory."""
import ctypes as C
import os

import numpy as np
import pytest

from voxelhashing_amd import synth, vhtypes as T

MINF = np.float32(-np.inf)
TRACK_SPHERES = synth.S3_SPHERES  # S1's big sphere is centred on the orbit: no geometric tracker can see that motion


def maps_for(O, spheres, pose, cp):
    """what CUDARGBDSensor hands to the tracker for a frame taken at `pose`: camera-space positions and normals"""
    depth, _ = O.synth_frame(spheres, 0, pose, cp)
    cam = O.image_op("convert_depth_float_to_camera_space_float4", depth, cp.m_imageWidth, cp.m_imageHeight, cp, out_channels=4)
    return depth, cam, O.compute_normals(cam)


def oracle_model(O, hp, cp, rp, poses, spheres):
    sc = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=False))
    for p in poses:
        d, c = O.synth_frame(spheres, 0, p, cp)
        sc.integrate(p, d, c)
    return sc


def pose_error(a, b):
    a, b = np.asarray(a, np.float64).reshape(4, 4), np.asarray(b, np.float64).reshape(4, 4)
    rel = np.linalg.inv(a) @ b
    ang = np.degrees(np.arccos(np.clip(0.5 * (np.trace(rel[:3, :3]) - 1.0), -1, 1)))
    return float(np.linalg.norm(rel[:3, 3])), float(ang)


def setup_small(w=160, h=120):
    hp = T.make_hash_params(1 << 14, 1 << 13, **synth.PARAM_SETS["P2"])
    cp = T.make_depth_camera_params(w, h)
    rp = T.make_raycast_params(hp, cp)
    return hp, cp, rp


# ---------------------------------------------------------------------------- CPU

def test_oracle_icp_recovers_a_known_motion(oracle_lib):
    from oracle import icp
    O = oracle_lib
    hp, cp, rp = setup_small()
    poses = [synth.orbit_pose(k, n_frames=400) for k in range(4)]  # 0.9 degrees / 3.9 cm per frame
    sc = oracle_model(O, hp, cp, rp, poses[:3], TRACK_SPHERES)
    model = sc.render(poses[2])
    _, cam, nrm = maps_for(O, TRACK_SPHERES, poses[3], cp)
    ts = T.make_tracking_state()
    got, info = icp.apply_ct(cam, nrm, model["depth4"], model["normals"], poses[2], ts, np.eye(4, dtype=np.float32), cp, 3)
    assert got is not None and info["numCorr"] > 3000 and info["matrixCondition"] < 1e5
    dt, da = pose_error(got, poses[3])
    dt0, da0 = pose_error(poses[2], poses[3])
    assert dt < 0.004 and da < 0.1 and dt < 0.15 * dt0, (dt, da, dt0, da0)
    # a step the level thresholds forbid -> lost; no correspondences at all -> lost
    tight = T.make_tracking_state(dist_trans=1e-4)
    lost, _ = icp.apply_ct(cam, nrm, model["depth4"], model["normals"], poses[2], tight, np.eye(4, dtype=np.float32), cp, 3)
    assert lost is None
    empty = np.full_like(cam, MINF)
    lost, info = icp.apply_ct(empty, empty, model["depth4"], model["normals"], poses[2], ts, np.eye(4, dtype=np.float32), cp, 3)
    assert lost is None and info["numCorr"] == 0


def test_tracking_parameter_file():
    from voxelhashing_amd import lib
    L = lib.load()
    ts = T.TrackingState()
    text = b"""s_maxLevels = 2;
s_maxOuterIter[0] = 8;   s_maxOuterIter[9] = 1;
s_maxInnerIter[0] = 2;
s_distThres[0] = 0.15f;
s_normalThres[0] = 0.97f;
s_angleTransThres[0] = 1.0f;// radians
s_distTransThres[0] = 0.5f;
s_residualEarlyOut[0] = 0.01;
s_maxOuterIter[1] = 6;
s_distThres[1] = 0.2f;
"""
    lib.check(L.vh_tracking_state_parse(text, C.byref(ts)), "parse")
    assert ts.s_maxLevels == 2 and ts.numLevelsFound == 2 and list(ts.s_maxOuterIter)[:3] == [8, 6, 0]
    assert ts.s_maxInnerIter[0] == 2 and ts.s_maxInnerIter[1] == 0 and ts.s_distThres[1] == np.float32(0.2)
    assert ts.s_distTransThres[0] == 0.5 and ts.s_residualEarlyOut[0] == np.float32(0.01)
    path = "/root/reference/zParametersTrackingDefault.txt"
    if os.path.exists(path):  # the reference's own file, read as data
        lib.check(L.vh_tracking_state_read(path.encode(), C.byref(ts)), "read")
        assert ts.s_maxLevels == 3 and ts.numLevelsFound == 4 and list(ts.s_maxOuterIter)[:4] == [8, 6, 4, 4]
        assert all(ts.s_distThres[i] == np.float32(0.15) and ts.s_normalThres[i] == np.float32(0.97) for i in range(4))
        want = T.make_tracking_state()
        for name in ("s_maxOuterIter", "s_maxInnerIter", "s_distThres", "s_normalThres", "s_angleTransThres", "s_distTransThres", "s_residualEarlyOut"):
            assert list(getattr(ts, name))[:3] == list(getattr(want, name))[:3], name


# ---------------------------------------------------------------------------- GPU

class GpuRig:
    """scene + ray caster + sensor + tracker on the GPU for a synthetic sequence"""

    def __init__(self, E, hp, cp, rp, levels=3):
        self.E, self.hp, self.cp = E, hp, cp
        self.scene = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=True, gc=False))
        self.ray = E.CUDARayCastSDF(rp)
        W, H = cp.m_imageWidth, cp.m_imageHeight
        self.sensor = E.CUDARGBDSensor((W, H), (W, H), (W, H), cp.fx, cp.fy, cp.mx, cp.my, cp.m_sensorDepthWorldMin, cp.m_sensorDepthWorldMax)
        self.tracker = E.CUDACameraTrackingMultiRes(W, H, levels)

    def feed(self, O, spheres, pose):
        depth, color = O.synth_frame(spheres, 0, pose, self.cp)
        rgbx = np.ascontiguousarray(np.clip(color * 255.0, 0, 255).astype(np.uint8))
        rgbx[..., 3] = 255
        self.sensor.process(depth, rgbx)

    def integrate(self, pose):
        cam = self.sensor.getDepthCameraData()
        self.scene.integrate(pose, self.E.DepthFrame(self.cp, depth_ptr=cam.d_depthData, color_ptr=cam.d_colorData), self.cp, None)

    def track(self, last_pose, ts):
        from voxelhashing_amd import lib
        L = lib.load()
        self.ray.render(self.scene.getHashData(), self.scene.getHashParams(), self.cp, last_pose)
        rd = self.ray.getRayCastData()
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        lib.check(L.vh_rgbd_sensor_get_maps(self.sensor.handle, C.byref(a), C.byref(b), C.byref(c)), "maps")
        return self.tracker.applyCT(a, b, rd.d_depth4, rd.d_normals, last_pose, ts, None, self.cp)


@pytest.mark.gpu
def test_gpu_apply_ct_equals_oracle_and_ground_truth(vh, oracle_lib):
    from oracle import icp
    from voxelhashing_amd import engine as E
    O = oracle_lib
    hp, cp, rp = setup_small()
    poses = [synth.orbit_pose(k, n_frames=400) for k in range(4)]
    rig = GpuRig(E, hp, cp, rp)
    for p in poses[:3]:
        rig.feed(O, TRACK_SPHERES, p)
        rig.integrate(p)
    rig.feed(O, TRACK_SPHERES, poses[3])
    ts = T.make_tracking_state()
    got, lost = rig.track(poses[2], ts)
    assert not lost
    # the oracle on the very same maps
    model = rig.ray.download()
    inp = rig.sensor.download()
    want, info = icp.apply_ct(inp["camera_space"], inp["normals"], model["depth4"], model["normals"], poses[2], ts, np.eye(4, dtype=np.float32), cp, 3)
    assert want is not None
    assert np.abs(got - want).max() < 1e-4, np.abs(got - want).max()  # tolerance: 1e-4 absolute per entry (north star)
    st = rig.tracker.state
    assert st.iterations == info["iterations"] and st.numCorr == info["numCorr"]
    assert abs(st.sumRegError - info["sumRegError"]) <= 1e-4 * max(1.0, info["sumRegError"]) and abs(st.matrixCondition / info["matrixCondition"] - 1.0) < 1e-3
    dt, da = pose_error(got, poses[3])
    assert dt < 0.004 and da < 0.1, (dt, da)
    # lost tracking: a threshold no step can meet, and an input without a single valid pixel
    _, lost = rig.track(poses[2], T.make_tracking_state(dist_trans=1e-4))
    assert lost and rig.tracker.state.lost == 1
    rig.sensor.process(np.full((cp.m_imageHeight, cp.m_imageWidth), MINF, np.float32), np.zeros((cp.m_imageHeight, cp.m_imageWidth, 4), np.uint8))
    got, lost = rig.track(poses[2], ts)
    assert lost and np.all(got == MINF) and rig.tracker.state.numCorr == 0


@pytest.mark.gpu
def test_gpu_icp_steps_match_oracle(vh, oracle_lib):
    """one level, launcher level: correspondences (which pixels pair up: exactly; values: the target's bits) and the
    summed linear system (1e-5 relative to the largest term)"""
    from oracle import icp
    from voxelhashing_amd import engine as E, lib
    O = oracle_lib
    L = lib.load()
    hp, cp, rp = setup_small()
    poses = [synth.orbit_pose(k, n_frames=400) for k in range(3)]
    sc = oracle_model(O, hp, cp, rp, poses[:2], TRACK_SPHERES)
    model = sc.render(poses[1])
    _, cam, nrm = maps_for(O, TRACK_SPHERES, poses[2], cp)
    W, H = cp.m_imageWidth, cp.m_imageHeight
    delta = np.eye(4, dtype=np.float32)
    delta[0, 3] = 0.003
    up = lambda a: lib.DeviceBuffer.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    d_in, d_inn, d_t, d_tn = up(cam), up(nrm), up(model["depth4"]), up(model["normals"])
    d_c, d_cn = lib.DeviceBuffer(W * H * 16), lib.DeviceBuffer(W * H * 16)
    d_state, d_delta = lib.DeviceBuffer(C.sizeof(T.IcpState)), up(delta)
    nP = L.vh_icp_num_partials(W, H)
    d_part = lib.DeviceBuffer(nP * 30 * 4)
    lib.check(L.vh_icp_begin(d_state.ptr, d_delta.ptr, None))
    lib.check(L.vh_icp_projective_correspondences(d_in.ptr, d_inn.ptr, d_t.ptr, d_tn.ptr, d_c.ptr, d_cn.ptr, W, H, 0.15, 0.97, 1.0, d_state.ptr, C.byref(cp), None))
    lib.check(L.vh_icp_build_linear_system(W, H, d_part.ptr, d_in.ptr, d_c.ptr, d_cn.ptr, d_state.ptr, None))
    corr = d_c.download(np.float32, W * H * 4).reshape(H, W, 4)
    corr_n = d_cn.download(np.float32, W * H * 4).reshape(H, W, 4)
    wc, wn = icp.correspondences(cam, nrm, model["depth4"], model["normals"], delta.reshape(16), 0.15, 0.97, 1.0, cp)
    assert np.array_equal(corr[..., 0] != MINF, wc[..., 0] != MINF) and (wc[..., 0] != MINF).sum() > 3000
    assert np.array_equal(corr.view(np.uint32), wc.view(np.uint32))
    ok = wn[..., 0] != MINF
    assert np.array_equal(corr_n[..., :3].view(np.uint32), wn[..., :3].view(np.uint32)) and np.allclose(corr_n[..., 3][ok], wn[..., 3][ok], rtol=1e-5, atol=1e-7)
    part = d_part.download(np.float32, nP * 30).reshape(nP, 30).astype(np.float64).sum(axis=0)
    ata, atb, err, wsum, ncorr = icp.build_system(cam, corr, corr_n, delta.reshape(16))
    want = np.concatenate([ata[np.triu_indices(6)], atb, [err, wsum, ncorr]])
    assert np.abs(part - want).max() <= 1e-5 * np.abs(want).max(), np.abs(part - want).max() / np.abs(want).max()


@pytest.mark.gpu
def test_gpu_closed_loop_tracking_and_fusion(vh, oracle_lib):
    """the reference's frame loop with tracking on: every pose after the first comes from applyCT"""
    from voxelhashing_amd import engine as E
    O = oracle_lib
    hp, cp, rp = setup_small()
    truth = [synth.orbit_pose(k, n_frames=400) for k in range(12)]
    rig = GpuRig(E, hp, cp, rp)
    ts = T.make_tracking_state()
    pose = truth[0]
    rig.feed(O, TRACK_SPHERES, truth[0])
    rig.integrate(pose)
    for k in range(1, len(truth)):
        rig.feed(O, TRACK_SPHERES, truth[k])
        pose, lost = rig.track(pose, ts)
        assert not lost, f"frame {k}"
        rig.integrate(pose)
    dt, da = pose_error(pose, truth[-1])
    path = sum(pose_error(truth[k - 1], truth[k])[0] for k in range(1, len(truth)))
    assert dt < 0.01 and da < 0.3 and dt < 0.03 * path, (dt, da, path)  # < 1 cm / 0.3 degrees after 43 cm of motion
    assert rig.scene.getNumOccupiedBlocks() > 100
