#!/usr/bin/env python3
"""Measures the closed loop with camera tracking on: sensor pre-processing -> raycast at the last
pose -> projective ICP (3 levels, the reference's default settings) -> integrate at the tracked pose, on the S3 scene
at 640x480 / 4 cm voxels (cfg2's sizes).  Wall time per frame including the one read-back of the pose, HIP-event time of
applyCT alone, and the drift against the true trajectory.  One JSON line.

    python tools/bench_tracking.py [--frames 120] [--width 640 --height 480]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--trajectory", action="store_true", help="poses from the true trajectory, no ICP: the host-fed (PCIe-inclusive) rate of the plain loop")
    args = ap.parse_args()
    import torch
    from oracle import oracle as O
    from voxelhashing_amd import engine as E, lib, synth, vhtypes as T
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    L = lib.load()
    W, H = args.width, args.height
    hp = T.make_hash_params(500000, 1 << 18, **synth.PARAM_SETS["P4"])
    cp = T.make_depth_camera_params(W, H)
    rp = T.make_raycast_params(hp, cp)
    spheres, inside, radius = synth.scene("S3")
    truth = [synth.orbit_pose(k, 1000, radius) for k in range(args.frames)]
    frames = []
    for p in truth:  # the "sensor": depth in metres + RGBX bytes on the host
        d, c = O.synth_frame(spheres, inside, p, cp)
        rgbx = np.ascontiguousarray(np.clip(c * 255.0, 0, 255).astype(np.uint8))
        rgbx[..., 3] = 255
        frames.append((d, rgbx))
    scene = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=False, gc=False))
    ray = E.CUDARayCastSDF(rp)
    sensor = E.CUDARGBDSensor((W, H), (W, H), (W, H), cp.fx, cp.fy, cp.mx, cp.my, cp.m_sensorDepthWorldMin, cp.m_sensorDepthWorldMax)
    tracker = E.CUDACameraTrackingMultiRes(W, H, 3)
    ts = T.make_tracking_state()
    a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
    lib.check(L.vh_rgbd_sensor_get_maps(sensor.handle, C.byref(a), C.byref(b), C.byref(c)), "maps")
    cam = sensor.getDepthCameraData()
    frame = E.DepthFrame(cp, depth_ptr=cam.d_depthData, color_ptr=cam.d_colorData)
    pose = truth[0]
    sensor.process(*frames[0])
    scene.integrate(pose, frame, cp, None)
    icp_ms, lost_frames, iters = 0.0, 0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1, args.frames):
        sensor.process(*frames[k])
        ray.render(scene.getHashData(), scene.getHashParams(), cp, pose)
        rd = ray.getRayCastData()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if args.trajectory:
            pose = truth[k]
            scene.integrate(pose, frame, cp, None)
            continue
        e0.record()
        new_pose, lost = tracker.applyCT(a, b, rd.d_depth4, rd.d_normals, pose, ts, None, cp)
        e1.record()
        e1.synchronize()
        icp_ms += e0.elapsed_time(e1)
        iters += tracker.state.iterations
        if lost:
            lost_frames += 1
        else:
            pose = new_pose
        scene.integrate(pose, frame, cp, None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rel = np.linalg.inv(np.asarray(pose, np.float64).reshape(4, 4)) @ np.asarray(truth[-1], np.float64).reshape(4, 4)
    path = sum(np.linalg.norm(np.asarray(truth[k], np.float64).reshape(4, 4)[:3, 3] - np.asarray(truth[k - 1], np.float64).reshape(4, 4)[:3, 3]) for k in range(1, args.frames))
    n = args.frames - 1
    if args.trajectory:
        print(json.dumps(dict(metric="host-fed frames/sec: upload + sensor pre-processing + raycast + integrate, poses given", value=round(n / dt, 1), unit="frames/s",
                              ms_per_frame=round(1e3 * dt / n, 3), upload_bytes_per_frame=int(frames[0][0].nbytes + frames[0][1].nbytes),
                              config=dict(workload=f"S3 orbit, {W}x{H}, P4 voxels, float depth + RGBX bytes from pageable host memory every frame"))))
        return
    print(json.dumps(dict(metric="tracked frames/sec: sensor pre-processing + raycast + ICP + integrate", value=round(n / dt, 1), unit="frames/s",
                          ms_per_frame=round(1e3 * dt / n, 3), icp_ms_per_frame=round(icp_ms / n, 3), icp_systems_per_frame=round(iters / n, 2), lost_frames=lost_frames,
                          drift_m=round(float(np.linalg.norm(rel[:3, 3])), 5),
                          drift_deg=round(float(np.degrees(np.arccos(np.clip(0.5 * (np.trace(rel[:3, :3]) - 1), -1, 1)))), 4), path_m=round(float(path), 3),
                          config=dict(workload=f"S3 orbit, {W}x{H}, P4 voxels, 3 pyramid levels, reference default tracking settings, host-fed frames"))))


if __name__ == "__main__":
    main()
