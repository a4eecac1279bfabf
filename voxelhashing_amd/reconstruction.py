"""The reference's frame loop for a recorded sequence, over the C ABI: `reconstruction()`
(DepthSensingCUDA/Source/DepthSensing.cpp:720-924) with the SensorDataReader as the sensor, plus the two things
the application does around it -- recording what was processed (RGBDSensor::recordFrame / recordTrajectory /
saveRecordedFramesToFile, RGBDSensor.cpp:275-389) and StopScanningAndExtractIsoSurfaceMC (DepthSensing.cpp:284-373).

Everything is configured the way the application is: a GlobalAppState parameter file (zParameters*.txt) and a
tracking parameter file.  The D3D window, the GUI and the live sensors are out of scope."""
import ctypes as C
import os

import numpy as np

from . import engine as E
from . import sensor_data as SD
from . import vhtypes as T
from .lib import check, load

MINF = np.float32(-np.inf)


def read_app_state(path_or_text):
    """zParameters*.txt (a path, or the text itself as bytes) -> AppState"""
    L = load()
    g = T.AppState()
    if isinstance(path_or_text, bytes):
        check(L.vh_app_state_parse(path_or_text, C.byref(g)), "vh_app_state_parse")
    else:
        check(L.vh_app_state_read(str(path_or_text).encode(), C.byref(g)), "vh_app_state_read")
    return g


def read_tracking_state(path_or_text):
    L = load()
    t = T.TrackingState()
    if isinstance(path_or_text, bytes):
        check(L.vh_tracking_state_parse(path_or_text, C.byref(t)), "vh_tracking_state_parse")
    else:
        check(L.vh_tracking_state_read(str(path_or_text).encode(), C.byref(t)), "vh_tracking_state_read")
    return t


class Reconstruction:
    """One scene fed from `.sens` files.  `frame()` is one pass of the reference's render callback with
    reconstruction enabled: read a frame, pre-process it, ray-cast the model at the last pose, find the new pose
    (recorded trajectory or projective ICP), stream, integrate."""

    def __init__(self, app_state, tracking_state=None, sens_files=None, stream=None):
        self.L = load()
        self.gas = app_state
        self.tracking = tracking_state if tracking_state is not None else T.make_tracking_state()
        if sens_files is None:
            sens_files = [bytes(app_state.s_binaryDumpSensorFile[i].value).decode() for i in range(app_state.s_numBinaryDumpSensorFiles)]
        if not sens_files:
            raise ValueError("need to specify s_binaryDumpSensorFile[0]")  # SensorDataReader.cpp:43
        self.sens_files = list(sens_files)
        self.file_idx = 0
        self.reader = SD.SensorDataReader(self.sens_files[0])
        h = self.reader.header
        g = app_state
        self.adapter_size = (g.s_adapterWidth, g.s_adapterHeight)
        di = np.array(h.m_depthIntrinsic[:], dtype=np.float32).reshape(4, 4)
        # RGBDSensor::init + initializeDepthIntrinsics (SensorDataReader.cpp:56-58); the adapter rescales them
        self.sensor = E.CUDARGBDSensor((h.m_depthWidth, h.m_depthHeight), (max(h.m_colorWidth, 1), max(h.m_colorHeight, 1)), self.adapter_size,
                                       float(di[0, 0]), float(di[1, 1]), float(di[0, 2]), float(di[1, 2]), g.s_sensorDepthMin, g.s_sensorDepthMax, stream=stream)
        if g.s_depthFilter:
            self.sensor.setFiterDepthValues(True, g.s_depthSigmaD, g.s_depthSigmaR)
        if g.s_colorFilter:
            self.sensor.setFiterIntensityValues(True, g.s_colorSigmaD, g.s_colorSigmaR)
        self.cp = self.sensor.getDepthCameraParams()
        hp, opt, rp, mp = T.HashParams(), T.SceneOptions(), T.RayCastParams(), T.MarchingCubesParams()
        self.L.vh_hash_params_from_app_state(C.byref(g), C.byref(hp))
        self.L.vh_scene_options_from_app_state(C.byref(g), C.byref(opt))
        intr = np.eye(4, dtype=np.float32)
        intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2] = self.cp.fx, self.cp.fy, self.cp.mx, self.cp.my
        inv = np.linalg.inv(intr.astype(np.float64)).astype(np.float32)
        fp = lambda a: np.ascontiguousarray(a, dtype=np.float32).reshape(16).ctypes.data_as(C.POINTER(C.c_float))
        self.L.vh_raycast_params_from_app_state(C.byref(g), fp(intr), fp(inv), C.byref(rp))
        self.L.vh_marching_cubes_params_from_app_state(C.byref(g), C.byref(mp))
        self.hp, self.rp, self.mp = hp, rp, mp
        self.scene = E.CUDASceneRepHashSDF(hp, opt, stream=stream)
        self.ray = E.CUDARayCastSDF(rp, stream=stream)
        self.chunk_grid = None
        if g.s_streamingEnabled:
            self.chunk_grid = E.CUDASceneRepChunkGrid(self.scene, tuple(g.s_streamingVoxelExtents), tuple(g.s_streamingGridDimensions),
                                                      tuple(g.s_streamingMinGridPos), g.s_streamingInitialChunkListSize, False, g.s_streamingOutParts)
        self.tracker = E.CUDACameraTrackingMultiRes(self.adapter_size[0], self.adapter_size[1], self.tracking.s_maxLevels, stream=stream)
        self.marching_cubes = None
        cam = self.sensor.getDepthCameraData()
        self.frame_data = E.DepthFrame(self.cp, depth_ptr=cam.d_depthData, color_ptr=cam.d_colorData)
        self.frame_number = 0  # g_RGBDAdapter.getFrameNumber()
        self.trajectory = []   # the pose every processed frame was integrated at (recordTrajectory)
        self.recorded = None
        self.lost_frames = 0

    # -- the sensor side ------------------------------------------------------------------------------------------
    def _next_frame(self):
        """processDepth + loadNextSensFile (SensorDataReader.cpp:79-165)"""
        got = self.reader.processDepth()
        while got is None and self.file_idx + 1 < len(self.sens_files):
            self.file_idx += 1
            self.reader.close()
            self.reader = SD.SensorDataReader(self.sens_files[self.file_idx])
            got = self.reader.processDepth()
        return got

    def _record(self, depth, color):
        """recordFrame, RGBDSensor.cpp:275-323.  The reference stores JPEG colour; no encoder is built in, so the
        colour goes in raw (a reader of either side opens both)."""
        h = self.reader.header
        if self.recorded is None:
            self.recorded = SD.SensorData.create((h.m_depthWidth, h.m_depthHeight), (max(h.m_colorWidth, 1), max(h.m_colorHeight, 1)),
                                                 np.array(h.m_depthIntrinsic[:]), np.array(h.m_colorIntrinsic[:]), 1000.0,
                                                 bytes(h.m_sensorName).decode(), SD.TYPE_RAW, SD.TYPE_ZLIB_USHORT,
                                                 np.array(h.m_depthExtrinsic[:]), np.array(h.m_colorExtrinsic[:]))
        d = np.where(np.isfinite(depth), depth, 0.0).astype(np.float64)
        self.recorded.addFrame(color[..., :3], np.floor(1000.0 * d + 0.5).clip(0, 65535).astype(np.uint16))

    # -- reconstruction(), DepthSensing.cpp:720-924 ---------------------------------------------------------------
    def frame(self):
        """-> the camera-to-world pose the frame was integrated at, or None when the input is exhausted"""
        g = self.gas
        got = self._next_frame()
        if got is None:
            return None
        depth, color = got
        self.sensor.process(depth, color)
        self.frame_number += 1
        if g.s_recordData:
            self._record(depth, color)
        use_trajectory = bool(g.s_binaryDumpSensorUseTrajectory)
        only_init = bool(g.s_binaryDumpSensorUseTrajectoryOnlyInit)
        transformation = np.eye(4, dtype=np.float32)
        if use_trajectory:
            transformation = self.reader.getRigidTransform().reshape(4, 4)
            if transformation[0, 0] == MINF or np.isnan(transformation[0, 0]):
                return self._done(None)  # "INVALID FRAME"
        if self.frame_number > 1:
            render_transform = self.scene.getLastRigidTransform().reshape(4, 4)
            if use_trajectory and only_init:
                delta = np.linalg.inv(self.reader.getRigidTransform(-1).reshape(4, 4).astype(np.float64)).astype(np.float32) @ transformation
                render_transform = render_transform @ delta
                self.scene.setLastRigidTransformAndCompactify(render_transform, self.cp)
            self.ray.render(self.scene.getHashData(), self.scene.getHashParams(), self.cp, render_transform)
            if not g.s_trackingEnabled:
                transformation = np.eye(4, dtype=np.float32)
            elif use_trajectory and not only_init:
                pass  # the recorded pose is the pose
            else:
                a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
                check(self.L.vh_rgbd_sensor_get_maps(self.sensor.handle, C.byref(a), C.byref(b), C.byref(c)), "maps")
                rd = self.ray.getRayCastData()
                transformation, lost = self.tracker.applyCT(a, b, rd.d_depth4, rd.d_normals, self.scene.getLastRigidTransform(), self.tracking, None, self.cp)
                if lost:
                    self.lost_frames += 1
                    return self._done(None)  # "!!! TRACKING LOST !!!": the frame is not integrated
        if self.chunk_grid is not None:
            p = (transformation.reshape(4, 4) @ np.array(list(g.s_streamingPos) + [1.0], dtype=np.float32))[:3]
            if g.s_offlineProcessing:
                for _ in range(g.s_streamingOutParts):
                    self.chunk_grid.streamOutToCPUPass0GPU(p, g.s_streamingRadius, True, False)
                    self.chunk_grid.streamOutToCPUPass1CPU(False)
                self.chunk_grid.streamInToGPUAll(p, g.s_streamingRadius, True)
            else:
                self.chunk_grid.streamOutToCPU(p, g.s_streamingRadius, True)
                self.chunk_grid.streamInToGPU(p, g.s_streamingRadius, True)
        mask = self.chunk_grid.getBitMaskGPU() if self.chunk_grid is not None else None
        if g.s_integrationEnabled:
            self.scene.integrate(transformation, self.frame_data, self.cp, mask)
        else:
            self.scene.setLastRigidTransformAndCompactify(transformation, self.cp)
        return self._done(np.ascontiguousarray(transformation, dtype=np.float32).reshape(4, 4))

    def _done(self, pose):
        if self.gas.s_recordData:
            self.trajectory.append(pose if pose is not None else np.full((4, 4), MINF, dtype=np.float32))
        elif pose is not None:
            self.trajectory.append(pose)
        return pose if pose is not None else np.full((4, 4), MINF, dtype=np.float32)

    def run(self, max_frames=None):
        """-> number of frames read"""
        n = 0
        while max_frames is None or n < max_frames:
            if self.frame() is None:
                break
            n += 1
        return n

    # -- around the loop ------------------------------------------------------------------------------------------
    def saveRecordedFramesToFile(self, filename=None):
        """RGBDSensor.cpp:339-389: poses from the run, time stamps and IMU records from the input"""
        if self.recorded is None:
            return None
        filename = filename or bytes(self.gas.s_recordDataFile).decode()
        d = os.path.dirname(filename)
        if d:
            os.makedirs(d, exist_ok=True)
        n = self.recorded.info().m_numFrames
        if n != len(self.trajectory):
            raise RuntimeError("num frames and trajectory size doesn't match")
        src = SD.SensorData.loadFromFile(self.sens_files[0]) if len(self.sens_files) == 1 else None
        out = SD.SensorData.create(*self._recorded_header())
        for i in range(n):
            f = self.recorded.frame(i)
            ts = src.frame(i, depth=False, color=False)["timeStamps"] if src is not None and i < src.info().m_numFrames else (0, 0)
            out.addFrame(f["color"], f["depth"], self.trajectory[i], ts[0], ts[1])
        out.saveToFile(filename)
        return filename

    def _recorded_header(self):
        h = self.recorded.info()
        return ((h.m_depthWidth, h.m_depthHeight), (h.m_colorWidth, h.m_colorHeight), np.array(h.m_depthIntrinsic[:]), np.array(h.m_colorIntrinsic[:]),
                h.m_depthShift, bytes(h.m_sensorName).decode(), h.m_colorCompressionType, h.m_depthCompressionType,
                np.array(h.m_depthExtrinsic[:]), np.array(h.m_colorExtrinsic[:]))

    def extractIsoSurface(self, filename=None):
        """StopScanningAndExtractIsoSurfaceMC: marching cubes over the whole scene (through the chunk grid when
        streaming is on) -> (vertices, colours, faces); written as a PLY when a file name is given"""
        if self.marching_cubes is None:
            self.marching_cubes = E.CUDAMarchingCubesHashSDF(self.mp)
            # offline: every batch is merged and de-duplicated as it arrives; otherwise the buffer holds the triangle
            # soup (three vertices per triangle, no indices) until saveMesh merges it (.cpp:31-86)
            self.marching_cubes.setOfflineProcessing(bool(self.gas.s_offlineProcessing))
        mc = self.marching_cubes
        mc.clearMeshBuffer()
        if self.chunk_grid is not None:
            pos = (self.scene.getLastRigidTransform().reshape(4, 4) @ np.array(list(self.gas.s_streamingPos) + [1.0], dtype=np.float32))[:3]
            mc.extractIsoSurfaceChunkGrid(self.chunk_grid, pos, self.gas.s_streamingRadius)
        else:
            mc.extractIsoSurface(self.scene.getHashData(), self.scene.getHashParams())
        mesh = mc.mesh()
        if filename:
            mc.saveMesh(filename, None, True)  # merges close vertices, writes the PLY and clears the buffer (.cpp:126-144)
        return mesh
