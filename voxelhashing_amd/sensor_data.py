"""Recorded sequences: mirrors of ml::SensorData (the `.sens` container,
DepthSensingCUDA/Source/sensorData/sensorData.h:608-830) and SensorDataReader
(DepthSensingCUDA/Source/SensorDataReader.cpp:39-179) over the C ABI.  Host side only."""
import ctypes as C

import numpy as np

from . import vhtypes as T
from .lib import check, load

TYPE_RAW, TYPE_PNG, TYPE_JPEG = 0, 1, 2                      # COMPRESSION_TYPE_COLOR, sensorData.h:217-221
TYPE_RAW_USHORT, TYPE_ZLIB_USHORT, TYPE_OCCI_USHORT = 0, 1, 2  # COMPRESSION_TYPE_DEPTH, :222-226


def _f16(m):
    a = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def make_intrinsic_matrix(fx, fy, mx, my):
    """CalibrationData::makeIntrinsicMatrix :168-175"""
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 2], m[1, 1], m[1, 2] = fx, mx, fy, my
    return m


class SensorData:
    """ml::SensorData: header + compressed RGB-D frames (+ IMU records)"""

    def __init__(self, handle):
        self.L = load()
        self.handle = handle

    @classmethod
    def create(cls, depth_size, color_size, depth_intrinsic, color_intrinsic=None, depth_shift=1000.0, sensor_name="Unknown",
               color_type=TYPE_RAW, depth_type=TYPE_ZLIB_USHORT, depth_extrinsic=None, color_extrinsic=None):
        L = load()
        h = T.SensorDataInfo()
        h.m_versionNumber = 4
        h.m_colorCompressionType, h.m_depthCompressionType = color_type, depth_type
        h.m_depthWidth, h.m_depthHeight = depth_size
        h.m_colorWidth, h.m_colorHeight = color_size
        h.m_depthShift = depth_shift
        h.m_sensorName = sensor_name.encode()
        eye = np.eye(4, dtype=np.float32)
        for name, m in (("m_depthIntrinsic", depth_intrinsic), ("m_colorIntrinsic", color_intrinsic if color_intrinsic is not None else depth_intrinsic),
                        ("m_depthExtrinsic", depth_extrinsic if depth_extrinsic is not None else eye),
                        ("m_colorExtrinsic", color_extrinsic if color_extrinsic is not None else eye)):
            getattr(h, name)[:] = [float(x) for x in np.asarray(m, dtype=np.float32).reshape(16)]
        out = C.c_void_p()
        check(L.vh_sensor_data_create(C.byref(h), C.byref(out)), "vh_sensor_data_create")
        return cls(out)

    @classmethod
    def loadFromFile(cls, filename):
        out = C.c_void_p()
        check(load().vh_sensor_data_load(str(filename).encode(), C.byref(out)), "loadFromFile")
        return cls(out)

    def close(self):
        if getattr(self, "handle", None):
            self.L.vh_sensor_data_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def saveToFile(self, filename):
        check(self.L.vh_sensor_data_save(self.handle, str(filename).encode()), "saveToFile")

    def info(self):
        out = T.SensorDataInfo()
        check(self.L.vh_sensor_data_info(self.handle, C.byref(out)), "info")
        return out

    def addFrame(self, color_rgb, depth_u16, camera_to_world=None, time_stamp_color=0, time_stamp_depth=0):
        """color_rgb: [H,W,3] u8, already compressed bytes (PNG / JPEG, when the header says so) or None"""
        i = self.info()
        d = None if depth_u16 is None else np.ascontiguousarray(depth_u16, dtype=np.uint16)
        if d is not None and d.size != i.m_depthWidth * i.m_depthHeight:
            raise ValueError("depth frame has the wrong size")
        pose, pp = _f16(camera_to_world if camera_to_world is not None else np.eye(4))
        dp = None if d is None else d.ctypes.data
        if isinstance(color_rgb, (bytes, bytearray)):
            buf = np.frombuffer(bytes(color_rgb), dtype=np.uint8)
            check(self.L.vh_sensor_data_add_frame_compressed(self.handle, buf.ctypes.data, len(buf), dp, pp, time_stamp_color, time_stamp_depth), "addFrame")
            return
        c = None if color_rgb is None else np.ascontiguousarray(color_rgb, dtype=np.uint8)
        if c is not None and c.size != 3 * i.m_colorWidth * i.m_colorHeight:
            raise ValueError("colour frame has the wrong size")
        check(self.L.vh_sensor_data_add_frame(self.handle, None if c is None else c.ctypes.data, dp, pp, time_stamp_color, time_stamp_depth), "addFrame")

    def addIMUFrame(self, values15, time_stamp=0):
        v = np.ascontiguousarray(values15, dtype=np.float64).reshape(15)
        check(self.L.vh_sensor_data_add_imu_frame(self.handle, v.ctypes.data_as(C.POINTER(C.c_double)), time_stamp), "addIMUFrame")

    def frame(self, idx, depth=True, color=True):
        """-> dict(depth [H,W] u16 | None, color [H,W,3] u8 | None, cameraToWorld [16] f32, timeStamps (color, depth))"""
        i = self.info()
        d = np.empty((i.m_depthHeight, i.m_depthWidth), dtype=np.uint16) if depth else None
        c = np.empty((i.m_colorHeight, i.m_colorWidth, 3), dtype=np.uint8) if color else None
        pose = np.empty(16, dtype=np.float32)
        ts = (C.c_uint64 * 2)()
        check(self.L.vh_sensor_data_get_frame(self.handle, idx, None if d is None else d.ctypes.data, None if c is None else c.ctypes.data,
                                              pose.ctypes.data_as(C.POINTER(C.c_float)), ts), "frame")
        return dict(depth=d, color=c, cameraToWorld=pose, timeStamps=(int(ts[0]), int(ts[1])))


class SensorDataReader:
    """SensorDataReader: the sensor the frame loop polls when a recorded sequence is played"""

    def __init__(self, filename):
        self.L = load()
        h = C.c_void_p()
        check(self.L.vh_sensor_data_reader_create(str(filename).encode(), C.byref(h)), "createFirstConnected")
        self.handle = h
        self.header = T.SensorDataInfo()
        check(self.L.vh_sensor_data_reader_info(self.handle, C.byref(self.header)), "info")

    def close(self):
        if getattr(self, "handle", None):
            self.L.vh_sensor_data_reader_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def processDepth(self):
        """-> (depth [H,W] f32 metres, colour [H,W,4] u8) of the next frame, or None when the sequence is complete.
        The arrays are copies."""
        got, d, c = C.c_int(0), C.c_void_p(), C.c_void_p()
        check(self.L.vh_sensor_data_reader_process_depth(self.handle, C.byref(got), C.byref(d), C.byref(c)), "processDepth")
        if not got.value:
            return None
        h = self.header
        cw, ch = max(h.m_colorWidth, 1), max(h.m_colorHeight, 1)
        depth = np.ctypeslib.as_array(C.cast(d, C.POINTER(C.c_float)), shape=(h.m_depthHeight, h.m_depthWidth)).copy()
        color = np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_uint8)), shape=(ch, cw, 4)).copy()
        return depth, color

    def getRigidTransform(self, offset=0):
        out = np.empty(16, dtype=np.float32)
        check(self.L.vh_sensor_data_reader_get_rigid_transform(self.handle, offset, out.ctypes.data_as(C.POINTER(C.c_float))), "getRigidTransform")
        return out

    def getCurrFrame(self):
        a, b = C.c_uint32(), C.c_uint32()
        check(self.L.vh_sensor_data_reader_get_curr_frame(self.handle, C.byref(a), C.byref(b)), "getCurrFrame")
        return a.value

    def getNumFrames(self):
        return int(self.header.m_numFrames)
