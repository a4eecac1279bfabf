"""ctypes wrapper of the CPU oracle (libvh_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
PARITY UNPINNED: see oracle/vh_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from voxelhashing_amd import vhtypes as T

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib(omp=False):
    """the checker (serial build); omp=True: the same source built with OpenMP, bench.py's all-core CPU baseline"""
    if omp in _LIBS:
        return _LIBS[omp]
    path = os.path.join(_HERE, "libvh_oracle_omp.so" if omp else "libvh_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.vho_num_threads.restype = C.c_int
    L.vho_set_num_threads.argtypes = [C.c_int]
    P = C.POINTER
    L.vho_hash_data_alloc.argtypes = [P(T.HashData), P(T.HashParams)]
    L.vho_hash_data_alloc.restype = C.c_int
    L.vho_hash_data_free.argtypes = [P(T.HashData)]
    L.vho_mat4_inverse.argtypes = [P(C.c_float), P(C.c_float)]
    L.vho_reset.argtypes = [P(T.HashData), P(T.HashParams)]
    L.vho_reset_bucket_mutex.argtypes = [P(T.HashData), P(T.HashParams)]
    L.vho_alloc.argtypes = [P(T.HashData), P(T.HashParams), P(T.DepthCameraData), P(T.DepthCameraParams), C.c_void_p]
    L.vho_compactify.argtypes = [P(T.HashData), P(T.HashParams), P(T.DepthCameraParams)]
    L.vho_compactify.restype = C.c_uint32
    L.vho_integrate.argtypes = [P(T.HashData), P(T.HashParams), P(T.DepthCameraData), P(T.DepthCameraParams)]
    L.vho_starve.argtypes = [P(T.HashData), P(T.HashParams)]
    L.vho_gc_identify.argtypes = [P(T.HashData), P(T.HashParams), P(T.DepthCameraParams)]
    L.vho_gc_free.argtypes = [P(T.HashData), P(T.HashParams)]
    L.vho_render.argtypes = [P(T.HashData), P(T.HashParams), P(T.RayCastData), P(T.DepthCameraParams), P(T.RayCastParams)]
    L.vho_compute_normals.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    L.vho_stream_out_pass1.argtypes = [P(T.HashData), P(T.HashParams), C.c_uint32, C.c_uint32, C.c_float,
                                       P(C.c_float), C.c_void_p, C.c_uint32]
    L.vho_stream_out_pass1.restype = C.c_uint32
    L.vho_stream_out_pass2.argtypes = [P(T.HashData), P(T.HashParams), C.c_void_p, C.c_void_p, C.c_uint32]
    L.vho_stream_in_pass1.argtypes = [P(T.HashData), P(T.HashParams), C.c_uint32, C.c_uint32, C.c_void_p]
    L.vho_stream_in_pass1.restype = C.c_uint32
    L.vho_stream_in_pass2.argtypes = [P(T.HashData), P(T.HashParams), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    for _n, _a in (('vho_convert_color_raw_to_float4', [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
                   ('vho_resample_float_map', [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]),
                   ('vho_resample_float4_map', [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]),
                   ('vho_convert_color_to_intensity_float', [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
                   ('vho_convert_depth_float_to_camera_space_float4', [C.c_void_p, C.c_void_p, P(T.DepthCameraParams), C.c_uint32, C.c_uint32]),
                   ('vho_gauss_filter_float_map', [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_uint32]),
                   ('vho_gauss_filter_float4_map', [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_uint32]),
                   ('vho_bilateral_filter_float_map', [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_uint32]),
                   ('vho_erode_depth_map', [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_float, C.c_float])):
        getattr(L, _n).argtypes = _a
        getattr(L, _n).restype = None
    L.vho_extract_iso_surface.argtypes = [P(T.HashData), P(T.HashParams), P(T.MarchingCubesParams), C.c_void_p, C.c_uint32]
    L.vho_extract_iso_surface.restype = C.c_uint32
    L.vho_alloc_block.argtypes = [P(T.HashData), P(T.HashParams), P(C.c_int32)]
    L.vho_delete_hash_entry_element.argtypes = [P(T.HashData), P(T.HashParams), P(C.c_int32)]
    L.vho_delete_hash_entry_element.restype = C.c_int
    L.vho_insert_hash_entry.argtypes = [P(T.HashData), P(T.HashParams), P(T.HashEntry)]
    L.vho_insert_hash_entry.restype = C.c_int
    L.vho_get_hash_entry.argtypes = [P(T.HashData), P(T.HashParams), P(C.c_int32)]
    L.vho_get_hash_entry.restype = T.HashEntry
    L.vho_compute_hash_pos.argtypes = [P(T.HashParams), P(C.c_int32)]
    L.vho_compute_hash_pos.restype = C.c_uint32
    L.vho_world_to_virtual_voxel_pos.argtypes = [P(T.HashParams), P(C.c_float), P(C.c_int32)]
    L.vho_virtual_voxel_pos_to_sdf_block.argtypes = [P(C.c_int32), P(C.c_int32)]
    L.vho_is_block_in_frustum.argtypes = [P(T.HashParams), P(T.DepthCameraParams), P(C.c_int32)]
    L.vho_is_block_in_frustum.restype = C.c_int
    L.vho_camera_to_screen_int.argtypes = [P(T.DepthCameraParams), P(C.c_float), P(C.c_int32)]
    L.vho_combine_voxel.argtypes = [P(T.HashParams), T.Voxel, T.Voxel]
    L.vho_combine_voxel.restype = T.Voxel
    L.vho_scene_integrate.argtypes = [P(T.HashData), P(T.HashParams), P(T.SceneOptions), P(C.c_uint32), P(C.c_float),
                                      P(T.DepthCameraData), P(T.DepthCameraParams), C.c_void_p]
    L.vho_raycast_render.argtypes = [P(T.HashData), P(T.HashParams), P(T.RayCastData), P(T.DepthCameraParams),
                                     P(T.RayCastParams), P(C.c_float)]
    L.vho_synth_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_float), P(T.DepthCameraParams), C.c_void_p, C.c_void_p]
    _LIBS[omp] = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _copy_struct(s):
    out = type(s)()
    C.memmove(C.byref(out), C.byref(s), C.sizeof(s))
    return out


def mat4_inverse(m):
    m = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
    out = np.empty(16, dtype=np.float32)
    lib().vho_mat4_inverse(_fp(m), _fp(out))
    return out


def synth_frame(spheres, inside, cam_to_world, cam_params):
    """-> (depth[H,W] f32, color[H,W,4] f32)"""
    sp = np.ascontiguousarray(spheres, dtype=np.float64)
    Tm = np.ascontiguousarray(cam_to_world, dtype=np.float32).reshape(16)
    H, W = cam_params.m_imageHeight, cam_params.m_imageWidth
    depth = np.empty((H, W), dtype=np.float32)
    color = np.empty((H, W, 4), dtype=np.float32)
    lib().vho_synth_frame(sp.ctypes.data, sp.shape[0], int(inside), _fp(Tm), C.byref(cam_params),
                          depth.ctypes.data, color.ctypes.data)
    return depth, color


def compute_normals(depth4):
    H, W, _ = depth4.shape
    d4 = np.ascontiguousarray(depth4, dtype=np.float32)
    out = np.empty_like(d4)
    lib().vho_compute_normals(out.ctypes.data, d4.ctypes.data, W, H)
    return out


class OracleScene:
    """Oracle twin of CUDASceneRepHashSDF + CUDARayCastSDF on host memory."""

    def __init__(self, hash_params, cam_params, ray_params=None, options=None, omp=False):
        self.L = lib(omp)
        self.hp = _copy_struct(hash_params)
        self.cp = _copy_struct(cam_params)
        self.rp = _copy_struct(ray_params) if ray_params is not None else T.make_raycast_params(self.hp, self.cp)
        self.opt = _copy_struct(options) if options is not None else T.make_scene_options()
        self.hd = T.HashData()
        if self.L.vho_hash_data_alloc(C.byref(self.hd), C.byref(self.hp)) != 0:
            raise MemoryError("oracle hash data")
        self.frames = C.c_uint32(0)
        H, W = self.cp.m_imageHeight, self.cp.m_imageWidth
        self.rc_depth = np.empty((H, W), dtype=np.float32)
        self.rc_depth4 = np.empty((H, W, 4), dtype=np.float32)
        self.rc_normals = np.empty((H, W, 4), dtype=np.float32)
        self.rc_colors = np.empty((H, W, 4), dtype=np.float32)
        self.rd = T.RayCastData(self.rc_depth.ctypes.data, self.rc_depth4.ctypes.data,
                                self.rc_normals.ctypes.data, self.rc_colors.ctypes.data)
        self.reset()

    def close(self):
        if self.hd.d_hash:
            self.L.vho_hash_data_free(C.byref(self.hd))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- CUDASceneRepHashSDF ------------------------------------------------
    def reset(self):
        self.frames = C.c_uint32(0)
        self.hp.m_rigidTransform = (C.c_float * 16)(*T.IDENTITY16)
        self.hp.m_rigidTransformInverse = (C.c_float * 16)(*T.IDENTITY16)
        self.hp.m_numOccupiedBlocks = 0
        self.L.vho_reset(C.byref(self.hd), C.byref(self.hp))

    def _cam(self, depth, color):
        self._depth = np.ascontiguousarray(depth, dtype=np.float32)
        self._color = None if color is None else np.ascontiguousarray(color, dtype=np.float32)
        return T.DepthCameraData(self._depth.ctypes.data, None if self._color is None else self._color.ctypes.data)

    def set_transform(self, transform):
        m = np.ascontiguousarray(transform, dtype=np.float32).reshape(16)
        self.hp.m_rigidTransform = (C.c_float * 16)(*m.tolist())
        inv = mat4_inverse(m)
        self.hp.m_rigidTransformInverse = (C.c_float * 16)(*inv.tolist())

    def integrate(self, transform, depth, color, bitmask=None):
        cam = self._cam(depth, color)
        m = np.ascontiguousarray(transform, dtype=np.float32).reshape(16)
        bm = None if bitmask is None else bitmask.ctypes.data
        self.L.vho_scene_integrate(C.byref(self.hd), C.byref(self.hp), C.byref(self.opt), C.byref(self.frames),
                                   _fp(m), C.byref(cam), C.byref(self.cp), bm)

    # individual launchers
    def reset_mutex(self):
        self.L.vho_reset_bucket_mutex(C.byref(self.hd), C.byref(self.hp))

    def alloc(self, depth, color=None, bitmask=None):
        cam = self._cam(depth, color)
        bm = None if bitmask is None else bitmask.ctypes.data
        self.L.vho_alloc(C.byref(self.hd), C.byref(self.hp), C.byref(cam), C.byref(self.cp), bm)

    def compactify(self):
        n = self.L.vho_compactify(C.byref(self.hd), C.byref(self.hp), C.byref(self.cp))
        self.hp.m_numOccupiedBlocks = n
        return n

    def integrate_depth_map(self, depth, color):
        cam = self._cam(depth, color)
        self.L.vho_integrate(C.byref(self.hd), C.byref(self.hp), C.byref(cam), C.byref(self.cp))

    def starve(self):
        self.L.vho_starve(C.byref(self.hd), C.byref(self.hp))

    def gc_identify(self):
        self.L.vho_gc_identify(C.byref(self.hd), C.byref(self.hp), C.byref(self.cp))

    def gc_free(self):
        self.L.vho_gc_free(C.byref(self.hd), C.byref(self.hp))

    def heap_free_count(self):
        return int(self.array("d_heapCounter", np.uint32, 1)[0]) + 1

    # -- CUDARayCastSDF -------------------------------------------------------
    def render(self, last_rigid_transform):
        m = np.ascontiguousarray(last_rigid_transform, dtype=np.float32).reshape(16)
        self.L.vho_raycast_render(C.byref(self.hd), C.byref(self.hp), C.byref(self.rd), C.byref(self.cp),
                                  C.byref(self.rp), _fp(m))
        return dict(depth=self.rc_depth.copy(), depth4=self.rc_depth4.copy(),
                    normals=self.rc_normals.copy(), colors=self.rc_colors.copy())

    # -- single hash operations ---------------------------------------------
    def alloc_block(self, pos):
        p = np.asarray(pos, dtype=np.int32)
        self.L.vho_alloc_block(C.byref(self.hd), C.byref(self.hp), _ip(p))

    def delete_block(self, pos):
        p = np.asarray(pos, dtype=np.int32)
        return self.L.vho_delete_hash_entry_element(C.byref(self.hd), C.byref(self.hp), _ip(p))

    def insert_entry(self, pos, ptr):
        e = T.HashEntry()
        e.pos = (C.c_int32 * 3)(*[int(v) for v in pos])
        e.ptr = int(ptr)
        e.offset = 0
        return self.L.vho_insert_hash_entry(C.byref(self.hd), C.byref(self.hp), C.byref(e))

    def get_entry(self, pos):
        p = np.asarray(pos, dtype=np.int32)
        e = self.L.vho_get_hash_entry(C.byref(self.hd), C.byref(self.hp), _ip(p))
        return (tuple(e.pos), e.ptr, e.offset)

    # -- streaming launchers ---------------------------------------------------
    def stream_out_pass1(self, threads_per_part, start, radius, cam_pos, capacity=100000):
        out = np.zeros(capacity, dtype=T.DESC_DTYPE)
        cpv = np.asarray(cam_pos, dtype=np.float32)
        n = self.L.vho_stream_out_pass1(C.byref(self.hd), C.byref(self.hp), threads_per_part, start,
                                        C.c_float(radius), _fp(cpv), out.ctypes.data, capacity)
        return out[:n].copy()

    def stream_out_pass2(self, descs):
        descs = np.ascontiguousarray(descs, dtype=T.DESC_DTYPE)
        out = np.zeros((len(descs), T.SDF_BLOCK_VOXELS), dtype=T.VOXEL_DTYPE)
        self.L.vho_stream_out_pass2(C.byref(self.hd), C.byref(self.hp), descs.ctypes.data, out.ctypes.data, len(descs))
        return out

    def stream_in(self, descs, blocks):
        descs = np.ascontiguousarray(descs, dtype=T.DESC_DTYPE)
        blocks = np.ascontiguousarray(blocks, dtype=T.VOXEL_DTYPE)
        n = len(descs)
        hc = self.array("d_heapCounter", np.uint32, 1)
        prev = int(hc[0])
        failed = self.L.vho_stream_in_pass1(C.byref(self.hd), C.byref(self.hp), n, prev, descs.ctypes.data)
        self.L.vho_stream_in_pass2(C.byref(self.hd), C.byref(self.hp), n, prev, descs.ctypes.data, blocks.ctypes.data)
        hc[0] = prev - n
        return failed

    # -- state access -----------------------------------------------------------
    def array(self, field, dtype, count):
        ptr = getattr(self.hd, field)
        dt = np.dtype(dtype)
        buf = (C.c_char * (dt.itemsize * count)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt, count=count)

    def num_entries(self):
        return self.hp.m_hashNumBuckets * T.HASH_BUCKET_SIZE

    def hash_table(self):
        return self.array("d_hash", T.HASH_ENTRY_DTYPE, self.num_entries())

    def extract_iso_surface(self, mc_params, max_triangles=None):
        """marching cubes over every allocated block -> triangles (T.TRIANGLE_DTYPE), in the reference's serial order"""
        cap = int(max_triangles if max_triangles is not None else mc_params.m_maxNumTriangles)
        out = np.zeros(cap, dtype=T.TRIANGLE_DTYPE)
        n = self.L.vho_extract_iso_surface(C.byref(self.hd), C.byref(self.hp), C.byref(mc_params), out.ctypes.data, cap)
        return out[:min(n, cap)], int(n)

    def compactified(self):
        return self.array("d_hashCompactified", T.HASH_ENTRY_DTYPE, self.hp.m_numOccupiedBlocks)

    def sdf_blocks(self):
        return self.array("d_SDFBlocks", T.VOXEL_DTYPE, self.hp.m_numSDFBlocks * T.SDF_BLOCK_VOXELS)

    def heap(self):
        return self.array("d_heap", np.uint32, self.hp.m_numSDFBlocks)

    def decisions(self):
        return self.array("d_hashDecision", np.int32, self.hp.m_numOccupiedBlocks)

    def state(self):
        """Canonical snapshot: see voxelhashing_amd.canonical.snapshot"""
        from voxelhashing_amd import canonical
        return canonical.snapshot(self.hash_table(), self.sdf_blocks(), self.heap(),
                                  int(self.array("d_heapCounter", np.uint32, 1)[0]), self.hp)


# ---- sensor pre-processing (DSC/CameraUtil.cu) ----

def image_op(name, src, width, height, *args, out_channels=1, out_size=None, prefill=None):
    """oracle twin of voxelhashing_amd.engine.image_op"""
    L = lib()
    src = np.ascontiguousarray(src)
    ow, oh = out_size if out_size else (width, height)
    out = np.zeros(ow * oh * out_channels, dtype=np.float32)
    if prefill is not None:
        out[:] = np.ascontiguousarray(prefill, dtype=np.float32).ravel()
    fn = getattr(L, "vho_" + name)
    if name in ("resample_float_map", "resample_float4_map"):
        fn(out.ctypes.data, ow, oh, src.ctypes.data, width, height)
    elif name == "convert_depth_float_to_camera_space_float4":
        fn(out.ctypes.data, src.ctypes.data, C.byref(args[0]), width, height)
    elif name == "erode_depth_map":
        fn(out.ctypes.data, src.ctypes.data, int(args[0]), width, height, float(args[1]), float(args[2]))
    elif name in ("gauss_filter_float_map", "gauss_filter_float4_map", "bilateral_filter_float_map"):
        fn(out.ctypes.data, src.ctypes.data, float(args[0]), float(args[1]), width, height)
    else:
        fn(out.ctypes.data, src.ctypes.data, width, height)
    return out.reshape((oh, ow, out_channels)) if out_channels > 1 else out.reshape((oh, ow))


def compute_normals(camera_space4):
    L = lib()
    H, W = camera_space4.shape[:2]
    src = np.ascontiguousarray(camera_space4, dtype=np.float32)
    out = np.zeros_like(src)
    L.vho_compute_normals(out.ctypes.data, src.ctypes.data, W, H)
    return out
