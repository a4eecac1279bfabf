"""Python mirror of the reference host classes over the C ABI:

    CUDASceneRepHashSDF   (DepthSensingCUDA/Source/CUDASceneRepHashSDF.h:28)
    CUDARayCastSDF        (DepthSensingCUDA/Source/CUDARayCastSDF.h:13)
    CUDASceneRepChunkGrid (DepthSensingCUDA/Source/CUDASceneRepChunkGrid.h:152)

Same method names and argument meaning as the reference (camelCase kept), so
tests read like the reference's frame loop (DepthSensing.cpp:720-924).  All
compute happens in libvoxelhashing_amd.so on the GPU; this file only marshals.
"""
import ctypes as C

import numpy as np

from . import canonical
from . import vhtypes as T
from .lib import DeviceBuffer, check, download, f16, load


def _copy_struct(s):
    out = type(s)()
    C.memmove(C.byref(out), C.byref(s), C.sizeof(s))
    return out


class DepthFrame:
    """Device-resident depth + colour image pair (DepthCameraData,
    DepthSensingCUDA/Source/DepthCameraUtil.h:17)."""

    def __init__(self, cam_params, depth=None, color=None, depth_ptr=None, color_ptr=None, stream=None):
        self.cp = cam_params
        n = cam_params.m_imageWidth * cam_params.m_imageHeight
        self._own = []
        if depth_ptr is None:
            buf = DeviceBuffer(4 * n)
            self._own.append(buf)
            depth_ptr = buf.ptr
            if depth is not None:
                buf.upload(np.ascontiguousarray(depth, dtype=np.float32), stream)
        if color_ptr is None and color is not False:
            buf = DeviceBuffer(16 * n)
            self._own.append(buf)
            color_ptr = buf.ptr
            if color is not None:
                buf.upload(np.ascontiguousarray(color, dtype=np.float32), stream)
        self.depth_ptr = depth_ptr
        self.color_ptr = color_ptr if color is not False else None
        self.data = T.DepthCameraData(self.depth_ptr, self.color_ptr)

    def download(self):
        H, W = self.cp.m_imageHeight, self.cp.m_imageWidth
        d = download(self.depth_ptr, np.float32, H * W).reshape(H, W)
        c = download(self.color_ptr, np.float32, H * W * 4).reshape(H, W, 4) if self.color_ptr else None
        return d, c


def synth_frame(spheres, inside, cam_to_world, cam_params, out=None, stream=None):
    """generate a synthetic frame on the device (vh_synth_frame)"""
    fr = out if out is not None else DepthFrame(cam_params)
    sp = np.ascontiguousarray(spheres, dtype=np.float64)
    check(load().vh_synth_frame(sp.ctypes.data, sp.shape[0], int(inside), f16(cam_to_world), C.byref(cam_params),
                                fr.depth_ptr, fr.color_ptr, stream), "vh_synth_frame")
    return fr


class CUDASceneRepHashSDF:
    def __init__(self, params, options=None, stream=None):
        self.L = load()
        self.stream = stream
        self._params = _copy_struct(params)
        self._options = _copy_struct(options) if options is not None else T.make_scene_options(offline=False)
        h = C.c_void_p()
        check(self.L.vh_scene_rep_create(C.byref(self._params), C.byref(self._options), stream, C.byref(h)), "vh_scene_rep_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.L.vh_scene_rep_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference API -------------------------------------------------------
    def integrate(self, lastRigidTransform, depthCameraData, depthCameraParams, d_bitMask=None):
        data = depthCameraData.data if isinstance(depthCameraData, DepthFrame) else depthCameraData
        check(self.L.vh_scene_rep_integrate(self.handle, f16(lastRigidTransform), C.byref(data), C.byref(depthCameraParams), d_bitMask),
              "CUDASceneRepHashSDF::integrate")

    def integrateAhead(self, lastRigidTransform, depthCameraData, depthCameraParams, d_bitMask=None):
        """-> the frame's alloc + compactify job (a pointer for CUDARayCastSDF.render(..., coLaunch=job)) or None"""
        data = depthCameraData.data if isinstance(depthCameraData, DepthFrame) else depthCameraData
        job = C.POINTER(T.FrameJob)()
        check(self.L.vh_scene_rep_integrate_ahead(self.handle, f16(lastRigidTransform), C.byref(data), C.byref(depthCameraParams), d_bitMask,
                                                  C.byref(job)), "CUDASceneRepHashSDF::integrateAhead")
        return job if job else None

    def integrateFinish(self, depthCameraData, depthCameraParams):
        data = depthCameraData.data if isinstance(depthCameraData, DepthFrame) else depthCameraData
        check(self.L.vh_scene_rep_integrate_finish(self.handle, C.byref(data), C.byref(depthCameraParams)), "CUDASceneRepHashSDF::integrateFinish")

    def setLastRigidTransformAndCompactify(self, lastRigidTransform, depthCameraParams):
        check(self.L.vh_scene_rep_set_last_rigid_transform_and_compactify(self.handle, f16(lastRigidTransform), C.byref(depthCameraParams)),
              "setLastRigidTransformAndCompactify")

    def reset(self):
        check(self.L.vh_scene_rep_reset(self.handle), "reset")

    def getHashData(self):
        hd = T.HashData()
        check(self.L.vh_scene_rep_get_hash_data(self.handle, C.byref(hd)), "getHashData")
        return hd

    def getHashParams(self):
        hp = T.HashParams()
        check(self.L.vh_scene_rep_get_hash_params(self.handle, C.byref(hp)), "getHashParams")
        return hp

    def getLastRigidTransform(self):
        return np.array(self.getHashParams().m_rigidTransform, dtype=np.float32).reshape(4, 4)

    def getHeapFreeCount(self):
        n = C.c_uint32()
        check(self.L.vh_scene_rep_get_heap_free_count(self.handle, C.byref(n)), "getHeapFreeCount")
        return n.value

    def getNumOccupiedBlocks(self):
        n = C.c_uint32()
        check(self.L.vh_scene_rep_get_num_occupied_blocks(self.handle, C.byref(n)), "getNumOccupiedBlocks")
        return n.value

    def debugHash(self):
        rep = (C.c_uint32 * 4)()
        check(self.L.vh_scene_rep_debug_hash(self.handle, rep), "debugHash")
        return dict(numOccupied=rep[0], numFree=rep[1], duplicates=rep[2], lockEntries=rep[3])

    # ---- additions ---------------------------------------------------------------
    def setOptions(self, options):
        self._options = _copy_struct(options)
        check(self.L.vh_scene_rep_set_options(self.handle, C.byref(self._options)), "setOptions")

    def getState(self):
        out = (C.c_uint32 * T.STATE_WORDS)()
        check(self.L.vh_scene_rep_get_state(self.handle, out), "getState")
        return np.array(out, dtype=np.uint32)

    def getTimings(self):
        out = (C.c_double * 4)()
        check(self.L.vh_scene_rep_get_timings(self.handle, out), "getTimings")
        return dict(alloc_ms=out[0], compactify_ms=out[1], integrate_ms=out[2], frames=int(out[3]))

    def synchronize(self):
        check(self.L.vh_stream_synchronize(self.stream), "synchronize")

    # ---- downloads (test support) ---------------------------------------------------
    def download(self, with_voxels=True):
        """-> dict with the raw tables (hash, heap, counters, optionally voxels)"""
        hp = self.getHashParams()
        hd = self.getHashData()
        ne = hp.m_hashNumBuckets * T.HASH_BUCKET_SIZE
        s = self.stream
        out = dict(
            params=hp,
            hash=download(hd.d_hash, T.HASH_ENTRY_DTYPE, ne, s),
            heap=download(hd.d_heap, np.uint32, hp.m_numSDFBlocks, s),
            heap_counter=int(download(hd.d_heapCounter, np.uint32, 1, s)[0]),
            bucket_count=download(hd.d_bucketCount, np.uint32, hp.m_hashNumBuckets, s),
            bucket_bits=download(hd.d_bucketBits, np.uint32, (hp.m_hashNumBuckets + 31) // 32, s),
            compact_count=int(download(hd.d_hashCompactifiedCounter, np.int32, 1, s)[0]),
        )
        out["compactified"] = download(hd.d_hashCompactified, T.HASH_ENTRY_DTYPE, out["compact_count"], s)
        out["decisions"] = download(hd.d_hashDecision, np.int32, out["compact_count"], s)
        if with_voxels:
            out["sdf_blocks"] = download(hd.d_SDFBlocks, T.VOXEL_DTYPE, hp.m_numSDFBlocks * T.SDF_BLOCK_VOXELS, s)
        return out

    def state(self, with_voxels=True, check_invariants=True):
        """canonical snapshot (voxelhashing_amd.canonical) + invariant checks"""
        d = self.download(with_voxels)
        hp = d["params"]
        if check_invariants:
            canonical.check_invariants(d["hash"], d["heap"], d["heap_counter"], hp, d.get("sdf_blocks"))
            canonical.check_bucket_summary(d["hash"], d["bucket_count"], d["bucket_bits"], hp)
        snap = canonical.snapshot(d["hash"], d.get("sdf_blocks"), d["heap"], d["heap_counter"], hp, with_voxels)
        snap["compactified"] = d["compactified"]
        snap["decisions"] = d["decisions"]
        return snap


class CUDARayCastSDF:
    def __init__(self, params, stream=None):
        self.L = load()
        self.stream = stream
        self._params = _copy_struct(params)
        h = C.c_void_p()
        check(self.L.vh_raycast_create(C.byref(self._params), stream, C.byref(h)), "vh_raycast_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.L.vh_raycast_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, hashData, hashParams, depthCameraParams, lastRigidTransform, coLaunch=None):
        if coLaunch is not None:
            check(self.L.vh_raycast_render_co(self.handle, C.byref(hashData), C.byref(hashParams), C.byref(depthCameraParams),
                                              f16(lastRigidTransform), coLaunch), "CUDARayCastSDF::render")
            return
        check(self.L.vh_raycast_render(self.handle, C.byref(hashData), C.byref(hashParams), C.byref(depthCameraParams),
                                       f16(lastRigidTransform)), "CUDARayCastSDF::render")

    def getRayCastData(self):
        rd = T.RayCastData()
        check(self.L.vh_raycast_get_data(self.handle, C.byref(rd)), "getRayCastData")
        return rd

    def getRayCastParams(self):
        rp = T.RayCastParams()
        check(self.L.vh_raycast_get_params(self.handle, C.byref(rp)), "getRayCastParams")
        return rp

    def setTiming(self, on, march_only=False, stride=1):
        check(self.L.vh_raycast_set_timing_stride(self.handle, (2 if march_only else 1) if on else 0, stride), "setTiming")

    def setIntervalSplatting(self, on):
        check(self.L.vh_raycast_set_interval_splatting(self.handle, 1 if on else 0), "setIntervalSplatting")

    def getTimings(self):
        out = (C.c_double * 4)()
        check(self.L.vh_raycast_get_timings(self.handle, out), "getTimings")
        return dict(raycast_ms=out[0], normals_ms=out[1], frames=int(out[2]), splat_ms=out[3])

    def getEventPairOverheadMs(self):
        out = C.c_double(0.0)
        check(self.L.vh_raycast_get_event_pair_overhead(self.handle, C.byref(out)), "getEventPairOverhead")
        return out.value

    def download(self):
        rd = self.getRayCastData()
        W, H = self._params.m_width, self._params.m_height
        s = self.stream
        return dict(
            depth=download(rd.d_depth, np.float32, H * W, s).reshape(H, W),
            depth4=download(rd.d_depth4, np.float32, H * W * 4, s).reshape(H, W, 4),
            normals=download(rd.d_normals, np.float32, H * W * 4, s).reshape(H, W, 4),
            colors=download(rd.d_colors, np.float32, H * W * 4, s).reshape(H, W, 4),
        )


class CUDASceneRepChunkGrid:
    def __init__(self, sceneRepHashSDF, voxelExtends, gridDimensions, minGridPos, initialChunkListSize,
                 streamingEnabled, streamOutParts):
        self.L = load()
        self.scene = sceneRepHashSDF
        h = C.c_void_p()
        ext = np.asarray(voxelExtends, dtype=np.float32)
        dims = (C.c_int32 * 3)(*[int(v) for v in gridDimensions])
        mn = (C.c_int32 * 3)(*[int(v) for v in minGridPos])
        check(self.L.vh_chunk_grid_create(sceneRepHashSDF.handle, f16(ext), dims, mn, initialChunkListSize,
                                          1 if streamingEnabled else 0, streamOutParts, C.byref(h)), "vh_chunk_grid_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.L.vh_chunk_grid_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def streamOutToCPUPass0GPU(self, posCamera, radius, useParts=True, multiThreaded=True):
        check(self.L.vh_chunk_grid_stream_out_to_cpu_pass0_gpu(self.handle, f16(posCamera), radius, int(useParts), int(multiThreaded)),
              "streamOutToCPUPass0GPU")

    def streamOutToCPUPass1CPU(self, multiThreaded=True):
        check(self.L.vh_chunk_grid_stream_out_to_cpu_pass1_cpu(self.handle, int(multiThreaded)), "streamOutToCPUPass1CPU")

    def streamInToGPUPass0CPU(self, posCamera, radius, useParts=True, multiThreaded=True):
        check(self.L.vh_chunk_grid_stream_in_to_gpu_pass0_cpu(self.handle, f16(posCamera), radius, int(useParts), int(multiThreaded)),
              "streamInToGPUPass0CPU")

    def streamInToGPUPass1GPU(self, multiThreaded=True):
        check(self.L.vh_chunk_grid_stream_in_to_gpu_pass1_gpu(self.handle, int(multiThreaded)), "streamInToGPUPass1GPU")

    def streamOutToCPU(self, posCamera, radius, useParts=True):
        n = C.c_uint32()
        check(self.L.vh_chunk_grid_stream_out_to_cpu(self.handle, f16(posCamera), radius, int(useParts), C.byref(n)), "streamOutToCPU")
        return n.value

    def streamInToGPU(self, posCamera, radius, useParts=True):
        n = C.c_uint32()
        check(self.L.vh_chunk_grid_stream_in_to_gpu(self.handle, f16(posCamera), radius, int(useParts), C.byref(n)), "streamInToGPU")
        return n.value

    def streamOutToCPUAll(self):
        check(self.L.vh_chunk_grid_stream_out_to_cpu_all(self.handle), "streamOutToCPUAll")

    def streamInToGPUAll(self, posCamera, radius, useParts=True):
        n = C.c_uint32()
        check(self.L.vh_chunk_grid_stream_in_to_gpu_all(self.handle, f16(posCamera), radius, int(useParts), C.byref(n)), "streamInToGPUAll")
        return n.value

    def getBitMaskGPU(self):
        p = C.c_void_p()
        check(self.L.vh_chunk_grid_get_bit_mask_gpu(self.handle, C.byref(p)), "getBitMaskGPU")
        return p

    def reset(self):
        check(self.L.vh_chunk_grid_reset(self.handle), "reset")

    def debugCheckForDuplicates(self):
        check(self.L.vh_chunk_grid_debug_check_for_duplicates(self.handle), "debugCheckForDuplicates")

    def getStatistics(self):
        out = (C.c_uint32 * 3)()
        check(self.L.vh_chunk_grid_get_statistics(self.handle, out), "getStatistics")
        return dict(chunks=out[0], blocks=out[1], bits=out[2])

    def getNumFailedInserts(self):
        out = C.c_uint32(0)
        check(self.L.vh_chunk_grid_get_num_failed_inserts(self.handle, C.byref(out)), "getNumFailedInserts")
        return out.value

    def downloadHostBlocks(self):
        n = C.c_uint32()
        check(self.L.vh_chunk_grid_download_host_blocks(self.handle, None, None, 0, C.byref(n)), "downloadHostBlocks")
        descs = np.zeros(n.value, dtype=T.DESC_DTYPE)
        blocks = np.zeros((n.value, T.SDF_BLOCK_VOXELS), dtype=T.VOXEL_DTYPE)
        if n.value:
            check(self.L.vh_chunk_grid_download_host_blocks(self.handle, descs.ctypes.data, blocks.ctypes.data, n.value, C.byref(n)),
                  "downloadHostBlocks")
        return descs, blocks

    def saveToFile(self, filename, camPos, radius):
        check(self.L.vh_chunk_grid_save_to_file(self.handle, filename.encode(), f16(camPos), radius), "saveToFile")

    def loadFromFile(self, filename, camPos, radius):
        check(self.L.vh_chunk_grid_load_from_file(self.handle, filename.encode(), f16(camPos), radius), "loadFromFile")


class Reconstruction:
    """The frame loop reconstruction() (DepthSensingCUDA/Source/DepthSensing.cpp:720-924) for a recorded sequence at
    given poses, native behind the C ABI: run() enqueues any number of frames with one call."""

    def __init__(self, sceneRep, rayCast, chunkGrid, depthCameraParams, options=None):
        self.L = load()
        self.scene, self.ray, self.grid = sceneRep, rayCast, chunkGrid  # kept alive as long as the loop
        self._cp = _copy_struct(depthCameraParams)
        self._options = _copy_struct(options) if options is not None else self.defaultOptions()
        h = C.c_void_p()
        check(self.L.vh_reconstruction_create(sceneRep.handle, rayCast.handle if rayCast is not None else None,
                                              chunkGrid.handle if chunkGrid is not None else None, C.byref(self._cp),
                                              C.byref(self._options), C.byref(h)), "vh_reconstruction_create")
        self.handle = h

    @staticmethod
    def defaultOptions(**overrides):
        o = T.ReconstructionOptions()
        load().vh_reconstruction_default_options(C.byref(o))
        for k, v in overrides.items():
            if k in ("s_streamingPos",):
                o.s_streamingPos[:] = [float(x) for x in v]
            else:
                setattr(o, k, v)
        return o

    @staticmethod
    def makeFrames(poses, depth_ptrs, color_ptrs):
        """-> ctypes array of VhSequenceFrame (device pointers, or host pointers for s_framesOnHost)"""
        n = len(poses)
        arr = (T.SequenceFrame * n)()
        for k in range(n):
            arr[k].rigidTransform[:] = [float(v) for v in np.asarray(poses[k], dtype=np.float32).reshape(-1)]
            arr[k].depth = depth_ptrs[k]
            arr[k].color = color_ptrs[k] if color_ptrs is not None else None
        return arr

    def close(self):
        if getattr(self, "handle", None):
            self.L.vh_reconstruction_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, frames, first=0, count=None, lookahead=False):
        """frames: a VhSequenceFrame array (makeFrames); processes frames[first:first+count].  lookahead: the loop may
        read the pose of frames[first+count], the frame a later call will bring (vh_reconstruction_run_ahead)"""
        n = len(frames) - first if count is None else count
        if n <= 0:
            return
        ptr = C.cast(C.byref(frames, first * C.sizeof(T.SequenceFrame)), C.POINTER(T.SequenceFrame))
        if lookahead and first + n < len(frames):
            nxt = C.cast(C.byref(frames, (first + n) * C.sizeof(T.SequenceFrame)), C.POINTER(T.SequenceFrame))
            check(self.L.vh_reconstruction_run_ahead(self.handle, ptr, n, nxt), "Reconstruction::run")
        else:
            check(self.L.vh_reconstruction_run(self.handle, ptr, n), "Reconstruction::run")

    def synchronize(self):
        check(self.L.vh_reconstruction_synchronize(self.handle), "Reconstruction::synchronize")

    def reset(self):
        check(self.L.vh_reconstruction_reset(self.handle), "Reconstruction::reset")

    def debugFailRender(self, nth_render_from_now):
        check(self.L.vh_reconstruction_debug_fail_render(self.handle, int(nth_render_from_now)), "Reconstruction::debugFailRender")

    def getStats(self):
        st = T.ReconstructionStats()
        check(self.L.vh_reconstruction_get_stats(self.handle, C.byref(st)), "Reconstruction::getStats")
        return {k: getattr(st, k) for k, _ in T.ReconstructionStats._fields_}


class LauncherScene:
    """HashData + the launcher-level C ABI (the twins of the reference's
    extern "C" launchers, DepthSensingCUDA/Source/CUDASceneRepHashSDF.h:15-26 and
    CUDASceneRepChunkGrid.h:142-146), one call per kernel, for tests that pin
    each launcher on its own."""

    def __init__(self, params, stream=None):
        self.L = load()
        self.stream = stream
        self.hp = _copy_struct(params)
        self.hd = T.HashData()
        check(self.L.vh_hash_data_alloc(C.byref(self.hd), C.byref(self.hp)), "vh_hash_data_alloc")
        self.reset()

    def close(self):
        if getattr(self, "hd", None) is not None and self.hd.d_hash:
            self.L.vh_stream_synchronize(self.stream)
            self.L.vh_hash_data_free(C.byref(self.hd))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_transform(self, transform, inverse):
        self.hp.m_rigidTransform = T.mat16(transform)
        self.hp.m_rigidTransformInverse = T.mat16(inverse)

    def reset(self):
        check(self.L.vh_reset(C.byref(self.hd), C.byref(self.hp), self.stream), "vh_reset")

    def reset_mutex(self):
        check(self.L.vh_reset_bucket_mutex(C.byref(self.hd), C.byref(self.hp), self.stream), "vh_reset_bucket_mutex")

    def alloc(self, frame, cp, bitmask_ptr=None, lock_token=T.LOCK_ENTRY):
        check(self.L.vh_alloc(C.byref(self.hd), C.byref(self.hp), C.byref(frame.data), C.byref(cp), bitmask_ptr, lock_token, self.stream), "vh_alloc")

    def compactify(self, cp):
        n = C.c_uint32()
        check(self.L.vh_compactify(C.byref(self.hd), C.byref(self.hp), C.byref(cp), C.byref(n), 0, self.stream), "vh_compactify")
        self.hp.m_numOccupiedBlocks = n.value
        return n.value

    def integrate(self, frame, cp):
        check(self.L.vh_integrate(C.byref(self.hd), C.byref(self.hp), C.byref(frame.data), C.byref(cp), self.stream), "vh_integrate")

    def integrate_fused(self, frame, cp, flags, lock_token, packed_ptr=None):
        check(self.L.vh_integrate_fused(C.byref(self.hd), C.byref(self.hp), C.byref(frame.data), C.byref(cp), flags, lock_token, None, 0,
                                        packed_ptr, self.stream), "vh_integrate_fused")

    def frame_job(self, frame, cp, bitmask_ptr=None, lock_token=T.LOCK_ENTRY, packed_ptr=None):
        """the alloc + compactify passes of a frame as a VhFrameJob (vh_alloc_job / vh_compactify_job / the co-launches)"""
        job = T.FrameJob()
        C.memmove(C.byref(job.hashData), C.byref(self.hd), C.sizeof(self.hd))
        C.memmove(C.byref(job.hashParams), C.byref(self.hp), C.sizeof(self.hp))
        C.memmove(C.byref(job.cam), C.byref(frame.data), C.sizeof(frame.data))
        C.memmove(C.byref(job.cp), C.byref(cp), C.sizeof(cp))
        job.d_bitMask = bitmask_ptr
        job.d_packedFrame = packed_ptr
        job.lockToken = lock_token
        return job

    def alloc_job(self, job):
        check(self.L.vh_alloc_job(C.byref(job), self.stream), "vh_alloc_job")

    def compactify_job(self, job):
        check(self.L.vh_compactify_job(C.byref(job), self.stream), "vh_compactify_job")

    def starve(self):
        check(self.L.vh_starve(C.byref(self.hd), C.byref(self.hp), self.stream), "vh_starve")

    def gc_identify(self, cp):
        check(self.L.vh_gc_identify(C.byref(self.hd), C.byref(self.hp), C.byref(cp), self.stream), "vh_gc_identify")

    def gc_free(self, lock_token=T.LOCK_ENTRY):
        check(self.L.vh_gc_free(C.byref(self.hd), C.byref(self.hp), lock_token, self.stream), "vh_gc_free")

    def hash_ops(self, ops):
        """ops: [n,5] int32 {op, x, y, z, arg} executed serially by one thread -> results[n]"""
        ops = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 5)
        d_ops = DeviceBuffer.from_numpy(ops, self.stream)
        d_res = DeviceBuffer(4 * len(ops))
        check(self.L.vh_debug_hash_ops(C.byref(self.hd), C.byref(self.hp), d_ops.ptr, d_res.ptr, len(ops), self.stream), "vh_debug_hash_ops")
        return d_res.download(np.int32, len(ops), self.stream)

    def stream_out(self, threads_per_part, start, radius, cam_pos, lock_token, capacity=4096):
        """pass 1 + pass 2 -> (descs, blocks)"""
        cnt = DeviceBuffer(4)
        check(self.L.vh_memset(cnt.ptr, 0, 4, self.stream), "memset")
        d_desc = DeviceBuffer(16 * capacity)
        check(self.L.vh_stream_out_pass1(C.byref(self.hd), C.byref(self.hp), threads_per_part, start, C.c_float(radius), f16(cam_pos),
                                         cnt.ptr, d_desc.ptr, capacity, lock_token, self.stream), "vh_stream_out_pass1")
        n = int(cnt.download(np.uint32, 1, self.stream)[0])
        assert n <= capacity
        d_blocks = DeviceBuffer(4096 * max(n, 1))
        check(self.L.vh_stream_out_pass2(C.byref(self.hd), C.byref(self.hp), d_desc.ptr, d_blocks.ptr, n, self.stream), "vh_stream_out_pass2")
        descs = d_desc.download(T.DESC_DTYPE, n, self.stream)
        blocks = d_blocks.download(T.VOXEL_DTYPE, n * T.SDF_BLOCK_VOXELS, self.stream).reshape(n, T.SDF_BLOCK_VOXELS)
        return descs, blocks

    def stream_in(self, descs, blocks, lock_token):
        descs = np.ascontiguousarray(descs, dtype=T.DESC_DTYPE)
        blocks = np.ascontiguousarray(blocks, dtype=T.VOXEL_DTYPE)
        n = len(descs)
        if n == 0:
            return
        d_desc, d_blocks = DeviceBuffer.from_numpy(descs, self.stream), DeviceBuffer.from_numpy(blocks, self.stream)
        prev = int(download(self.hd.d_heapCounter, np.uint32, 1, self.stream)[0])
        check(self.L.vh_stream_in_pass1(C.byref(self.hd), C.byref(self.hp), n, prev, d_desc.ptr, lock_token, self.stream), "vh_stream_in_pass1")
        check(self.L.vh_stream_in_pass2(C.byref(self.hd), C.byref(self.hp), n, prev, d_desc.ptr, d_blocks.ptr, self.stream), "vh_stream_in_pass2")
        new = np.array([prev - n], dtype=np.uint32)
        check(self.L.vh_memcpy_h2d(self.hd.d_heapCounter, new.ctypes.data, 4, self.stream), "heapCounter")

    def download(self, with_voxels=True):
        hp, hd, s = self.hp, self.hd, self.stream
        ne = hp.m_hashNumBuckets * T.HASH_BUCKET_SIZE
        out = dict(
            params=hp,
            hash=download(hd.d_hash, T.HASH_ENTRY_DTYPE, ne, s),
            heap=download(hd.d_heap, np.uint32, hp.m_numSDFBlocks, s),
            heap_counter=int(download(hd.d_heapCounter, np.uint32, 1, s)[0]),
            bucket_count=download(hd.d_bucketCount, np.uint32, hp.m_hashNumBuckets, s),
            bucket_bits=download(hd.d_bucketBits, np.uint32, (hp.m_hashNumBuckets + 31) // 32, s),
            state=download(hd.d_state, np.uint32, T.STATE_WORDS, s),
            compactified=download(hd.d_hashCompactified, T.HASH_ENTRY_DTYPE, hp.m_numOccupiedBlocks, s),
            decisions=download(hd.d_hashDecision, np.int32, hp.m_numOccupiedBlocks, s),
        )
        if with_voxels:
            out["sdf_blocks"] = download(hd.d_SDFBlocks, T.VOXEL_DTYPE, hp.m_numSDFBlocks * T.SDF_BLOCK_VOXELS, s)
        return out

    def state(self, with_voxels=True):
        d = self.download(with_voxels)
        canonical.check_invariants(d["hash"], d["heap"], d["heap_counter"], self.hp, d.get("sdf_blocks"))
        canonical.check_bucket_summary(d["hash"], d["bucket_count"], d["bucket_bits"], self.hp)
        snap = canonical.snapshot(d["hash"], d.get("sdf_blocks"), d["heap"], d["heap_counter"], self.hp, with_voxels)
        snap.update(compactified=d["compactified"], decisions=d["decisions"], raw=d)
        return snap


class CUDAMarchingCubesHashSDF:
    """Mirror of DSC/CUDAMarchingCubesHashSDF.h:8-67 over the C ABI."""

    def __init__(self, params, stream=None):
        self.L = load()
        self._params = params
        self.stream = stream
        h = C.c_void_p()
        check(self.L.vh_marching_cubes_create(C.byref(params), stream, C.byref(h)), "vh_marching_cubes_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.L.vh_marching_cubes_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setOfflineProcessing(self, on):
        check(self.L.vh_marching_cubes_set_offline_processing(self.handle, 1 if on else 0), "setOfflineProcessing")

    def extractIsoSurface(self, hashData, hashParams, minCorner=(0, 0, 0), maxCorner=(0, 0, 0), boxEnabled=False, copy=True):
        check(self.L.vh_marching_cubes_extract_iso_surface(self.handle, C.byref(hashData), C.byref(hashParams), f16(minCorner),
                                                           f16(maxCorner), int(boxEnabled), int(copy)), "extractIsoSurface")

    def extractIsoSurfaceWithoutCopy(self, hashData, hashParams, minCorner=(0, 0, 0), maxCorner=(0, 0, 0), boxEnabled=False):
        self.extractIsoSurface(hashData, hashParams, minCorner, maxCorner, boxEnabled, copy=False)

    def extractIsoSurfaceChunkGrid(self, chunkGrid, camPos, radius):
        check(self.L.vh_marching_cubes_extract_iso_surface_chunk_grid(self.handle, chunkGrid.handle, f16(camPos), radius),
              "extractIsoSurface(chunkGrid)")

    def copyTrianglesToCPU(self):
        check(self.L.vh_marching_cubes_copy_triangles_to_cpu(self.handle), "copyTrianglesToCPU")

    def clearMeshBuffer(self):
        check(self.L.vh_marching_cubes_clear_mesh_buffer(self.handle), "clearMeshBuffer")

    def counts(self):
        out = (C.c_uint32 * 2)()
        check(self.L.vh_marching_cubes_get_counts(self.handle, out), "get_counts")
        return dict(triangles=int(out[0]), occupied_blocks=int(out[1]))

    def triangles(self):
        """device triangle buffer of the last extraction -> numpy array of T.TRIANGLE_DTYPE"""
        n = min(self.counts()["triangles"], self._params.m_maxNumTriangles)
        out = np.zeros(n, dtype=T.TRIANGLE_DTYPE)
        if n:
            check(self.L.vh_marching_cubes_download_triangles(self.handle, out.ctypes.data, n), "download_triangles")
        return out

    def mesh(self):
        sz = (C.c_uint64 * 2)()
        check(self.L.vh_marching_cubes_get_mesh_size(self.handle, sz), "get_mesh_size")
        v = np.zeros((int(sz[0]), 3), dtype=np.float32)
        c = np.zeros((int(sz[0]), 4), dtype=np.float32)
        f = np.zeros(int(sz[1]), dtype=np.uint32)
        check(self.L.vh_marching_cubes_get_mesh(self.handle, v.ctypes.data, c.ctypes.data, f.ctypes.data), "get_mesh")
        return dict(vertices=v, colors=c, faces=f.reshape(-1, 3))

    def saveMesh(self, filename, transform=None, overwriteExistingFile=False):
        t = f16(transform) if transform is not None else None
        check(self.L.vh_marching_cubes_save_mesh(self.handle, filename.encode(), t, int(overwriteExistingFile)), "saveMesh")


# ---- sensor pre-processing (DSC/CameraUtil.cu) over the C ABI: numpy in, numpy out (tests, tools) ----

def image_op(name, src, width, height, *args, out_channels=1, out_size=None, prefill=None):
    """run vh_<name> on one source image; -> float32 array (height, width[, 4]) of the output size"""
    L = load()
    src = np.ascontiguousarray(src)
    d_in = DeviceBuffer.from_numpy(src)
    ow, oh = out_size if out_size else (width, height)
    n_out = ow * oh * out_channels
    d_out = DeviceBuffer(n_out * 4)
    if prefill is not None:
        d_out.upload(np.ascontiguousarray(prefill, dtype=np.float32))
    fn = getattr(L, "vh_" + name)
    if name in ("resample_float_map", "resample_float4_map"):
        check(fn(d_out.ptr, ow, oh, d_in.ptr, width, height, None), name)
    elif name == "convert_depth_float_to_camera_space_float4":
        check(fn(d_out.ptr, d_in.ptr, C.byref(args[0]), width, height, None), name)
    elif name == "erode_depth_map":
        check(fn(d_out.ptr, d_in.ptr, int(args[0]), width, height, float(args[1]), float(args[2]), None), name)
    elif name in ("gauss_filter_float_map", "gauss_filter_float4_map", "bilateral_filter_float_map"):
        check(fn(d_out.ptr, d_in.ptr, float(args[0]), float(args[1]), width, height, None), name)
    elif name == "set_invalid_float_map":
        check(fn(d_out.ptr, width, height, None), name)
    else:
        check(fn(d_out.ptr, d_in.ptr, width, height, None), name)
    out = d_out.download(np.float32, n_out)
    return out.reshape((oh, ow, out_channels)) if out_channels > 1 else out.reshape((oh, ow))


class CUDARGBDSensor:
    """Mirror of CUDARGBDSensor over CUDARGBDAdapter (include/vh.hpp) over the C ABI."""

    def __init__(self, depth_size, color_size, adapter_size, fx, fy, mx, my, depth_min, depth_max, stream=None):
        self.L = load()
        self.size = tuple(adapter_size)
        sizes = (C.c_uint32 * 6)(depth_size[0], depth_size[1], color_size[0], color_size[1], adapter_size[0], adapter_size[1])
        intr = (C.c_float * 6)(fx, fy, mx, my, depth_min, depth_max)
        h = C.c_void_p()
        check(self.L.vh_rgbd_sensor_create(sizes, intr, stream, C.byref(h)), "vh_rgbd_sensor_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.L.vh_rgbd_sensor_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setFiterDepthValues(self, b=True, sigmaD=1.0, sigmaR=1.0):
        check(self.L.vh_rgbd_sensor_set_filter_depth_values(self.handle, int(b), sigmaD, sigmaR), "setFiterDepthValues")

    def setFiterIntensityValues(self, b=True, sigmaD=1.0, sigmaR=1.0):
        check(self.L.vh_rgbd_sensor_set_filter_intensity_values(self.handle, int(b), sigmaD, sigmaR), "setFiterIntensityValues")

    def process(self, depth_float, color_rgbx):
        d = np.ascontiguousarray(depth_float, dtype=np.float32)
        c = np.ascontiguousarray(color_rgbx, dtype=np.uint8)
        check(self.L.vh_rgbd_sensor_process(self.handle, d.ctypes.data, c.ctypes.data), "process")

    def getDepthCameraData(self):
        out = T.DepthCameraData()
        check(self.L.vh_rgbd_sensor_get_depth_camera_data(self.handle, C.byref(out)), "getDepthCameraData")
        return out

    def getDepthCameraParams(self):
        out = T.DepthCameraParams()
        check(self.L.vh_rgbd_sensor_get_depth_camera_params(self.handle, C.byref(out)), "getDepthCameraParams")
        return out

    def download(self):
        W, H = self.size
        cam = self.getDepthCameraData()
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self.L.vh_rgbd_sensor_get_maps(self.handle, C.byref(a), C.byref(b), C.byref(c)), "get_maps")
        return dict(
            depth=download(cam.d_depthData, np.float32, W * H).reshape(H, W),
            color=download(cam.d_colorData, np.float32, W * H * 4).reshape(H, W, 4),
            camera_space=download(a.value, np.float32, W * H * 4).reshape(H, W, 4),
            normals=download(b.value, np.float32, W * H * 4).reshape(H, W, 4),
            intensity=download(c.value, np.float32, W * H).reshape(H, W),
        )


class CUDACameraTrackingMultiRes:
    """Mirror of DSC/CUDACameraTrackingMultiRes.h:17-78 over the C ABI (device pointers in, 4x4 pose out)."""

    def __init__(self, imageWidth, imageHeight, levels, stream=None):
        self.L = load()
        h = C.c_void_p()
        check(self.L.vh_camera_tracking_create(imageWidth, imageHeight, levels, stream, C.byref(h)), "vh_camera_tracking_create")
        self.handle = h
        self.state = T.IcpState()

    def close(self):
        if getattr(self, "handle", None):
            self.L.vh_camera_tracking_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def applyCT(self, d_input, d_inputNormals, d_model, d_modelNormals, lastTransform, settings, deltaTransformEstimate, cameraParams):
        """-> (4x4 float32 pose = lastTransform * delta, lost flag); the final VhIcpState is kept in self.state"""
        out = (C.c_float * 16)()
        lost = C.c_int(0)
        est = f16(deltaTransformEstimate) if deltaTransformEstimate is not None else None
        check(self.L.vh_camera_tracking_apply_ct(self.handle, d_input, d_inputNormals, d_model, d_modelNormals, f16(lastTransform), C.byref(settings),
                                                 est, C.byref(cameraParams), out, C.byref(lost), C.byref(self.state)), "applyCT")
        return np.array(out, dtype=np.float32).reshape(4, 4), bool(lost.value)
