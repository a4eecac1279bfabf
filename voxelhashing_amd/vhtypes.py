"""ctypes mirrors of include/vh_types.h plus the parameter builders of the
reference host classes.

Reference: HashParams  <- CUDASceneRepHashSDF::parametersFromGlobalAppState
           (DepthSensingCUDA/Source/CUDASceneRepHashSDF.h:38-58),
           RayCastParams <- CUDARayCastSDF::parametersFromGlobalAppState
           (DepthSensingCUDA/Source/CUDARayCastSDF.h:24-40),
           DepthCameraParams <- CUDARGBDSensor (CUDARGBDSensor.cpp:133-142).
"""
import ctypes as C

import numpy as np

SDF_BLOCK_SIZE = 8
SDF_BLOCK_VOXELS = 512
HASH_BUCKET_SIZE = 10
LOCK_ENTRY = -1
FREE_ENTRY = -2
STATE_WORDS = 16
STATE_HEAP_UNDERFLOW = 0
STATE_INSERT_FAILED = 1
STATE_ALLOC_LOCK_LOST = 2
STATE_RIDER_GAVE_UP = 3

MINF = np.float32(-np.inf)


class HashEntry(C.Structure):
    _fields_ = [("pos", C.c_int32 * 3), ("ptr", C.c_int32), ("offset", C.c_uint32), ("_pad", C.c_uint32 * 3)]


class Voxel(C.Structure):
    _fields_ = [("sdf", C.c_float), ("color", C.c_uint8 * 3), ("weight", C.c_uint8)]


class HashParams(C.Structure):
    _fields_ = [
        ("m_rigidTransform", C.c_float * 16),
        ("m_rigidTransformInverse", C.c_float * 16),
        ("m_hashNumBuckets", C.c_uint32),
        ("m_hashBucketSize", C.c_uint32),
        ("m_hashMaxCollisionLinkedListSize", C.c_uint32),
        ("m_numSDFBlocks", C.c_uint32),
        ("m_SDFBlockSize", C.c_int32),
        ("m_virtualVoxelSize", C.c_float),
        ("m_numOccupiedBlocks", C.c_uint32),
        ("m_maxIntegrationDistance", C.c_float),
        ("m_truncScale", C.c_float),
        ("m_truncation", C.c_float),
        ("m_integrationWeightSample", C.c_uint32),
        ("m_integrationWeightMax", C.c_uint32),
        ("m_streamingVoxelExtents", C.c_float * 3),
        ("m_streamingGridDimensions", C.c_int32 * 3),
        ("m_streamingMinGridPos", C.c_int32 * 3),
        ("m_streamingInitialChunkListSize", C.c_uint32),
        ("m_dummy", C.c_uint32 * 2),
    ]


class DepthCameraParams(C.Structure):
    _fields_ = [
        ("fx", C.c_float), ("fy", C.c_float), ("mx", C.c_float), ("my", C.c_float),
        ("m_imageWidth", C.c_uint32), ("m_imageHeight", C.c_uint32),
        ("m_sensorDepthWorldMin", C.c_float), ("m_sensorDepthWorldMax", C.c_float),
    ]


class RayCastParams(C.Structure):
    _fields_ = [
        ("m_viewMatrix", C.c_float * 16),
        ("m_viewMatrixInverse", C.c_float * 16),
        ("m_intrinsics", C.c_float * 16),
        ("m_intrinsicsInverse", C.c_float * 16),
        ("m_width", C.c_uint32), ("m_height", C.c_uint32),
        ("m_numOccupiedSDFBlocks", C.c_uint32), ("m_maxNumVertices", C.c_uint32),
        ("m_splatMinimum", C.c_int32),
        ("m_minDepth", C.c_float), ("m_maxDepth", C.c_float), ("m_rayIncrement", C.c_float),
        ("m_thresSampleDist", C.c_float), ("m_thresDist", C.c_float),
        ("m_useGradients", C.c_uint8), ("_pad0", C.c_uint8 * 3),
        ("dummy0", C.c_uint32),
    ]


class HashData(C.Structure):
    _fields_ = [
        ("d_heap", C.c_void_p),
        ("d_heapCounter", C.c_void_p),
        ("d_hashDecision", C.c_void_p),
        ("d_hashDecisionPrefix", C.c_void_p),
        ("d_hash", C.c_void_p),
        ("d_hashCompactified", C.c_void_p),
        ("d_hashCompactifiedCounter", C.c_void_p),
        ("d_SDFBlocks", C.c_void_p),
        ("d_hashBucketMutex", C.c_void_p),
        ("m_bIsOnGPU", C.c_uint8), ("_pad0", C.c_uint8 * 7),
        ("d_bucketCount", C.c_void_p),
        ("d_bucketBits", C.c_void_p),
        ("d_state", C.c_void_p),
    ]


class DepthCameraData(C.Structure):
    _fields_ = [("d_depthData", C.c_void_p), ("d_colorData", C.c_void_p)]


class RayCastData(C.Structure):
    _fields_ = [("d_depth", C.c_void_p), ("d_depth4", C.c_void_p), ("d_normals", C.c_void_p), ("d_colors", C.c_void_p)]


class SDFBlockDesc(C.Structure):
    _fields_ = [("pos", C.c_int32 * 3), ("ptr", C.c_int32)]


class MarchingCubesParams(C.Structure):
    _fields_ = [
        ("m_boxEnabled", C.c_uint32), ("m_minCorner", C.c_float * 3),
        ("m_maxNumTriangles", C.c_uint32), ("m_maxCorner", C.c_float * 3),
        ("m_sdfBlockSize", C.c_uint32), ("m_hashNumBuckets", C.c_uint32), ("m_hashBucketSize", C.c_uint32),
        ("m_threshMarchingCubes", C.c_float), ("m_threshMarchingCubes2", C.c_float), ("dummy", C.c_float * 3),
    ]


class MarchingCubesData(C.Structure):
    _fields_ = [
        ("d_params", C.c_void_p), ("d_numOccupiedBlocks", C.c_void_p), ("d_occupiedBlocks", C.c_void_p),
        ("d_numTriangles", C.c_void_p), ("d_triangles", C.c_void_p), ("m_bIsOnGPU", C.c_uint8),
    ]


# MarchingCubesData::Triangle = 3 x {float3 p, float3 c} (72 B)
TRIANGLE_DTYPE = np.dtype([("v", [("p", np.float32, 3), ("c", np.float32, 3)], 3)])


class AppState(C.Structure):
    _fields_ = [
        ("s_adapterWidth", C.c_uint32), ("s_adapterHeight", C.c_uint32),
        ("s_sensorDepthMax", C.c_float), ("s_sensorDepthMin", C.c_float),
        ("s_SDFVoxelSize", C.c_float), ("s_SDFMarchingCubeThreshFactor", C.c_float), ("s_SDFTruncation", C.c_float),
        ("s_SDFTruncationScale", C.c_float), ("s_SDFMaxIntegrationDistance", C.c_float),
        ("s_SDFIntegrationWeightSample", C.c_uint32), ("s_SDFIntegrationWeightMax", C.c_uint32),
        ("s_hashNumBuckets", C.c_uint32), ("s_hashNumSDFBlocks", C.c_uint32), ("s_hashMaxCollisionLinkedListSize", C.c_uint32),
        ("s_SDFRayIncrementFactor", C.c_float), ("s_SDFRayThresSampleDistFactor", C.c_float), ("s_SDFRayThresDistFactor", C.c_float),
        ("s_SDFUseGradients", C.c_uint32),
        ("s_depthSigmaD", C.c_float), ("s_depthSigmaR", C.c_float), ("s_depthFilter", C.c_uint32),
        ("s_colorSigmaD", C.c_float), ("s_colorSigmaR", C.c_float), ("s_colorFilter", C.c_uint32),
        ("s_integrationEnabled", C.c_uint32), ("s_trackingEnabled", C.c_uint32), ("s_timingsDetailledEnabled", C.c_uint32),
        ("s_timingsTotalEnabled", C.c_uint32), ("s_garbageCollectionEnabled", C.c_uint32), ("s_garbageCollectionStarve", C.c_uint32),
        ("s_marchingCubesMaxNumTriangles", C.c_uint32), ("s_streamingEnabled", C.c_uint32),
        ("s_streamingVoxelExtents", C.c_float * 3), ("s_streamingGridDimensions", C.c_int32 * 3), ("s_streamingMinGridPos", C.c_int32 * 3),
        ("s_streamingInitialChunkListSize", C.c_uint32), ("s_streamingRadius", C.c_float), ("s_streamingPos", C.c_float * 3),
        ("s_streamingOutParts", C.c_uint32), ("s_offlineProcessing", C.c_uint32), ("s_sensorIdx", C.c_uint32),
        ("s_binaryDumpSensorUseTrajectory", C.c_uint32), ("s_binaryDumpSensorUseTrajectoryOnlyInit", C.c_uint32),
        ("s_playData", C.c_uint32), ("s_recordData", C.c_uint32), ("s_recordCompression", C.c_uint32), ("s_reconstructionEnabled", C.c_uint32),
        ("s_numBinaryDumpSensorFiles", C.c_uint32), ("s_binaryDumpSensorFile", (C.c_char * 256) * 8), ("s_recordDataFile", C.c_char * 256),
        ("numKeysFound", C.c_uint32),
    ]


class SensorDataInfo(C.Structure):
    """VhSensorDataInfo: the header of a `.sens` sequence"""
    _fields_ = [
        ("m_versionNumber", C.c_uint32), ("m_colorCompressionType", C.c_int32), ("m_depthCompressionType", C.c_int32),
        ("m_colorWidth", C.c_uint32), ("m_colorHeight", C.c_uint32), ("m_depthWidth", C.c_uint32), ("m_depthHeight", C.c_uint32),
        ("m_depthShift", C.c_float), ("m_numFrames", C.c_uint64), ("m_numIMUFrames", C.c_uint64),
        ("m_colorIntrinsic", C.c_float * 16), ("m_colorExtrinsic", C.c_float * 16),
        ("m_depthIntrinsic", C.c_float * 16), ("m_depthExtrinsic", C.c_float * 16), ("m_sensorName", C.c_char * 64),
    ]


class TrackingState(C.Structure):
    _fields_ = [
        ("s_maxLevels", C.c_uint32), ("s_maxOuterIter", C.c_uint32 * 8), ("s_maxInnerIter", C.c_uint32 * 8),
        ("s_distThres", C.c_float * 8), ("s_normalThres", C.c_float * 8), ("s_angleTransThres", C.c_float * 8),
        ("s_distTransThres", C.c_float * 8), ("s_residualEarlyOut", C.c_float * 8), ("numLevelsFound", C.c_uint32),
    ]


class IcpState(C.Structure):
    _fields_ = [
        ("delta", C.c_float * 16), ("lastError", C.c_float), ("done", C.c_uint32), ("lost", C.c_uint32),
        ("sumRegError", C.c_float), ("sumRegWeight", C.c_float), ("numCorr", C.c_uint32), ("matrixCondition", C.c_float),
        ("iterations", C.c_uint32), ("pad", C.c_uint32 * 8),
    ]


def make_tracking_state(levels=3, outer=(8, 6, 4), inner=(1, 1, 1), dist=0.15, normal=0.97, angle_trans=1.0, dist_trans=1.0, early_out=0.01):
    """the reference's zParametersTrackingDefault.txt"""
    t = TrackingState()
    t.s_maxLevels = levels
    for i in range(levels):
        t.s_maxOuterIter[i] = outer[i]
        t.s_maxInnerIter[i] = inner[i]
        t.s_distThres[i], t.s_normalThres[i] = dist, normal
        t.s_angleTransThres[i], t.s_distTransThres[i], t.s_residualEarlyOut[i] = angle_trans, dist_trans, early_out
    t.numLevelsFound = levels
    return t


class SceneOptions(C.Structure):
    _fields_ = [
        ("s_offlineProcessing", C.c_uint8),
        ("s_garbageCollectionEnabled", C.c_uint8),
        ("s_timingsDetailledEnabled", C.c_uint8),
        ("s_useReferenceLaunchSequence", C.c_uint8),
        ("s_garbageCollectionStarve", C.c_uint32),
        ("s_streamingOutParts", C.c_uint32),
    ]


class FrameJob(C.Structure):
    _fields_ = [
        ("hashData", HashData),
        ("hashParams", HashParams),
        ("cam", DepthCameraData),
        ("cp", DepthCameraParams),
        ("d_bitMask", C.c_void_p),
        ("d_packedFrame", C.c_void_p),
        ("lockToken", C.c_int32),
        ("allocLaunched", C.c_uint8),
        ("compactifyLaunched", C.c_uint8),
        ("pad0", C.c_uint8 * 2),
        ("frameNumber", C.c_uint32),
        ("tableEpoch", C.c_uint32),
        ("d_riderDone", C.c_void_p),
        ("listDoneTotal", C.c_uint32),
        ("listClassTotal", C.c_uint32),
        ("fusedFlags", C.c_uint32),
        ("fusedLockToken", C.c_int32),
        ("d_countMirror", C.c_void_p),
        ("mirrorTag", C.c_uint32),
        ("fusedPrepared", C.c_uint8),
        ("fusedLaunched", C.c_uint8),
        ("pad1", C.c_uint8 * 2),
    ]


class ReconstructionOptions(C.Structure):
    _fields_ = [
        ("s_streamingEnabled", C.c_uint8),
        ("s_integrationEnabled", C.c_uint8),
        ("s_offlineProcessing", C.c_uint8),
        ("s_renderEnabled", C.c_uint8),
        ("s_allocAhead", C.c_uint8),
        ("s_framesOnHost", C.c_uint8),
        ("pad0", C.c_uint8 * 2),
        ("s_maxFramesInFlight", C.c_uint32),
        ("s_streamingPos", C.c_float * 3),
        ("s_streamingRadius", C.c_float),
    ]


class SequenceFrame(C.Structure):
    _fields_ = [("rigidTransform", C.c_float * 16), ("depth", C.c_void_p), ("color", C.c_void_p)]


class ReconstructionStats(C.Structure):
    _fields_ = [
        ("frames", C.c_uint64),
        ("invalidFrames", C.c_uint64),
        ("blocksStreamedOut", C.c_uint64),
        ("blocksStreamedIn", C.c_uint64),
        ("hostEnqueueSeconds", C.c_double),
        ("hostWaitSeconds", C.c_double),
        ("uploadMs", C.c_double),
        ("uploadsTimed", C.c_uint64),
        ("uploadBytes", C.c_uint64),
        ("streamingStepsSkipped", C.c_uint64),
        ("heapUnderflows", C.c_uint64),
        ("failedInserts", C.c_uint64),
        ("framesWithRiders", C.c_uint64),
        ("splatsMadeAheadUsed", C.c_uint64),
        ("streamingFramesPipelined", C.c_uint64),
        ("framesInTwoLaunches", C.c_uint64),
    ]


assert C.sizeof(HashEntry) == 32
assert C.sizeof(Voxel) == 8
assert C.sizeof(HashParams) == 224
assert C.sizeof(DepthCameraParams) == 32
assert C.sizeof(RayCastParams) == 304
assert C.sizeof(SDFBlockDesc) == 16

# numpy views of the same layouts (for downloads)
HASH_ENTRY_DTYPE = np.dtype([("pos", np.int32, 3), ("ptr", np.int32), ("offset", np.uint32), ("_pad", np.uint32, 3)])
VOXEL_DTYPE = np.dtype([("sdf", np.float32), ("color", np.uint8, 3), ("weight", np.uint8)])
DESC_DTYPE = np.dtype([("pos", np.int32, 3), ("ptr", np.int32)])
assert HASH_ENTRY_DTYPE.itemsize == 32 and VOXEL_DTYPE.itemsize == 8 and DESC_DTYPE.itemsize == 16

IDENTITY16 = (1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0)


def mat16(m):
    """row-major 4x4 -> (c_float*16)"""
    a = np.asarray(m, dtype=np.float32).reshape(16)
    return (C.c_float * 16)(*a.tolist())


def make_hash_params(num_buckets, num_sdf_blocks, voxel_size, truncation=None, trunc_scale=None,
                     max_integration_distance=4.0, weight_sample=10, weight_max=255,
                     max_collision_list=7, streaming_extents=(1.0, 1.0, 1.0),
                     streaming_dims=(257, 257, 257), streaming_min=(-128, -128, -128),
                     streaming_list_size=2000):
    """HashParams as parametersFromGlobalAppState builds it.  truncation defaults
    to 5*voxel and trunc_scale to 2.5*voxel (zParametersManolisScan.txt:31-32)."""
    p = HashParams()
    p.m_rigidTransform = (C.c_float * 16)(*IDENTITY16)
    p.m_rigidTransformInverse = (C.c_float * 16)(*IDENTITY16)
    p.m_hashNumBuckets = num_buckets
    p.m_hashBucketSize = HASH_BUCKET_SIZE
    p.m_hashMaxCollisionLinkedListSize = max_collision_list
    p.m_numSDFBlocks = num_sdf_blocks
    p.m_SDFBlockSize = SDF_BLOCK_SIZE
    p.m_virtualVoxelSize = voxel_size
    p.m_numOccupiedBlocks = 0
    p.m_maxIntegrationDistance = max_integration_distance
    p.m_truncation = float(np.float32(5.0) * np.float32(voxel_size)) if truncation is None else truncation
    p.m_truncScale = float(np.float32(2.5) * np.float32(voxel_size)) if trunc_scale is None else trunc_scale
    p.m_integrationWeightSample = weight_sample
    p.m_integrationWeightMax = weight_max
    p.m_streamingVoxelExtents = (C.c_float * 3)(*streaming_extents)
    p.m_streamingGridDimensions = (C.c_int32 * 3)(*streaming_dims)
    p.m_streamingMinGridPos = (C.c_int32 * 3)(*streaming_min)
    p.m_streamingInitialChunkListSize = streaming_list_size
    return p


def make_depth_camera_params(width, height, depth_min=0.5, depth_max=5.0, fx=None, fy=None, mx=None, my=None):
    """Intrinsics of SURVEY.md section 8(d): fx = fy = 525*W/640, principal point
    at the image centre."""
    p = DepthCameraParams()
    p.fx = 525.0 * width / 640.0 if fx is None else fx
    p.fy = 525.0 * width / 640.0 if fy is None else fy
    p.mx = (width - 1) / 2.0 if mx is None else mx
    p.my = (height - 1) / 2.0 if my is None else my
    p.m_imageWidth = width
    p.m_imageHeight = height
    p.m_sensorDepthWorldMin = depth_min
    p.m_sensorDepthWorldMax = depth_max
    return p


def make_raycast_params(hash_params, cam_params, ray_increment_factor=0.8, thres_sample_dist_factor=50.5,
                        thres_dist_factor=50.0, use_gradients=False):
    """RayCastParams as CUDARayCastSDF::parametersFromGlobalAppState builds it
    (float32 arithmetic as in the reference)."""
    p = RayCastParams()
    p.m_viewMatrix = (C.c_float * 16)(*IDENTITY16)
    p.m_viewMatrixInverse = (C.c_float * 16)(*IDENTITY16)
    fx, fy, mx, my = cam_params.fx, cam_params.fy, cam_params.mx, cam_params.my
    p.m_intrinsics = mat16([[fx, 0, mx, 0], [0, fy, my, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    p.m_intrinsicsInverse = mat16([[1 / fx, 0, -mx / fx, 0], [0, 1 / fy, -my / fy, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    p.m_width = cam_params.m_imageWidth
    p.m_height = cam_params.m_imageHeight
    p.m_numOccupiedSDFBlocks = 0
    p.m_maxNumVertices = hash_params.m_numSDFBlocks * 6
    p.m_splatMinimum = 0
    p.m_minDepth = cam_params.m_sensorDepthWorldMin
    p.m_maxDepth = cam_params.m_sensorDepthWorldMax
    inc = np.float32(ray_increment_factor) * np.float32(hash_params.m_truncation)
    p.m_rayIncrement = float(inc)
    p.m_thresSampleDist = float(np.float32(thres_sample_dist_factor) * inc)
    p.m_thresDist = float(np.float32(thres_dist_factor) * inc)
    p.m_useGradients = 1 if use_gradients else 0
    return p


def make_marching_cubes_params(hash_params, max_num_triangles=1 << 20, thresh_factor=10.0):
    """parametersFromGlobalAppState, DSC/CUDAMarchingCubesHashSDF.h:19-28 (s_SDFMarchingCubeThreshFactor = 10 in
    the reference's zParameters files)"""
    p = MarchingCubesParams()
    p.m_maxNumTriangles = max_num_triangles
    p.m_threshMarchingCubes = np.float32(thresh_factor) * np.float32(hash_params.m_virtualVoxelSize)
    p.m_threshMarchingCubes2 = np.float32(thresh_factor) * np.float32(hash_params.m_virtualVoxelSize)
    p.m_sdfBlockSize = 8
    p.m_hashBucketSize = 10
    p.m_hashNumBuckets = hash_params.m_hashNumBuckets
    return p


def make_scene_options(offline=True, gc=True, starve=15, timings=False, streaming_out_parts=80,
                       reference_launch_sequence=False):
    o = SceneOptions()
    o.s_offlineProcessing = 1 if offline else 0
    o.s_garbageCollectionEnabled = 1 if gc else 0
    o.s_timingsDetailledEnabled = 1 if timings else 0
    o.s_useReferenceLaunchSequence = 1 if reference_launch_sequence else 0
    o.s_garbageCollectionStarve = starve
    o.s_streamingOutParts = streaming_out_parts
    return o
