/*
 * vh_types.h -- plain-old-data types of the voxel-hashing TSDF hot path.
 *
 * C-compatible (included by the HIP library, by the C oracle and by FFI users).
 * Every struct names the reference type it replaces.  Field names and field
 * order follow the reference so that a maintainer can memcpy between the two.
 *
 *   DSC/ = /root/reference/DepthSensingCUDA/Source/
 */
#ifndef VH_TYPES_H
#define VH_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Compile-time constants of the reference (DSC/VoxelUtilHashSDF.h:39-54). */
#define VH_SDF_BLOCK_SIZE 8
#define VH_SDF_BLOCK_VOXELS 512 /* 8*8*8 */
#define VH_HASH_BUCKET_SIZE 10
#define VH_LOCK_ENTRY (-1)
#define VH_FREE_ENTRY (-2)
#define VH_NO_OFFSET 0

/* HashEntry, DSC/VoxelUtilHashSDF.h:56-74.  20 B payload; 32 B stride is what
 * __align__(16) yields on the reference toolchain and gives 16 B-aligned
 * {pos,ptr} quads for single dwordx4 probes on gfx950. */
typedef struct VhHashEntry {
    int32_t pos[3];   /* SDF-block coordinates (x,y,z) */
    int32_t ptr;      /* first voxel index (= block id * 512), or FREE/LOCK */
    uint32_t offset;  /* collision-list hop, relative to the home bucket's last slot */
    uint32_t _pad[3];
} VhHashEntry;

/* Voxel, DSC/VoxelUtilHashSDF.h:76-88.  8 B, moved as one 64-bit word. */
typedef struct VhVoxel {
    float sdf;
    uint8_t color[3];
    uint8_t weight;
} VhVoxel;

/* HashParams, DSC/CUDAHashParams.h:8-38 (224 B). Matrices are row-major
 * (DSC/cuda_SimpleMatrixUtil.h:1127-1137: entries[16] = m11,m12,...,m44). */
typedef struct VhHashParams {
    float m_rigidTransform[16];
    float m_rigidTransformInverse[16];

    uint32_t m_hashNumBuckets;
    uint32_t m_hashBucketSize;
    uint32_t m_hashMaxCollisionLinkedListSize;
    uint32_t m_numSDFBlocks;

    int32_t m_SDFBlockSize;
    float m_virtualVoxelSize;
    uint32_t m_numOccupiedBlocks; /* occupied blocks in the viewing frustum */

    float m_maxIntegrationDistance;
    float m_truncScale;
    float m_truncation;
    uint32_t m_integrationWeightSample;
    uint32_t m_integrationWeightMax;

    float m_streamingVoxelExtents[3];
    int32_t m_streamingGridDimensions[3];
    int32_t m_streamingMinGridPos[3];
    uint32_t m_streamingInitialChunkListSize;
    uint32_t m_dummy[2];
} VhHashParams;

/* DepthCameraParams, DSC/CUDADepthCameraParams.h:8-19 (32 B). */
typedef struct VhDepthCameraParams {
    float fx, fy, mx, my;
    uint32_t m_imageWidth;
    uint32_t m_imageHeight;
    float m_sensorDepthWorldMin;
    float m_sensorDepthWorldMax;
} VhDepthCameraParams;

/* RayCastParams, DSC/CUDARayCastParams.h:7-28 (304 B). */
typedef struct VhRayCastParams {
    float m_viewMatrix[16];
    float m_viewMatrixInverse[16];
    float m_intrinsics[16];
    float m_intrinsicsInverse[16];

    uint32_t m_width;
    uint32_t m_height;

    uint32_t m_numOccupiedSDFBlocks;
    uint32_t m_maxNumVertices;
    int32_t m_splatMinimum;

    float m_minDepth;
    float m_maxDepth;
    float m_rayIncrement;
    float m_thresSampleDist;
    float m_thresDist;
    uint8_t m_useGradients; /* bool in the reference */
    uint8_t _pad0[3];

    uint32_t dummy0;
} VhRayCastParams;

/* HashData, DSC/VoxelUtilHashSDF.h:813-823: the nine device buffers, in the
 * reference's order.  Two extension buffers follow (not in the reference):
 * a per-bucket occupancy count and a 1-bit-per-bucket summary that let the
 * compaction and the ray caster skip empty buckets without touching d_hash.
 * They are owned by vh_hash_data_alloc()/vh_hash_data_free() like the rest. */
typedef struct VhHashData {
    uint32_t* d_heap;                  /* [numSDFBlocks] free-list of block ids */
    uint32_t* d_heapCounter;           /* [1] index of the top free element */
    int32_t* d_hashDecision;           /* [numEntries] GC decisions (first No used) */
    int32_t* d_hashDecisionPrefix;     /* [numEntries] kept for layout parity; unused */
    VhHashEntry* d_hash;               /* [numEntries] */
    VhHashEntry* d_hashCompactified;   /* [numEntries] in-frustum occupied entries */
    int32_t* d_hashCompactifiedCounter;/* [1] */
    VhVoxel* d_SDFBlocks;              /* [numSDFBlocks*512] */
    int32_t* d_hashBucketMutex;        /* [numBuckets] */
    uint8_t m_bIsOnGPU;
    uint8_t _pad0[7];
    /* --- extensions --- */
    uint32_t* d_bucketCount;           /* [numBuckets] occupied slots physically in bucket */
    uint32_t* d_bucketBits;            /* [ceil(numBuckets/32)] bit b = (d_bucketCount[b] != 0) */
    uint32_t* d_state;                 /* [VH_STATE_WORDS] device-side status words */
} VhHashData;

/* words of VhHashData::d_state */
enum {
    VH_STATE_HEAP_UNDERFLOW = 0, /* consumeHeap found the heap empty */
    VH_STATE_INSERT_FAILED = 1,  /* stream-in: insertHashEntry returned false */
    VH_STATE_ALLOC_LOCK_LOST = 2,/* alloc requests that lost a bucket lock this pass */
    VH_STATE_RIDER_GAVE_UP = 3,  /* workgroups of a co-launched pass over the voxels that gave up waiting for the launch's other riders (must stay 0) */
    VH_STATE_WORDS = 16
};

/* DepthCameraData, DSC/DepthCameraUtil.h:17-159: the two image buffers the
 * path reads (the cudaArray/texture twins of the reference are not needed:
 * images are read with plain cached loads). */
typedef struct VhDepthCameraData {
    const float* d_depthData; /* [H*W] metres; invalid = -inf (MINF) or 0 */
    const float* d_colorData; /* [H*W*4] float4 rgba in [0,1]; invalid = MINF in .x; may be NULL */
} VhDepthCameraData;

/* RayCastData, DSC/RayCastSDFUtil.h:266-274: the four output maps. */
typedef struct VhRayCastData {
    float* d_depth;   /* [H*W]   */
    float* d_depth4;  /* [H*W*4] camera-space position, w=1 */
    float* d_normals; /* [H*W*4] */
    float* d_colors;  /* [H*W*4] */
} VhRayCastData;

/* SDFBlockDesc, DSC/CUDASceneRepChunkGrid.cu:13-16 / .h:24-51 (16 B):
 * streaming wire/disk unit, together with a 4096 B block of 512 voxels. */
/* ray-interval splatting (vh_ray_interval_splat): one entry of a tile's block list */
#define VH_TILE_LIST_CAPACITY 64        /* small tile tables (default) */
#define VH_TILE_LIST_CAPACITY_LARGE 128 /* large tile tables: fine voxels */
typedef struct VhTileBlock {
    int32_t pos[3]; /* SDF block position */
    int32_t ptr;    /* its voxel pointer (HashEntry::ptr) */
} VhTileBlock;

typedef struct VhSDFBlockDesc {
    int32_t pos[3];
    int32_t ptr;
} VhSDFBlockDesc;

/* MarchingCubesParams, DSC/MarchingCubesSDFUtil.h:9-23 (64 B; the reference's leading `bool` is the low byte of
 * m_boxEnabled, its padding the rest). */
typedef struct VhMarchingCubesParams {
    uint32_t m_boxEnabled;
    float m_minCorner[3];
    uint32_t m_maxNumTriangles;
    float m_maxCorner[3];
    uint32_t m_sdfBlockSize;
    uint32_t m_hashNumBuckets;
    uint32_t m_hashBucketSize;
    float m_threshMarchingCubes;
    float m_threshMarchingCubes2;
    float dummy[3];
} VhMarchingCubesParams;

/* MarchingCubesData::Vertex / ::Triangle, DSC/MarchingCubesSDFUtil.h:33-44 (24 B / 72 B) */
typedef struct VhVertex {
    float p[3];
    float c[3];
} VhVertex;
typedef struct VhTriangle {
    VhVertex v0, v1, v2;
} VhTriangle;

/* MarchingCubesData, DSC/MarchingCubesSDFUtil.h:27-326 (device buffers) */
typedef struct VhMarchingCubesData {
    VhMarchingCubesParams* d_params;
    uint32_t* d_numOccupiedBlocks;
    uint32_t* d_occupiedBlocks; /* hash entry indices, Ne of them */
    uint32_t* d_numTriangles;
    VhTriangle* d_triangles;    /* m_maxNumTriangles */
    uint8_t m_bIsOnGPU;
} VhMarchingCubesData;

/* The five GlobalAppState flags the reference host classes read
 * (DSC/CUDASceneRepHashSDF.h:249,251,329,331; DSC/CUDASceneRepChunkGrid.cpp:13). */
typedef struct VhSceneOptions {
    uint8_t s_offlineProcessing;        /* alloc until fixed point (host loop) */
    uint8_t s_garbageCollectionEnabled;
    uint8_t s_timingsDetailledEnabled;  /* record per-stage HIP events */
    uint8_t s_useReferenceLaunchSequence; /* not in the reference: 1 = run integrate / starve / identify /
                                             mutex reset / free as separate launches with host-side counts
                                             (the reference's sequence) instead of the fused kernel */
    uint32_t s_garbageCollectionStarve; /* starve every n-th frame */
    uint32_t s_streamingOutParts;
} VhSceneOptions;

/* alloc + compactify of one frame as a job (CUDASceneRepHashSDF::integrateAhead prepares it): whoever holds it may
 * launch the two passes -- CUDARayCastSDF::render does so INSIDE its own two launches (vh_render_intervals_co,
 * vh_compute_normals_co), CUDASceneRepHashSDF::integrateFinish launches whatever is still open. */
typedef struct VhFrameJob {
    VhHashData hashData;
    VhHashParams hashParams; /* with the frame's pose */
    VhDepthCameraData cam;
    VhDepthCameraParams cp;
    const uint32_t* d_bitMask;
    void* d_packedFrame;     /* width*height*8 bytes, written by the alloc pass (see vh_alloc_job), or NULL */
    int32_t lockToken;
    uint8_t allocLaunched, compactifyLaunched, pad0[2];
    uint32_t frameNumber; /* frames the scene had integrated when the job was made */
    uint32_t tableEpoch;  /* bumped by everything that edits the table outside integrate(): reset, streaming */
    /* The frame's pass over the voxels (integrate + starve + GC, vh_integrate_fused) as a third rider of computeNormals' launch:
     * its workgroups come last in the grid; they start when the launch's compactify workgroups have counted themselves off
     * (vh_compute_normals_co2).  Prepared by integrateAhead(); integrateFinish()
     * launches the pass itself if nobody did. */
    uint32_t* d_riderDone;   /* VH_RIDER_DONE_WORDS words (the scene's): the flags and counters of the compactify workgroups */
    uint32_t listDoneTotal, listClassTotal;   /* what their top counter / class counters read when every launch enqueued so far has finished (kept by the launcher) */
    uint32_t fusedFlags;     /* VH_FUSED_* of the frame's pass */
    int32_t fusedLockToken;
    uint32_t* d_countMirror; /* the scene's mapped {block count, frame number} words, or NULL */
    uint32_t mirrorTag;
    uint8_t fusedPrepared, fusedLaunched, pad1[2];
} VhFrameJob;
#define VH_RIDER_DONE_COUNTERS 32 /* copies of a flag, 128 bytes apart */
#define VH_RIDER_DONE_WORDS ((2 * VH_RIDER_DONE_COUNTERS + 1) * 32) /* flags, class counters, top counter */

/* The switches reconstruction() reads (DSC/DepthSensing.cpp:720-924) when it runs headless over a recorded sequence at
 * given poses (s_binaryDumpSensorUseTrajectory = true, s_binaryDumpSensorUseTrajectoryOnlyInit = false), plus what is
 * not in the reference: how far the host may run ahead of the device, and where the frames live. */
typedef struct VhReconstructionOptions {
    uint8_t s_streamingEnabled;   /* :881-900: stream out / in around the camera every frame (needs a chunk grid) */
    uint8_t s_integrationEnabled; /* :903-908: 0 = setLastRigidTransformAndCompactify instead of integrate */
    uint8_t s_offlineProcessing;  /* :885-891: streaming moves every part out and everything in range in, each frame */
    uint8_t s_renderEnabled;      /* 0 = skip the ray cast of the previous pose (:763) */
    uint8_t s_allocAhead;         /* not in the reference: alloc of frame k rides in the launch of the ray cast of pose k-1, compactify
                                     and the next pose's interval splat in computeNormals' (CUDASceneRepHashSDF::integrateAhead).
                                     With streaming on: in the frames whose streaming step is known a frame ahead to be a no-op
                                     (vh_stream_out_probe), and in frames where blocks only LEAVE (the pass runs behind the ray cast) */
    uint8_t s_framesOnHost;       /* not in the reference: VhSequenceFrame pointers are HOST memory (pinned for an
                                     asynchronous copy): float depth + RGBX bytes as a sensor delivers them
                                     (RGBDSensor::getDepthFloat / getColorRGBX); uploaded by two copy streams (depth, colour) into
                                     a ring of four staging slots, beside the previous frames' work */
    uint8_t pad0[2];
    uint32_t s_maxFramesInFlight; /* frames the host may enqueue ahead of the device (0 = no bound) */
    float s_streamingPos[3];      /* DSC/DepthSensing.cpp:1340-1355: in camera space */
    float s_streamingRadius;
} VhReconstructionOptions;

/* one frame of a recorded sequence */
typedef struct VhSequenceFrame {
    float rigidTransform[16]; /* camera-to-world pose of the trajectory; [0] = -inf or NaN marks an invalid frame (:738) */
    const float* depth;       /* width*height floats (device pointer, or host pointer with s_framesOnHost) */
    const void* color;        /* device: float4 per pixel (may be NULL); host: RGBX bytes, 4 per pixel */
} VhSequenceFrame;

typedef struct VhReconstructionStats {
    uint64_t frames;             /* frames processed since creation / reset */
    uint64_t invalidFrames;      /* skipped: invalid pose */
    uint64_t blocksStreamedOut, blocksStreamedIn;
    double hostEnqueueSeconds;   /* host time inside run() spent enqueueing work (waits for the device excluded) */
    double hostWaitSeconds;      /* host time inside run() spent waiting: run-ahead bound, streaming read-backs */
    double uploadMs;             /* device time of the timed frame uploads: depth copy, colour copy and conversion (HIP events; the pair
                                    spans both copy streams) */
    uint64_t uploadsTimed;
    uint64_t uploadBytes;        /* bytes per frame upload */
    uint64_t streamingStepsSkipped; /* frames whose streaming step was known ahead to move nothing and that ran like a frame without streaming */
    uint64_t heapUnderflows;     /* the scene's status words as get_stats() found them (VH_STATE_*): alloc requests that found */
    uint64_t failedInserts;      /* the voxel pool empty; stream-in inserts that found no slot (the blocks went back to the host grid) */
    uint64_t framesWithRiders;   /* frames whose alloc pass rode in the ray caster's launch and whose compactify pass rode in computeNormals' */
    uint64_t splatsMadeAheadUsed; /* ray casts that ran on an interval splat made ahead (inside the previous computeNormals launch): such a
                                     frame is three launches -- k_render, k_compute_normals, k_integrate_fused -- or two (framesInTwoLaunches) */
    uint64_t streamingFramesPipelined; /* frames whose streaming step ran without a host wait (CUDASceneRepChunkGrid's pipeline: counts on the
                                          device, the chunk that comes in chosen a frame ahead); streamingStepsSkipped of them moved nothing */
    uint64_t framesInTwoLaunches; /* frames whose pass over the voxels rode in computeNormals' launch too (VhFrameJob::fusedLaunched): k_render and
                                     k_compute_normals are all the frame launches */
} VhReconstructionStats;

/* The GlobalAppState members (DSC/GlobalAppState.h:28-101) that the path reads, as filled from a zParameters*.txt
 * file by vh_app_state_read.  A key the file does not hold is value-initialised (0 / false), as readMembers() does
 * (DSC/GlobalAppState.h:139-142). */
typedef struct VhAppState {
    uint32_t s_adapterWidth, s_adapterHeight;
    float s_sensorDepthMax, s_sensorDepthMin;
    float s_SDFVoxelSize, s_SDFMarchingCubeThreshFactor, s_SDFTruncation, s_SDFTruncationScale, s_SDFMaxIntegrationDistance;
    uint32_t s_SDFIntegrationWeightSample, s_SDFIntegrationWeightMax;
    uint32_t s_hashNumBuckets, s_hashNumSDFBlocks, s_hashMaxCollisionLinkedListSize;
    float s_SDFRayIncrementFactor, s_SDFRayThresSampleDistFactor, s_SDFRayThresDistFactor;
    uint32_t s_SDFUseGradients;
    float s_depthSigmaD, s_depthSigmaR;
    uint32_t s_depthFilter;
    float s_colorSigmaD, s_colorSigmaR;
    uint32_t s_colorFilter;
    uint32_t s_integrationEnabled, s_trackingEnabled, s_timingsDetailledEnabled, s_timingsTotalEnabled;
    uint32_t s_garbageCollectionEnabled, s_garbageCollectionStarve;
    uint32_t s_marchingCubesMaxNumTriangles;
    uint32_t s_streamingEnabled;
    float s_streamingVoxelExtents[3];
    int32_t s_streamingGridDimensions[3];
    int32_t s_streamingMinGridPos[3];
    uint32_t s_streamingInitialChunkListSize;
    float s_streamingRadius;
    float s_streamingPos[3];
    uint32_t s_streamingOutParts;
    uint32_t s_offlineProcessing;
    uint32_t s_sensorIdx;
    /* recorded sequences: played when s_sensorIdx selects the SensorDataReader (DSC/GlobalAppState.h:51-53,80,95-98) */
    uint32_t s_binaryDumpSensorUseTrajectory, s_binaryDumpSensorUseTrajectoryOnlyInit;
    uint32_t s_playData, s_recordData, s_recordCompression, s_reconstructionEnabled;
    uint32_t s_numBinaryDumpSensorFiles; /* entries s_binaryDumpSensorFile[0..n) the file held, at most 8 kept */
    char s_binaryDumpSensorFile[8][256];
    char s_recordDataFile[256];
    uint32_t numKeysFound; /* how many of the members above the file held (the file list counts once) */
} VhAppState;

/* GlobalCameraTrackingState (DSC/GlobalCameraTrackingState.h:14-25), per pyramid level; from zParametersTracking*.txt */
#define VH_TRACKING_MAX_LEVELS 8
typedef struct VhTrackingState {
    uint32_t s_maxLevels;
    uint32_t s_maxOuterIter[VH_TRACKING_MAX_LEVELS];
    uint32_t s_maxInnerIter[VH_TRACKING_MAX_LEVELS];
    float s_distThres[VH_TRACKING_MAX_LEVELS];
    float s_normalThres[VH_TRACKING_MAX_LEVELS];
    float s_angleTransThres[VH_TRACKING_MAX_LEVELS];
    float s_distTransThres[VH_TRACKING_MAX_LEVELS];
    float s_residualEarlyOut[VH_TRACKING_MAX_LEVELS];
    uint32_t numLevelsFound; /* how many levels of s_maxOuterIter the file held */
} VhTrackingState;

/* Header of a recorded sequence (`.sens`, ml::SensorData, DSC/sensorData/sensorData.h:608-830): what
 * SensorDataReader::createFirstConnected hands to RGBDSensor::init and the intrinsics / extrinsics setters. */
typedef struct VhSensorDataInfo {
    uint32_t m_versionNumber;
    int32_t m_colorCompressionType; /* 0 raw, 1 PNG, 2 JPEG */
    int32_t m_depthCompressionType; /* 0 raw u16, 1 zlib u16, 2 uplink (refused) */
    uint32_t m_colorWidth, m_colorHeight, m_depthWidth, m_depthHeight;
    float m_depthShift; /* metres = sample / m_depthShift */
    uint64_t m_numFrames, m_numIMUFrames;
    float m_colorIntrinsic[16], m_colorExtrinsic[16], m_depthIntrinsic[16], m_depthExtrinsic[16];
    char m_sensorName[64];
} VhSensorDataInfo;

/* Device-resident state of one camera-tracking solve (vh_icp_*): the delta transform being refined and what the
 * reference keeps in LinearSystemConfidence (DSC/ICPErrorLog.h:16-58). */
typedef struct VhIcpState {
    float delta[16];        /* row-major; input points are moved by it */
    float lastError;        /* lastICPError of the current level (-1 at its start) */
    uint32_t done;          /* current level left its outer loop early (residual early-out) */
    uint32_t lost;          /* tracking lost: singular system or a step beyond the level's thresholds */
    float sumRegError;
    float sumRegWeight;
    uint32_t numCorr;
    float matrixCondition;
    uint32_t iterations;    /* linear systems solved so far */
    uint32_t pad[8];
} VhIcpState;

/* Error codes of the C ABI: 0 ok; <0 = -(hipError_t); >0 logical. */
enum {
    VH_OK = 0,
    VH_ERR_HEAP_EXHAUSTED = 1,
    VH_ERR_STAGING_OVERFLOW = 2,
    VH_ERR_INSERT_FAILED = 3,
    VH_ERR_BAD_ARGUMENT = 4,
    VH_ERR_VERSION_MISMATCH = 5,
    VH_ERR_IO = 6,
    VH_ERR_TIMEOUT = 7 /* the device made no progress for 30 s (hung or faulted): not a misuse.  Work may still be enqueued against the
                          caller's frame buffers: synchronize() or destroy the loop before freeing them */
};

#ifdef __cplusplus
}
#endif

#endif /* VH_TYPES_H */
