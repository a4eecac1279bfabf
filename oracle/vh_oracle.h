/*
 * vh_oracle.h -- CPU oracle for the voxel-hashing TSDF hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a single-threaded, IEEE-fp32, serial-order
 * restatement in plain C of the reference algorithms (file:line cited at each
 * function in vh_oracle.c).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product library never does.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
 * this path and cannot be built in this image without
 * writing stand-ins for the CUDA toolkit headers it includes (cuda_runtime.h,
 * texture references), which the build rules forbid.  The only reference-run
 * facts available are the block/hit counts recorded in SURVEY.md section 6 and
 * section 8(c); tests/test_oracle_known_answers.py checks the oracle against
 * those.  Everything else is pinned by this restatement alone.
 *
 * All pointers in VhHashData / VhDepthCameraData / VhRayCastData are HOST
 * pointers here.  The extension buffers of VhHashData are ignored.
 */
#ifndef VH_ORACLE_H
#define VH_ORACLE_H

#include "../include/vh_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* memory (HashData::allocate/free, DSC/VoxelUtilHashSDF.h:113-181) */
int vho_hash_data_alloc(VhHashData* hd, const VhHashParams* hp);
void vho_hash_data_free(VhHashData* hd);

/* float4x4::getInverse (DSC/cuda_SimpleMatrixUtil.h:944-1069) */
void vho_set_num_threads(int n);
int vho_num_threads(void); /* 1 unless built with -fopenmp (libvh_oracle_omp.so, bench baseline only) */
void vho_mat4_inverse(const float m[16], float out[16]);

/* launchers of DSC/CUDASceneRepHashSDF.cu */
void vho_reset(VhHashData* hd, const VhHashParams* hp);
void vho_reset_bucket_mutex(VhHashData* hd, const VhHashParams* hp);
void vho_alloc(VhHashData* hd, const VhHashParams* hp, const VhDepthCameraData* cam,
               const VhDepthCameraParams* cp, const uint32_t* bitMask);
uint32_t vho_compactify(VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp);
void vho_integrate(VhHashData* hd, const VhHashParams* hp, const VhDepthCameraData* cam,
                   const VhDepthCameraParams* cp);
void vho_starve(VhHashData* hd, const VhHashParams* hp);
void vho_gc_identify(VhHashData* hd, const VhHashParams* hp, const VhDepthCameraParams* cp);
void vho_gc_free(VhHashData* hd, const VhHashParams* hp);

/* launchers of DSC/CUDARayCastSDF.cu and DSC/CameraUtil.cu:669-711 */
void vho_render(const VhHashData* hd, const VhHashParams* hp, const VhRayCastData* rd,
                const VhDepthCameraParams* cp, const VhRayCastParams* rp);
void vho_compute_normals(float* out4, const float* in4, uint32_t width, uint32_t height);

/* launchers of DSC/CUDASceneRepChunkGrid.cu */
uint32_t vho_stream_out_pass1(VhHashData* hd, const VhHashParams* hp, uint32_t threadsPerPart,
                              uint32_t start, float radius, const float camPos[3],
                              VhSDFBlockDesc* out, uint32_t outCapacity);
void vho_stream_out_pass2(VhHashData* hd, const VhHashParams* hp, const VhSDFBlockDesc* descs,
                          VhVoxel* out, uint32_t n);
uint32_t vho_stream_in_pass1(VhHashData* hd, const VhHashParams* hp, uint32_t n,
                             uint32_t heapCountPrev, const VhSDFBlockDesc* descs);
void vho_stream_in_pass2(VhHashData* hd, const VhHashParams* hp, uint32_t n,
                         uint32_t heapCountPrev, const VhSDFBlockDesc* descs, const VhVoxel* blocks);

/* single hash operations, for the collision-list tests
 * (DSC/VoxelUtilHashSDF.h:424-468, 533-638, 643-717, 723-809) */
void vho_alloc_block(VhHashData* hd, const VhHashParams* hp, const int32_t pos[3]);
int vho_delete_hash_entry_element(VhHashData* hd, const VhHashParams* hp, const int32_t pos[3]);
int vho_insert_hash_entry(VhHashData* hd, const VhHashParams* hp, const VhHashEntry* e);
VhHashEntry vho_get_hash_entry(const VhHashData* hd, const VhHashParams* hp, const int32_t pos[3]);

/* scalar helpers exposed for the math unit tests */
uint32_t vho_compute_hash_pos(const VhHashParams* hp, const int32_t pos[3]);
void vho_world_to_virtual_voxel_pos(const VhHashParams* hp, const float p[3], int32_t out[3]);
void vho_virtual_voxel_pos_to_sdf_block(const int32_t v[3], int32_t out[3]);
int vho_is_block_in_frustum(const VhHashParams* hp, const VhDepthCameraParams* cp, const int32_t blk[3]);
void vho_camera_to_screen_int(const VhDepthCameraParams* cp, const float p[3], int32_t out[2]);
VhVoxel vho_combine_voxel(const VhHashParams* hp, VhVoxel v0, VhVoxel v1);

/* CUDASceneRepHashSDF::integrate (DSC/CUDASceneRepHashSDF.h:64-83) on an
 * oracle scene: hp is updated in place (transform, inverse, numOccupied).
 * numIntegratedFrames is the caller-held frame counter (in/out). */
void vho_scene_integrate(VhHashData* hd, VhHashParams* hp, const VhSceneOptions* opt,
                         uint32_t* numIntegratedFrames, const float rigidTransform[16],
                         const VhDepthCameraData* cam, const VhDepthCameraParams* cp,
                         const uint32_t* bitMask);

/* CUDARayCastSDF::render (DSC/CUDARayCastSDF.cpp:38-72) */
void vho_raycast_render(const VhHashData* hd, const VhHashParams* hp, const VhRayCastData* rd,
                        const VhDepthCameraParams* cp, VhRayCastParams* rp,
                        const float lastRigidTransform[16]);

/* Sensor pre-processing, DSC/CameraUtil.cu (host buffers; float4 maps as float[4*n]) */
void vho_convert_color_raw_to_float4(float* out4, const uint8_t* rgbx, uint32_t width, uint32_t height);                 /* :137-152 */
void vho_resample_float_map(float* out, uint32_t outW, uint32_t outH, const float* in, uint32_t inW, uint32_t inH);       /* :1071-1118 */
void vho_resample_float4_map(float* out4, uint32_t outW, uint32_t outH, const float* in4, uint32_t inW, uint32_t inH);    /* :1136-1186 */
void vho_convert_color_to_intensity_float(float* out, const float* in4, uint32_t width, uint32_t height);                 /* :258-267 */
void vho_convert_depth_float_to_camera_space_float4(float* out4, const float* in, const VhDepthCameraParams* cp, uint32_t width, uint32_t height); /* :390-407 */
void vho_gauss_filter_float_map(float* out, const float* in, float sigmaD, float sigmaR, uint32_t width, uint32_t height);   /* :555-593 */
void vho_gauss_filter_float4_map(float* out4, const float* in4, float sigmaD, float sigmaR, uint32_t width, uint32_t height); /* :611-651 */
void vho_bilateral_filter_float_map(float* out, const float* in, float sigmaD, float sigmaR, uint32_t width, uint32_t height); /* :446-483 */
void vho_erode_depth_map(float* out, const float* in, int structureSize, uint32_t width, uint32_t height, float dThresh, float fracReq); /* :1632-1670 */

/* Marching cubes: extractIsoSurfacePass1Kernel + extractIsoSurfacePass2Kernel (DSC/CUDAMarchingCubesSDF.cu:65-121)
 * with extractIsoSurfaceAtPosition / vertexInterp (DSC/MarchingCubesSDFUtil.h:154-262), serially: entries in table
 * order, voxels of a block in thread order (x fastest).  Stores at most maxTriangles triangles and returns the
 * number produced. */
uint32_t vho_extract_iso_surface(const VhHashData* hd, const VhHashParams* hp, const VhMarchingCubesParams* mp,
                                 VhTriangle* out, uint32_t maxTriangles);

/* Synthetic scenes of SURVEY.md section 8(d): analytic spheres, double
 * precision, rounded once to float.  spheres = n x {cx,cy,cz,r}; inside != 0
 * renders the far intersection (camera inside the sphere, scene S2). */
void vho_synth_frame(const double* spheres, int nSpheres, int inside, const float camToWorld[16],
                     const VhDepthCameraParams* cp, float* depth, float* color4);

#ifdef __cplusplus
}
#endif
#endif
