"""Loader and ctypes prototypes of libvoxelhashing_amd.so (include/vh_api.h).

There is no CPU fallback: if the HIP library is missing or a call fails, this
module raises.  (Building needs only hipcc, not a GPU: voxelhashing_amd.build.)
"""
import ctypes as C
import os

import numpy as np

from . import vhtypes as T

_HERE = os.path.dirname(os.path.abspath(__file__))
# VH_LIB_PATH: a measurement build of the same library (voxelhashing_amd.build --out ...), for tools/ only
LIB_PATH = os.environ.get("VH_LIB_PATH") or os.path.join(_HERE, "libvoxelhashing_amd.so")
_LIB = None

P = C.POINTER
_F16 = P(C.c_float)
_VP = C.c_void_p

# name -> (restype, argtypes); every symbol include/vh_api.h declares
PROTOTYPES = {
    "vh_version": (C.c_char_p, []),
    "vh_error_string": (C.c_char_p, [C.c_int]),
    "vh_last_error_message": (C.c_char_p, []),
    "vh_malloc": (C.c_int, [P(_VP), C.c_size_t]),
    "vh_free": (C.c_int, [_VP]),
    "vh_malloc_host": (C.c_int, [P(_VP), C.c_size_t]),
    "vh_free_host": (C.c_int, [_VP]),
    "vh_memcpy_h2d": (C.c_int, [_VP, _VP, C.c_size_t, _VP]),
    "vh_memcpy_d2h": (C.c_int, [_VP, _VP, C.c_size_t, _VP]),
    "vh_memset": (C.c_int, [_VP, C.c_int, C.c_size_t, _VP]),
    "vh_time_next_launch": (C.c_int, [_VP, _VP]),
    "vh_stream_create": (C.c_int, [P(_VP)]),
    "vh_stream_destroy": (C.c_int, [_VP]),
    "vh_stream_synchronize": (C.c_int, [_VP]),
    "vh_device_synchronize": (C.c_int, []),
    "vh_hash_data_alloc": (C.c_int, [P(T.HashData), P(T.HashParams)]),
    "vh_hash_data_free": (C.c_int, [P(T.HashData)]),
    "vh_reset": (C.c_int, [P(T.HashData), P(T.HashParams), _VP]),
    "vh_reset_bucket_mutex": (C.c_int, [P(T.HashData), P(T.HashParams), _VP]),
    "vh_alloc": (C.c_int, [P(T.HashData), P(T.HashParams), P(T.DepthCameraData), P(T.DepthCameraParams), _VP, C.c_int32, _VP]),
    "vh_compactify": (C.c_int, [P(T.HashData), P(T.HashParams), P(T.DepthCameraParams), P(C.c_uint32), C.c_uint32, _VP]),
    "vh_integrate": (C.c_int, [P(T.HashData), P(T.HashParams), P(T.DepthCameraData), P(T.DepthCameraParams), _VP]),
    "vh_starve": (C.c_int, [P(T.HashData), P(T.HashParams), _VP]),
    "vh_gc_identify": (C.c_int, [P(T.HashData), P(T.HashParams), P(T.DepthCameraParams), _VP]),
    "vh_gc_free": (C.c_int, [P(T.HashData), P(T.HashParams), C.c_int32, _VP]),
    "vh_bind_input_depth_color_textures": (C.c_int, [P(T.DepthCameraData)]),
    "vh_integrate_fused": (C.c_int, [P(T.HashData), P(T.HashParams), P(T.DepthCameraData), P(T.DepthCameraParams), C.c_uint32, C.c_int32, _VP, C.c_uint32, _VP, _VP]),
    "vh_alloc_job": (C.c_int, [P(T.FrameJob), _VP]),
    "vh_compactify_job": (C.c_int, [P(T.FrameJob), _VP]),
    "vh_render": (C.c_int, [P(T.HashData), P(T.HashParams), P(T.RayCastData), P(T.DepthCameraParams), P(T.RayCastParams), _VP]),
    "vh_ray_interval_clear": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _VP]),
    "vh_ray_interval_splat": (C.c_int, [P(T.HashData), P(T.HashParams), P(T.DepthCameraParams), P(T.RayCastParams), _VP, _VP, C.c_uint32, _VP, C.c_uint32, _VP, _VP]),
    "vh_render_intervals": (C.c_int, [P(T.HashData), P(T.HashParams), P(T.RayCastData), P(T.DepthCameraParams), P(T.RayCastParams), _VP, _VP, C.c_uint32, _VP, C.c_uint32, _VP]),
    "vh_render_schedule_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "vh_render_split_tiles": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "vh_compute_normals": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, _VP]),
    "vh_render_intervals_co": (C.c_int, [P(T.HashData), P(T.HashParams), P(T.RayCastData), P(T.DepthCameraParams), P(T.RayCastParams), _VP, _VP, C.c_uint32, _VP, C.c_uint32, P(T.FrameJob), _VP]),
    "vh_compute_normals_co": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, P(T.FrameJob), _VP]),
    "vh_compute_normals_co2": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, P(T.FrameJob), P(T.RayCastParams), _VP, _VP, C.c_uint32, _VP, C.c_uint32, _VP, _VP]),
    "vh_stream_out_pass1": (C.c_int, [P(T.HashData), P(T.HashParams), C.c_uint32, C.c_uint32, C.c_float, _F16, _VP, _VP, C.c_uint32, C.c_int32, _VP]),
    "vh_stream_out_pass2": (C.c_int, [P(T.HashData), P(T.HashParams), _VP, _VP, C.c_uint32, _VP]),
    "vh_stream_out_device": (C.c_int, [P(T.HashData), P(T.HashParams), C.c_uint32, C.c_uint32, C.c_float, _F16, _VP, _VP, _VP, C.c_uint32, C.c_int32, _VP, _VP]),
    "vh_publish_count": (C.c_int, [_VP, _VP, C.c_uint32, _VP]),
    "vh_stream_in_device": (C.c_int, [P(T.HashData), P(T.HashParams), C.c_uint32, _VP, _VP, C.c_int32, _VP, _VP, C.c_uint32, _VP, C.c_uint32, _VP]),
    "vh_stream_in_pass1": (C.c_int, [P(T.HashData), P(T.HashParams), C.c_uint32, C.c_uint32, _VP, C.c_int32, _VP]),
    "vh_stream_in_pass1_report": (C.c_int, [P(T.HashData), P(T.HashParams), C.c_uint32, C.c_uint32, _VP, C.c_int32, _VP, _VP]),
    "vh_stream_in_pass2": (C.c_int, [P(T.HashData), P(T.HashParams), C.c_uint32, C.c_uint32, _VP, _VP, _VP]),
    "vh_synth_frame": (C.c_int, [_VP, C.c_int, C.c_int, _F16, P(T.DepthCameraParams), _VP, _VP, _VP]),
    "vh_debug_hash_ops": (C.c_int, [P(T.HashData), P(T.HashParams), _VP, _VP, C.c_uint32, _VP]),
    "vh_debug_check_fast_math": (C.c_int, [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, _VP, _VP]),
    "vh_debug_valu_probe": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, _VP, P(C.c_uint32), _VP]),
    "vh_debug_check_refined_division": (C.c_int, [C.c_uint32, C.c_uint32, _VP, _VP]),
    "vh_publish_words": (C.c_int, [_VP, _VP, _VP, C.c_uint32, _VP]),
    "vh_stream_out_probe": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.c_float, _VP, _VP, _VP, C.c_uint32, _VP]),
    "vh_scene_rep_create": (C.c_int, [P(T.HashParams), P(T.SceneOptions), _VP, P(_VP)]),
    "vh_scene_rep_destroy": (None, [_VP]),
    "vh_scene_rep_integrate": (C.c_int, [_VP, _F16, P(T.DepthCameraData), P(T.DepthCameraParams), _VP]),
    "vh_scene_rep_set_last_rigid_transform_and_compactify": (C.c_int, [_VP, _F16, P(T.DepthCameraParams)]),
    "vh_scene_rep_reset": (C.c_int, [_VP]),
    "vh_scene_rep_get_hash_data": (C.c_int, [_VP, P(T.HashData)]),
    "vh_scene_rep_get_hash_params": (C.c_int, [_VP, P(T.HashParams)]),
    "vh_scene_rep_get_heap_free_count": (C.c_int, [_VP, P(C.c_uint32)]),
    "vh_scene_rep_get_num_occupied_blocks": (C.c_int, [_VP, P(C.c_uint32)]),
    "vh_scene_rep_debug_hash": (C.c_int, [_VP, P(C.c_uint32)]),
    "vh_scene_rep_get_state": (C.c_int, [_VP, P(C.c_uint32)]),
    "vh_scene_rep_get_timings": (C.c_int, [_VP, P(C.c_double)]),
    "vh_scene_rep_set_options": (C.c_int, [_VP, P(T.SceneOptions)]),
    "vh_scene_rep_integrate_ahead": (C.c_int, [_VP, _F16, P(T.DepthCameraData), P(T.DepthCameraParams), _VP, P(P(T.FrameJob))]),
    "vh_scene_rep_integrate_finish": (C.c_int, [_VP, P(T.DepthCameraData), P(T.DepthCameraParams)]),
    "vh_raycast_create": (C.c_int, [P(T.RayCastParams), _VP, P(_VP)]),
    "vh_raycast_destroy": (None, [_VP]),
    "vh_raycast_render": (C.c_int, [_VP, P(T.HashData), P(T.HashParams), P(T.DepthCameraParams), _F16]),
    "vh_raycast_render_co": (C.c_int, [_VP, P(T.HashData), P(T.HashParams), P(T.DepthCameraParams), _F16, P(T.FrameJob)]),
    "vh_raycast_get_data": (C.c_int, [_VP, P(T.RayCastData)]),
    "vh_raycast_get_params": (C.c_int, [_VP, P(T.RayCastParams)]),
    "vh_raycast_get_timings": (C.c_int, [_VP, P(C.c_double)]),
    "vh_raycast_get_event_pair_overhead": (C.c_int, [_VP, P(C.c_double)]),
    "vh_raycast_set_timing": (C.c_int, [_VP, C.c_int]),
    "vh_raycast_set_timing_stride": (C.c_int, [_VP, C.c_int, C.c_uint32]),
    "vh_raycast_set_interval_splatting": (C.c_int, [_VP, C.c_int]),
    "vh_reconstruction_default_options": (None, [P(T.ReconstructionOptions)]),
    "vh_reconstruction_create": (C.c_int, [_VP, _VP, _VP, P(T.DepthCameraParams), P(T.ReconstructionOptions), P(_VP)]),
    "vh_reconstruction_destroy": (None, [_VP]),
    "vh_reconstruction_run": (C.c_int, [_VP, P(T.SequenceFrame), C.c_uint32]),
    "vh_reconstruction_run_ahead": (C.c_int, [_VP, P(T.SequenceFrame), C.c_uint32, P(T.SequenceFrame)]),
    "vh_reconstruction_synchronize": (C.c_int, [_VP]),
    "vh_reconstruction_debug_fail_render": (C.c_int, [_VP, C.c_uint32]),
    "vh_reconstruction_get_stats": (C.c_int, [_VP, P(T.ReconstructionStats)]),
    "vh_reconstruction_reset": (C.c_int, [_VP]),
    "vh_convert_color_raw_to_float4": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, _VP]),
    "vh_upload_frame": (C.c_int, [_VP, _VP, _VP, _VP, C.c_uint32, C.c_uint32, _VP]),
    "vh_resample_float_map": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _VP, C.c_uint32, C.c_uint32, _VP]),
    "vh_resample_float4_map": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _VP, C.c_uint32, C.c_uint32, _VP]),
    "vh_copy_float_map": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, _VP]),
    "vh_copy_float4_map": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, _VP]),
    "vh_set_invalid_float_map": (C.c_int, [_VP, C.c_uint32, C.c_uint32, _VP]),
    "vh_convert_color_to_intensity_float": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, _VP]),
    "vh_convert_depth_float_to_camera_space_float4": (C.c_int, [_VP, _VP, P(T.DepthCameraParams), C.c_uint32, C.c_uint32, _VP]),
    "vh_gauss_filter_float_map": (C.c_int, [_VP, _VP, C.c_float, C.c_float, C.c_uint32, C.c_uint32, _VP]),
    "vh_gauss_filter_float4_map": (C.c_int, [_VP, _VP, C.c_float, C.c_float, C.c_uint32, C.c_uint32, _VP]),
    "vh_bilateral_filter_float_map": (C.c_int, [_VP, _VP, C.c_float, C.c_float, C.c_uint32, C.c_uint32, _VP]),
    "vh_erode_depth_map": (C.c_int, [_VP, _VP, C.c_int32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, _VP]),
    "vh_icp_begin": (C.c_int, [_VP, _VP, _VP]),
    "vh_icp_begin_level": (C.c_int, [_VP, _VP]),
    "vh_icp_projective_correspondences": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, _VP, P(T.DepthCameraParams), _VP]),
    "vh_icp_num_partials": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "vh_icp_build_linear_system": (C.c_int, [C.c_uint32, C.c_uint32, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vh_icp_solve": (C.c_int, [_VP, _VP, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_int, _VP]),
    "vh_tracking_state_read": (C.c_int, [C.c_char_p, P(T.TrackingState)]),
    "vh_tracking_state_parse": (C.c_int, [C.c_char_p, P(T.TrackingState)]),
    "vh_camera_tracking_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, _VP, P(_VP)]),
    "vh_camera_tracking_destroy": (None, [_VP]),
    "vh_camera_tracking_apply_ct": (C.c_int, [_VP, _VP, _VP, _VP, _VP, P(C.c_float), P(T.TrackingState), P(C.c_float), P(T.DepthCameraParams), P(C.c_float), P(C.c_int), P(T.IcpState)]),
    "vh_app_state_read": (C.c_int, [C.c_char_p, P(T.AppState)]),
    "vh_app_state_parse": (C.c_int, [C.c_char_p, P(T.AppState)]),
    "vh_hash_params_from_app_state": (None, [P(T.AppState), P(T.HashParams)]),
    "vh_raycast_params_from_app_state": (None, [P(T.AppState), P(C.c_float), P(C.c_float), P(T.RayCastParams)]),
    "vh_marching_cubes_params_from_app_state": (None, [P(T.AppState), P(T.MarchingCubesParams)]),
    "vh_scene_options_from_app_state": (None, [P(T.AppState), P(T.SceneOptions)]),
    "vh_sensor_data_create": (C.c_int, [P(T.SensorDataInfo), P(_VP)]),
    "vh_sensor_data_load": (C.c_int, [C.c_char_p, P(_VP)]),
    "vh_sensor_data_destroy": (None, [_VP]),
    "vh_sensor_data_save": (C.c_int, [_VP, C.c_char_p]),
    "vh_sensor_data_info": (C.c_int, [_VP, P(T.SensorDataInfo)]),
    "vh_sensor_data_add_frame": (C.c_int, [_VP, _VP, _VP, P(C.c_float), C.c_uint64, C.c_uint64]),
    "vh_sensor_data_add_frame_compressed": (C.c_int, [_VP, _VP, C.c_uint64, _VP, P(C.c_float), C.c_uint64, C.c_uint64]),
    "vh_sensor_data_add_imu_frame": (C.c_int, [_VP, P(C.c_double), C.c_uint64]),
    "vh_sensor_data_get_frame": (C.c_int, [_VP, C.c_uint64, _VP, _VP, P(C.c_float), P(C.c_uint64)]),
    "vh_sensor_data_reader_create": (C.c_int, [C.c_char_p, P(_VP)]),
    "vh_sensor_data_reader_destroy": (None, [_VP]),
    "vh_sensor_data_reader_info": (C.c_int, [_VP, P(T.SensorDataInfo)]),
    "vh_sensor_data_reader_process_depth": (C.c_int, [_VP, P(C.c_int), P(_VP), P(_VP)]),
    "vh_sensor_data_reader_get_rigid_transform": (C.c_int, [_VP, C.c_int, P(C.c_float)]),
    "vh_sensor_data_reader_get_curr_frame": (C.c_int, [_VP, P(C.c_uint32), P(C.c_uint32)]),
    "vh_rgbd_sensor_create": (C.c_int, [P(C.c_uint32), P(C.c_float), _VP, P(_VP)]),
    "vh_rgbd_sensor_destroy": (None, [_VP]),
    "vh_rgbd_sensor_set_filter_depth_values": (C.c_int, [_VP, C.c_int, C.c_float, C.c_float]),
    "vh_rgbd_sensor_set_filter_intensity_values": (C.c_int, [_VP, C.c_int, C.c_float, C.c_float]),
    "vh_rgbd_sensor_process": (C.c_int, [_VP, _VP, _VP]),
    "vh_rgbd_sensor_get_depth_camera_data": (C.c_int, [_VP, P(T.DepthCameraData)]),
    "vh_rgbd_sensor_get_depth_camera_params": (C.c_int, [_VP, P(T.DepthCameraParams)]),
    "vh_rgbd_sensor_get_maps": (C.c_int, [_VP, P(_VP), P(_VP), P(_VP)]),
    "vh_marching_cubes_data_alloc": (C.c_int, [P(T.MarchingCubesData), P(T.MarchingCubesParams)]),
    "vh_marching_cubes_data_free": (None, [P(T.MarchingCubesData)]),
    "vh_marching_cubes_update_params": (C.c_int, [P(T.MarchingCubesData), P(T.MarchingCubesParams), _VP]),
    "vh_reset_marching_cubes": (C.c_int, [P(T.MarchingCubesData), _VP]),
    "vh_extract_iso_surface_pass1": (C.c_int, [P(T.HashData), P(T.HashParams), P(T.MarchingCubesData), _VP]),
    "vh_extract_iso_surface_pass2": (C.c_int, [P(T.HashData), P(T.HashParams), P(T.MarchingCubesData), C.c_uint32, _VP]),
    "vh_marching_cubes_create": (C.c_int, [P(T.MarchingCubesParams), _VP, P(_VP)]),
    "vh_marching_cubes_destroy": (None, [_VP]),
    "vh_marching_cubes_parameters": (C.c_int, [C.c_uint32, C.c_float, C.c_float, C.c_uint32, P(T.MarchingCubesParams)]),
    "vh_marching_cubes_set_offline_processing": (C.c_int, [_VP, C.c_int]),
    "vh_marching_cubes_extract_iso_surface": (C.c_int, [_VP, P(T.HashData), P(T.HashParams), P(C.c_float), P(C.c_float), C.c_int, C.c_int]),
    "vh_marching_cubes_extract_iso_surface_chunk_grid": (C.c_int, [_VP, _VP, P(C.c_float), C.c_float]),
    "vh_marching_cubes_copy_triangles_to_cpu": (C.c_int, [_VP]),
    "vh_marching_cubes_clear_mesh_buffer": (C.c_int, [_VP]),
    "vh_marching_cubes_get_counts": (C.c_int, [_VP, P(C.c_uint32)]),
    "vh_marching_cubes_download_triangles": (C.c_int, [_VP, _VP, C.c_uint32]),
    "vh_marching_cubes_get_mesh_size": (C.c_int, [_VP, P(C.c_uint64)]),
    "vh_marching_cubes_get_mesh": (C.c_int, [_VP, _VP, _VP, _VP]),
    "vh_marching_cubes_save_mesh": (C.c_int, [_VP, C.c_char_p, P(C.c_float), C.c_int]),
    "vh_chunk_grid_create": (C.c_int, [_VP, _F16, P(C.c_int32), P(C.c_int32), C.c_uint32, C.c_int, C.c_uint32, P(_VP)]),
    "vh_chunk_grid_destroy": (None, [_VP]),
    "vh_chunk_grid_stream_out_to_cpu_pass0_gpu": (C.c_int, [_VP, _F16, C.c_float, C.c_int, C.c_int]),
    "vh_chunk_grid_stream_out_to_cpu_pass1_cpu": (C.c_int, [_VP, C.c_int]),
    "vh_chunk_grid_stream_in_to_gpu_pass0_cpu": (C.c_int, [_VP, _F16, C.c_float, C.c_int, C.c_int]),
    "vh_chunk_grid_stream_in_to_gpu_pass1_gpu": (C.c_int, [_VP, C.c_int]),
    "vh_chunk_grid_stream_out_to_cpu": (C.c_int, [_VP, _F16, C.c_float, C.c_int, P(C.c_uint32)]),
    "vh_chunk_grid_stream_in_to_gpu": (C.c_int, [_VP, _F16, C.c_float, C.c_int, P(C.c_uint32)]),
    "vh_chunk_grid_stream_out_to_cpu_all": (C.c_int, [_VP]),
    "vh_chunk_grid_stream_in_to_gpu_all": (C.c_int, [_VP, _F16, C.c_float, C.c_int, P(C.c_uint32)]),
    "vh_chunk_grid_get_bit_mask_gpu": (C.c_int, [_VP, P(_VP)]),
    "vh_chunk_grid_reset": (C.c_int, [_VP]),
    "vh_chunk_grid_debug_check_for_duplicates": (C.c_int, [_VP]),
    "vh_chunk_grid_get_statistics": (C.c_int, [_VP, P(C.c_uint32)]),
    "vh_chunk_grid_get_num_failed_inserts": (C.c_int, [_VP, P(C.c_uint32)]),
    "vh_chunk_grid_download_host_blocks": (C.c_int, [_VP, _VP, _VP, C.c_uint32, P(C.c_uint32)]),
    "vh_chunk_grid_save_to_file": (C.c_int, [_VP, C.c_char_p, _F16, C.c_float]),
    "vh_chunk_grid_load_from_file": (C.c_int, [_VP, C.c_char_p, _F16, C.c_float]),
}


class VhError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(what)


def load():
    """dlopen the in-tree HIP library; raises if it has not been built"""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m voxelhashing_amd.build` "
                           "(hipcc, gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def check(code, what=""):
    if code != 0:
        L = load()
        msg = L.vh_error_string(code).decode()
        extra = L.vh_last_error_message().decode()
        raise VhError(code, f"{what}: {msg} (code {code}){' -- ' + extra if extra else ''}")


def f16(m):
    a = np.ascontiguousarray(m, dtype=np.float32).reshape(-1)
    return a.ctypes.data_as(_F16)


class DeviceBuffer:
    """A hipMalloc'ed buffer owned through vh_malloc/vh_free."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(load().vh_malloc(C.byref(p), max(self.nbytes, 1)), "vh_malloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a, stream=None):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        b.upload(a, stream)
        return b

    def upload(self, a, stream=None):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        if a.nbytes:
            check(load().vh_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes, stream), "vh_memcpy_h2d")

    def download(self, dtype, count=None, stream=None):
        dt = np.dtype(dtype)
        n = self.nbytes // dt.itemsize if count is None else count
        out = np.empty(n, dtype=dt)
        if out.nbytes:
            check(load().vh_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, stream), "vh_memcpy_d2h")
        return out

    def free(self):
        if getattr(self, "ptr", None):
            load().vh_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """A numpy array in pinned, device-visible host memory (vh_malloc_host)."""

    def __init__(self, shape, dtype):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(int(v) for v in np.atleast_1d(shape))
        n = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        check(load().vh_malloc_host(C.byref(p), max(n, 1)), "vh_malloc_host")
        self.ptr = p.value
        buf = (C.c_char * max(n, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        p = cls(a.shape, a.dtype)
        p.array[...] = a
        return p

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            load().vh_free_host(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def download(ptr, dtype, count, stream=None):
    """copy `count` items of `dtype` from a raw device pointer"""
    out = np.empty(count, dtype=np.dtype(dtype))
    if out.nbytes:
        check(load().vh_memcpy_d2h(out.ctypes.data, ptr, out.nbytes, stream), "vh_memcpy_d2h")
    return out
