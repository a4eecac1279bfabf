"""MI355X-native voxel-hashing TSDF fusion + raycast engine (hot path of
MicroYY/VoxelHashing's DepthSensingCUDA): HIP kernels behind a C ABI, with a
Python mirror of the reference host classes for tests and benchmarks.

Submodules: vhtypes (POD mirrors + parameter builders), synth (synthetic scenes),
canonical (parity forms + invariants), lib (C-ABI loader), engine (host-class
mirror: CUDASceneRepHashSDF / CUDARayCastSDF / CUDASceneRepChunkGrid).
"""
__version__ = "0.1.0"
