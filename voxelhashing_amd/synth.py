"""Synthetic scenes, trajectories and parameter sets of SURVEY.md section 8(d).

Host-side numpy only: poses are computed in double and rounded once to float32
(row-major camera-to-world), exactly as the measurement contract states, so the
oracle, the HIP path and the bench all see bit-identical inputs.
"""
import math

import numpy as np

from . import vhtypes as T

# scene S1 "orbiting sphere": sphere A (0,0,0) r=1.0 and sphere B (0.9,-0.5,0.6) r=0.3
S1_SPHERES = np.array([[0.0, 0.0, 0.0, 1.0], [0.9, -0.5, 0.6, 0.3]], dtype=np.float64)
S1_ORBIT_RADIUS = 2.5
# single sphere seen from 2.5 m (the frame SURVEY.md section 6 was measured on)
SPHERE_A = np.array([[0.0, 0.0, 0.0, 1.0]], dtype=np.float64)
# scene S2 "inside-out room": camera orbits at 0.5 m inside a sphere of r=3.0
S2_SPHERES = np.array([[0.0, 0.0, 0.0, 3.0]], dtype=np.float64)
S2_ORBIT_RADIUS = 0.5
# scene S3 "tracking": four spheres off the orbit centre (S1's big sphere is centred on it: the view does not change
# along the orbit and a geometric tracker cannot see the motion)
S3_SPHERES = np.array([[-0.6, -0.3, 0.2, 0.5], [0.5, 0.2, -0.1, 0.45], [0.0, 0.45, 0.5, 0.35], [0.1, -0.5, -0.4, 0.4]], dtype=np.float64)

# parameter sets: voxel, truncation, truncScale (truncation = 5*voxel, truncScale = 2.5*voxel)
PARAM_SETS = {
    "P4": dict(voxel_size=0.04, truncation=0.20, trunc_scale=0.10),
    "P2": dict(voxel_size=0.02, truncation=0.10, trunc_scale=0.05),
    "P1": dict(voxel_size=0.01, truncation=0.05, trunc_scale=0.025),
    "P04": dict(voxel_size=0.004, truncation=0.02, trunc_scale=0.01),  # reference default (zParametersDefault.txt:25-28)
}

# BASELINE.json configs restated (SURVEY.md section 8(d) "Configs restated")
CONFIGS = {
    "cfg1": dict(width=640, height=480, params="P4", num_buckets=1 << 18, num_sdf_blocks=1 << 17, frames=1, scene="S1"),
    "cfg2": dict(width=640, height=480, params="P4", num_buckets=500000, num_sdf_blocks=1000000, frames=1000, scene="S1"),
    "cfg3": dict(width=640, height=480, params="P1", num_buckets=2000000, num_sdf_blocks=2097152, frames=1000, scene="S1",
                 streaming=True),
    "cfg4": dict(width=1920, height=1080, params="P2", num_buckets=500000, num_sdf_blocks=1000000, frames=1000, scene="S1"),
}


def orbit_pose(k, n_frames=1000, radius=S1_ORBIT_RADIUS, phase=0.0):
    """Camera-to-world matrix T_k (row-major float32[16]) of the orbit:
    theta = 2*pi*k/n + phase, centre c = (r sin t, 0, -r cos t),
    right = (cos t, 0, sin t), down = (0,1,0), fwd = (-sin t, 0, cos t);
    T = [right down fwd c; 0 0 0 1] with the axes as columns."""
    th = 2.0 * math.pi * k / n_frames + phase
    s, c = math.sin(th), math.cos(th)
    right = (c, 0.0, s)
    down = (0.0, 1.0, 0.0)
    fwd = (-s, 0.0, c)
    ctr = (radius * s, 0.0, -radius * c)
    m = np.array([
        [right[0], down[0], fwd[0], ctr[0]],
        [right[1], down[1], fwd[1], ctr[1]],
        [right[2], down[2], fwd[2], ctr[2]],
        [0.0, 0.0, 0.0, 1.0],
    ], dtype=np.float64)
    return m.astype(np.float32).reshape(16)


def orbit_poses(n, n_frames=1000, radius=S1_ORBIT_RADIUS, phase=0.0):
    return np.stack([orbit_pose(k, n_frames, radius, phase) for k in range(n)])


def scene(name):
    """-> (spheres[n,4] float64, inside flag, orbit radius)"""
    if name == "S1":
        return S1_SPHERES, 0, S1_ORBIT_RADIUS
    if name == "A":
        return SPHERE_A, 0, S1_ORBIT_RADIUS
    if name == "S2":
        return S2_SPHERES, 1, S2_ORBIT_RADIUS
    if name == "S3":
        return S3_SPHERES, 0, S1_ORBIT_RADIUS
    raise KeyError(name)


def config_params(cfg, **overrides):
    """-> (HashParams, DepthCameraParams, RayCastParams) for a named config."""
    c = dict(CONFIGS[cfg]) if isinstance(cfg, str) else dict(cfg)
    c.update(overrides)
    ps = PARAM_SETS[c["params"]]
    hp = T.make_hash_params(c["num_buckets"], c["num_sdf_blocks"], **ps)
    cp = T.make_depth_camera_params(c["width"], c["height"])
    rp = T.make_raycast_params(hp, cp)
    return hp, cp, rp


def streaming_sphere(hp, cp):
    """Streaming centre (camera space) and radius as DepthSensing.cpp:1340-1355
    recomputes them at startup (float32 arithmetic)."""
    f = np.float32
    ext = max(hp.m_streamingVoxelExtents[0], hp.m_streamingVoxelExtents[1], hp.m_streamingVoxelExtents[2])
    chunk_radius = f(0.5) * f(ext) * f(math.sqrt(f(3.0)))
    frust_ext = f(hp.m_maxIntegrationDistance) - f(cp.m_sensorDepthWorldMin)
    frust_radius = f(0.5) * frust_ext * f(math.sqrt(f(3.0)))
    pos = np.array([0.0, 0.0, f(cp.m_sensorDepthWorldMin) + f(0.5) * frust_ext], dtype=np.float32)
    return pos, float(frust_radius + chunk_radius)
