"""shared helpers of the parity tests"""
import numpy as np

from voxelhashing_amd import synth, vhtypes as T


def small_config(width=160, height=120, params="P4", num_buckets=1 << 14, num_sdf_blocks=1 << 13, **hp_over):
    ps = dict(synth.PARAM_SETS[params])
    ps.update(hp_over)
    hp = T.make_hash_params(num_buckets, num_sdf_blocks, **ps)
    cp = T.make_depth_camera_params(width, height)
    rp = T.make_raycast_params(hp, cp)
    return hp, cp, rp


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_maps_equal(got, want, what=""):
    """bit-exact comparison of raycast output maps (depth / depth4 / normals / colors)"""
    for k in ("depth", "depth4", "colors", "normals"):
        g, w = bits(got[k]), bits(want[k])
        if not np.array_equal(g, w):
            bad = np.argwhere(g != w)
            i = tuple(bad[0])
            raise AssertionError(f"{what}: map '{k}' differs at {len(bad)} elements, first {i}: got {got[k][i]!r} want {want[k][i]!r}")
