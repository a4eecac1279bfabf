"""The C oracle's scalar functions against an independent numpy-float32
restatement written from the same reference lines (two codings of one spec):
hash, world->voxel rounding, floor division, frustum test, projection,
combineVoxel, 4x4 inverse."""
import ctypes as C

import numpy as np

from voxelhashing_amd import synth, vhtypes as T

f32 = np.float32


def np_hash(x, y, z, nb):
    # computeHashPos, DepthSensingCUDA/Source/VoxelUtilHashSDF.h:217-225
    h = (np.int64(x) * 73856093) & 0xFFFFFFFF
    h ^= (np.int64(y) * 19349669) & 0xFFFFFFFF
    h ^= (np.int64(z) * 83492791) & 0xFFFFFFFF
    return int(h % nb)


def np_sign(v):
    return int(v > 0) - int(v < 0)


def np_world_to_vvp(p, vs):
    # worldToVirtualVoxelPos :266-270
    out = []
    for c in p:
        q = f32(c) / f32(vs)
        out.append(int(np.trunc(f32(q + f32(np_sign(q)) * f32(0.5)))))
    return out


def np_block(v):
    # virtualVoxelPosToSDFBlock :273-282
    return [(c - 7) // 8 if c < 0 and False else int(np.floor(c / 8.0)) for c in v]


def test_hash_rounding_blocks(oracle_lib):
    L = oracle_lib.lib()
    rng = np.random.default_rng(7)
    hp = T.make_hash_params(500000, 1024, **synth.PARAM_SETS["P4"])
    for _ in range(2000):
        pos = rng.integers(-5000, 5000, 3).astype(np.int32)
        got = L.vho_compute_hash_pos(C.byref(hp), pos.ctypes.data_as(C.POINTER(C.c_int32)))
        assert got == np_hash(pos[0], pos[1], pos[2], 500000)
    big = np.array([2 ** 30, -2 ** 30, 123456789], dtype=np.int32)  # wrapping products
    assert L.vho_compute_hash_pos(C.byref(hp), big.ctypes.data_as(C.POINTER(C.c_int32))) == np_hash(*[int(v) for v in big], 500000)
    for vs in (0.04, 0.01, 0.004):
        hp.m_virtualVoxelSize = vs
        pts = rng.uniform(-6, 6, (2000, 3)).astype(np.float32)
        pts[:50] = (rng.integers(-100, 100, (50, 3)) * np.float32(vs) * np.float32(0.5)).astype(np.float32)  # on the .5 cliffs
        for p in pts:
            out = (C.c_int32 * 3)()
            L.vho_world_to_virtual_voxel_pos(C.byref(hp), p.ctypes.data_as(C.POINTER(C.c_float)), out)
            assert list(out) == np_world_to_vvp(p, hp.m_virtualVoxelSize)
    for _ in range(2000):
        v = rng.integers(-100000, 100000, 3).astype(np.int32)
        out = (C.c_int32 * 3)()
        L.vho_virtual_voxel_pos_to_sdf_block(v.ctypes.data_as(C.POINTER(C.c_int32)), out)
        assert list(out) == [int(c) // 8 for c in v]  # python // floors


def np_frustum(hp, cp, blk):
    # isSDFBlockInCameraFrustumApprox :306-309 + DepthCameraUtil.h:99-110,141-147 in float32
    vs = f32(hp.m_virtualVoxelSize)
    off = f32(f32(vs * f32(0.5)) * f32(7.0))
    pw = [f32(f32(f32(b * 8) * vs) + off) for b in blk]
    m = np.array(hp.m_rigidTransformInverse, dtype=np.float32)
    pc = [f32(f32(f32(f32(m[4 * r] * pw[0]) + f32(m[4 * r + 1] * pw[1])) + f32(m[4 * r + 2] * pw[2])) + f32(m[4 * r + 3] * f32(1.0))) for r in range(3)]
    with np.errstate(all="ignore"):
        px = f32(f32(f32(pc[0] * f32(cp.fx)) / pc[2]) + f32(cp.mx))
        py = f32(f32(f32(pc[1] * f32(cp.fy)) / pc[2]) + f32(cp.my))
        wm1, hm1 = f32(f32(cp.m_imageWidth) - f32(1)), f32(f32(cp.m_imageHeight) - f32(1))
        x = f32(f32(f32(f32(2) * px) - wm1) / wm1) * f32(0.95)
        y = f32(f32(hm1 - f32(f32(2) * py)) / hm1) * f32(0.95)
        z = f32(f32(pc[2] - f32(cp.m_sensorDepthWorldMin)) / f32(f32(cp.m_sensorDepthWorldMax) - f32(cp.m_sensorDepthWorldMin))) * f32(0.95)
    return int(not (x < -1 or x > 1 or y < -1 or y > 1 or z < 0 or z > 1))


def test_frustum_and_projection(oracle_lib):
    O = oracle_lib
    L = O.lib()
    rng = np.random.default_rng(11)
    hp, cp, _ = synth.config_params("cfg1")
    pose = synth.orbit_pose(137)
    hp.m_rigidTransform = T.mat16(pose)
    hp.m_rigidTransformInverse = T.mat16(O.mat4_inverse(pose))
    n_in = 0
    for _ in range(3000):
        blk = rng.integers(-12, 12, 3).astype(np.int32)
        got = L.vho_is_block_in_frustum(C.byref(hp), C.byref(cp), blk.ctypes.data_as(C.POINTER(C.c_int32)))
        assert got == np_frustum(hp, cp, [int(b) for b in blk])
        n_in += got
    assert 50 < n_in < 2950
    for _ in range(2000):
        p = rng.uniform(-3, 3, 3).astype(np.float32)
        p[2] = abs(p[2]) + np.float32(0.3)
        out = (C.c_int32 * 2)()
        L.vho_camera_to_screen_int(C.byref(cp), p.ctypes.data_as(C.POINTER(C.c_float)), out)
        sx = f32(f32(f32(p[0] * f32(cp.fx)) / p[2]) + f32(cp.mx))
        sy = f32(f32(f32(p[1] * f32(cp.fy)) / p[2]) + f32(cp.my))
        assert list(out) == [int(np.trunc(f32(sx + f32(0.5)))), int(np.trunc(f32(sy + f32(0.5))))]


def test_combine_voxel(oracle_lib):
    L = oracle_lib.lib()
    rng = np.random.default_rng(3)
    hp = T.make_hash_params(1024, 1024, **synth.PARAM_SETS["P4"])
    for _ in range(3000):
        v0, v1 = T.Voxel(), T.Voxel()
        v0.sdf, v1.sdf = float(f32(rng.uniform(-0.5, 0.5))), float(f32(rng.uniform(-0.5, 0.5)))
        v0.weight, v1.weight = int(rng.integers(0, 256)), int(rng.integers(1, 16))
        for c in range(3):
            v0.color[c], v1.color[c] = int(rng.integers(0, 256)), int(rng.integers(0, 256))
        out = L.vho_combine_voxel(C.byref(hp), v0, v1)
        # combineVoxel, VoxelUtilHashSDF.h:229-250
        w0, w1 = f32(v0.weight), f32(v1.weight)
        sdf = f32(f32(f32(f32(v0.sdf) * w0) + f32(f32(v1.sdf) * w1)) / f32(w0 + w1))
        assert np.float32(out.sdf).view(np.uint32) == sdf.view(np.uint32)
        assert out.weight == min(255, v0.weight + v1.weight)
        for c in range(3):
            res = f32(f32(f32(0.5) * f32(v0.color[c])) + f32(f32(0.5) * f32(v1.color[c])))
            assert out.color[c] == int(np.trunc(f32(res + f32(0.5))))


def test_mat4_inverse(oracle_lib):
    O = oracle_lib
    for k in (0, 1, 250, 777):
        m = synth.orbit_pose(k).reshape(4, 4)
        inv = O.mat4_inverse(m).reshape(4, 4)
        assert np.allclose(inv.astype(np.float64) @ m.astype(np.float64), np.eye(4), atol=2e-6)
        # rigid transform: inverse = [R^T, -R^T c]
        assert np.allclose(inv[:3, :3], m[:3, :3].T, atol=1e-6)
    assert np.array_equal(O.mat4_inverse(np.eye(4, dtype=np.float32)).reshape(4, 4), np.eye(4, dtype=np.float32))


def test_colour_average_identity_used_by_the_hip_path():
    """combineVoxel's colour (DSC/VoxelUtilHashSDF.h:236-240), uchar(0.5f*c0 + 0.5f*c1 + 0.5f), equals the byte-wise
    average rounded up, (a | b) - (((a ^ b) >> 1) & 0x7f), for every pair of bytes: the HIP kernels use the integer
    form on the packed colour word (vh_device.hpp combine_voxel)."""
    a, b = np.meshgrid(np.arange(256, dtype=np.uint32), np.arange(256, dtype=np.uint32), indexing="ij")
    f = np.float32
    ref = (f(0.5) * a.astype(np.float32) + f(0.5) * b.astype(np.float32) + f(0.5)).astype(np.float32)
    ref = np.minimum(np.maximum(np.trunc(ref), 0), 255).astype(np.uint32)
    got = (a | b) - (((a ^ b) >> 1) & 0x7F)
    assert np.array_equal(ref, got)
    # packed: three channels side by side do not disturb each other (no borrow crosses a byte: (a|b) >= ((a^b)>>1) per byte)
    rng = np.random.default_rng(5)
    wa, wb = rng.integers(0, 1 << 24, 100000, dtype=np.uint32), rng.integers(0, 1 << 24, 100000, dtype=np.uint32)
    packed = (wa | wb) - (((wa ^ wb) >> 1) & 0x7F7F7F7F)
    for sh in (0, 8, 16):
        ca, cb = (wa >> sh) & 0xFF, (wb >> sh) & 0xFF
        assert np.array_equal((packed >> sh) & 0xFF, (ca | cb) - (((ca ^ cb) >> 1) & 0x7F))
