"""The fused integrate pass picks its shape on the device from the block count (vh_kernels.hip, k_integrate_fused):
one workgroup per block when blocks are few, one wave per block in several rounds when they are many, and in that
shape a block's screen footprint is staged in LDS when it fits 32 x 31 pixels.  Each shape and each branch against
the oracle, bit for bit, over frames with garbage collection and starving."""
import numpy as np
import pytest

from helpers import small_config
from voxelhashing_amd import canonical, synth, vhtypes as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E(vh):
    from voxelhashing_amd import engine
    return engine


def run(E, O, hp, cp, rp, scene_name, n_frames, n_orbit, opt):
    spheres, inside, radius = synth.scene(scene_name)
    scene, ref = E.CUDASceneRepHashSDF(hp, opt), O.OracleScene(hp, cp, rp, opt)
    frame = E.DepthFrame(cp)
    most = 0
    for k in range(n_frames):
        pose = synth.orbit_pose(k, n_orbit, radius)
        E.synth_frame(spheres, inside, pose, cp, out=frame)
        d, c = O.synth_frame(spheres, inside, pose, cp)
        scene.integrate(pose, frame, cp, None)
        ref.integrate(pose, d, c)
        most = max(most, scene.getNumOccupiedBlocks())
        canonical.assert_same_scene(scene.state(), ref.state(), f"{scene_name} frame {k}")
    st = scene.getState()
    assert st[T.STATE_HEAP_UNDERFLOW] == 0
    return most


@pytest.mark.parametrize("width,height,what", [(160, 120, "footprints of ~17 pixels: staged"), (320, 240, "footprints of ~34 pixels: gathered")])
def test_wave_per_block_shape_small_pool(E, oracle_lib, width, height, what):
    """a pool of 512 blocks launches 128 workgroups: ~170 blocks in the frustum take the wave-per-block shape"""
    hp, cp, rp = small_config(width, height, params="P4", num_buckets=1 << 14, num_sdf_blocks=512)
    opt = T.make_scene_options(offline=True, gc=True, starve=3)
    most = run(E, oracle_lib, hp, cp, rp, "S1", 7, 150, opt)
    assert 128 < most <= 512, most


def test_wave_per_block_shape_many_rounds(E, oracle_lib):
    """camera inside the 3 m sphere, 1 cm voxels: thousands of small blocks, every pixel valid -- several rounds of
    blocks per wave, every footprint staged"""
    hp, cp, rp = small_config(320, 240, params="P1", num_buckets=1 << 17, num_sdf_blocks=1 << 14)
    opt = T.make_scene_options(offline=True, gc=True, starve=2)
    most = run(E, oracle_lib, hp, cp, rp, "S2", 4, 400, opt)
    assert most > 5120 + 64, most  # more than one round (kIntegrateWavesMost = 5120)


def test_workgroup_per_block_shape_without_gc(E, oracle_lib):
    """few blocks, garbage collection off: the reduction and both barriers are skipped"""
    hp, cp, rp = small_config(160, 120, params="P4")
    opt = T.make_scene_options(offline=True, gc=False)
    most = run(E, oracle_lib, hp, cp, rp, "S3", 5, 100, opt)
    assert 50 < most < 2048


def test_certified_blocks_with_crafted_voxels(E, oracle_lib):
    """Blocks whose corners pass the certificate take integrate_block_certified (shared refined reciprocals, packed
    arithmetic, pixel from the staged tile).  Its blend division leaves the fast route for numerators outside
    [2^-100, 2^90): blocks streamed in with denormal, tiny, huge and signed-zero sdf values and every kind of weight sit
    in the free space in front of a wall, where every voxel integrates.  Launcher level, against the oracle."""
    import ctypes as C
    from voxelhashing_amd.lib import DeviceBuffer
    O = oracle_lib
    hp, cp, rp = small_config(160, 120, params="P2", num_buckets=1 << 12, num_sdf_blocks=2048)
    rng = np.random.default_rng(20260)
    W, H = cp.m_imageWidth, cp.m_imageHeight
    depth = np.full((H, W), 2.5, np.float32) + rng.uniform(-0.02, 0.02, (H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.03] = -np.inf
    color = np.concatenate([rng.random((H, W, 3), dtype=np.float32), np.ones((H, W, 1), np.float32)], axis=2)
    color[rng.random((H, W)) < 0.02, :3] = -np.inf
    # a camera pose with every rotation axis in play, away from the origin
    a, b, c = np.radians([17.0, -23.0, 9.0])
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    pose = np.eye(4)
    pose[:3, :3] = rz @ ry @ rx
    pose[:3, 3] = [3.1, -1.7, 2.3]
    pose = pose.astype(np.float32).reshape(16)

    # crafted blocks in front of the wall
    z = rng.uniform(0.9, 2.1, 600)
    pts = np.stack([rng.uniform(-0.4, 0.4, 600) * z, rng.uniform(-0.3, 0.3, 600) * z, z, np.ones(600)], axis=1)
    ids = np.unique(np.floor((pts @ pose.reshape(4, 4).astype(np.float64).T)[:, :3] / (8 * hp.m_virtualVoxelSize)).astype(np.int32), axis=0)[:200]
    assert len(ids) > 150, len(ids)
    special = np.array([0.0, -0.0, 1e-45, -1e-40, 1e-35, -3e-31, 5e-31, 1e-30, 1.5, -0.07, 0.2, 1e26, -4.7e24, 6e24, 2e28, -1e38, 3e38], np.float32)
    blocks = np.zeros((len(ids), T.SDF_BLOCK_VOXELS), T.VOXEL_DTYPE)
    pick = rng.random(blocks.shape)
    blocks["sdf"] = np.where(pick < 0.5, special[rng.integers(0, len(special), blocks.shape)], rng.normal(0, 0.2, blocks.shape).astype(np.float32))
    blocks["weight"] = np.array([0, 1, 2, 3, 127, 128, 254, 255], np.uint8)[rng.integers(0, 8, blocks.shape)]
    blocks["color"] = rng.integers(0, 256, blocks.shape + (3,), dtype=np.uint8)
    descs = np.zeros(len(ids), T.DESC_DTYPE)
    descs["pos"] = ids

    g = E.LauncherScene(hp)
    o = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=True))
    g.set_transform(pose, O.mat4_inverse(pose))
    o.set_transform(pose)
    g.reset_mutex()
    g.stream_in(descs, blocks, T.LOCK_ENTRY)
    assert o.stream_in(descs, blocks) == 0
    frame = E.DepthFrame(cp, depth, color)
    packed = DeviceBuffer(8 * W * H)
    job = g.frame_job(frame, cp, packed_ptr=packed.ptr)
    prev = -1
    while True:  # alloc until the heap stops changing; every pass packs the frame as well
        g.reset_mutex()
        g.alloc_job(job)
        cur = g.download(with_voxels=False)["heap_counter"]
        if cur == prev:
            break
        prev = cur
    prev = -1
    while True:
        o.reset_mutex()
        o.alloc(depth, color)
        cur = o.heap_free_count()
        if cur == prev:
            break
        prev = cur
    n = g.compactify(cp)
    assert n == o.compactify() and 512 < n <= 2048, n  # 512 workgroups: the wave-per-block shape
    canonical.assert_same_scene(g.state(), o.state(), "before the pass")
    g.reset_mutex()
    g.integrate_fused(frame, cp, 3, T.LOCK_ENTRY, packed.ptr)  # VH_FUSED_GC | VH_FUSED_STARVE
    o.integrate_depth_map(depth, color)
    o.starve()
    o.gc_identify()
    o.reset_mutex()
    o.gc_free()
    gs, os_ = g.state(), o.state()
    canonical.assert_same_scene(gs, os_, "after the fused pass")
    # the crafted values were really blended: most crafted voxels changed
    assert gs["num_occupied"] > 100


def test_compactify_queue_overflow_and_boxes(E, oracle_lib):
    """4096 buckets are ONE compactify workgroup; a dense scene leaves more in-frustum blocks than its LDS queue takes
    (768): the rest go straight to the list.  Every entry of the list carries its block's screen box, flags and the
    frame's tag; the list as a set and the pass over it equal the oracle's."""
    O = oracle_lib
    hp, cp, rp = small_config(160, 120, params="P2", num_buckets=1 << 12, num_sdf_blocks=1 << 13)
    opt = T.make_scene_options(offline=True, gc=True, starve=2)
    spheres, inside, radius = synth.scene("S2")
    scene, ref = E.CUDASceneRepHashSDF(hp, opt), O.OracleScene(hp, cp, rp, opt)
    frame = E.DepthFrame(cp)
    for k in range(3):
        pose = synth.orbit_pose(k, 300, radius)
        E.synth_frame(spheres, inside, pose, cp, out=frame)
        d, c = O.synth_frame(spheres, inside, pose, cp)
        scene.integrate(pose, frame, cp, None)
        ref.integrate(pose, d, c)
        canonical.assert_same_scene(scene.state(), ref.state(), f"frame {k}")
    raw = scene.download(with_voxels=False)
    n = raw["compact_count"]
    assert n > 768 + 64, n
    comp = raw["compactified"]
    pad = comp["_pad"].view(np.uint32)
    assert len(np.unique(pad[:, 2])) == 1 and pad[0, 2] != 0, "one tag, not zero, on every entry"
    x0, y0 = pad[:, 0] & 0xFFFF, pad[:, 0] >> 16
    w, h, flags = pad[:, 1] & 0xFF, (pad[:, 1] >> 8) & 0xFF, pad[:, 1] >> 16
    staged = (flags & 1) != 0
    assert staged.sum() > n // 2 and ((flags & 2) != 0).sum() > n // 2
    assert ((x0[staged] & 1) == 0).all() and (x0[staged] + w[staged] <= cp.m_imageWidth).all() and (y0[staged] + h[staged] <= cp.m_imageHeight).all()
    assert (w[staged] >= 1).all() and (w[staged] <= 32).all() and (h[staged] >= 1).all() and (h[staged] <= 29).all()
    assert (w[~staged] == 0).all() and (h[~staged] == 0).all()


def test_list_made_for_another_transform_takes_the_plain_code(E, oracle_lib):
    """launcher level: compactify under one pose, the fused pass under another (nobody does that; the list's boxes and
    certificates are then for the wrong transform and the tag says so): every block goes through the plain code, and the
    result is the oracle's for the same two calls"""
    from voxelhashing_amd.lib import DeviceBuffer
    O = oracle_lib
    hp, cp, rp = small_config(160, 120, params="P2", num_buckets=1 << 12, num_sdf_blocks=4096)  # 1024 workgroups
    spheres, inside, radius = synth.scene("S2")
    pose_a, pose_b = synth.orbit_pose(0, 300, radius), synth.orbit_pose(2, 300, radius)
    g = E.LauncherScene(hp)
    o = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=True))
    frame = E.DepthFrame(cp)
    E.synth_frame(spheres, inside, pose_a, cp, out=frame)
    depth, color = O.synth_frame(spheres, inside, pose_a, cp)
    g.set_transform(pose_a, O.mat4_inverse(pose_a))
    o.set_transform(pose_a)
    packed = DeviceBuffer(8 * cp.m_imageWidth * cp.m_imageHeight)
    job = g.frame_job(frame, cp, packed_ptr=packed.ptr)
    prev = -1
    while True:  # alloc until the heap stops changing; every pass packs the frame as well
        g.reset_mutex()
        g.alloc_job(job)
        cur = g.download(with_voxels=False)["heap_counter"]
        if cur == prev:
            break
        prev = cur
    prev = -1
    while True:
        o.reset_mutex()
        o.alloc(depth, color)
        cur = o.heap_free_count()
        if cur == prev:
            break
        prev = cur
    canonical.assert_same_scene(g.state(), o.state(), "after alloc")
    n = g.compactify(cp)
    assert n == o.compactify() and 1024 < n < 4096, n  # the wave-per-block shape
    g.set_transform(pose_b, O.mat4_inverse(pose_b))
    o.set_transform(pose_b)
    g.reset_mutex()
    g.integrate_fused(frame, cp, 3, T.LOCK_ENTRY, packed.ptr)
    o.integrate_depth_map(depth, color)
    o.starve()
    o.gc_identify()
    o.reset_mutex()
    o.gc_free()
    canonical.assert_same_scene(g.state(), o.state(), "after the fused pass under the other pose")
