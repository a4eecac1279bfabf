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
