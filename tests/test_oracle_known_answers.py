"""The oracle against the only reference-run facts that exist for this path:
the counts SURVEY.md sections 6 / 7 (hard part 6) / 8 recorded when the
reference's own device code was run serially during the survey (one analytic
sphere frame, r = 1 m seen from 2.5 m; truncation = 5*voxel, truncScale =
2.5*voxel; offline alloc loop).  Everything else about the oracle is unpinned
(the reference has no tests or fixtures)."""
import numpy as np
import pytest

from helpers import small_config
from voxelhashing_amd import canonical, synth, vhtypes as T


def sphere_frame_scene(O, width, height, params, num_buckets, num_sdf_blocks, gc=False):
    hp = T.make_hash_params(num_buckets, num_sdf_blocks, **synth.PARAM_SETS[params])
    cp = T.make_depth_camera_params(width, height)
    sc = O.OracleScene(hp, cp, options=T.make_scene_options(offline=True, gc=gc))
    pose = synth.orbit_pose(0)
    depth, color = O.synth_frame(synth.SPHERE_A, 0, pose, cp)
    return sc, pose, depth, color


def test_cfg1_sphere_frame_block_count_and_gc_churn(oracle_lib):
    """640x480, 4 cm, 2^18 buckets: 152 blocks, max bucket fill 2; GC flags 38 of them,
    frees 38, and the next frame's alloc re-allocates 38"""
    O = oracle_lib
    sc, pose, depth, color = sphere_frame_scene(O, 640, 480, "P4", 1 << 18, 1 << 12)
    sc.integrate(pose, depth, color)
    s = sc.state()
    assert s["num_occupied"] == 152
    assert sc.hp.m_numOccupiedBlocks == 152
    assert s["bucket_counts"].max() == 2
    canonical.check_invariants(sc.hash_table(), sc.heap(), sc.array("d_heapCounter", np.uint32, 1)[0], sc.hp, sc.sdf_blocks())
    sc.gc_identify()
    assert int(sc.decisions().sum()) == 38
    sc.reset_mutex()
    sc.gc_free()
    assert sc.state()["num_occupied"] == 152 - 38
    sc.integrate(pose, depth, color)
    assert sc.state()["num_occupied"] == 152
    canonical.check_invariants(sc.hash_table(), sc.heap(), sc.array("d_heapCounter", np.uint32, 1)[0], sc.hp, sc.sdf_blocks())


def test_5m_entry_table_same_block_count(oracle_lib):
    """500 k buckets (5 M entries): still 152 blocks, max fill 2"""
    sc, pose, depth, color = sphere_frame_scene(oracle_lib, 640, 480, "P4", 500000, 1 << 12)
    sc.integrate(pose, depth, color)
    s = sc.state()
    assert s["num_occupied"] == 152 and s["bucket_counts"].max() == 2


def test_1cm_sphere_frame_block_count(oracle_lib):
    """640x480, 1 cm, 2 M buckets: 1 697 blocks"""
    sc, pose, depth, color = sphere_frame_scene(oracle_lib, 640, 480, "P1", 2000000, 1 << 12)
    sc.integrate(pose, depth, color)
    s = sc.state()
    assert s["num_occupied"] == 1697 and s["bucket_counts"].max() == 2


@pytest.mark.slow
def test_1080p_2cm_block_count(oracle_lib):
    """1920x1080, 2 cm, 500 k buckets: 360 blocks"""
    sc, pose, depth, color = sphere_frame_scene(oracle_lib, 1920, 1080, "P2", 500000, 1 << 12)
    sc.integrate(pose, depth, color)
    assert sc.state()["num_occupied"] == 360


def test_plane_160x120_reference_default_truncation(oracle_lib):
    """fronto-parallel plane at 2 m, 160x120, 4 cm voxels with the reference's default truncation
    (0.02 + 0.01*z): one layer of 8 x 6 blocks = 48.  The voxel and
    hit counts quoted there depend on parameters the survey does not state, so only the block count is used."""
    O = oracle_lib
    hp = T.make_hash_params(1 << 14, 1 << 12, voxel_size=0.04, truncation=0.02, trunc_scale=0.01)
    cp = T.make_depth_camera_params(160, 120)
    sc = O.OracleScene(hp, cp, options=T.make_scene_options(offline=True, gc=False))
    depth = np.full((120, 160), 2.0, np.float32)
    color = np.full((120, 160, 4), 0.5, np.float32)
    eye = np.eye(4, dtype=np.float32)
    sc.integrate(eye, depth, color)
    s = sc.state()
    assert s["num_occupied"] == 48
    r = sc.render(eye)
    hits = r["depth"] != -np.inf
    assert hits.sum() > 17000
    assert abs(float(r["depth"][hits].mean(dtype=np.float64)) - 2.0) < 1e-5


def test_all_core_baseline_build_equals_the_checker(oracle_lib):
    """libvh_oracle_omp.so (bench.py's all-core CPU baseline; the same source with OpenMP) leaves the same hash
    table, heap, voxels and ray-cast maps as the serial checker, byte for byte -- including the slot and heap-pointer
    assignment, because its alloc pass replays the serial order"""
    O = oracle_lib
    assert O.lib().vho_num_threads() == 1
    hp, cp, rp = small_config(96, 72, "P2", 1 << 12, 1 << 12)
    opt = T.make_scene_options(offline=True, gc=True, starve=2)
    a, b = O.OracleScene(hp, cp, rp, opt), O.OracleScene(hp, cp, rp, opt, omp=True)
    last = None
    for k in range(5):
        pose = synth.orbit_pose(k, 60)
        depth, color = O.synth_frame(synth.S1_SPHERES, 0, pose, cp)
        if last is not None:
            ra, rb = a.render(last), b.render(last)
            for m in ra:
                assert ra[m].tobytes() == rb[m].tobytes(), (k, m)
        a.integrate(pose, depth, color)
        b.integrate(pose, depth, color)
        assert a.hp.m_numOccupiedBlocks == b.hp.m_numOccupiedBlocks > 50
        assert a.hash_table().tobytes() == b.hash_table().tobytes(), k
        assert a.sdf_blocks().tobytes() == b.sdf_blocks().tobytes(), k
        assert a.heap().tobytes() == b.heap().tobytes(), k
        n = a.hp.m_numOccupiedBlocks
        assert a.array("d_hashCompactified", T.HASH_ENTRY_DTYPE, n).tobytes() == b.array("d_hashCompactified", T.HASH_ENTRY_DTYPE, n).tobytes()
        last = pose
