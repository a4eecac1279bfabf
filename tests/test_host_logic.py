"""Host-side logic that needs no GPU: trajectories, parameter builders, the
canonical forms and the debugHash-style invariant checker (including that it
really detects corruption)."""
import numpy as np
import pytest

from voxelhashing_amd import canonical, synth, vhtypes as T


def test_orbit_poses_are_rigid_and_look_at_the_origin():
    for k in (0, 1, 250, 500, 999):
        m = synth.orbit_pose(k).reshape(4, 4).astype(np.float64)
        R, c = m[:3, :3], m[:3, 3]
        assert np.allclose(R.T @ R, np.eye(3), atol=1e-6) and abs(np.linalg.det(R) - 1) < 1e-6
        assert abs(np.linalg.norm(c) - 2.5) < 1e-6
        fwd = R[:, 2]
        assert np.allclose(fwd, -c / np.linalg.norm(c), atol=1e-6)  # camera looks at the origin
    assert np.array_equal(synth.orbit_pose(0).reshape(4, 4)[:3, :3], np.eye(3, dtype=np.float32))
    assert not np.array_equal(synth.orbit_pose(5), synth.orbit_pose(5, phase=0.7))


def test_parameter_builders_follow_the_reference_rules():
    hp, cp, rp = synth.config_params("cfg2")
    assert (hp.m_hashNumBuckets, hp.m_hashBucketSize, hp.m_numSDFBlocks) == (500000, 10, 1000000)
    assert np.float32(hp.m_virtualVoxelSize) == np.float32(0.04) and np.float32(hp.m_truncation) == np.float32(0.2)
    assert (cp.m_imageWidth, cp.m_imageHeight) == (640, 480) and cp.fx == 525.0 and cp.mx == 319.5
    inc = np.float32(0.8) * np.float32(0.2)
    assert np.float32(rp.m_rayIncrement) == inc
    assert np.float32(rp.m_thresSampleDist) == np.float32(50.5) * inc and np.float32(rp.m_thresDist) == np.float32(50.0) * inc
    assert rp.m_maxNumVertices == hp.m_numSDFBlocks * 6 and rp.m_useGradients == 0
    pos, radius = synth.streaming_sphere(hp, cp)
    # DepthSensing.cpp:1340-1355: centre (0,0,dmin + (maxInt-dmin)/2), radius = frustum radius + chunk radius
    assert np.allclose(pos, [0, 0, 0.5 + 0.5 * 3.5]) and abs(radius - (0.5 * 3.5 * 3 ** 0.5 + 0.5 * 3 ** 0.5)) < 1e-5


def tiny_scene(oracle_lib):
    hp = T.make_hash_params(1 << 10, 1 << 9, **synth.PARAM_SETS["P4"])
    cp = T.make_depth_camera_params(64, 48)
    sc = oracle_lib.OracleScene(hp, cp, options=T.make_scene_options(offline=True, gc=True))
    pose = synth.orbit_pose(0)
    d, c = oracle_lib.synth_frame(synth.S1_SPHERES, 0, pose, cp)
    sc.integrate(pose, d, c)
    return sc


def test_invariant_checker_detects_corruption(oracle_lib):
    sc = tiny_scene(oracle_lib)
    hc = int(sc.array("d_heapCounter", np.uint32, 1)[0])
    table, heap, blocks = sc.hash_table().copy(), sc.heap().copy(), sc.sdf_blocks().copy()
    rep = canonical.check_invariants(table, heap, hc, sc.hp, blocks)
    assert rep["num_occupied"] > 20 and rep["num_occupied"] + rep["heap_free"] == sc.hp.m_numSDFBlocks
    occ = np.nonzero(table["ptr"] != T.FREE_ENTRY)[0]
    # duplicate free pointer
    h2 = heap.copy(); h2[0] = h2[1]
    with pytest.raises(AssertionError, match="duplicate free"):
        canonical.check_invariants(table, h2, hc, sc.hp)
    # entry freed without returning its block: leak
    t2 = table.copy(); t2[occ[0]]["ptr"] = T.FREE_ENTRY; t2[occ[0]]["pos"] = 0
    with pytest.raises(AssertionError, match="neither free nor allocated"):
        canonical.check_invariants(t2, heap, hc, sc.hp)
    # block both free and allocated
    h3 = heap.copy(); h3[hc] = table[occ[0]]["ptr"] // 512
    with pytest.raises(AssertionError):
        canonical.check_invariants(table, h3, hc, sc.hp)
    # duplicate position
    t3 = table.copy(); t3[occ[1]]["pos"] = t3[occ[0]]["pos"]
    with pytest.raises(AssertionError, match="duplicate block positions"):
        canonical.check_invariants(t3, heap, hc, sc.hp)
    # free block not cleared
    b2 = blocks.copy(); b2[int(heap[0]) * 512 + 3]["weight"] = 1
    with pytest.raises(AssertionError, match="not cleared"):
        canonical.check_invariants(table, heap, hc, sc.hp, b2)


def test_canonical_snapshot_is_order_independent(oracle_lib):
    sc = tiny_scene(oracle_lib)
    hc = int(sc.array("d_heapCounter", np.uint32, 1)[0])
    a = canonical.snapshot(sc.hash_table(), sc.sdf_blocks(), sc.heap(), hc, sc.hp)
    # move every occupied entry to another free slot of its bucket and give it another SDF block:
    # the canonical form must not change
    table, blocks, heap = sc.hash_table().copy(), sc.sdf_blocks().copy(), sc.heap().copy()
    occ = np.nonzero(table["ptr"] != T.FREE_ENTRY)[0]
    ptrs = table["ptr"][occ].copy()
    perm = np.roll(np.arange(len(occ)), 1)
    vox = blocks.reshape(-1, 512)
    newvox = vox.copy()
    for i, j in zip(range(len(occ)), perm):
        newvox[ptrs[j] // 512] = vox[ptrs[i] // 512]
        table[occ[i]]["ptr"] = ptrs[j]
    b = canonical.snapshot(table, newvox.reshape(-1), heap, hc, sc.hp)
    canonical.assert_same_scene(a, b, "permuted")
    # and a changed voxel must be noticed
    newvox[ptrs[0] // 512][7]["sdf"] += np.float32(1e-3)
    c = canonical.snapshot(table, newvox.reshape(-1), heap, hc, sc.hp)
    with pytest.raises(AssertionError):
        canonical.assert_same_scene(a, c, "changed")


def test_committed_bench_line_has_the_contract_fields():
    """the last bench line kept under profiles/ (written by bench.py on the GPU box) carries every field the driver's
    contract names, with the types and relations it states"""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r01_g_bench_line.json")
    d = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-3  # one frame per step, one GPU
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 0.5
    assert abs(r["avg_launch_us"] - (r["event_pair_us"] - r["event_pair_overhead_us"])) < 1e-2
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "frames/s" and c["value"] > c["single_thread"] > 0
