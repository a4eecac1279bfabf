"""Launcher-level parity (one vh_* call per reference launcher) and the
collision-list paths of the hash table, executed serially on the GPU so that
slot order, pointers and heap order must equal the serial oracle EXACTLY."""
import numpy as np
import pytest

from helpers import bits, small_config
from voxelhashing_amd import canonical, synth, vhtypes as T

pytestmark = pytest.mark.gpu

OP_ALLOC, OP_DELETE, OP_INSERT, OP_LOOKUP, OP_NEW_PASS = 0, 1, 2, 3, 4


def hash_bucket(p, nb):
    h = ((int(p[0]) * 73856093) & 0xFFFFFFFF) ^ ((int(p[1]) * 19349669) & 0xFFFFFFFF) ^ ((int(p[2]) * 83492791) & 0xFFFFFFFF)
    return h % nb


def positions_in_bucket(bucket, nb, count, rng):
    out = []
    while len(out) < count:
        p = rng.integers(-40, 40, 3)
        if hash_bucket(p, nb) == bucket and not any((p == q).all() for q in out):
            out.append(p)
    return out


def test_collision_lists_match_oracle_slot_by_slot(vh, oracle_lib):
    """7 buckets, 16 blocks forced into bucket 3 (overflow into a collision list), other buckets filled so that the
    linear probe has to skip last slots; then lookups, deletes of in-bucket / list-head / list elements, re-allocs and
    stream-in style inserts.  Every op is its own lock pass."""
    from voxelhashing_amd import engine as E
    rng = np.random.default_rng(5)
    nb = 7
    hp = T.make_hash_params(nb, 64, **synth.PARAM_SETS["P4"])
    g = E.LauncherScene(hp)
    o = oracle_lib.OracleScene(hp, T.make_depth_camera_params(8, 8))
    a = positions_in_bucket(3, nb, 16, rng)
    b = positions_in_bucket(4, nb, 8, rng)
    ops = []
    for p in a[:12] + b[:8] + a[12:]:
        ops += [(OP_ALLOC, *p, 0), (OP_NEW_PASS, 0, 0, 0, 0)]
    for p in a + b + [np.array([99, 99, 99])]:
        ops.append((OP_LOOKUP, *p, 0))
    for p in [a[9], a[13], a[0], a[11], a[15], b[2], a[9]]:  # slot 9 with offset, list nodes, plain slots, a miss
        ops += [(OP_DELETE, *p, 0), (OP_NEW_PASS, 0, 0, 0, 0)]
    for p in a + b:
        ops.append((OP_LOOKUP, *p, 0))
    for p in [a[9], a[0], a[13]]:
        ops += [(OP_ALLOC, *p, 0), (OP_NEW_PASS, 0, 0, 0, 0)]
    ops = np.array(ops, dtype=np.int32)
    got = g.hash_ops(ops)

    want = np.zeros(len(ops), dtype=np.int32)
    for i, (op, x, y, z, arg) in enumerate(ops):
        if op == OP_ALLOC:
            before = o.heap_free_count()
            o.alloc_block((x, y, z))
            want[i] = 1 if o.heap_free_count() != before else got[i]  # 0 / 2 / 3 are not distinguishable on the oracle
        elif op == OP_DELETE:
            want[i] = o.delete_block((x, y, z))
        elif op == OP_LOOKUP:
            want[i] = o.get_entry((x, y, z))[1]
        elif op == OP_NEW_PASS:
            o.reset_mutex()
    assert np.array_equal(got, want)
    d = g.download()
    ot = o.hash_table()
    for f in ("pos", "ptr", "offset"):
        assert np.array_equal(d["hash"][f], ot[f]), f"hash table field {f} differs from the serial oracle"
    assert d["heap_counter"] == int(o.array("d_heapCounter", np.uint32, 1)[0])
    assert np.array_equal(d["heap"][: d["heap_counter"] + 1], o.heap()[: d["heap_counter"] + 1])
    canonical.check_invariants(d["hash"], d["heap"], d["heap_counter"], g.hp, d["sdf_blocks"])
    canonical.check_bucket_summary(d["hash"], d["bucket_count"], d["bucket_bits"], g.hp)
    assert (d["hash"]["offset"] != 0).sum() >= 2, "the scenario must leave a collision list behind"


def test_insert_hash_entry_in_bucket_and_overflow(vh, oracle_lib):
    """insertHashEntry: in-bucket CAS path as the reference; overflow path with the defined (fenced) behaviour"""
    from voxelhashing_amd import engine as E
    rng = np.random.default_rng(9)
    nb = 5
    hp = T.make_hash_params(nb, 64, **synth.PARAM_SETS["P4"])
    g = E.LauncherScene(hp)
    o = oracle_lib.OracleScene(hp, T.make_depth_camera_params(8, 8))
    a = positions_in_bucket(1, nb, 13, rng)
    ops = []
    for i, p in enumerate(a):
        ops += [(OP_INSERT, *p, 512 * (40 + i)), (OP_NEW_PASS, 0, 0, 0, 0)]
    for p in a:
        ops.append((OP_LOOKUP, *p, 0))
    ops = np.array(ops, dtype=np.int32)
    got = g.hash_ops(ops)
    want = np.zeros(len(ops), dtype=np.int32)
    for i, (op, x, y, z, arg) in enumerate(ops):
        if op == OP_INSERT:
            want[i] = o.insert_entry((x, y, z), arg)
        elif op == OP_LOOKUP:
            want[i] = o.get_entry((x, y, z))[1]
        else:
            o.reset_mutex()
    assert np.array_equal(got, want)
    d = g.download(with_voxels=False)
    for f in ("pos", "ptr", "offset"):
        assert np.array_equal(d["hash"][f], o.hash_table()[f])
    assert got[: 2 * len(a) : 2].sum() == len(a)  # all 13 found room (10 in the bucket, 3 through the list)
    canonical.check_bucket_summary(d["hash"], d["bucket_count"], d["bucket_bits"], g.hp)


def test_each_launcher_against_oracle(vh, oracle_lib):
    """reset / alloc (fixed point, reference lock protocol: LOCK_ENTRY + mutex reset) / compactify / integrate /
    starve / gc identify / gc free, one launcher at a time"""
    from voxelhashing_amd import engine as E
    O = oracle_lib
    hp, cp, rp = small_config(128, 96)
    g = E.LauncherScene(hp)
    o = O.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=True))
    frame = E.DepthFrame(cp)
    for k in (0, 7):
        pose = synth.orbit_pose(k, 100)
        inv = O.mat4_inverse(pose)
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        depth, color = O.synth_frame(synth.S1_SPHERES, 0, pose, cp)
        g.set_transform(pose, inv)
        o.set_transform(pose)
        assert np.array_equal(np.array(g.hp.m_rigidTransformInverse), np.array(o.hp.m_rigidTransformInverse))
        # alloc until the heap stops changing (CUDASceneRepHashSDF::alloc, offline branch)
        prev = -1
        while True:
            g.reset_mutex()
            g.alloc(frame, cp, None, T.LOCK_ENTRY)
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
        gs, os_ = g.state(with_voxels=False), o.state()
        assert np.array_equal(gs["positions"], os_["positions"]) and gs["heap_free"] == os_["heap_free"]
        n = g.compactify(cp)
        assert n == o.compactify()
        gs = g.state()
        assert np.array_equal(canonical.compactified_set(gs["compactified"]), canonical.compactified_set(o.compactified()))
        g.integrate(frame, cp)
        o.integrate_depth_map(depth, color)
        canonical.assert_same_scene(g.state(), o.state(), f"integrate k={k}")
        g.starve()
        o.starve()
        canonical.assert_same_scene(g.state(), o.state(), f"starve k={k}")
        g.gc_identify(cp)
        o.gc_identify()
        gs = g.state()
        dec_g = {tuple(p): int(d) for p, d in zip(gs["compactified"]["pos"], gs["decisions"])}
        dec_o = {tuple(p): int(d) for p, d in zip(o.compactified()["pos"], o.decisions())}
        assert dec_g == dec_o and sum(dec_g.values()) > 0
        g.reset_mutex()
        g.gc_free(T.LOCK_ENTRY)
        o.reset_mutex()
        o.gc_free()
        canonical.assert_same_scene(g.state(), o.state(), f"gc free k={k}")


def test_float_to_int_cliffs_on_device(vh, oracle_lib):
    """block ids of extreme / non-finite depth values: NaN, +-inf and huge depths must neither fault nor allocate"""
    from voxelhashing_amd import engine as E
    hp, cp, rp = small_config(64, 48)
    g = E.LauncherScene(hp)
    depth = np.full((48, 64), 2.0, np.float32)
    depth[0, :] = np.nan
    depth[1, :] = np.inf
    depth[2, :] = -np.inf
    depth[3, :] = 3.9999998
    depth[4, :] = 0.0
    depth[5, :] = 1e30
    depth[6, :] = -5.0
    color = np.full((48, 64, 4), 0.25, np.float32)
    frame = E.DepthFrame(cp, depth, color)
    o = oracle_lib.OracleScene(hp, cp, rp, T.make_scene_options(offline=True, gc=False))
    pose = synth.orbit_pose(0)
    g.set_transform(pose, oracle_lib.mat4_inverse(pose))
    o.set_transform(pose)
    for _ in range(3):
        g.reset_mutex()
        g.alloc(frame, cp)
        o.reset_mutex()
        o.alloc(depth, color)
    gs, os_ = g.state(with_voxels=False), o.state()
    assert np.array_equal(gs["positions"], os_["positions"]) and gs["num_occupied"] > 10
