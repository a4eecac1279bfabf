"""Canonical forms and structural invariants of a voxel-hash scene.

The reference's block->heap-slot assignment, slot order inside a bucket and
compactified order depend on thread scheduling
(DepthSensingCUDA/Source/VoxelUtilHashSDF.h:587-595), so parity is defined on
canonical forms: the sorted set of block positions, voxel payloads keyed by
position, per-bucket occupancy counts and the heap free count.

check_invariants() restates CUDASceneRepHashSDF::debugHash
(DepthSensingCUDA/Source/CUDASceneRepHashSDF.h:129-233) and
CUDASceneRepChunkGrid::debugCheckForDuplicates (CUDASceneRepChunkGrid.cpp:313-341).
"""
import numpy as np

from . import vhtypes as T


def lexsort_pos(pos):
    """order that sorts [n,3] int positions by (x, y, z)"""
    if len(pos) == 0:
        return np.zeros(0, dtype=np.int64)
    return np.lexsort((pos[:, 2], pos[:, 1], pos[:, 0]))


def block_positions(hash_table):
    """sorted positions of the allocated blocks of a hash table"""
    occ = hash_table["ptr"] != T.FREE_ENTRY
    pos = np.ascontiguousarray(hash_table["pos"][occ])
    return pos[lexsort_pos(pos)]


def snapshot(hash_table, sdf_blocks, heap, heap_counter, hp, with_voxels=True):
    occ = hash_table["ptr"] != T.FREE_ENTRY
    idx = np.nonzero(occ)[0]
    pos = hash_table["pos"][idx]
    ptr = hash_table["ptr"][idx]
    order = lexsort_pos(pos)
    pos = np.ascontiguousarray(pos[order])
    ptr = ptr[order]
    bucket_counts = np.bincount(idx // T.HASH_BUCKET_SIZE, minlength=hp.m_hashNumBuckets).astype(np.uint32)
    snap = dict(
        positions=pos,
        ptrs=ptr,
        slots=idx[order],
        bucket_counts=bucket_counts,
        heap_free=(int(heap_counter) + 1) & 0xFFFFFFFF,  # the counter is the top index: -1 (wrapped) when empty
        num_occupied=int(len(idx)),
    )
    if with_voxels:
        if len(ptr):
            vox = sdf_blocks.reshape(-1, T.SDF_BLOCK_VOXELS)[ptr // T.SDF_BLOCK_VOXELS]
        else:
            vox = np.zeros((0, T.SDF_BLOCK_VOXELS), dtype=T.VOXEL_DTYPE)
        snap["voxels"] = np.ascontiguousarray(vox)
    return snap


def check_invariants(hash_table, heap, heap_counter, hp, sdf_blocks=None):
    """debugHash: free-list has no duplicates; no block is both free and
    allocated; every block is free or allocated; no duplicate positions; no
    LOCK_ENTRY left behind.  With sdf_blocks: every free block is all-zero."""
    n_blocks = hp.m_numSDFBlocks
    n_free = (int(heap_counter) + 1) & 0xFFFFFFFF  # the counter is the top index: -1 (wrapped) when empty
    assert 0 <= n_free <= n_blocks, f"heap counter out of range: {heap_counter}"
    free_ids = heap[:n_free].astype(np.int64)
    assert free_ids.min(initial=0) >= 0 and free_ids.max(initial=0) < n_blocks
    assert len(np.unique(free_ids)) == n_free, "duplicate free pointers in heap array"
    occ = hash_table["ptr"] != T.FREE_ENTRY
    ptrs = hash_table["ptr"][occ].astype(np.int64)
    assert np.all(ptrs != T.LOCK_ENTRY), "LOCK_ENTRY left in the table"
    assert np.all(ptrs % T.SDF_BLOCK_VOXELS == 0)
    used_ids = ptrs // T.SDF_BLOCK_VOXELS
    assert len(np.unique(used_ids)) == len(used_ids), "two entries share one SDF block"
    state = np.zeros(n_blocks, dtype=np.int8)
    state[free_ids] += 1
    state[used_ids] += 2
    assert not np.any(state == 3), "ptr is on the free heap but also marked as an allocated entry"
    assert not np.any(state == 0), "memory leak: block neither free nor allocated"
    pos = hash_table["pos"][occ]
    if len(pos):
        assert len(np.unique(pos, axis=0)) == len(pos), "duplicate block positions in hash"
    # free entries are fully reset (deleteHashEntry, VoxelUtilHashSDF.h:365-369)
    free_e = hash_table[~occ]
    assert not free_e["offset"].any() and not free_e["pos"].any(), "free entry not reset"
    if sdf_blocks is not None and n_free:
        raw = sdf_blocks.view(np.uint64).reshape(n_blocks, T.SDF_BLOCK_VOXELS)
        assert not raw[free_ids].any(), "free SDF block is not cleared"
    return dict(num_occupied=int(occ.sum()), heap_free=n_free)


def check_bucket_summary(hash_table, bucket_count, bucket_bits, hp):
    """extension buffers: d_bucketCount[b] = occupied slots physically in
    bucket b; bit b of d_bucketBits = (count != 0)"""
    occ = hash_table["ptr"] != T.FREE_ENTRY
    want = np.bincount(np.nonzero(occ)[0] // T.HASH_BUCKET_SIZE, minlength=hp.m_hashNumBuckets).astype(np.uint32)
    assert np.array_equal(bucket_count, want), "d_bucketCount out of sync with d_hash"
    bits = np.unpackbits(bucket_bits.view(np.uint8), bitorder="little")[: hp.m_hashNumBuckets].astype(bool)
    assert np.array_equal(bits, want != 0), "d_bucketBits out of sync with d_bucketCount"


def assert_same_scene(a, b, what=""):
    """exact equality of two snapshots on the canonical forms"""
    assert a["num_occupied"] == b["num_occupied"], f"{what}: occupied {a['num_occupied']} != {b['num_occupied']}"
    assert np.array_equal(a["positions"], b["positions"]), f"{what}: block position sets differ"
    assert a["heap_free"] == b["heap_free"], f"{what}: heap free {a['heap_free']} != {b['heap_free']}"
    assert np.array_equal(a["bucket_counts"], b["bucket_counts"]), f"{what}: per-bucket occupancy differs"
    if "voxels" in a and "voxels" in b:
        va, vb = a["voxels"], b["voxels"]
        assert np.array_equal(va["weight"], vb["weight"]), f"{what}: voxel weights differ"
        assert np.array_equal(va["color"], vb["color"]), f"{what}: voxel colours differ"
        assert np.array_equal(va["sdf"].view(np.uint32), vb["sdf"].view(np.uint32)), f"{what}: voxel sdf bits differ"


def compactified_set(entries):
    """sorted positions of a compactified entry list (its order is arbitrary)"""
    pos = np.ascontiguousarray(entries["pos"])
    return pos[lexsort_pos(pos)]
