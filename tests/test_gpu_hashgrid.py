"""The .hashgrid file (CUDASceneRepChunkGrid::saveToFile / loadFromFile, DSC/CUDASceneRepChunkGrid.h:458-548) against
an independent restatement of its byte layout, written here with struct from the reference's source:

    u32 version (1) | f32 voxel size | 3 f32 voxel extents | 3 i32 grid dimensions | 3 i32 min grid pos |
    3 i32 max grid pos | u32 initial chunk list size | u32 number of occupied chunks, then per chunk
    u32 chunk index | u64 n | n x SDFBlock (512 x {f32 sdf, u8 r, g, b, u8 weight}) | u64 n | n x {3 i32 pos, i32 ptr}

(:470-489: the header; operator<< of ChunkDesc :124-129: blocks first, then descriptors; a std::vector is a UINT64
count followed by the elements, mLib binaryDataStream.h:156-163).  Plus: files that lie about their counts are
refused, and blocks a stream-in pass cannot insert are neither lost nor leaked."""
import struct

import numpy as np
import pytest

from helpers import small_config
from voxelhashing_amd import canonical, synth, vhtypes as T

pytestmark = pytest.mark.gpu

EXT, DIMS, MINP, LIST = (0.5, 0.5, 0.5), (65, 65, 65), (-32, -32, -32), 16
MAXP = tuple(m + d for m, d in zip(MINP, DIMS))


@pytest.fixture(scope="module")
def E(vh):
    from voxelhashing_amd import engine
    return engine


def header(voxel, ext=EXT, dims=DIMS, minp=MINP, maxp=MAXP, lst=LIST, version=1):
    return struct.pack("<If3f3i3i3iI", version, voxel, *ext, *dims, *minp, *maxp, lst)


def chunk_of(pos, voxel, ext=EXT, minp=MINP, dims=DIMS):
    """worldToChunks + linearizeChunkPos (DSC/CUDASceneRepChunkGrid.h:570-606) for a block position"""
    f = np.float32
    c = []
    for a in range(3):
        w = f(f(pos[a] * 8) * f(voxel))
        p = f(w / f(ext[a]))
        c.append(int(np.trunc(p + f(np.sign(p)) * f(0.5))))
    q = [c[a] - minp[a] for a in range(3)]
    return q[2] * dims[0] * dims[1] + q[1] * dims[0] + q[0]


def make_blocks(rng, n):
    v = np.zeros((n, 512), dtype=T.VOXEL_DTYPE)
    v["sdf"] = rng.uniform(-0.2, 0.2, (n, 512)).astype(np.float32)
    v["color"] = rng.integers(0, 256, (n, 512, 3), dtype=np.uint8)
    v["weight"] = rng.integers(1, 255, (n, 512), dtype=np.uint8)
    return v


def write_file(path, voxel, chunks, **hdr):
    """chunks: {chunk index: (positions [n,3], voxels [n,512])}"""
    with open(path, "wb") as f:
        f.write(header(voxel, **hdr))
        f.write(struct.pack("<I", len(chunks)))
        for index in sorted(chunks):
            pos, vox = chunks[index]
            f.write(struct.pack("<IQ", index, len(pos)))
            f.write(np.ascontiguousarray(vox).tobytes())
            f.write(struct.pack("<Q", len(pos)))
            for p in pos:
                f.write(struct.pack("<3ii", int(p[0]), int(p[1]), int(p[2]), 0))


def parse_file(path):
    raw = open(path, "rb").read()
    at = 0

    def take(fmt):
        nonlocal at
        v = struct.unpack_from("<" + fmt, raw, at)
        at += struct.calcsize("<" + fmt)
        return v

    version, voxel = take("If")
    ext, dims, minp, maxp = take("3f"), take("3i"), take("3i"), take("3i")
    lst, n_chunks = take("II")
    chunks = {}
    for _ in range(n_chunks):
        index, nb = take("IQ")
        vox = np.frombuffer(raw, dtype=T.VOXEL_DTYPE, count=nb * 512, offset=at).reshape(nb, 512)
        at += nb * 4096
        (nd,) = take("Q")
        desc = np.frombuffer(raw, dtype=T.DESC_DTYPE, count=nd, offset=at)
        at += nd * 16
        chunks[index] = (desc["pos"].copy(), vox.copy())
    assert at == len(raw), "bytes left over"
    return dict(version=version, voxel=voxel, ext=ext, dims=dims, minp=minp, maxp=maxp, list=lst, chunks=chunks)


def scene_and_grid(E, threaded=False, **cfg):
    hp, cp, rp = small_config(64, 48, streaming_extents=EXT, streaming_dims=DIMS, streaming_min=MINP, **cfg)
    # (with the worker thread the scene must be in online mode: in offline mode the worker prepares no stream-in pass)
    scene = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=not threaded, gc=False, streaming_out_parts=4))
    grid = E.CUDASceneRepChunkGrid(scene, EXT, DIMS, MINP, LIST, threaded, 4)
    return hp, cp, scene, grid


CENTRE, BIG = np.zeros(3, np.float32), 1000.0


def test_a_file_written_by_hand_loads_into_the_scene(E, tmp_path):
    rng = np.random.default_rng(11)
    hp, cp, scene, grid = scene_and_grid(E)
    voxel = hp.m_virtualVoxelSize
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [-1, -1, -1], [3, 2, 1], [3, 2, 2], [-4, 5, -6], [7, 7, 7]], dtype=np.int32)
    vox = make_blocks(rng, len(pos))
    chunks = {}
    for p, v in zip(pos, vox):
        chunks.setdefault(chunk_of(p, voxel), ([], []))
        chunks[chunk_of(p, voxel)][0].append(p)
        chunks[chunk_of(p, voxel)][1].append(v)
    chunks = {k: (np.array(a), np.array(b)) for k, (a, b) in chunks.items()}
    assert len(chunks) >= 3
    path = str(tmp_path / "hand.hashgrid")
    write_file(path, voxel, chunks)
    grid.loadFromFile(path, CENTRE, BIG)
    st = grid.getStatistics()
    assert st["blocks"] == len(pos) and st["chunks"] == len(chunks) and st["bits"] == len(chunks)
    assert grid.streamInToGPUAll(CENTRE, BIG, True) == len(pos)
    s = scene.state()  # (runs the invariants: heap and table partition the pool, free blocks are zero)
    order = canonical.lexsort_pos(pos)
    assert np.array_equal(s["positions"], pos[order])
    assert s["voxels"].tobytes() == np.ascontiguousarray(vox[order]).tobytes()


def test_a_file_the_library_wrote_parses_with_the_layout(E, tmp_path):
    hp, cp, scene, grid = scene_and_grid(E)
    frame = E.DepthFrame(cp)
    for k in range(3):
        pose = synth.orbit_pose(k, 50)
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        scene.integrate(pose, frame, cp, grid.getBitMaskGPU())
    before = scene.state()
    path = str(tmp_path / "lib.hashgrid")
    grid.saveToFile(path, CENTRE, BIG)
    f = parse_file(path)
    assert f["version"] == 1 and f["voxel"] == np.float32(hp.m_virtualVoxelSize)
    assert f["ext"] == tuple(np.float32(e) for e in EXT) and f["dims"] == DIMS and f["minp"] == MINP and f["maxp"] == MAXP and f["list"] == LIST
    pos = np.concatenate([c[0] for c in f["chunks"].values()])
    vox = np.concatenate([c[1] for c in f["chunks"].values()])
    for index, (p, _) in f["chunks"].items():
        assert all(chunk_of(q, f["voxel"]) == index for q in p), "a block sits in the wrong chunk"
    order = canonical.lexsort_pos(pos)
    assert np.array_equal(pos[order], before["positions"])
    assert np.ascontiguousarray(vox[order]).tobytes() == before["voxels"].tobytes()
    canonical.assert_same_scene(before, scene.state(), "saveToFile streams everything back in")


@pytest.mark.parametrize("what", ["fewer descriptors than blocks", "truncated", "count beyond the file", "chunk twice", "chunk index out of range"])
def test_files_that_lie_are_refused_and_the_worker_survives(E, tmp_path, what):
    rng = np.random.default_rng(5)
    hp, cp, scene, grid = scene_and_grid(E, threaded=True)
    voxel = hp.m_virtualVoxelSize
    pos = np.array([[0, 0, 0], [1, 0, 0]], dtype=np.int32)
    vox = make_blocks(rng, 2)
    index = chunk_of(pos[0], voxel)  # (every one of these files is refused before a block's chunk would matter)
    path = str(tmp_path / "bad.hashgrid")
    body = struct.pack("<IQ", index, 2) + vox.tobytes()
    descs = b"".join(struct.pack("<3ii", *map(int, p), 0) for p in pos)
    if what == "fewer descriptors than blocks":
        raw = header(voxel) + struct.pack("<I", 1) + body + struct.pack("<Q", 1) + descs[:16]
    elif what == "truncated":
        raw = (header(voxel) + struct.pack("<I", 1) + body + struct.pack("<Q", 2) + descs)[:-7]
    elif what == "count beyond the file":
        raw = header(voxel) + struct.pack("<I", 1) + struct.pack("<IQ", index, 1 << 40) + vox.tobytes()
    elif what == "chunk twice":
        one = body + struct.pack("<Q", 2) + descs
        raw = header(voxel) + struct.pack("<I", 2) + one + one
    else:
        raw = header(voxel) + struct.pack("<I", 1) + struct.pack("<IQ", DIMS[0] * DIMS[1] * DIMS[2], 0) + struct.pack("<Q", 0)
    open(path, "wb").write(raw)
    with pytest.raises(Exception):
        grid.loadFromFile(path, CENTRE, BIG)
    assert grid.getStatistics()["blocks"] == 0, "a refused file must not leave half a grid behind"
    # the worker thread is running again: the two-thread protocol still turns over
    frame = E.DepthFrame(cp)
    for k in range(4):
        pose = synth.orbit_pose(k, 40)
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        grid.streamOutToCPUPass0GPU(np.zeros(3, np.float32), 1.0, True, True)
        grid.streamInToGPUPass1GPU(True)
        scene.integrate(pose, frame, cp, grid.getBitMaskGPU())
    grid.reset()
    assert scene.debugHash()["duplicates"] == 0
    grid.close()


def bucket_of(p, nb):
    M = 0xFFFFFFFF
    x, y, z = [int(v) & M for v in p]
    return ((((x * 73856093) & M) ^ ((y * 19349669) & M) ^ ((z * 83492791) & M)) & M) % nb


def test_blocks_that_find_no_slot_go_back_to_the_host_grid(E, tmp_path):
    """fourteen blocks of one bucket in one chunk: ten fill the bucket, the eleventh opens its list, and every further
    one of the same pass finds the bucket taken (the reference's overflow branch is an unported remnant,
    DSC/VoxelUtilHashSDF.h:682-713).  They return to the host grid with their SDF blocks back on the heap, and
    streamInToGPUAll's next passes bring them in: nothing lost, nothing leaked."""
    rng = np.random.default_rng(3)
    ext, dims, minp = (4.0, 4.0, 4.0), (9, 9, 9), (-4, -4, -4)
    hp, cp, rp = small_config(64, 48, num_buckets=7, num_sdf_blocks=64, streaming_extents=ext, streaming_dims=dims, streaming_min=minp)
    scene = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=True, gc=False, streaming_out_parts=1))
    grid = E.CUDASceneRepChunkGrid(scene, ext, dims, minp, LIST, False, 1)
    voxel = hp.m_virtualVoxelSize
    cand = [(x, y, z) for x in range(-5, 6) for y in range(-5, 6) for z in range(-5, 6)]
    same = [p for p in cand if bucket_of(p, 7) == 3 and chunk_of(p, voxel, ext, minp, dims) == chunk_of((0, 0, 0), voxel, ext, minp, dims)][:14]
    assert len(same) == 14
    pos = np.array(same, dtype=np.int32)
    vox = make_blocks(rng, len(pos))
    path = str(tmp_path / "crowded.hashgrid")
    maxp = tuple(m + d for m, d in zip(minp, dims))
    write_file(path, voxel, {chunk_of((0, 0, 0), voxel, ext, minp, dims): (pos, vox)}, ext=ext, dims=dims, minp=minp, maxp=maxp)
    grid.loadFromFile(path, CENTRE, BIG)
    total = grid.streamInToGPUAll(CENTRE, BIG, True)
    assert total == 14
    assert grid.getNumFailedInserts() > 0, "the pass was meant to overflow one bucket twice"
    assert grid.getStatistics()["blocks"] == 0
    s = scene.state()  # invariants: no block both free and used, none lost, free ones zero
    order = canonical.lexsort_pos(pos)
    assert np.array_equal(s["positions"], pos[order])
    assert s["voxels"].tobytes() == np.ascontiguousarray(vox[order]).tobytes()
    assert s["heap_free"] == 64 - 14
    assert scene.debugHash()["duplicates"] == 0


@pytest.mark.timeout(300)
def test_blocks_that_find_no_slot_in_a_pipelined_frame_go_back_too(E, tmp_path):
    """The same crowded chunk, brought in by the native loop's pipelined streaming step (the grid's worker chooses and uploads
    the chunk a frame ahead, the insert runs without the host reading anything back): the pass reports the blocks that found
    no slot through mapped memory, the loop repairs them a frame later (back into the host grid, their SDF blocks back on the
    heap, the bit mask set right again), and later frames bring them in.  Nothing lost, nothing leaked, and the loop never
    waited for the device in the frames in between."""
    rng = np.random.default_rng(7)
    ext, dims, minp = (4.0, 4.0, 4.0), (9, 9, 9), (-4, -4, -4)
    hp, cp, rp = small_config(64, 48, num_buckets=7, num_sdf_blocks=64, streaming_extents=ext, streaming_dims=dims, streaming_min=minp)
    scene = E.CUDASceneRepHashSDF(hp, T.make_scene_options(offline=False, gc=False, streaming_out_parts=1))
    ray = E.CUDARayCastSDF(rp)
    grid = E.CUDASceneRepChunkGrid(scene, ext, dims, minp, LIST, True, 1)  # worker thread running
    voxel = hp.m_virtualVoxelSize
    cand = [(x, y, z) for x in range(-5, 6) for y in range(-5, 6) for z in range(-5, 6)]
    same = [p for p in cand if bucket_of(p, 7) == 3 and chunk_of(p, voxel, ext, minp, dims) == chunk_of((0, 0, 0), voxel, ext, minp, dims)][:14]
    assert len(same) == 14
    pos = np.array(same, dtype=np.int32)
    vox = make_blocks(rng, len(pos))
    path = str(tmp_path / "crowded.hashgrid")
    maxp = tuple(m + d for m, d in zip(minp, dims))
    write_file(path, voxel, {chunk_of((0, 0, 0), voxel, ext, minp, dims): (pos, vox)}, ext=ext, dims=dims, minp=minp, maxp=maxp)
    grid.loadFromFile(path, np.array([100.0, 0.0, 0.0], np.float32), 1.0)  # (a sphere far away: nothing comes in yet)
    assert grid.getStatistics()["blocks"] == 14
    # frames that see nothing (no depth): alloc adds no block, the table holds what streaming brings
    W, H = cp.m_imageWidth, cp.m_imageHeight
    frame = E.DepthFrame(cp, depth=np.full((H, W), -np.inf, np.float32), color=np.full((H, W, 4), -np.inf, np.float32))
    n = 12

    def pose_at(x):
        m = np.eye(4, dtype=np.float32)
        m[0, 3] = x
        return m.reshape(16)

    poses = [pose_at(100.0)] + [pose_at(0.01 * k) for k in range(1, n)]  # the first sphere holds no chunk, the others the crowded one
    recon = E.Reconstruction(scene, ray, grid, cp, E.Reconstruction.defaultOptions(s_streamingEnabled=1, s_streamingPos=(0.0, 0.0, 0.0), s_streamingRadius=10.0,
                                                                                  s_allocAhead=1, s_maxFramesInFlight=4))
    seq = E.Reconstruction.makeFrames(poses, [frame.depth_ptr] * n, [frame.color_ptr] * n)
    recon.run(seq)
    recon.synchronize()
    st = recon.getStats()
    assert st["frames"] == n and st["streamingFramesPipelined"] == n - 1, st
    assert grid.getNumFailedInserts() > 0, "the pipelined pass was meant to overflow one bucket twice"
    assert grid.getStatistics()["blocks"] == 0, "every block must have come in by now"
    assert st["blocksStreamedIn"] == 14 and st["blocksStreamedOut"] == 0, st
    s = scene.state()  # invariants: no block both free and used, none lost, free ones zero
    order = canonical.lexsort_pos(pos)
    assert np.array_equal(s["positions"], pos[order])
    assert s["voxels"].tobytes() == np.ascontiguousarray(vox[order]).tobytes()
    assert s["heap_free"] == 64 - 14
    assert scene.debugHash()["duplicates"] == 0
    grid.debugCheckForDuplicates()
    recon.close()
    grid.close()
