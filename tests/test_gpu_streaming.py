"""World-chunk streaming (CUDASceneRepChunkGrid): blocks outside a sphere
around the camera move to the host chunk grid, chunks entirely inside come
back; alloc is suppressed for streamed-out chunks through the bit mask.
GPU engine vs the oracle twin (oracle/chunk_grid.py), single-threaded passes
for determinism, plus a smoke run of the worker-thread protocol."""
import os
import tempfile

import numpy as np
import pytest

from helpers import small_config
from voxelhashing_amd import canonical, synth, vhtypes as T

pytestmark = pytest.mark.gpu

EXT, DIMS, MINP, PARTS = (0.5, 0.5, 0.5), (65, 65, 65), (-32, -32, -32), 4
STREAM_POS = np.array([0.0, 0.0, 1.6, 1.0], dtype=np.float32)
RADIUS = 1.2


def sorted_blocks(descs, blocks):
    order = canonical.lexsort_pos(np.ascontiguousarray(descs["pos"]))
    return np.ascontiguousarray(descs["pos"][order]), np.ascontiguousarray(blocks[order])


def make_pair(E, O, offline=True):
    from oracle.chunk_grid import OracleChunkGrid
    hp, cp, rp = small_config(64, 48, streaming_extents=EXT, streaming_dims=DIMS, streaming_min=MINP)
    opt = T.make_scene_options(offline=offline, gc=True, starve=15, streaming_out_parts=PARTS)
    gs = E.CUDASceneRepHashSDF(hp, opt)
    gg = E.CUDASceneRepChunkGrid(gs, EXT, DIMS, MINP, 16, False, PARTS)
    os_ = O.OracleScene(hp, cp, rp, opt)
    og = OracleChunkGrid(os_, EXT, DIMS, MINP, PARTS)
    return hp, cp, rp, gs, gg, os_, og


def compare(gs, gg, os_, og, what):
    canonical.assert_same_scene(gs.state(), os_.state(), what)
    gd, gb = sorted_blocks(*gg.downloadHostBlocks())
    od, ob = sorted_blocks(*og.host_blocks())
    assert np.array_equal(gd, od), f"{what}: host chunk grid holds different blocks"
    assert gb.tobytes() == ob.tobytes(), f"{what}: host voxel payloads differ"
    st = gg.getStatistics()
    assert st["blocks"] == len(od) and st["bits"] == og.statistics()["bits"]
    gg.debugCheckForDuplicates()


def test_stream_out_in_sequence(vh, oracle_lib):
    from voxelhashing_amd import engine as E
    O = oracle_lib
    hp, cp, rp, gs, gg, os_, og = make_pair(E, O)
    frame = E.DepthFrame(cp)
    moved_out = moved_in = 0
    for k in range(24):
        pose = synth.orbit_pose(k, 40)
        p = (pose.reshape(4, 4) @ STREAM_POS)[:3]  # DepthSensing.cpp:882-883
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        depth, color = O.synth_frame(synth.S1_SPHERES, 0, pose, cp)
        n_out, n_in = gg.streamOutToCPU(p, RADIUS, True), gg.streamInToGPU(p, RADIUS, True)
        assert (n_out, n_in) == (og.stream_out_to_cpu(p, RADIUS, True), og.stream_in_to_gpu(p, RADIUS, True)), f"frame {k}"
        moved_out += n_out
        moved_in += n_in
        mask = gg.getBitMaskGPU()
        gs.integrate(pose, frame, cp, mask)
        os_.integrate(pose, depth, color, og.bitmask)
        compare(gs, gg, os_, og, f"frame {k}")
    assert moved_out > 20 and moved_in > 5, (moved_out, moved_in)
    st = gs.getState()
    assert st[T.STATE_INSERT_FAILED] == 0 and st[T.STATE_HEAP_UNDERFLOW] == 0


def test_stream_everything_out_and_back_and_hashgrid_file(vh, oracle_lib):
    """streamOutToCPUAll empties the GPU hash; saveToFile/loadFromFile round-trip the .hashgrid;
    streaming everything back restores the scene bit for bit"""
    from voxelhashing_amd import engine as E
    O = oracle_lib
    hp, cp, rp, gs, gg, os_, og = make_pair(E, O)
    frame = E.DepthFrame(cp)
    for k in range(3):
        pose = synth.orbit_pose(k, 40)
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        gs.integrate(pose, frame, cp, gg.getBitMaskGPU())
    before = gs.state()
    assert before["num_occupied"] > 30
    gg.streamOutToCPUAll()
    empty = gs.state()
    assert empty["num_occupied"] == 0 and empty["heap_free"] == hp.m_numSDFBlocks
    assert gg.getStatistics()["blocks"] == before["num_occupied"]
    hd, hb = sorted_blocks(*gg.downloadHostBlocks())
    assert np.array_equal(hd, before["positions"]) and hb.tobytes() == before["voxels"].tobytes()

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scene.hashgrid")
        centre, big = np.zeros(3, np.float32), 1000.0
        gg.saveToFile(path, centre, big)  # streams the sphere around `centre` back in afterwards
        size = os.path.getsize(path)
        # header 4+4+12+12+12+12+4 + count 4, then per chunk 4 + 8 + n*4096 + 8 + n*16 (mLib BinaryDataStreamFile)
        n_blocks = before["num_occupied"]
        assert (size - 64 - n_blocks * (4096 + 16)) % 20 == 0
        after_save = gs.state()
        canonical.assert_same_scene(before, after_save, "after saveToFile")
        gg.loadFromFile(path, centre, big)  # streams everything out, then reads the file into the host grid
        assert gs.state()["num_occupied"] == 0 and gg.getStatistics()["blocks"] == n_blocks
        n = gg.streamInToGPUAll(centre, big, True)
        assert n == n_blocks
        canonical.assert_same_scene(before, gs.state(), "after loadFromFile + streamInToGPUAll")
        # a wrong version is refused
        raw = bytearray(open(path, "rb").read())
        raw[0] = 9
        open(path, "wb").write(raw)
        with pytest.raises(Exception, match="version"):
            gg.loadFromFile(path, centre, big)


def test_worker_thread_protocol_smoke(vh, oracle_lib):
    """the two-thread producer/consumer hand-off (streamOutToCPUPass0GPU / streamInToGPUPass1GPU on the caller,
    pass 1 / pass 0 on the worker): no deadlock, no block lost or duplicated"""
    from voxelhashing_amd import engine as E
    hp, cp, rp = small_config(64, 48, streaming_extents=EXT, streaming_dims=DIMS, streaming_min=MINP)
    opt = T.make_scene_options(offline=False, gc=False, streaming_out_parts=PARTS)
    gs = E.CUDASceneRepHashSDF(hp, opt)
    gg = E.CUDASceneRepChunkGrid(gs, EXT, DIMS, MINP, 16, True, PARTS)  # worker thread running
    frame = E.DepthFrame(cp)
    for k in range(30):
        pose = synth.orbit_pose(k, 40)
        p = (pose.reshape(4, 4) @ STREAM_POS)[:3]
        E.synth_frame(synth.S1_SPHERES, 0, pose, cp, out=frame)
        gg.streamOutToCPUPass0GPU(p, RADIUS, True, True)
        gg.streamInToGPUPass1GPU(True)
        gs.integrate(pose, frame, cp, gg.getBitMaskGPU())
    gg.reset()  # stops the worker, clears the grid, restarts it
    gg.close()
    s = gs.state(with_voxels=False)
    assert s["num_occupied"] > 10
    assert gs.debugHash()["duplicates"] == 0
