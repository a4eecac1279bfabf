"""zParameters*.txt -> VhAppState -> parameter structs (host logic, no GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

from voxelhashing_amd import lib, vhtypes as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def read(path):
    L = lib.load()
    gas = T.AppState()
    lib.check(L.vh_app_state_read(path.encode(), C.byref(gas)), "vh_app_state_read")
    return L, gas


def test_sample_file_and_builders():
    L, g = read(os.path.join(ROOT, "tests", "data", "zParametersSample.txt"))
    assert (g.s_sensorIdx, g.s_adapterWidth, g.s_adapterHeight) == (8, 640, 480)
    assert g.s_sensorDepthMax == f32(4.5) and g.s_sensorDepthMin == f32(0.4) and g.s_SDFVoxelSize == f32(0.01)
    assert g.s_hashNumBuckets == 2000000 and g.s_hashNumSDFBlocks == 2097152          # the later line wins
    assert g.s_depthFilter == 1 and g.s_colorFilter == 0 and g.s_SDFUseGradients == 0 and g.s_garbageCollectionEnabled == 1
    assert list(g.s_streamingVoxelExtents) == [1.0, 1.0, 0.5] and list(g.s_streamingGridDimensions) == [257, 257, 129]
    assert list(g.s_streamingMinGridPos) == [-128, -128, -64]
    assert g.s_trackingEnabled == 0 and g.s_streamingRadius == 0.0                      # absent keys are value-initialised
    assert g.numKeysFound == 33  # of the members VhAppState has; the two junk lines are not among them
    assert g.s_recordDataFile == b"Dump/test.sens" and g.s_recordData == 0 and g.s_numBinaryDumpSensorFiles == 0
    hp = T.HashParams()
    L.vh_hash_params_from_app_state(C.byref(g), C.byref(hp))
    assert hp.m_hashNumBuckets == 2000000 and hp.m_hashBucketSize == 10 and hp.m_SDFBlockSize == 8 and hp.m_numSDFBlocks == 2097152
    assert hp.m_virtualVoxelSize == f32(0.01) and hp.m_truncation == f32(0.05) and hp.m_truncScale == f32(0.025)
    assert list(hp.m_rigidTransform) == list(T.IDENTITY16) and list(hp.m_streamingGridDimensions) == [257, 257, 129]
    # the same struct as the tests' own helper builds for this configuration
    want = T.make_hash_params(2000000, 2097152, voxel_size=0.01, truncation=0.05, trunc_scale=0.025, streaming_extents=(1.0, 1.0, 0.5),
                              streaming_dims=(257, 257, 129), streaming_min=(-128, -128, -64))
    assert bytes(hp) == bytes(want)
    rp = T.RayCastParams()
    L.vh_raycast_params_from_app_state(C.byref(g), None, None, C.byref(rp))
    assert (rp.m_width, rp.m_height) == (640, 480) and rp.m_minDepth == f32(0.4) and rp.m_maxDepth == f32(4.5)
    inc = f32(0.8) * f32(0.05)
    assert rp.m_rayIncrement == inc and rp.m_thresSampleDist == f32(50.5) * inc and rp.m_thresDist == f32(50.0) * inc
    assert rp.m_maxNumVertices == 2097152 * 6 and rp.m_useGradients == 0
    mp = T.MarchingCubesParams()
    L.vh_marching_cubes_params_from_app_state(C.byref(g), C.byref(mp))
    assert mp.m_maxNumTriangles == 2500000 and mp.m_threshMarchingCubes == f32(10.0) * f32(0.01) and mp.m_hashNumBuckets == 2000000
    opt = T.SceneOptions()
    L.vh_scene_options_from_app_state(C.byref(g), C.byref(opt))
    assert opt.s_garbageCollectionEnabled == 1 and opt.s_garbageCollectionStarve == 15 and opt.s_streamingOutParts == 80 and opt.s_offlineProcessing == 0


def test_parsing_rules_of_the_reference_reader():
    L = lib.load()
    g = T.AppState()
    text = b'''
s_adapterWidth=320;s_adapterHeight=240;
	 s_adapterHeight	 =	 "200" ;
s_sensorDepthMax = 5.0f // metres
s_depthFilter = False
s_colorFilter = yes
# s_hashNumBuckets = 7;
s_hashNumSDFBlocks = 12abc;
s_SDFVoxelSize = .004f
'''
    lib.check(L.vh_app_state_parse(text, C.byref(g)), "parse")
    assert g.s_adapterWidth == 320       # the ";" ends the line: what follows it on the same line is dropped
    assert g.s_adapterHeight == 200      # blanks, tabs and quotes are stripped
    assert g.s_sensorDepthMax == f32(5.0) and g.s_depthFilter == 0 and g.s_colorFilter == 1  # only false / False / 0 are false
    assert g.s_hashNumBuckets == 0 and g.s_hashNumSDFBlocks == 12 and g.s_SDFVoxelSize == f32(0.004)
    assert L.vh_app_state_read(b"/nonexistent/zParameters.txt", C.byref(g)) == 6  # VH_ERR_IO


@pytest.mark.parametrize("name", ["zParametersDefault.txt", "zParametersManolisScan.txt", "zParametersTrackingDefault.txt"])
def test_reference_parameter_files(name):
    """the reference's own files, where its tree is mounted (read as data)"""
    path = os.path.join("/root/reference", name)
    if not os.path.exists(path):
        pytest.skip("reference tree not present (GPU box)")
    L, g = read(path)
    if name == "zParametersTrackingDefault.txt":
        assert g.numKeysFound == 0 or g.s_hashNumBuckets == 0  # the tracking file holds GlobalCameraTrackingState keys
        return
    assert g.numKeysFound >= 35 and g.s_hashNumBuckets >= 100000 and g.s_hashNumSDFBlocks >= 100000
    assert 0.001 <= g.s_SDFVoxelSize <= 0.05 and g.s_SDFTruncation >= 2 * g.s_SDFVoxelSize
    assert g.s_adapterWidth in (320, 640) and g.s_hashMaxCollisionLinkedListSize == 7 and g.s_SDFRayIncrementFactor == f32(0.8)
    dims, mins = list(g.s_streamingGridDimensions), list(g.s_streamingMinGridPos)
    assert dims[0] == dims[1] == dims[2] and dims[0] % 2 == 1 and mins == [-(dims[0] // 2)] * 3  # odd, centred on the origin
    if name == "zParametersDefault.txt":
        assert (g.s_adapterWidth, g.s_adapterHeight) == (320, 240) and g.s_SDFVoxelSize == f32(0.004) and g.s_hashNumBuckets == 500000
        assert g.s_hashNumSDFBlocks == 1000000 and g.s_depthFilter == 1 and g.s_garbageCollectionEnabled == 0 and g.s_marchingCubesMaxNumTriangles == 2500000


def test_recorded_sequence_keys():
    """the .sens file list is read as mLib reads a std::vector: name[0], name[1], ... until one is missing
    (parameterFile.h:92-108)"""
    L = lib.load()
    g = T.AppState()
    text = b"""s_sensorIdx = 8;
s_binaryDumpSensorFile[0] = "./DumpOutput/a_0.sens";
s_binaryDumpSensorFile[1] = "./DumpOutput/a_1.sens";
s_binaryDumpSensorFile[3] = "never reached: index 2 is missing";
s_binaryDumpSensorUseTrajectory = true;
s_binaryDumpSensorUseTrajectoryOnlyInit = false;	//only valid if prev is true
s_playData = true;
s_recordData = false;
s_reconstructionEnabled = true;
"""
    lib.check(L.vh_app_state_parse(text, C.byref(g)), "parse")
    assert g.s_numBinaryDumpSensorFiles == 2
    assert bytes(g.s_binaryDumpSensorFile[0].value) == b"./DumpOutput/a_0.sens" and bytes(g.s_binaryDumpSensorFile[1].value) == b"./DumpOutput/a_1.sens"
    assert (g.s_binaryDumpSensorUseTrajectory, g.s_binaryDumpSensorUseTrajectoryOnlyInit, g.s_playData, g.s_recordData, g.s_reconstructionEnabled) == (1, 0, 1, 0, 1)
    assert g.numKeysFound == 7  # the list counts once
