"""Recorded sequences: the `.sens` container and SensorDataReader.  Host-side code: no GPU.

The byte layout is checked against an independent packing of the reference's saveToFile / loadFromFile order
(sensorData.h:502-510, 756-830) written here with `struct`; the lossless decoders (zlib depth, PNG colour) exactly
against Python's own; the baseline JPEG decoder against Pillow's within 3 grey levels (JPEG leaves the IDCT and the
chroma upsampling to the decoder; stb_image, which the reference uses, differs from libjpeg by as much)."""
import io
import struct
import zlib

import numpy as np
import pytest

from voxelhashing_amd import sensor_data as SD
from voxelhashing_amd.lib import VhError

DW, DH, CW, CH = 40, 30, 48, 36


def frames(n, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        depth = (rng.integers(400, 5000, size=(DH, DW))).astype(np.uint16)
        depth[rng.random((DH, DW)) < 0.1] = 0
        y, x = np.mgrid[0:CH, 0:CW]
        color = np.stack([(4 * x + 10 * k) % 256, (6 * y) % 256, (x + y + 30 * k) % 256], axis=-1).astype(np.uint8)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 3] = [0.01 * k, -0.02 * k, 0.5 + 0.1 * k]
        out.append((depth, color, pose))
    return out


def pack_file(name, ctype, dtype, depth_shift, recs, imu_count_field, imu_records, color_bytes, depth_bytes):
    """the reference's write order, restated with struct"""
    intr = SD.make_intrinsic_matrix(50.0, 51.0, 19.5, 14.5).astype("<f4").tobytes()
    eye = np.eye(4, dtype="<f4").tobytes()
    b = struct.pack("<IQ", 4, len(name)) + name + intr + eye + intr + eye
    b += struct.pack("<iiIIIIf", ctype, dtype, CW, CH, DW, DH, depth_shift)
    b += struct.pack("<Q", len(recs))
    for (depth, color, pose), cb, db, k in zip(recs, color_bytes, depth_bytes, range(len(recs))):
        b += pose.astype("<f4").tobytes() + struct.pack("<QQQQ", 1000 + k, 2000 + k, len(cb), len(db)) + cb + db
    b += struct.pack("<Q", imu_count_field)
    for r in imu_records:
        b += struct.pack("<15dQ", *r)
    return b


def test_write_then_read_round_trip_and_byte_layout(tmp_path):
    recs = frames(3)
    intr = SD.make_intrinsic_matrix(50.0, 51.0, 19.5, 14.5)
    sd = SD.SensorData.create((DW, DH), (CW, CH), intr, depth_shift=1000.0, sensor_name="StructureSensor",
                              color_type=SD.TYPE_RAW, depth_type=SD.TYPE_RAW_USHORT)
    for k, (depth, color, pose) in enumerate(recs):
        sd.addFrame(color, depth, pose, 1000 + k, 2000 + k)
    imu = [tuple(float(i + 10 * j) for i in range(15)) + (77 + j,) for j in range(2)]
    for r in imu:
        sd.addIMUFrame(r[:15], r[15])
    path = tmp_path / "raw.sens"
    sd.saveToFile(path)
    want = pack_file(b"StructureSensor", 0, 0, 1000.0, recs, 2, imu, [c.tobytes() for _, c, _ in recs], [d.tobytes() for d, _, _ in recs])
    assert path.read_bytes() == want
    back = SD.SensorData.loadFromFile(path)
    i = back.info()
    assert (i.m_versionNumber, i.m_numFrames, i.m_numIMUFrames, i.m_depthWidth, i.m_depthHeight, i.m_colorWidth, i.m_colorHeight) == (4, 3, 2, DW, DH, CW, CH)
    assert i.m_sensorName == b"StructureSensor" and i.m_depthShift == 1000.0
    assert np.array_equal(np.array(i.m_depthIntrinsic[:]).reshape(4, 4), intr)
    for k, (depth, color, pose) in enumerate(recs):
        f = back.frame(k)
        assert np.array_equal(f["depth"], depth) and np.array_equal(f["color"], color)
        assert np.array_equal(f["cameraToWorld"].reshape(4, 4), pose) and f["timeStamps"] == (1000 + k, 2000 + k)
    with pytest.raises(VhError, match="out of bounds"):
        back.frame(3)


def test_reads_a_file_as_the_reference_writes_it_zlib_depth_png_colour_and_the_imu_count_quirk(tmp_path):
    """zlib depth + PNG colour streams made by Python's zlib / Pillow, and the IMU count field holding the RGB-D
    frame count with no IMU record behind it (sensorData.h:781)"""
    from PIL import Image
    recs = frames(4, seed=3)
    cbytes, dbytes = [], []
    for depth, color, _ in recs:
        buf = io.BytesIO()
        Image.fromarray(color).save(buf, format="PNG")
        cbytes.append(buf.getvalue())
        dbytes.append(zlib.compress(depth.tobytes(), 6))
    path = tmp_path / "ref_style.sens"
    path.write_bytes(pack_file(b"Kinect", 1, 1, 1000.0, recs, len(recs), [], cbytes, dbytes))
    sd = SD.SensorData.loadFromFile(path)
    i = sd.info()
    assert (i.m_numFrames, i.m_numIMUFrames, i.m_colorCompressionType, i.m_depthCompressionType) == (4, 0, 1, 1)
    for k, (depth, color, _) in enumerate(recs):
        f = sd.frame(k)
        assert np.array_equal(f["depth"], depth) and np.array_equal(f["color"], color)

    rd = SD.SensorDataReader(path)
    assert rd.getNumFrames() == 4
    with pytest.raises(VhError, match="invalid trajectory index"):
        rd.getRigidTransform()  # no frame decoded yet: index -1 wraps, as in the reference
    for k, (depth, color, pose) in enumerate(recs):
        d, c = rd.processDepth()
        assert d.dtype == np.float32 and np.array_equal(d, depth.astype(np.float32) / np.float32(1000.0))  # :129-131
        assert np.array_equal(c[..., :3], color) and np.all(c[..., 3] == 1)  # vec4uc(vec3uc): w = 1
        assert np.array_equal(rd.getRigidTransform().reshape(4, 4), pose) and rd.getCurrFrame() == k + 1
    assert rd.processDepth() is None  # sequence complete
    assert np.array_equal(rd.getRigidTransform(-1).reshape(4, 4), recs[2][2])


def test_zlib_depth_written_here_is_read_by_any_inflate(tmp_path):
    recs = frames(2, seed=5)
    sd = SD.SensorData.create((DW, DH), (CW, CH), np.eye(4), depth_type=SD.TYPE_ZLIB_USHORT)
    for depth, color, pose in recs:
        sd.addFrame(None, depth, pose)
    path = tmp_path / "z.sens"
    sd.saveToFile(path)
    raw = path.read_bytes()
    off = 4 + 8 + len(b"Unknown") + 4 * 64 + 8 + 16 + 4 + 8  # header up to the first frame
    off += 64 + 16
    csize, dsize = struct.unpack_from("<QQ", raw, off)
    assert csize == 0 and 0 < dsize < DW * DH * 2
    assert zlib.decompress(raw[off + 16:off + 16 + dsize]) == recs[0][0].tobytes()
    rd = SD.SensorDataReader(path)
    d, c = rd.processDepth()
    assert np.array_equal(d, recs[0][0].astype(np.float32) / np.float32(1000.0)) and not c.any()  # no colour data: zeros


@pytest.mark.parametrize("subsampling,quality", [(0, 95), (2, 90), (1, 85), (2, 60)])
def test_jpeg_colour_frames_match_pillow_within_three_levels(tmp_path, subsampling, quality):
    from PIL import Image
    y, x = np.mgrid[0:CH, 0:CW]
    img = np.stack([128 + 100 * np.sin(x / 7.0), 128 + 90 * np.cos(y / 5.0), 40 + 3 * x + 2 * y], axis=-1).clip(0, 255).astype(np.uint8)
    img[10:20, 12:30] = (250, 20, 30)  # a hard chroma edge
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG", quality=quality, subsampling=subsampling)
    want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
    sd = SD.SensorData.create((DW, DH), (CW, CH), np.eye(4), color_type=SD.TYPE_JPEG, depth_type=SD.TYPE_RAW_USHORT)
    sd.addFrame(buf.getvalue(), np.zeros((DH, DW), np.uint16))
    path = tmp_path / "j.sens"
    sd.saveToFile(path)
    got = SD.SensorData.loadFromFile(path).frame(0)["color"]
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 3 and diff.mean() < 0.6, (diff.max(), diff.mean())


def test_grey_jpeg_restart_intervals_and_odd_sizes(tmp_path):
    from PIL import Image
    w, h = 37, 21
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(5 * x) % 256, (9 * y) % 256, (3 * x + 4 * y) % 256], axis=-1).astype(np.uint8)
    for mode, kw in (("RGB", dict(subsampling=2)), ("L", {})):
        buf = io.BytesIO()
        Image.fromarray(img).convert(mode).save(buf, format="JPEG", quality=92, restart_marker_blocks=2, **kw)
        want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
        sd = SD.SensorData.create((4, 4), (w, h), np.eye(4), color_type=SD.TYPE_JPEG, depth_type=SD.TYPE_RAW_USHORT)
        sd.addFrame(buf.getvalue(), np.zeros((4, 4), np.uint16))
        got = sd.frame(0)["color"]
        diff = np.abs(got.astype(int) - want.astype(int))
        assert diff.max() <= 3, (mode, diff.max())


def test_errors(tmp_path):
    with pytest.raises(VhError, match="could not open file"):
        SD.SensorData.loadFromFile(tmp_path / "missing.sens")
    recs = frames(1)
    good = pack_file(b"x", 0, 0, 1000.0, recs, 0, [], [recs[0][1].tobytes()], [recs[0][0].tobytes()])
    bad = tmp_path / "v.sens"
    bad.write_bytes(struct.pack("<I", 3) + good[4:])
    with pytest.raises(VhError, match="Invalid file version -- found 3"):
        SD.SensorData.loadFromFile(bad)
    bad.write_bytes(good[:len(good) // 2])
    with pytest.raises(VhError, match="file ends inside"):
        SD.SensorData.loadFromFile(bad)
    # uplink depth is compiled out in the reference too
    occ = tmp_path / "o.sens"
    occ.write_bytes(pack_file(b"x", 0, 2, 1000.0, recs, 0, [], [recs[0][1].tobytes()], [b"1234"]))
    with pytest.raises(VhError, match="UPLINK"):
        SD.SensorData.loadFromFile(occ).frame(0)
    # damaged zlib stream, progressive JPEG
    z = tmp_path / "z.sens"
    z.write_bytes(pack_file(b"x", 0, 1, 1000.0, recs, 0, [], [recs[0][1].tobytes()], [b"not zlib at all"]))
    with pytest.raises(VhError, match="zlib"):
        SD.SensorData.loadFromFile(z).frame(0)
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(recs[0][1]).save(buf, format="JPEG", progressive=True)
    sd = SD.SensorData.create((DW, DH), (CW, CH), np.eye(4), color_type=SD.TYPE_JPEG, depth_type=SD.TYPE_RAW_USHORT)
    sd.addFrame(buf.getvalue(), recs[0][0])
    with pytest.raises(VhError, match="progressive"):
        sd.frame(0)
    with pytest.raises(VhError, match="raw"):
        sd.addFrame(recs[0][1], recs[0][0])  # no JPEG encoder: raw pixels cannot go into a JPEG sequence


def test_damaged_colour_and_depth_streams_are_refused_not_crashed(tmp_path):
    """random damage to JPEG / PNG / zlib frames: every outcome is a decoded image or a VhError"""
    from PIL import Image
    rng = np.random.default_rng(7)
    y, x = np.mgrid[0:CH, 0:CW]
    img = np.stack([(5 * x) % 256, (9 * y) % 256, (3 * x + 4 * y) % 256], axis=-1).astype(np.uint8)
    streams = {}
    for name, kw in (("JPEG", dict(quality=90, subsampling=2)), ("PNG", {})):
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format=name, **kw)
        streams[name] = buf.getvalue()
    depth = np.arange(DW * DH, dtype=np.uint16).reshape(DH, DW)
    outcomes = {"ok": 0, "refused": 0}
    for name, ctype in (("JPEG", SD.TYPE_JPEG), ("PNG", SD.TYPE_PNG)):
        good = bytearray(streams[name])
        for trial in range(150):
            b = bytearray(good)
            kind = trial % 3
            if kind == 0:    # flip a few bytes
                for _ in range(1 + trial % 5):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif kind == 1:  # truncate
                b = b[:int(rng.integers(1, len(b)))]
            else:            # splice garbage in
                at = int(rng.integers(0, len(b)))
                b[at:at] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8))
            sd = SD.SensorData.create((DW, DH), (CW, CH), np.eye(4), color_type=ctype, depth_type=SD.TYPE_RAW_USHORT)
            sd.addFrame(bytes(b), depth)
            try:
                out = sd.frame(0)["color"]
                assert out.shape == (CH, CW, 3)
                outcomes["ok"] += 1
            except VhError:
                outcomes["refused"] += 1
    assert outcomes["refused"] > 50 and outcomes["ok"] + outcomes["refused"] == 300
    # a damaged zlib depth frame in a file
    recs = frames(1)
    z = bytearray(zlib.compress(recs[0][0].tobytes()))
    for trial in range(40):
        b = bytearray(z)
        b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        path = tmp_path / "d.sens"
        path.write_bytes(pack_file(b"x", 0, 1, 1000.0, recs, 0, [], [recs[0][1].tobytes()], [bytes(b)]))
        try:
            SD.SensorData.loadFromFile(path).frame(0)
        except VhError:
            pass
