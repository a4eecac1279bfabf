// vh_sensor_data.cpp -- the recorded-sequence input side: ml::SensorData, the `.sens` container
// (DSC/sensorData/sensorData.h), and SensorDataReader (DSC/SensorDataReader.{h,cpp}), which turns its frames into the
// float depth map, RGBX colour and camera-to-world pose the frame loop consumes.  Host code: the images go to the
// device through CUDARGBDSensor::process (vh_sensor.cpp).
//
// File layout (sensorData.h:756-787, little endian, no padding):
//   u32 version (4) | u64 len, sensor name | colour calibration, depth calibration: intrinsic mat4f + extrinsic mat4f
//   each (:188-196) | i32 colour compression, i32 depth compression | u32 colour w, h, depth w, h | f32 depth shift |
//   u64 #frames, frames | u64 #IMU frames, IMU frames (15 doubles + u64 each, :573-580)
//   frame (:502-510): mat4f cameraToWorld | u64 colour time stamp, depth time stamp | u64 colour bytes, depth bytes |
//                     colour data | depth data
// The reference's writer stores the RGB-D frame count in the IMU count field (:781) whatever the number of IMU records
// that follow, so the IMU section of a file is read until it ends.
//
// Decoders.  The reference decodes through stb_image (zlib depth, PNG / JPEG colour); that file is third-party code
// vendored in the reference and is neither copied nor linked here.  Depth (raw / zlib u16) is lossless: any inflate
// gives the same samples; the system zlib does it.  PNG colour is lossless too (8-bit, non-interlaced decoded here).
// JPEG leaves the inverse DCT and the chroma upsampling to the decoder: the baseline decoder below (double-precision
// IDCT, the common 3:1 triangle upsampling) matches other decoders within a few grey levels per channel, which is
// the most that can be said of any pair of JPEG decoders; depth, poses and therefore geometry are unaffected.
// uplink "OCCI" depth is compiled out in the reference (_USE_UPLINK_COMPRESSION) and refused here.
#include "vh.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

namespace vh {

namespace {

[[noreturn]] void fail(int code, const std::string& what) { throw Error(code, what); }

struct FileReader {
    std::FILE* f;
    std::string name;
    uint64_t size = 0;
    explicit FileReader(const std::string& path) : f(std::fopen(path.c_str(), "rb")), name(path)
    {
        if (!f) fail(VH_ERR_IO, "could not open file " + path); // sensorData.h:792-794
        if (std::fseek(f, 0, SEEK_END) == 0) {
            const long n = std::ftell(f);
            size = n > 0 ? (uint64_t)n : 0;
        }
        std::rewind(f);
    }
    uint64_t remaining() const { const long at = std::ftell(f); return at >= 0 && (uint64_t)at <= size ? size - (uint64_t)at : 0; }
    ~FileReader() { if (f) std::fclose(f); }
    bool tryRead(void* dst, size_t n) { return n == 0 || std::fread(dst, 1, n, f) == n; }
    void read(void* dst, size_t n, const char* what)
    {
        if (!tryRead(dst, n)) fail(VH_ERR_IO, name + ": file ends inside " + what);
    }
    template <class T> T get(const char* what) { T v; read(&v, sizeof(T), what); return v; }
};

struct FileWriter {
    std::FILE* f;
    std::string name;
    explicit FileWriter(const std::string& path) : f(std::fopen(path.c_str(), "wb")), name(path)
    {
        if (!f) fail(VH_ERR_IO, "could not open file " + path + " for writing");
    }
    ~FileWriter() { if (f) std::fclose(f); }
    void write(const void* src, size_t n)
    {
        if (n && std::fwrite(src, 1, n, f) != n) fail(VH_ERR_IO, name + ": write failed");
    }
    template <class T> void put(const T& v) { write(&v, sizeof(T)); }
};

void inflateTo(const uint8_t* src, size_t srcBytes, uint8_t* dst, size_t dstBytes, const char* what)
{
    uLongf got = (uLongf)dstBytes;
    const int rc = ::uncompress(dst, &got, src, (uLong)srcBytes);
    if (rc != Z_OK || got != dstBytes) fail(VH_ERR_IO, std::string("zlib stream of ") + what + " is damaged or has the wrong size");
}

std::vector<uint8_t> inflateAll(const uint8_t* src, size_t srcBytes, const char* what)
{
    std::vector<uint8_t> out;
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) fail(VH_ERR_IO, "zlib init failed");
    zs.next_in = const_cast<Bytef*>(src);
    zs.avail_in = (uInt)srcBytes;
    uint8_t buf[1 << 16];
    int rc = Z_OK;
    while (rc == Z_OK) {
        zs.next_out = buf;
        zs.avail_out = sizeof(buf);
        rc = inflate(&zs, Z_NO_FLUSH);
        out.insert(out.end(), buf, buf + (sizeof(buf) - zs.avail_out));
    }
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) fail(VH_ERR_IO, std::string("zlib stream of ") + what + " is damaged");
    return out;
}

// ---------------------------------------------------------------------------------------------------------------
// PNG: 8-bit grey / grey+alpha / RGB / RGBA, non-interlaced -> RGB
// ---------------------------------------------------------------------------------------------------------------
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

void decodePng(const uint8_t* data, size_t n, uint32_t width, uint32_t height, uint8_t* rgb)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (n < 8 || std::memcmp(data, sig, 8) != 0) fail(VH_ERR_IO, "colour frame is not a PNG stream");
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int channels = 0;
    std::vector<uint8_t> idat, palette;
    bool end = false;
    while (!end && pos + 12 <= n) {
        const uint32_t len = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if ((size_t)len > n - pos - 12) fail(VH_ERR_IO, "PNG chunk runs past the frame");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) fail(VH_ERR_IO, "PNG header too short");
            w = be32(body); h = be32(body + 4);
            const int depth = body[8], colour = body[9], interlace = body[12];
            if (depth != 8 || interlace != 0) fail(VH_ERR_IO, "PNG colour frames must be 8-bit, non-interlaced");
            channels = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? -1 : colour == 4 ? 2 : colour == 6 ? 4 : 0;
            if (!channels) fail(VH_ERR_IO, "PNG colour type not understood");
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (w != width || h != height) fail(VH_ERR_IO, "PNG colour frame has a different size than the file header says");
    const int bpp = channels < 0 ? 1 : channels;
    const size_t stride = (size_t)w * bpp;
    std::vector<uint8_t> raw = inflateAll(idat.data(), idat.size(), "a PNG colour frame");
    if (raw.size() != (stride + 1) * h) fail(VH_ERR_IO, "PNG colour frame has the wrong amount of pixel data");
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* in = &raw[(stride + 1) * y];
        const int filter = in[0];
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            int pred = 0;
            switch (filter) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: fail(VH_ERR_IO, "PNG filter type not understood");
            }
            cur[i] = (uint8_t)(in[1 + i] + pred);
        }
        uint8_t* out = rgb + (size_t)3 * w * y;
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t* s = &cur[(size_t)x * bpp];
            if (channels == -1) {
                if ((size_t)s[0] * 3 + 2 >= palette.size()) fail(VH_ERR_IO, "PNG palette index out of range");
                out[3 * x] = palette[3 * s[0]]; out[3 * x + 1] = palette[3 * s[0] + 1]; out[3 * x + 2] = palette[3 * s[0] + 2];
            } else if (channels <= 2) {
                out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = s[0];
            } else {
                out[3 * x] = s[0]; out[3 * x + 1] = s[1]; out[3 * x + 2] = s[2];
            }
        }
        prev.swap(cur);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// JPEG: baseline / extended sequential Huffman (SOF0, SOF1), 8-bit, 1 or 3 components -> RGB
// ---------------------------------------------------------------------------------------------------------------
const uint8_t kZigZag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                              15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct JpegDecoder {
    const uint8_t* d;
    size_t n, pos = 0;
    uint16_t qt[4][64];
    bool haveQt[4] = { false, false, false, false };
    struct Huff {
        bool present = false;
        uint8_t vals[256];
        int minCode[17], maxCode[18], valPtr[17];
    } dc[4], ac[4];
    struct Comp {
        int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
        int planeW = 0, planeH = 0;
        std::vector<uint8_t> plane;
    } comp[3];
    int nComp = 0, width = 0, height = 0, hMax = 1, vMax = 1, restartInterval = 0;
    uint32_t bitBuf = 0;
    int bitCnt = 0;
    bool hitMarker = false;
    double cosT[8][8];

    JpegDecoder(const uint8_t* data, size_t bytes) : d(data), n(bytes)
    {
        for (int x = 0; x < 8; x++)
            for (int u = 0; u < 8; u++) cosT[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * M_PI / 16.0);
    }

    [[noreturn]] static void bad(const char* what) { fail(VH_ERR_IO, std::string("JPEG colour frame: ") + what); }
    uint8_t byte() { if (pos >= n) bad("stream ends early"); return d[pos++]; }
    int word() { const int a = byte(); return (a << 8) | byte(); }

    void readDqt(int len)
    {
        while (len > 0) {
            const int pq = byte(), prec = pq >> 4, id = pq & 15;
            if (id > 3) bad("quantisation table id out of range");
            for (int i = 0; i < 64; i++) qt[id][kZigZag[i]] = (uint16_t)(prec ? word() : byte());
            haveQt[id] = true;
            len -= 1 + 64 * (prec ? 2 : 1);
        }
    }
    void readDht(int len)
    {
        while (len > 0) {
            const int tc = byte(), cls = tc >> 4, id = tc & 15;
            if (cls > 1 || id > 3) bad("Huffman table id out of range");
            Huff& h = cls ? ac[id] : dc[id];
            uint8_t bits[17];
            int total = 0;
            for (int i = 1; i <= 16; i++) { bits[i] = byte(); total += bits[i]; }
            if (total > 256) bad("Huffman table too large");
            for (int i = 0; i < total; i++) h.vals[i] = byte();
            int code = 0, k = 0;
            for (int l = 1; l <= 16; l++) {
                h.valPtr[l] = k;
                h.minCode[l] = code;
                code += bits[l];
                k += bits[l];
                h.maxCode[l] = bits[l] ? code - 1 : -1;
                code <<= 1;
            }
            h.maxCode[17] = 0x7fffffff;
            h.present = true;
            len -= 17 + total;
        }
    }
    void readSof(int len)
    {
        if (byte() != 8) bad("only 8-bit samples are supported");
        height = word(); width = word(); nComp = byte();
        if (nComp != 1 && nComp != 3) bad("only grey and YCbCr images are supported");
        if (len != 6 + 3 * nComp) bad("frame header has the wrong length");
        for (int i = 0; i < nComp; i++) {
            comp[i].id = byte();
            const int hv = byte();
            comp[i].h = hv >> 4; comp[i].v = hv & 15; comp[i].tq = byte();
            if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2 || comp[i].tq > 3) bad("sampling factors above 2 are not supported");
            hMax = std::max(hMax, comp[i].h); vMax = std::max(vMax, comp[i].v);
        }
    }

    // entropy-coded segment: bit reader that un-stuffs FF00 and stops feeding at a marker
    void fill()
    {
        while (bitCnt <= 24) {
            int b = 0;
            if (!hitMarker && pos < n) {
                b = d[pos];
                if (b == 0xFF) {
                    const int b2 = pos + 1 < n ? d[pos + 1] : 0xD9;
                    if (b2 == 0) pos += 2;
                    else { hitMarker = true; b = 0; }
                } else pos++;
            }
            bitBuf |= (uint32_t)b << (24 - bitCnt);
            bitCnt += 8;
        }
    }
    int bits(int c)
    {
        if (c == 0) return 0;
        if (bitCnt < c) fill();
        const int v = (int)(bitBuf >> (32 - c));
        bitBuf <<= c; bitCnt -= c;
        return v;
    }
    int decode(const Huff& h)
    {
        if (!h.present) bad("scan uses a Huffman table that was never defined");
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | bits(1);
            if (h.maxCode[l] >= 0 && code <= h.maxCode[l] && code >= h.minCode[l]) return h.vals[h.valPtr[l] + code - h.minCode[l]];
        }
        bad("invalid Huffman code");
    }
    static int extend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

    void block(Comp& c, int bx, int by)
    {
        double coef[64] = { 0 };
        const uint16_t* q = qt[c.tq];
        const int s = decode(dc[c.td]);
        if (s > 11) bad("DC difference too large");
        c.pred += extend(bits(s), s);
        coef[0] = (double)c.pred * q[0];
        for (int k = 1; k < 64;) {
            const int rs = decode(ac[c.ta]), r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r != 15) break;
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) bad("AC coefficient index out of range");
            coef[kZigZag[k]] = (double)extend(bits(sz), sz) * q[kZigZag[k]];
            k++;
        }
        // separable inverse DCT in double, then level shift, round, clamp
        double tmp[64];
        for (int y = 0; y < 8; y++)
            for (int u = 0; u < 8; u++) {
                double a = 0.0;
                for (int v = 0; v < 8; v++) a += cosT[y][v] * coef[8 * v + u];
                tmp[8 * y + u] = a;
            }
        for (int y = 0; y < 8; y++) {
            uint8_t* out = &c.plane[(size_t)(by * 8 + y) * c.planeW + bx * 8];
            for (int x = 0; x < 8; x++) {
                double a = 0.0;
                for (int u = 0; u < 8; u++) a += cosT[x][u] * tmp[8 * y + u];
                const long r = std::lround(a + 128.0);
                out[x] = (uint8_t)std::min(255L, std::max(0L, r));
            }
        }
    }

    void readScan()
    {
        const int len = word(), ns = byte();
        if (ns != nComp || len != 6 + 2 * ns) bad("only single-scan (non-progressive, interleaved) images are supported");
        for (int i = 0; i < ns; i++) {
            const int id = byte(), t = byte();
            if (id != comp[i].id) bad("scan component order differs from the frame header");
            comp[i].td = t >> 4; comp[i].ta = t & 15;
            if (comp[i].td > 3 || comp[i].ta > 3) bad("Huffman table selector out of range");
            if (!haveQt[comp[i].tq]) bad("quantisation table missing");
        }
        pos += 3; // spectral selection and approximation: fixed for sequential scans
        const int mcuW = 8 * hMax, mcuH = 8 * vMax;
        const int mcusX = (width + mcuW - 1) / mcuW, mcusY = (height + mcuH - 1) / mcuH;
        for (int i = 0; i < nComp; i++) {
            comp[i].planeW = mcusX * comp[i].h * 8; comp[i].planeH = mcusY * comp[i].v * 8;
            comp[i].plane.assign((size_t)comp[i].planeW * comp[i].planeH, 0);
            comp[i].pred = 0;
        }
        bitBuf = 0; bitCnt = 0; hitMarker = false;
        int untilRestart = restartInterval;
        for (int my = 0; my < mcusY; my++)
            for (int mx = 0; mx < mcusX; mx++) {
                if (restartInterval && untilRestart == 0) {
                    // byte-align, expect RSTn
                    bitBuf = 0; bitCnt = 0; hitMarker = false;
                    while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7)) pos++;
                    if (pos + 1 >= n) bad("restart marker missing");
                    pos += 2;
                    for (int i = 0; i < nComp; i++) comp[i].pred = 0;
                    untilRestart = restartInterval;
                }
                for (int i = 0; i < nComp; i++)
                    for (int v = 0; v < comp[i].v; v++)
                        for (int h = 0; h < comp[i].h; h++) block(comp[i], mx * comp[i].h + h, my * comp[i].v + v);
                untilRestart--;
            }
    }

    // chroma plane -> full resolution (width x height), triangle filter in each subsampled direction
    std::vector<uint8_t> upsample(const Comp& c) const
    {
        const int fx = hMax / c.h, fy = vMax / c.v;
        const int cw = (width * c.h + hMax - 1) / hMax, ch = (height * c.v + vMax - 1) / vMax; // samples that carry image
        std::vector<uint8_t> out((size_t)width * height);
        if (fx == 1 && fy == 1) {
            for (int y = 0; y < height; y++) std::memcpy(&out[(size_t)y * width], &c.plane[(size_t)y * c.planeW], (size_t)width);
            return out;
        }
        std::vector<int> rowSum((size_t)cw);
        for (int y = 0; y < height; y++) {
            // vertical: 3/4 nearest row + 1/4 next-nearest (scaled by 4), or 4x the row when not subsampled vertically
            const int sy = fy == 2 ? y >> 1 : y;
            const int oy = fy == 2 ? ((y & 1) ? std::min(sy + 1, ch - 1) : std::max(sy - 1, 0)) : sy;
            const uint8_t* r0 = &c.plane[(size_t)sy * c.planeW];
            const uint8_t* r1 = &c.plane[(size_t)oy * c.planeW];
            for (int x = 0; x < cw; x++) rowSum[x] = fy == 2 ? 3 * r0[x] + r1[x] : 4 * r0[x];
            uint8_t* o = &out[(size_t)y * width];
            if (fx == 1) {
                for (int x = 0; x < width; x++) o[x] = (uint8_t)((rowSum[x] + 2) >> 2);
            } else {
                for (int x = 0; x < width; x++) {
                    const int sx = x >> 1;
                    const int ox = (x & 1) ? std::min(sx + 1, cw - 1) : std::max(sx - 1, 0);
                    // rounding as the common decoders: 8 / 7 after both filters, 1 / 2 (x4) after the horizontal one alone
                    const int bias = fy == 2 ? ((x & 1) ? 7 : 8) : ((x & 1) ? 8 : 4);
                    o[x] = (uint8_t)((3 * rowSum[sx] + rowSum[ox] + bias) >> 4);
                }
            }
        }
        return out;
    }

    void run(uint32_t wantW, uint32_t wantH, uint8_t* rgb)
    {
        if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) bad("missing SOI");
        pos = 2;
        bool scanned = false;
        while (!scanned) {
            int b = byte();
            if (b != 0xFF) continue;
            int m = byte();
            while (m == 0xFF) m = byte();
            if (m == 0xD9) break;
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (m == 0xDA) {
                if (!width) bad("scan before frame header");
                readScan();
                scanned = true;
                break;
            }
            const int len = word() - 2;
            if (len < 0 || (size_t)len > n - pos) bad("segment runs past the frame");
            const size_t next = pos + (size_t)len;
            if (m == 0xDB) readDqt(len);
            else if (m == 0xC4) readDht(len);
            else if (m == 0xC0 || m == 0xC1) {
                readSof(len);
                // before anything is allocated for it
                if ((uint32_t)width != wantW || (uint32_t)height != wantH) bad("frame has a different size than the file header says");
            }
            else if (m == 0xC2) bad("progressive JPEG is not supported");
            else if (m >= 0xC3 && m <= 0xCF && m != 0xC8 && m != 0xCC) bad("lossless / arithmetic JPEG is not supported");
            else if (m == 0xDD) restartInterval = word();
            pos = next;
        }
        if (!scanned) bad("no image data");
        if (nComp == 1) {
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) {
                    const uint8_t v = comp[0].plane[(size_t)y * comp[0].planeW + x];
                    uint8_t* o = rgb + 3 * ((size_t)y * width + x);
                    o[0] = o[1] = o[2] = v;
                }
            return;
        }
        const std::vector<uint8_t> Y = upsample(comp[0]), Cb = upsample(comp[1]), Cr = upsample(comp[2]);
        auto clamp8 = [](double v) { const long r = std::lround(v); return (uint8_t)std::min(255L, std::max(0L, r)); };
        for (size_t i = 0; i < (size_t)width * height; i++) { // JFIF: full-range BT.601
            const double y = Y[i], cb = (double)Cb[i] - 128.0, cr = (double)Cr[i] - 128.0;
            rgb[3 * i + 0] = clamp8(y + 1.402 * cr);
            rgb[3 * i + 1] = clamp8(y - 0.344136 * cb - 0.714136 * cr);
            rgb[3 * i + 2] = clamp8(y + 1.772 * cb);
        }
    }
};

} // namespace

// ---------------------------------------------------------------------------------------------------------------
// SensorData
// ---------------------------------------------------------------------------------------------------------------
SensorData::SensorData()
{
    m_versionNumber = kVersion;
    m_sensorName = "Unknown";
    m_colorCompressionType = TYPE_RAW;
    m_depthCompressionType = TYPE_RAW_USHORT;
    m_colorWidth = m_colorHeight = m_depthWidth = m_depthHeight = 0;
    m_depthShift = 1000.0f; // sensorData.h:622-637
    m_colorIntrinsic = m_colorExtrinsic = m_depthIntrinsic = m_depthExtrinsic = mat4f::identity();
}

mat4f SensorData::makeIntrinsicMatrix(float fx, float fy, float mx, float my) // :168-175
{
    mat4f m = mat4f::identity();
    m(0, 0) = fx; m(0, 2) = mx;
    m(1, 1) = fy; m(1, 2) = my;
    return m;
}

void SensorData::loadFromFile(const std::string& filename) // :789-830
{
    FileReader in(filename);
    m_frames.clear();
    m_IMUFrames.clear();
    m_versionNumber = in.get<uint32_t>("the header");
    if (m_versionNumber != kVersion) // assertVersionNumber :647-650
        fail(VH_ERR_VERSION_MISMATCH, "Invalid file version -- found " + std::to_string(m_versionNumber) + " but expected " + std::to_string(kVersion));
    const uint64_t strLen = in.get<uint64_t>("the header");
    if (strLen > in.remaining()) fail(VH_ERR_IO, filename + ": file ends inside the sensor name");
    m_sensorName.assign((size_t)strLen, '\0');
    in.read(&m_sensorName[0], (size_t)strLen, "the sensor name");
    in.read(m_colorIntrinsic.m, 64, "the calibration"); in.read(m_colorExtrinsic.m, 64, "the calibration");
    in.read(m_depthIntrinsic.m, 64, "the calibration"); in.read(m_depthExtrinsic.m, 64, "the calibration");
    m_colorCompressionType = in.get<int32_t>("the header");
    m_depthCompressionType = in.get<int32_t>("the header");
    m_colorWidth = in.get<uint32_t>("the header"); m_colorHeight = in.get<uint32_t>("the header");
    m_depthWidth = in.get<uint32_t>("the header"); m_depthHeight = in.get<uint32_t>("the header");
    m_depthShift = in.get<float>("the header");
    if (m_colorCompressionType < TYPE_RAW || m_colorCompressionType > TYPE_JPEG || m_depthCompressionType < TYPE_RAW_USHORT || m_depthCompressionType > TYPE_OCCI_USHORT)
        fail(VH_ERR_IO, filename + ": compression type not understood");
    if ((uint64_t)m_depthWidth * m_depthHeight > (1ull << 28) || (uint64_t)m_colorWidth * m_colorHeight > (1ull << 28))
        fail(VH_ERR_IO, filename + ": image size is implausible");
    const uint64_t numFrames = in.get<uint64_t>("the header");
    m_frames.reserve((size_t)std::min<uint64_t>(numFrames, in.remaining() / 96)); // a frame is at least 96 bytes
    for (uint64_t i = 0; i < numFrames; i++) { // RGBDFrame::loadFromFile :512-523
        RGBDFrame f;
        in.read(f.m_cameraToWorld.m, 64, "a frame");
        f.m_timeStampColor = in.get<uint64_t>("a frame"); f.m_timeStampDepth = in.get<uint64_t>("a frame");
        const uint64_t colorBytes = in.get<uint64_t>("a frame"), depthBytes = in.get<uint64_t>("a frame");
        // a damaged size field must not turn into a giant allocation: the data has to be in the file
        if (colorBytes > in.remaining() || depthBytes > in.remaining() - colorBytes) fail(VH_ERR_IO, filename + ": file ends inside a frame's data");
        f.m_colorCompressed.resize((size_t)colorBytes); f.m_depthCompressed.resize((size_t)depthBytes);
        in.read(f.m_colorCompressed.data(), (size_t)colorBytes, "a frame's colour data");
        in.read(f.m_depthCompressed.data(), (size_t)depthBytes, "a frame's depth data");
        m_frames.push_back(std::move(f));
    }
    // IMU section: the count field is unreliable (see the file comment); read whole records until the file ends
    uint64_t numIMUFrames = 0;
    if (in.tryRead(&numIMUFrames, sizeof(numIMUFrames))) {
        for (uint64_t i = 0; i < numIMUFrames; i++) {
            IMUFrame f;
            if (!in.tryRead(&f, sizeof(f))) break;
            m_IMUFrames.push_back(f);
        }
    }
}

void SensorData::saveToFile(const std::string& filename) const // :756-787
{
    FileWriter out(filename);
    out.put<uint32_t>(m_versionNumber);
    out.put<uint64_t>(m_sensorName.size());
    out.write(m_sensorName.data(), m_sensorName.size());
    out.write(m_colorIntrinsic.m, 64); out.write(m_colorExtrinsic.m, 64);
    out.write(m_depthIntrinsic.m, 64); out.write(m_depthExtrinsic.m, 64);
    out.put<int32_t>(m_colorCompressionType); out.put<int32_t>(m_depthCompressionType);
    out.put<uint32_t>(m_colorWidth); out.put<uint32_t>(m_colorHeight);
    out.put<uint32_t>(m_depthWidth); out.put<uint32_t>(m_depthHeight);
    out.put<float>(m_depthShift);
    out.put<uint64_t>(m_frames.size());
    for (const RGBDFrame& f : m_frames) {
        out.write(f.m_cameraToWorld.m, 64);
        out.put<uint64_t>(f.m_timeStampColor); out.put<uint64_t>(f.m_timeStampDepth);
        out.put<uint64_t>(f.m_colorCompressed.size()); out.put<uint64_t>(f.m_depthCompressed.size());
        out.write(f.m_colorCompressed.data(), f.m_colorCompressed.size());
        out.write(f.m_depthCompressed.data(), f.m_depthCompressed.size());
    }
    // the reference writes m_frames.size() here (:781); a reader that trusts the field then reads past the end of its
    // own files.  The true count is written: files that carry IMU records for every frame are byte-identical.
    out.put<uint64_t>(m_IMUFrames.size());
    for (const IMUFrame& f : m_IMUFrames) out.put(f);
}

void SensorData::addFrame(const uint8_t* colorRGB, const uint16_t* depth, const mat4f& cameraToWorld, uint64_t timeStampColor, uint64_t timeStampDepth)
{ // addFrame :657-667 with RGBDFrame::compressColor / compressDepth :335-460
    RGBDFrame f;
    f.m_cameraToWorld = cameraToWorld;
    f.m_timeStampColor = timeStampColor; f.m_timeStampDepth = timeStampDepth;
    if (colorRGB) {
        if (m_colorCompressionType != TYPE_RAW) fail(VH_ERR_BAD_ARGUMENT, "colour frames can only be written raw (no PNG / JPEG encoder is built in)");
        f.m_colorCompressed.assign(colorRGB, colorRGB + (size_t)3 * m_colorWidth * m_colorHeight);
    }
    if (depth) {
        const size_t bytes = (size_t)2 * m_depthWidth * m_depthHeight;
        if (m_depthCompressionType == TYPE_RAW_USHORT) {
            f.m_depthCompressed.assign((const uint8_t*)depth, (const uint8_t*)depth + bytes);
        } else if (m_depthCompressionType == TYPE_ZLIB_USHORT) {
            uLongf cap = compressBound((uLong)bytes);
            f.m_depthCompressed.resize(cap);
            if (compress2(f.m_depthCompressed.data(), &cap, (const Bytef*)depth, (uLong)bytes, 8) != Z_OK) fail(VH_ERR_IO, "zlib compression failed");
            f.m_depthCompressed.resize(cap);
        } else {
            fail(VH_ERR_BAD_ARGUMENT, "depth frames can only be written raw or zlib-compressed");
        }
    }
    m_frames.push_back(std::move(f));
}

void SensorData::decompressDepth(size_t frameIdx, uint16_t* out) const // decompressDepthAlloc :462-500
{
    if (frameIdx >= m_frames.size()) fail(VH_ERR_BAD_ARGUMENT, "out of bounds"); // :703
    const RGBDFrame& f = m_frames[frameIdx];
    const size_t bytes = (size_t)2 * m_depthWidth * m_depthHeight;
    if (m_depthCompressionType == TYPE_RAW_USHORT) {
        if (f.m_depthCompressed.size() != bytes) fail(VH_ERR_IO, "invalid data");
        std::memcpy(out, f.m_depthCompressed.data(), bytes);
    } else if (m_depthCompressionType == TYPE_ZLIB_USHORT) {
        inflateTo(f.m_depthCompressed.data(), f.m_depthCompressed.size(), (uint8_t*)out, bytes, "a depth frame");
    } else {
        fail(VH_ERR_BAD_ARGUMENT, "need UPLINK_COMPRESSION"); // :489: compiled out in the reference as well
    }
}

void SensorData::decompressColor(size_t frameIdx, uint8_t* outRGB) const // decompressColorAlloc :369-415
{
    if (frameIdx >= m_frames.size()) fail(VH_ERR_BAD_ARGUMENT, "out of bounds");
    const RGBDFrame& f = m_frames[frameIdx];
    if (f.m_colorCompressed.empty()) fail(VH_ERR_IO, "decompression error"); // :380
    if (m_colorCompressionType == TYPE_RAW) {
        if (f.m_colorCompressed.size() != (size_t)3 * m_colorWidth * m_colorHeight) fail(VH_ERR_IO, "invalid data");
        std::memcpy(outRGB, f.m_colorCompressed.data(), f.m_colorCompressed.size());
    } else if (m_colorCompressionType == TYPE_PNG) {
        decodePng(f.m_colorCompressed.data(), f.m_colorCompressed.size(), m_colorWidth, m_colorHeight, outRGB);
    } else {
        JpegDecoder(f.m_colorCompressed.data(), f.m_colorCompressed.size()).run(m_colorWidth, m_colorHeight, outRGB);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// SensorDataReader
// ---------------------------------------------------------------------------------------------------------------
SensorDataReader::SensorDataReader() : m_numFrames(0), m_currFrame(0), m_bHasColorData(false) {}

void SensorDataReader::createFirstConnected(const std::string& filename) // SensorDataReader.cpp:39-77
{
    m_currFrame = 0;
    m_bHasColorData = false;
    m_sensorData.reset(new SensorData);
    m_sensorData->loadFromFile(filename);
    const SensorData& s = *m_sensorData;
    // RGBDSensor::init(depthWidth, depthHeight, max(colorWidth, 1), max(colorHeight, 1), 1)
    m_depthFloat.assign((size_t)s.m_depthWidth * s.m_depthHeight, 0.0f);
    m_colorRGBX.assign((size_t)4 * std::max(s.m_colorWidth, 1u) * std::max(s.m_colorHeight, 1u), 0);
    m_depthShorts.resize((size_t)s.m_depthWidth * s.m_depthHeight);
    m_colorRGB.resize((size_t)3 * s.m_colorWidth * s.m_colorHeight);
    m_numFrames = (unsigned int)s.m_frames.size();
    m_bHasColorData = m_numFrames > 0 && !s.m_frames[0].m_colorCompressed.empty();
}

bool SensorDataReader::processDepth() // :97-165; false = the sequence is complete (the reference then stops playing)
{
    if (!m_sensorData) fail(VH_ERR_BAD_ARGUMENT, "SensorDataReader: no file loaded");
    if (m_currFrame >= m_numFrames) return false;
    const SensorData& s = *m_sensorData;
    s.decompressDepth(m_currFrame, m_depthShorts.data());
    for (size_t i = 0; i < m_depthShorts.size(); i++) m_depthFloat[i] = (float)m_depthShorts[i] / s.m_depthShift; // :129-131
    if (m_bHasColorData) {
        s.decompressColor(m_currFrame, m_colorRGB.data());
        for (size_t i = 0; i < (size_t)s.m_colorWidth * s.m_colorHeight; i++) { // vec4uc(vec3uc): w = 1, point4d.h:42-47
            m_colorRGBX[4 * i + 0] = m_colorRGB[3 * i + 0]; m_colorRGBX[4 * i + 1] = m_colorRGB[3 * i + 1];
            m_colorRGBX[4 * i + 2] = m_colorRGB[3 * i + 2]; m_colorRGBX[4 * i + 3] = 1;
        }
    }
    m_currFrame++;
    return true;
}

mat4f SensorDataReader::getRigidTransform(int offset) const // :172-179
{
    const unsigned int idx = m_currFrame - 1 + (unsigned int)offset;
    if (!m_sensorData || idx >= m_sensorData->m_frames.size()) fail(VH_ERR_BAD_ARGUMENT, "invalid trajectory index " + std::to_string(idx));
    return m_sensorData->m_frames[idx].m_cameraToWorld;
}

const SensorData& SensorDataReader::getSensorData() const
{
    if (!m_sensorData) fail(VH_ERR_BAD_ARGUMENT, "SensorDataReader: no file loaded");
    return *m_sensorData;
}

} // namespace vh
