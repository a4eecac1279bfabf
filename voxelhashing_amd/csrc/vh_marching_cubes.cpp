// vh_marching_cubes.cpp -- host side of the iso-surface extraction: CUDAMarchingCubesHashSDF
// (DSC/CUDAMarchingCubesHashSDF.{h,cpp}), MarchingCubesData::allocate/free/copyToCPU
// (DSC/MarchingCubesSDFUtil.h:57-147) and the mesh container the reference takes from mLib (MeshDataf, MeshIOf).
//
// The mesh functions restate the mLib revision the reference vendors (DepthSensingCUDA/Include/mLib/include/
// core-mesh/meshData.{h,cpp}, meshIO.cpp, cited as MLIB/ below).  No reference output exists for them, so they are
// pinned by that source only ("parity unpinned" as far as outputs go).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <unordered_map>
#include <set>

#include "../../include/vh.hpp"
#include "vh_host_util.hpp"

namespace {

inline void check(int code, const char* what)
{
    if (code != 0) throw vh::Error(code, std::string(what) + ": " + vh_error_string(code));
}
inline void checkHip(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw vh::Error(-(int)e, std::string(what) + ": " + hipGetErrorString(e));
}

} // namespace

// ---------------------------------------------------------------------------
// launcher-level buffers
// ---------------------------------------------------------------------------

extern "C" {

// MarchingCubesData::allocate (GPU side), DSC/MarchingCubesSDFUtil.h:57-83
int vh_marching_cubes_data_alloc(VhMarchingCubesData* data, const VhMarchingCubesParams* params)
{
    if (!data || !params || params->m_maxNumTriangles == 0) return VH_ERR_BAD_ARGUMENT;
    std::memset(data, 0, sizeof(*data));
    const size_t maxBlocks = (size_t)params->m_hashNumBuckets * params->m_hashBucketSize;
    VH_HIP(hipMalloc((void**)&data->d_params, sizeof(VhMarchingCubesParams)));
    VH_HIP(hipMalloc((void**)&data->d_numOccupiedBlocks, sizeof(uint32_t)));
    VH_HIP(hipMalloc((void**)&data->d_occupiedBlocks, sizeof(uint32_t) * (maxBlocks ? maxBlocks : 1)));
    VH_HIP(hipMalloc((void**)&data->d_triangles, sizeof(VhTriangle) * (size_t)params->m_maxNumTriangles));
    VH_HIP(hipMalloc((void**)&data->d_numTriangles, sizeof(uint32_t)));
    VH_HIP(hipMemcpy(data->d_params, params, sizeof(*params), hipMemcpyHostToDevice));
    data->m_bIsOnGPU = 1;
    return VH_OK;
}

void vh_marching_cubes_data_free(VhMarchingCubesData* data)
{
    if (!data) return;
    if (data->d_params) (void)hipFree(data->d_params);
    if (data->d_numOccupiedBlocks) (void)hipFree(data->d_numOccupiedBlocks);
    if (data->d_occupiedBlocks) (void)hipFree(data->d_occupiedBlocks);
    if (data->d_triangles) (void)hipFree(data->d_triangles);
    if (data->d_numTriangles) (void)hipFree(data->d_numTriangles);
    std::memset(data, 0, sizeof(*data));
}

// MarchingCubesData::updateParams :85-93
int vh_marching_cubes_update_params(const VhMarchingCubesData* data, const VhMarchingCubesParams* params, vhStream_t stream)
{
    if (!data || !data->d_params || !params) return VH_ERR_BAD_ARGUMENT;
    VH_HIP(hipMemcpyAsync(data->d_params, params, sizeof(*params), hipMemcpyHostToDevice, (hipStream_t)stream));
    VH_HIP(hipStreamSynchronize((hipStream_t)stream)); // params is the caller's stack object
    return VH_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------
// vh::MeshData
// ---------------------------------------------------------------------------

void vh::MeshData::makeTriangleSoupIndices()
{
    m_FaceIndicesVertices.resize(m_Vertices.size() / 3 * 3);
    for (size_t i = 0; i < m_FaceIndicesVertices.size(); i++) m_FaceIndicesVertices[i] = (unsigned int)i;
}

namespace {
struct CellKey {
    int x, y, z;
    bool operator==(const CellKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct CellHash {
    size_t operator()(const CellKey& k) const
    {
        return ((size_t)(unsigned int)k.x * 73856093u) ^ ((size_t)(unsigned int)k.y * 19349669u) ^ ((size_t)(unsigned int)k.z * 83492791u);
    }
};
inline int signf(float v) { return (0.0f < v) - (v < 0.0f); }
} // namespace

// MLIB/core-mesh/meshData.cpp:216-289 with approx = true (the only mode the reference calls): a vertex goes to the
// cell int(v / thresh + 0.5 sign(v)) (meshData.h:732-734); the 27 cells around it are searched in x-major order and
// the first one that holds a vertex wins (meshData.cpp:197-212, no distance test); otherwise the vertex founds its
// cell.  Faces are re-indexed and those left with a repeated index are dropped (removeDegeneratedFaces :293-318).
void vh::MeshData::mergeCloseVertices(float thresh)
{
    if (thresh <= 0.0f) throw vh::Error(VH_ERR_BAD_ARGUMENT, "mergeCloseVertices: invalid thresh");
    if (m_Vertices.empty()) return;
    std::unordered_map<CellKey, unsigned int, CellHash> grid;
    grid.reserve(2 * m_Vertices.size());
    std::vector<unsigned int> lookUp(m_Vertices.size());
    std::vector<vec3f> verts;
    std::vector<float> cols;
    verts.reserve(m_Vertices.size());
    const bool hasColors = m_Colors.size() == 4 * m_Vertices.size();
    unsigned int cnt = 0;
    for (size_t v = 0; v < m_Vertices.size(); v++) {
        const vec3f& p = m_Vertices[v];
        const CellKey c = { (int)(p.x / thresh + 0.5f * (float)signf(p.x)), (int)(p.y / thresh + 0.5f * (float)signf(p.y)),
                            (int)(p.z / thresh + 0.5f * (float)signf(p.z)) };
        unsigned int nn = (unsigned int)-1;
        for (int i = -1; i <= 1 && nn == (unsigned int)-1; i++)
            for (int j = -1; j <= 1 && nn == (unsigned int)-1; j++)
                for (int k = -1; k <= 1 && nn == (unsigned int)-1; k++) {
                    auto it = grid.find(CellKey{ c.x + i, c.y + j, c.z + k });
                    if (it != grid.end()) nn = it->second;
                }
        if (nn == (unsigned int)-1) {
            grid[c] = cnt;
            verts.push_back(p);
            if (hasColors) cols.insert(cols.end(), m_Colors.begin() + 4 * v, m_Colors.begin() + 4 * v + 4);
            lookUp[v] = cnt++;
        } else {
            lookUp[v] = nn;
        }
    }
    for (auto& f : m_FaceIndicesVertices) f = lookUp[f];
    if (verts.size() != m_Vertices.size()) {
        m_Vertices.swap(verts);
        if (hasColors) m_Colors.swap(cols);
    }
    std::vector<unsigned int> faces;
    faces.reserve(m_FaceIndicesVertices.size());
    for (size_t f = 0; f + 2 < m_FaceIndicesVertices.size(); f += 3) {
        const unsigned int a = m_FaceIndicesVertices[f], b = m_FaceIndicesVertices[f + 1], c = m_FaceIndicesVertices[f + 2];
        if (a == b || b == c || a == c) continue;
        faces.push_back(a); faces.push_back(b); faces.push_back(c);
    }
    m_FaceIndicesVertices.swap(faces);
}

// MLIB/core-mesh/meshData.cpp:36-105: of the faces over one vertex set the first is kept, in its own winding
void vh::MeshData::removeDuplicateFaces()
{
    std::set<std::array<unsigned int, 3>> seen;
    std::vector<unsigned int> faces;
    faces.reserve(m_FaceIndicesVertices.size());
    for (size_t f = 0; f + 2 < m_FaceIndicesVertices.size(); f += 3) {
        std::array<unsigned int, 3> key = { m_FaceIndicesVertices[f], m_FaceIndicesVertices[f + 1], m_FaceIndicesVertices[f + 2] };
        std::sort(key.begin(), key.end());
        if (!seen.insert(key).second) continue;
        faces.push_back(m_FaceIndicesVertices[f]); faces.push_back(m_FaceIndicesVertices[f + 1]); faces.push_back(m_FaceIndicesVertices[f + 2]);
    }
    m_FaceIndicesVertices.swap(faces);
}

// MLIB/core-mesh/meshData.cpp:473-556.  The vendored revision returns without doing anything when *this is empty
// (the assignment is commented out, :479-482), which would leave the reference's offline path with an empty mesh
// for ever; here an empty mesh takes the other one over (fenced reference defects).
void vh::MeshData::merge(const MeshData& other)
{
    if (other.m_Vertices.empty()) return;
    if (m_Vertices.empty()) { *this = other; return; }
    if (hasVertexIndices() != other.hasVertexIndices()) throw vh::Error(VH_ERR_BAD_ARGUMENT, "invalid mesh conversion");
    const unsigned int base = (unsigned int)m_Vertices.size();
    m_Vertices.insert(m_Vertices.end(), other.m_Vertices.begin(), other.m_Vertices.end());
    m_Colors.insert(m_Colors.end(), other.m_Colors.begin(), other.m_Colors.end());
    for (unsigned int f : other.m_FaceIndicesVertices) m_FaceIndicesVertices.push_back(base + f);
}

// MLIB/core-mesh/meshData.h:471-479 with Matrix4x4 * point3d = implicit w = 1 and de-homogenisation
// (MLIB/core-math/matrix4x4.h:459-468)
void vh::MeshData::applyTransform(const mat4f& t)
{
    for (auto& v : m_Vertices) {
        const float x = t.m[0] * v.x + t.m[1] * v.y + t.m[2] * v.z + t.m[3];
        const float y = t.m[4] * v.x + t.m[5] * v.y + t.m[6] * v.z + t.m[7];
        const float z = t.m[8] * v.x + t.m[9] * v.y + t.m[10] * v.z + t.m[11];
        const float w = t.m[12] * v.x + t.m[13] * v.y + t.m[14] * v.z + t.m[15];
        v = { x / w, y / w, z / w };
    }
}

// MeshIO::saveToPLY, MLIB/core-mesh/meshIO.cpp:485-556
void vh::MeshData::saveToPLY(const std::string& filename) const
{
    std::ofstream f(filename, std::ios::binary);
    if (!f) throw vh::Error(VH_ERR_IO, "cannot write " + filename);
    const bool hasColors = m_Colors.size() == 4 * m_Vertices.size();
    const size_t nFaces = hasVertexIndices() ? m_FaceIndicesVertices.size() / 3 : m_Vertices.size() / 3;
    f << "ply\nformat binary_little_endian 1.0\ncomment MLIB generated\nelement vertex " << m_Vertices.size() << "\nproperty float x\nproperty float y\nproperty float z\n";
    if (hasColors) f << "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\n";
    f << "element face " << nFaces << "\nproperty list uchar int vertex_indices\nend_header\n";
    for (size_t i = 0; i < m_Vertices.size(); i++) {
        f.write((const char*)&m_Vertices[i], 12);
        if (hasColors) {
            unsigned char c[4];
            for (int k = 0; k < 4; k++) c[k] = (unsigned char)(int)std::min(255.0f, std::max(0.0f, m_Colors[4 * i + k] * 255)); // vec4uc(c * 255)
            f.write((const char*)c, 4);
        }
    }
    for (size_t i = 0; i < nFaces; i++) {
        const unsigned char three = 3;
        int idx[3];
        for (int k = 0; k < 3; k++) idx[k] = hasVertexIndices() ? (int)m_FaceIndicesVertices[3 * i + k] : (int)(3 * i + k);
        f.write((const char*)&three, 1);
        f.write((const char*)idx, 12);
    }
    if (!f) throw vh::Error(VH_ERR_IO, "write failed: " + filename);
}

// ---------------------------------------------------------------------------
// CUDAMarchingCubesHashSDF
// ---------------------------------------------------------------------------

MarchingCubesParams CUDAMarchingCubesHashSDF::parameters(unsigned int marchingCubesMaxNumTriangles, float SDFMarchingCubeThreshFactor,
                                                         float SDFVoxelSize, unsigned int hashNumBuckets)
{
    MarchingCubesParams p;
    std::memset(&p, 0, sizeof(p));
    p.m_maxNumTriangles = marchingCubesMaxNumTriangles;
    p.m_threshMarchingCubes = SDFMarchingCubeThreshFactor * SDFVoxelSize;
    p.m_threshMarchingCubes2 = SDFMarchingCubeThreshFactor * SDFVoxelSize;
    p.m_sdfBlockSize = VH_SDF_BLOCK_SIZE;
    p.m_hashBucketSize = VH_HASH_BUCKET_SIZE;
    p.m_hashNumBuckets = hashNumBuckets;
    return p;
}

CUDAMarchingCubesHashSDF::CUDAMarchingCubesHashSDF(const MarchingCubesParams& params, vhStream_t stream)
    : m_params(params), m_stream(stream), m_offline(false)
{
    std::memset(&m_data, 0, sizeof(m_data));
    check(vh_marching_cubes_data_alloc(&m_data, &m_params), "MarchingCubesData::allocate");
    check(vh_reset_marching_cubes(&m_data, m_stream), "resetMarchingCubesCUDA");
}

CUDAMarchingCubesHashSDF::~CUDAMarchingCubesHashSDF()
{
    (void)hipStreamSynchronize((hipStream_t)m_stream);
    vh_marching_cubes_data_free(&m_data);
}

unsigned int CUDAMarchingCubesHashSDF::getNumTriangles()
{
    unsigned int n = 0;
    checkHip(hipMemcpyAsync(&n, m_data.d_numTriangles, sizeof(n), hipMemcpyDeviceToHost, (hipStream_t)m_stream), "numTriangles");
    checkHip(hipStreamSynchronize((hipStream_t)m_stream), "numTriangles");
    return n;
}

unsigned int CUDAMarchingCubesHashSDF::getNumOccupiedBlocks()
{
    unsigned int n = 0;
    checkHip(hipMemcpyAsync(&n, m_data.d_numOccupiedBlocks, sizeof(n), hipMemcpyDeviceToHost, (hipStream_t)m_stream), "numOccupiedBlocks");
    checkHip(hipStreamSynchronize((hipStream_t)m_stream), "numOccupiedBlocks");
    return n;
}

void CUDAMarchingCubesHashSDF::downloadTriangles(VhTriangle* out, unsigned int n)
{
    if (n == 0) return;
    if (!out || n > m_params.m_maxNumTriangles) throw vh::Error(VH_ERR_BAD_ARGUMENT, "downloadTriangles");
    checkHip(hipMemcpyAsync(out, m_data.d_triangles, sizeof(VhTriangle) * (size_t)n, hipMemcpyDeviceToHost, (hipStream_t)m_stream), "triangles");
    checkHip(hipStreamSynchronize((hipStream_t)m_stream), "triangles");
}

// .cpp:194-224
void CUDAMarchingCubesHashSDF::extractIsoSurfaceWithoutCopy(const HashData& hashData, const HashParams& hashParams, const vh::vec3f& minCorner,
                                                            const vh::vec3f& maxCorner, bool boxEnabled)
{
    check(vh_reset_marching_cubes(&m_data, m_stream), "resetMarchingCubesCUDA");
    m_params.m_maxCorner[0] = maxCorner.x; m_params.m_maxCorner[1] = maxCorner.y; m_params.m_maxCorner[2] = maxCorner.z;
    m_params.m_minCorner[0] = minCorner.x; m_params.m_minCorner[1] = minCorner.y; m_params.m_minCorner[2] = minCorner.z;
    m_params.m_boxEnabled = boxEnabled ? 1u : 0u;
    check(vh_marching_cubes_update_params(&m_data, &m_params, m_stream), "MarchingCubesData::updateParams");
    check(vh_extract_iso_surface_pass1(&hashData, &hashParams, &m_data, m_stream), "extractIsoSurfacePass1CUDA");
    check(vh_extract_iso_surface_pass2(&hashData, &hashParams, &m_data, getNumOccupiedBlocks(), m_stream), "extractIsoSurfacePass2CUDA");
}

void CUDAMarchingCubesHashSDF::extractIsoSurface(const HashData& hashData, const HashParams& hashParams, const vh::vec3f& minCorner,
                                                 const vh::vec3f& maxCorner, bool boxEnabled)
{
    extractIsoSurfaceWithoutCopy(hashData, hashParams, minCorner, maxCorner, boxEnabled);
    copyTrianglesToCPU();
}

// .cpp:31-86
void CUDAMarchingCubesHashSDF::copyTrianglesToCPU()
{
    const unsigned int nTriangles = getNumTriangles();
    if (nTriangles >= m_params.m_maxNumTriangles)
        throw vh::Error(VH_ERR_STAGING_OVERFLOW, "not enough memory to store triangles for chunk; increase s_marchingCubesMaxNumTriangles");
    if (nTriangles == 0) return;
    std::vector<VhTriangle> tris(nTriangles);
    downloadTriangles(tris.data(), nTriangles);
    vh::MeshData md;
    md.m_Vertices.resize(3 * (size_t)nTriangles);
    md.m_Colors.resize(12 * (size_t)nTriangles);
    const VhVertex* vc = reinterpret_cast<const VhVertex*>(tris.data());
    for (size_t i = 0; i < 3 * (size_t)nTriangles; i++) {
        md.m_Vertices[i] = { vc[i].p[0], vc[i].p[1], vc[i].p[2] };
        md.m_Colors[4 * i + 0] = vc[i].c[0]; md.m_Colors[4 * i + 1] = vc[i].c[1]; md.m_Colors[4 * i + 2] = vc[i].c[2]; md.m_Colors[4 * i + 3] = 1.0f;
    }
    if (!m_offline) {
        // triangle soup appended as it is
        if (m_meshData.hasVertexIndices()) { md.makeTriangleSoupIndices(); m_meshData.merge(md); }
        else {
            m_meshData.m_Vertices.insert(m_meshData.m_Vertices.end(), md.m_Vertices.begin(), md.m_Vertices.end());
            m_meshData.m_Colors.insert(m_meshData.m_Colors.end(), md.m_Colors.begin(), md.m_Colors.end());
        }
    } else {
        // "some sequences exhaust cpu memory... -> merge first"
        md.makeTriangleSoupIndices();
        md.mergeCloseVertices(0.0001f);
        md.removeDuplicateFaces();
        if (!md.m_FaceIndicesVertices.empty()) m_meshData.merge(md);
    }
}

// .cpp:89-145 (the reference appends a numeric suffix instead of overwriting unless told to)
void CUDAMarchingCubesHashSDF::saveMesh(const std::string& filename, const vh::mat4f* transform, bool overwriteExistingFile)
{
    std::string actual = filename;
    if (!overwriteExistingFile) {
        unsigned int num = 0;
        for (;;) {
            std::ifstream probe(actual, std::ios::binary);
            if (!probe) break;
            const size_t dot = filename.find_last_of('.');
            const std::string stem = dot == std::string::npos ? filename : filename.substr(0, dot);
            const std::string ext = dot == std::string::npos ? "" : filename.substr(dot);
            actual = stem + std::to_string(++num) + ext;
        }
    }
    if (!m_meshData.hasVertexIndices()) m_meshData.makeTriangleSoupIndices();
    m_meshData.mergeCloseVertices(0.0001f);
    m_meshData.removeDuplicateFaces();
    if (transform) m_meshData.applyTransform(*transform);
    m_meshData.saveToPLY(actual);
    clearMeshBuffer();
}

// .cpp:149-192
void CUDAMarchingCubesHashSDF::extractIsoSurface(CUDASceneRepChunkGrid& chunkGrid, const vh::vec3f& camPos, float radius)
{
    chunkGrid.stopMultiThreading();
    const vh::vec3i minGridPos = chunkGrid.getMinGridPos(), maxGridPos = chunkGrid.getMaxGridPos();
    clearMeshBuffer();
    chunkGrid.streamOutToCPUAll();
    for (int x = minGridPos.x; x < maxGridPos.x; x++)
        for (int y = minGridPos.y; y < maxGridPos.y; y++)
            for (int z = minGridPos.z; z < maxGridPos.z; z++) {
                const vh::vec3i chunk = { x, y, z };
                if (!chunkGrid.containsSDFBlocksChunk(chunk)) continue;
                chunkGrid.streamInToGPUChunkNeighborhood(chunk, 1);
                const vh::vec3f c = chunkGrid.getWorldPosChunk(chunk), e = chunkGrid.getVoxelExtends();
                const HashParams hp = chunkGrid.getHashParams();
                const float pad = hp.m_virtualVoxelSize * (float)hp.m_SDFBlockSize;
                const vh::vec3f minCorner = { c.x - e.x / 2.0f - pad, c.y - e.y / 2.0f - pad, c.z - e.z / 2.0f - pad };
                const vh::vec3f maxCorner = { c.x + e.x / 2.0f + pad, c.y + e.y / 2.0f + pad, c.z + e.z / 2.0f + pad };
                extractIsoSurface(chunkGrid.getHashData(), hp, minCorner, maxCorner, true);
                chunkGrid.streamOutToCPUAll();
            }
    unsigned int nStreamedBlocks = 0;
    chunkGrid.streamInToGPUAll(camPos, radius, true, nStreamedBlocks);
    chunkGrid.startMultiThreading();
}
