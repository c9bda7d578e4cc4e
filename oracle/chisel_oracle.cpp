/*
 * chisel_oracle.cpp -- CPU ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Plain C++ restatement of the OpenChisel dense-TSDF path vendored in z619850002/CVIDS,
 * following /root/reference/OpenChisel/open_chisel line by line.  "ref:" comments give the
 * reference file:line each block restates (paths relative to OpenChisel/open_chisel/).
 *
 * Pinning status: see chisel_oracle.h.  Build: oracle/Makefile (g++ -O3 -ffp-contract=off,
 * no -march: the reference's flags, catkin.cmake:10-12).
 *
 * "Faithful mode" is the only mode: whole-frustum-AABB candidate enumeration, allocate every
 * missing candidate, integrate, erase new-and-untouched chunks, 16-byte AoS voxels, 16
 * std::threads over static blocks for the colour path, serial depth-only path.  That is what
 * makes it usable as the timed CPU baseline.
 *
 * fp32 operation order: Eigen >= 3.3 evaluates fixed-size 3-term reductions (dot(), the rows
 * of a 3x3 * 3x1 lazy product, squaredNorm()) as a0 + (a1 + a2) (redux_novec_unroller splits
 * [0,3) into [0,1) and [1,3)); Eigen 3.2's coefficient product was ((a0 + a1) + a2).  The
 * reference does not pin an Eigen version (catkin.cmake:7); this oracle fixes the 3.3 order.
 *
 * ALTERNATIVE READINGS (risk assessment only, tests/test_oracle_readings.py): three build switches turn the choices the
 * oracle had to make without being able to compile the reference's Eigen code into the other plausible reading --
 *   -DORACLE_ALT_SUM32    3-term reductions as (a0 + a1) + a2 (Eigen 3.2)
 *   -DORACLE_ALT_SQRTF    the band's voxel diagonal through sqrtf (float) instead of ::sqrt(double)
 *   -DORACLE_ALT_TRIGF    the frustum's atan2 / tan in float instead of double
 * so that the distance between the readings can be measured on the BASELINE scenes.  The product follows the default.
 */
#include "chisel_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "mc_table_data.inc"

namespace {

// ---------------------------------------------------------------------------------------------
// minimal fixed-size algebra with Eigen's evaluation order made explicit
// ---------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator()(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &operator()(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(const V3 &a, const V3 &b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3 &a, const V3 &b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(const V3 &a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(const V3 &a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, const V3 &a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(const V3 &a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
// Eigen 3.3 redux order for 3 terms
#ifdef ORACLE_ALT_SUM32
inline float sum3(float a0, float a1, float a2) { return (a0 + a1) + a2; }
#else
inline float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }
#endif
inline float dot(const V3 &a, const V3 &b) { return sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 cross(const V3 &a, const V3 &b) {
    // Eigen cross3: (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0)
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float squaredNorm(const V3 &a) { return sum3(a.x * a.x, a.y * a.y, a.z * a.z); }
inline float norm(const V3 &a) { return std::sqrt(squaredNorm(a)); }
inline V3 normalized(const V3 &a) {  // Eigen 3.3 MatrixBase::normalized(): z>0 ? n/sqrt(z) : n
    float z = squaredNorm(a);
    if (z > 0.0f) return a / std::sqrt(z);
    return a;
}

struct I3 {
    int x, y, z;
    I3() : x(0), y(0), z(0) {}
    I3(int a, int b, int c) : x(a), y(b), z(c) {}
    bool operator==(const I3 &o) const { return x == o.x && y == o.y && z == o.z; }
};
inline I3 operator+(const I3 &a, const I3 &b) { return I3(a.x + b.x, a.y + b.y, a.z + b.z); }

// camera->world rigid transform (Eigen::Affine3f): linear() = R (row-major here), translation() = t
struct Pose {
    float R[3][3];
    V3 t;
    static Pose fromRowMajor3x4(const float *p) {
        Pose q;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) q.R[r][c] = p[r * 4 + c];
        q.t = V3(p[3], p[7], p[11]);
        return q;
    }
    V3 col(int c) const { return V3(R[0][c], R[1][c], R[2][c]); }
    // linear().transpose() * v : row i of R^T is column i of R
    V3 transposeMul(const V3 &v) const {
        return V3(sum3(R[0][0] * v.x, R[1][0] * v.y, R[2][0] * v.z),
                  sum3(R[0][1] * v.x, R[1][1] * v.y, R[2][1] * v.z),
                  sum3(R[0][2] * v.x, R[1][2] * v.y, R[2][2] * v.z));
    }
};

// ---------------------------------------------------------------------------------------------
// voxels.  ref: DistVoxel.h:33-77, DistVoxel.cpp:27-31, ColorVoxel.h:33-100, ColorVoxel.cpp:27-31
// Both reference classes carry a vptr (virtual destructors) -> sizeof == 16; the pad member
// reproduces that footprint so the CPU baseline moves the same bytes.
// ---------------------------------------------------------------------------------------------
struct DistVoxel {
    void *vptr_pad;
    float sdf;
    float weight;
    DistVoxel() : vptr_pad(nullptr), sdf(99999), weight(0) {}
    inline void Integrate(const float &distUpdate, const float &weightUpdate) {  // DistVoxel.h:52-60
        float oldSDF = sdf;
        float oldWeight = weight;
        float newDist = (oldWeight * oldSDF + weightUpdate * distUpdate) / (weightUpdate + oldWeight);
        sdf = newDist;
        weight = oldWeight + weightUpdate;
    }
    inline void Reset() {  // DistVoxel.h:68-72 ; Carve() == Reset() (:62-66)
        sdf = 99999;
        weight = 0;
    }
};

struct ColorVoxel {
    void *vptr_pad;
    uint8_t red, green, blue, weight;
    ColorVoxel() : vptr_pad(nullptr), red(0), green(0), blue(0), weight(0) {}
    static inline float Saturate(float value) { return std::min(std::max(value, 0.0f), 255.0f); }
    inline void Integrate(const uint8_t &newRed, const uint8_t &newGreen, const uint8_t &newBlue,
                          const uint8_t &weightUpdate) {  // ColorVoxel.h:65-85
        if (weight >= std::numeric_limits<uint8_t>::max() - weightUpdate) return;
        float oldRed = static_cast<float>(red);
        float updatedRed = Saturate(static_cast<float>(weight * oldRed + weightUpdate * newRed) / (weightUpdate + weight));
        red = static_cast<uint8_t>(updatedRed);
        float oldGreen = static_cast<float>(green);
        float updatedGreen = Saturate(static_cast<float>(weight * oldGreen + weightUpdate * newGreen) / (weightUpdate + weight));
        green = static_cast<uint8_t>(updatedGreen);
        float oldBlue = static_cast<float>(blue);
        float updatedBlue = Saturate(static_cast<float>(weight * oldBlue + weightUpdate * newBlue) / (weightUpdate + weight));
        blue = static_cast<uint8_t>(updatedBlue);
        weight = weight + weightUpdate;
    }
};

// ---------------------------------------------------------------------------------------------
// truncation / weighting strategies.  ref: truncation/*.h, weighting/ConstantWeighter.h:43-46
// ---------------------------------------------------------------------------------------------
struct Truncator {
    int kind;
    float param;  // Constant: distance; Inverse/Quadratic: scalingFactor
    // InverseTruncator.h:48-52 member constants
    const float BASE_LINE = 0.10;
    const float FOCAL = 471.27;
    const float DEP_SAMPLE = 1.0f / (BASE_LINE * FOCAL);
    // QuadraticTruncator.h:65-67
    const float quadraticTerm = 0.0019 * 10;
    const float linearTerm = 0.00152 * 10;
    const float constantTerm = 0.001504 * 10;
    Truncator(int k, float p) : kind(k), param(p) {}
    float GetTruncationDistance(float reading) const {
        switch (kind) {
            case OC_TRUNC_CONSTANT:  // ConstantTruncator.h:48-51
                return param;
            case OC_TRUNC_INVERSE: {  // InverseTruncator.h:42-46
                float inv_reading = 1.0 / reading;
                return (DEP_SAMPLE / (inv_reading * inv_reading)) * param;
            }
            default:  // QuadraticTruncator.h:42-45 (double arithmetic through pow())
                return std::abs(quadraticTerm * ::pow((double)reading, 2.0) + linearTerm * reading + constantTerm) * param;
        }
    }
};
inline float ConstantWeight(float weight, float /*surfaceDist*/, float truncationDist) {
    return weight / (5 * truncationDist);  // ConstantWeighter.h:43-46
}

// ---------------------------------------------------------------------------------------------
// camera.  ref: camera/PinholeCamera.cpp:38-64, camera/Intrinsics.h:40-47
// ---------------------------------------------------------------------------------------------
struct Camera {
    float fx, fy, cx, cy;
    int width, height;
    float nearPlane, farPlane;
    V3 ProjectPoint(const V3 &point) const {  // PinholeCamera.cpp:38-45
        const float invZ = 1.0f / point.z;
        return V3(fx * point.x * invZ + cx, fy * point.y * invZ + cy, point.z);
    }
    bool IsPointOnImage(const V3 &point) const {  // PinholeCamera.cpp:61-64
        return point.x >= 0 && point.y >= 0 && point.x < width && point.y < height;
    }
};

// images.  ref: camera/DepthImage.h:54-77, camera/ColorImage.h:66-107
struct DepthView {
    const float *data;
    int width, height;
    float DepthAt(int row, int col) const { return data[col + row * width]; }
};
struct ColorView {
    const uint8_t *data;
    int width, height, numChannels;
    void At(int row, int col, uint8_t *rgba) const {  // ColorImage.h:72-107
        const int index = (col + row * width) * numChannels;
        switch (numChannels) {
            case 1:
                rgba[0] = data[index]; rgba[1] = rgba[0]; rgba[2] = rgba[0]; rgba[3] = rgba[0];
                break;
            case 2:
                rgba[0] = data[index]; rgba[1] = data[index + 1]; rgba[2] = rgba[1]; rgba[3] = rgba[1];
                break;
            case 3:
                rgba[0] = data[index + 2]; rgba[1] = data[index + 1]; rgba[2] = data[index]; rgba[3] = rgba[0];
                break;
            case 4:
                rgba[0] = data[index + 2]; rgba[1] = data[index + 1]; rgba[2] = data[index]; rgba[3] = data[index + 3];
                break;
            default:
                break;  // reference leaves the colour uninitialised; callers never use other counts
        }
    }
};

// ---------------------------------------------------------------------------------------------
// planes / frustum.  ref: geometry/Plane.cpp:44-52, geometry/Frustum.cpp:41-79,101-122,143-219
// ---------------------------------------------------------------------------------------------
struct Plane {
    V3 normal;
    float distance;
    Plane() : distance(0) {}
    Plane(const V3 &a, const V3 &b, const V3 &c) {  // Plane.cpp:44-52
        V3 ab = b - a;
        V3 ac = c - a;
        V3 cr = cross(ab, ac);
        normal = normalized(cr);
        distance = -(dot(cr, a));  // sic: un-normalised cross
    }
};
struct AABB {
    V3 min, max;
};
struct Frustum {
    V3 corners[8];
    Plane top, left, right, bottom, near_, far_;

    void SetFromVectors(const V3 &forward, const V3 &pos, const V3 &rightVec, const V3 &up, float nearDist,
                        float farDist, float fov, float aspect) {  // Frustum.cpp:155-219
#ifdef ORACLE_ALT_TRIGF
        float angleTangent = tanf(fov / 2);
#else
        float angleTangent = ::tan((double)(fov / 2));  // ::tan(double), narrowed
#endif
        float heightFar = angleTangent * farDist;
        float widthFar = heightFar * aspect;
        float heightNear = angleTangent * nearDist;
        float widthNear = heightNear * aspect;
        V3 farCenter = pos + forward * farDist;
        V3 farTopLeft = farCenter + (up * heightFar) - (rightVec * widthFar);
        V3 farTopRight = farCenter + (up * heightFar) + (rightVec * widthFar);
        V3 farBotLeft = farCenter - (up * heightFar) - (rightVec * widthFar);
        V3 farBotRight = farCenter - (up * heightFar) + (rightVec * widthFar);
        V3 nearCenter = pos + forward * nearDist;
        V3 nearTopLeft = nearCenter + (up * heightNear) - (rightVec * widthNear);
        V3 nearTopRight = nearCenter + (up * heightNear) + (rightVec * widthNear);
        V3 nearBotLeft = nearCenter - (up * heightNear) - (rightVec * widthNear);
        V3 nearBotRight = nearCenter - (up * heightNear) + (rightVec * widthNear);
        near_ = Plane(nearBotLeft, nearTopLeft, nearBotRight);
        far_ = Plane(farTopRight, farTopLeft, farBotRight);
        left = Plane(farTopLeft, nearTopLeft, farBotLeft);
        right = Plane(nearTopRight, farTopRight, nearBotRight);
        top = Plane(nearTopLeft, farTopLeft, nearTopRight);
        bottom = Plane(nearBotRight, farBotLeft, nearBotLeft);
        corners[0] = farTopLeft;
        corners[1] = farTopRight;
        corners[2] = farBotLeft;
        corners[3] = farBotRight;
        corners[4] = nearBotRight;
        corners[5] = nearTopLeft;
        corners[6] = nearTopRight;
        corners[7] = nearBotLeft;
    }
    void SetFromParams(const Pose &view, float nearDist, float farDist, float fx, float fy, float /*cx*/,
                       float cy, float imgWidth, float imgHeight) {  // Frustum.cpp:143-153
        V3 right_ = view.col(0);
        V3 up = -view.col(1);
        V3 d = view.col(2);
        V3 p = view.t;
        float aspect = (fx * imgWidth) / (fy * imgHeight);
#ifdef ORACLE_ALT_TRIGF
        float fov = atan2f(cy, fy) + atan2f(imgHeight - cy, fy);
#else
        float fov = ::atan2((double)cy, (double)fy) + ::atan2((double)(imgHeight - cy), (double)fy);  // double, narrowed
#endif
        SetFromVectors(d, p, right_, up, nearDist, farDist, fov, aspect);
    }
    void ComputeBoundingBox(AABB *box) const {  // Frustum.cpp:101-122
        float bigNum = std::numeric_limits<float>::max();
        V3 tempMin(bigNum, bigNum, bigNum);
        V3 tempMax(-bigNum, -bigNum, -bigNum);
        for (int i = 0; i < 8; i++) {
            const V3 &corner = corners[i];
            tempMin.x = std::min<float>(tempMin.x, corner.x);
            tempMin.y = std::min<float>(tempMin.y, corner.y);
            tempMin.z = std::min<float>(tempMin.z, corner.z);
            tempMax.x = std::max<float>(tempMax.x, corner.x);
            tempMax.y = std::max<float>(tempMax.y, corner.y);
            tempMax.z = std::max<float>(tempMax.z, corner.z);
        }
        box->min = tempMin;
        box->max = tempMax;
    }
    bool Intersects(const AABB &box) const {  // Frustum.cpp:41-79 (returns true on the FIRST plane that passes)
        const Plane *planes[] = {&far_, &near_, &top, &bottom, &left, &right};
        for (const Plane *plane : planes) {
            V3 axisVert;
            const V3 &normal = plane->normal;
            axisVert.x = (normal.x < 0.0f) ? box.min.x : box.max.x;
            axisVert.y = (normal.y < 0.0f) ? box.min.y : box.max.y;
            axisVert.z = (normal.z < 0.0f) ? box.min.z : box.max.z;
            if (dot(axisVert, normal) + plane->distance > 0.0f) return true;
        }
        return false;
    }
};
// PinholeCamera::SetupFrustum passes fy for BOTH focal arguments (PinholeCamera.cpp:55-59)
inline void SetupFrustum(const Camera &cam, const Pose &view, Frustum *frustum) {
    frustum->SetFromParams(view, cam.nearPlane, cam.farPlane, cam.fy, cam.fy, cam.cx, cam.cy, cam.width, cam.height);
}

// ---------------------------------------------------------------------------------------------
// chunks.  ref: Chunk.h:47-140, Chunk.cpp:33-44,72-86,118-136
// ---------------------------------------------------------------------------------------------
struct Chunk {
    I3 ID;
    I3 numVoxels;
    float voxelResolutionMeters;
    std::vector<DistVoxel> voxels;
    std::vector<ColorVoxel> colors;
    V3 origin;
    Chunk(const I3 &id, const I3 &nv, float r, bool useColor) : ID(id), numVoxels(nv), voxelResolutionMeters(r) {
        int total = nv.x * nv.y * nv.z;
        voxels.resize(total, DistVoxel());          // Chunk.cpp:51-56
        if (useColor) colors.resize(total, ColorVoxel());  // Chunk.cpp:58-63
        origin = V3(numVoxels.x * ID.x * voxelResolutionMeters, numVoxels.y * ID.y * voxelResolutionMeters,
                    numVoxels.z * ID.z * voxelResolutionMeters);  // Chunk.cpp:43
    }
    int GetTotalNumVoxels() const { return numVoxels.x * numVoxels.y * numVoxels.z; }
    // Chunk.h:81-84 (uses numVoxels(2) where numVoxels(1) is meant)
    int GetVoxelID(int x, int y, int z) const { return (z * numVoxels.z + y) * numVoxels.x + x; }
    bool IsCoordValid(int x, int y, int z) const {
        return (x >= 0 && x < numVoxels.x && y >= 0 && y < numVoxels.y && z >= 0 && z < numVoxels.z);
    }
    I3 GetVoxelCoords(const V3 &worldCoords) const {  // Chunk.cpp:72-81
        const float rf = 1.0f / (voxelResolutionMeters);
        return I3(static_cast<int>(std::floor(worldCoords.x * rf)), static_cast<int>(std::floor(worldCoords.y * rf)),
                  static_cast<int>(std::floor(worldCoords.z * rf)));
    }
    AABB ComputeBoundingBox() const {  // Chunk.cpp:65-70
        V3 size = V3((float)numVoxels.x, (float)numVoxels.y, (float)numVoxels.z) * voxelResolutionMeters;
        AABB b;
        b.min = origin;
        b.max = origin + size;
        return b;
    }
    V3 GetColorAt(const V3 &pos) const {  // Chunk.cpp:118-136 ; AABB::Contains geometry/AABB.h
        AABB b = ComputeBoundingBox();
        bool contains = pos.x >= b.min.x && pos.y >= b.min.y && pos.z >= b.min.z && pos.x <= b.max.x &&
                        pos.y <= b.max.y && pos.z <= b.max.z;
        if (contains) {
            V3 chunkPos = (pos - origin) / voxelResolutionMeters;
            int chunkX = static_cast<int>(chunkPos.x);
            int chunkY = static_cast<int>(chunkPos.y);
            int chunkZ = static_cast<int>(chunkPos.z);
            if (IsCoordValid(chunkX, chunkY, chunkZ)) {
                const ColorVoxel &color = colors.at(GetVoxelID(chunkX, chunkY, chunkZ));
                float maxVal = 255.0f;
                return V3(static_cast<float>(color.red) / maxVal, static_cast<float>(color.green) / maxVal,
                          static_cast<float>(color.blue) / maxVal);
            }
        }
        return V3(0, 0, 0);
    }
};
typedef std::shared_ptr<Chunk> ChunkPtr;

struct ChunkHasher {  // ChunkManager.h:40-52
    static constexpr size_t p1 = 73856093;
    static constexpr size_t p2 = 19349663;
    static constexpr size_t p3 = 8349279;
    std::size_t operator()(const I3 &key) const { return (key.x * p1 ^ key.y * p2 ^ key.z * p3); }
};
typedef std::unordered_map<I3, ChunkPtr, ChunkHasher> ChunkMap;
typedef std::unordered_map<I3, bool, ChunkHasher> ChunkSet;

struct Mesh {  // mesh/Mesh.h:54-58
    std::vector<V3> vertices;
    std::vector<size_t> indices;
    std::vector<V3> normals;
    std::vector<V3> colors;
    std::vector<V3> grids;
    void Clear() {
        vertices.clear(); indices.clear(); normals.clear(); colors.clear(); grids.clear();
    }
};
typedef std::shared_ptr<Mesh> MeshPtr;
typedef std::unordered_map<I3, MeshPtr, ChunkHasher> MeshMap;

// ---------------------------------------------------------------------------------------------
// marching cubes.  ref: marching_cubes/MarchingCubes.h:41-146, MarchingCubes.cpp:29-305
// ---------------------------------------------------------------------------------------------
struct TriTable {
    int rows[256][16];
    TriTable() {
        for (int c = 0; c < 256; c++) {
            const char *s = kOracleTriCases[c];
            int k = 0;
            for (; s[k]; k++) rows[c][k] = (s[k] <= '9') ? (s[k] - '0') : (s[k] - 'a' + 10);
            for (; k < 16; k++) rows[c][k] = -1;
        }
    }
};
const TriTable &triTable() {
    static const TriTable t;
    return t;
}
inline int CalculateVertexConfiguration(const float *s) {  // MarchingCubes.h:108-118
    int idx = 0;
    for (int i = 0; i < 8; i++) idx |= (s[i] < 0 ? (1 << i) : 0);
    return idx;
}
inline bool IsOccupied(const float *s) { return triTable().rows[CalculateVertexConfiguration(s)][0] != -1; }
inline V3 InterpolateVertex(const V3 &vertex1, const V3 &vertex2, const float &sdf1, const float &sdf2) {
    // MarchingCubes.h:135-146 ; note "vertex1 + 0.5 * vertex2" (sic) in the degenerate branch
    const float minDiff = 1e-6;
    const float sdfDiff = sdf1 - sdf2;
    if (fabs(sdfDiff) < minDiff) return vertex1 + 0.5f * vertex2;
    const float t = sdf1 / sdfDiff;
    return vertex1 + t * (vertex2 - vertex1);
}
inline void MeshCube(const V3 *vertexCoords, const float *vertexSDF, size_t *nextIDX, Mesh *mesh) {
    // MarchingCubes.h:73-106
    const int index = CalculateVertexConfiguration(vertexSDF);
    V3 edgeCoords[12];
    for (int i = 0; i < 12; ++i) {  // InterpolateEdgeVertices :120-131
        const int e0 = kOracleEdgeCorners[i][0];
        const int e1 = kOracleEdgeCorners[i][1];
        if ((vertexSDF[e0] < 0 && vertexSDF[e1] >= 0) || (vertexSDF[e0] >= 0 && vertexSDF[e1] < 0))
            edgeCoords[i] = InterpolateVertex(vertexCoords[e0], vertexCoords[e1], vertexSDF[e0], vertexSDF[e1]);
    }
    const int *table_row = triTable().rows[index];
    int table_col = 0;
    while (table_row[table_col] != -1) {
        mesh->vertices.push_back(edgeCoords[table_row[table_col + 2]]);
        mesh->vertices.push_back(edgeCoords[table_row[table_col + 1]]);
        mesh->vertices.push_back(edgeCoords[table_row[table_col]]);
        mesh->indices.push_back(*nextIDX);
        mesh->indices.push_back((*nextIDX) + 1);
        mesh->indices.push_back((*nextIDX) + 2);
        const V3 &p0 = mesh->vertices[*nextIDX];
        const V3 &p1 = mesh->vertices[*nextIDX + 1];
        const V3 &p2 = mesh->vertices[*nextIDX + 2];
        V3 px = (p1 - p0);
        V3 py = (p2 - p0);
        V3 n = normalized(cross(px, py));
        mesh->normals.push_back(n);
        mesh->normals.push_back(n);
        mesh->normals.push_back(n);
        *nextIDX += 3;
        table_col += 3;
    }
}

// cubeIndexOffsets.  ref: ChunkManager.cpp:67-69
const int kCubeOff[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};

}  // namespace

// ---------------------------------------------------------------------------------------------
// the map: ChunkManager + ProjectionIntegrator + Chisel in one object
// ---------------------------------------------------------------------------------------------
struct CloudTally {
    uint64_t hits = 0, sdf = 0, carved = 0;
};

struct oc_map {
    // ChunkManager state (ChunkManager.h:205-211)
    ChunkMap chunks;
    I3 chunkSize;
    float voxelResolutionMeters;
    std::vector<V3> centroids;
    MeshMap allMeshes;
    bool useColor;
    // Chisel state (Chisel.h:227-228)
    ChunkSet meshesToUpdate;
    int updateMeshesCalls;  // Chisel.cpp:53 "static int cnt"
    // ProjectionIntegrator state (ProjectionIntegrator.h:224-228)
    Truncator truncator;
    float weighterWeight;
    float carvingDist;
    bool enableVoxelCarving;
    int nThreads;
    // instrumentation (not in the reference)
    uint64_t counters[OC_NUM_COUNTERS];
    double phaseMs[4];

    oc_map(const I3 &size, float res, bool color)
        : chunkSize(size), voxelResolutionMeters(res), useColor(color), updateMeshesCalls(0),
          truncator(OC_TRUNC_INVERSE, 8.0f), weighterWeight(1.0f), carvingDist(0.05f), enableVoxelCarving(true),
          nThreads(16) {
        CacheCentroids();
        memset(counters, 0, sizeof(counters));
        memset(phaseMs, 0, sizeof(phaseMs));
    }

    void CacheCentroids() {  // ChunkManager.cpp:50-66
        V3 halfResolution = V3(voxelResolutionMeters, voxelResolutionMeters, voxelResolutionMeters) * 0.5f;
        centroids.resize(static_cast<size_t>(chunkSize.x * chunkSize.y * chunkSize.z));
        int i = 0;
        for (int z = 0; z < chunkSize.z; z++)
            for (int y = 0; y < chunkSize.y; y++)
                for (int x = 0; x < chunkSize.x; x++) {
                    centroids[i] = V3((float)x, (float)y, (float)z) * voxelResolutionMeters + halfResolution;
                    i++;
                }
    }
    bool HasChunk(const I3 &id) const { return chunks.find(id) != chunks.end(); }
    ChunkPtr GetChunk(const I3 &id) const { return chunks.at(id); }
    ChunkMap::iterator CreateChunk(const I3 &id) {  // ChunkManager.cpp:171-174, ChunkManager.h:89-92
        return chunks.insert(std::make_pair(id, std::make_shared<Chunk>(id, chunkSize, voxelResolutionMeters, useColor))).first;
    }
    // ChunkManager.h:136-145.  The reference caches the factors in function-local statics captured
    // from the FIRST manager of the process; per-instance here (identical for one map per process).
    I3 GetIDAt(const V3 &pos) const {
        const float rfx = 1.0f / (chunkSize.x * voxelResolutionMeters);
        const float rfy = 1.0f / (chunkSize.y * voxelResolutionMeters);
        const float rfz = 1.0f / (chunkSize.z * voxelResolutionMeters);
        return I3(static_cast<int>(std::floor(pos.x * rfx)), static_cast<int>(std::floor(pos.y * rfy)),
                  static_cast<int>(std::floor(pos.z * rfz)));
    }
    ChunkPtr GetChunkAt(const V3 &pos) const {
        I3 id = GetIDAt(pos);
        auto it = chunks.find(id);
        return it == chunks.end() ? ChunkPtr() : it->second;
    }

    void GetChunkIDsIntersecting(const Frustum &frustum, std::vector<I3> *chunkList) const {  // ChunkManager.cpp:182-212
        AABB frustumAABB;
        frustum.ComputeBoundingBox(&frustumAABB);
        I3 minID = GetIDAt(frustumAABB.min);
        I3 maxID = GetIDAt(frustumAABB.max) + I3(1, 1, 1);
        for (int x = minID.x - 1; x <= maxID.x + 1; x++)
            for (int y = minID.y - 1; y <= maxID.y + 1; y++)
                for (int z = minID.z - 1; z <= maxID.z + 1; z++) {
                    V3 mn = V3((float)(x * chunkSize.x), (float)(y * chunkSize.y), (float)(z * chunkSize.z)) * voxelResolutionMeters;
                    V3 mx = mn + V3((float)chunkSize.x, (float)chunkSize.y, (float)chunkSize.z) * voxelResolutionMeters;
                    AABB chunkBox;
                    chunkBox.min = mn;
                    chunkBox.max = mx;
                    if (frustum.Intersects(chunkBox)) chunkList->push_back(I3(x, y, z));
                }
    }

    struct Tally {
        uint64_t sdf = 0, col = 0, colsat = 0, probe = 0, carved = 0;
    };

    // ProjectionIntegrator::Integrate<float>.  ref: ProjectionIntegrator.h:51-99
    bool Integrate(const DepthView &depthImage, const Camera &camera, const Pose &cameraPose, Chunk *chunk,
                   bool chunkIsNew, Tally *tally) const {
        float resolution = chunk->voxelResolutionMeters;
        V3 origin = chunk->origin;
#ifdef ORACLE_ALT_SQRTF
        float diag = 2.0 * sqrtf(3.0f) * resolution;
#else
        float diag = 2.0 * ::sqrt((double)3.0f) * resolution;  // unqualified sqrt(float) binds to ::sqrt(double) under <cmath>
#endif
        V3 voxelCenter;
        bool updated = false;
        for (size_t i = 0; i < centroids.size(); i++) {
            voxelCenter = centroids[i] + origin;
            V3 voxelCenterInCamera = cameraPose.transposeMul(voxelCenter - cameraPose.t);
            V3 cameraPos = camera.ProjectPoint(voxelCenterInCamera);
            if (!camera.IsPointOnImage(cameraPos) || voxelCenterInCamera.z < 0) continue;
            float voxelDist = voxelCenterInCamera.z;
            float depth = depthImage.DepthAt((int)cameraPos.y, (int)cameraPos.x);
            if (depth > 50.) continue;
            float truncation = truncator.GetTruncationDistance(depth);
            float surfaceDist = depth - voxelDist;
            if (fabs(surfaceDist) < truncation + diag) {
                DistVoxel &voxel = chunk->voxels.at(i);
                voxel.Integrate(surfaceDist, 1.0f);
                updated = true;
                tally->sdf++;
            } else if (enableVoxelCarving && surfaceDist > truncation + carvingDist) {
                DistVoxel &voxel = chunk->voxels.at(i);
                if (!chunkIsNew) tally->probe++;
                if (voxel.weight > 0 && voxel.sdf < 1e-5) {
                    voxel.Reset();  // Carve()
                    updated = true;
                    tally->carved++;
                }
            }
        }
        return updated;
    }

    // ProjectionIntegrator::IntegrateColor<float,uint8_t>.  ref: ProjectionIntegrator.h:101-183
    bool IntegrateColor(const DepthView &depthImage, const Camera &depthCamera, const Pose &depthCameraPose,
                        const ColorView &colorImage, const Camera &colorCamera, const Pose &colorCameraPose,
                        Chunk *chunk, bool chunkIsNew, Tally *tally) const {
        float resolution = chunk->voxelResolutionMeters;
        V3 origin = chunk->origin;
#ifdef ORACLE_ALT_SQRTF
        float resolutionDiagonal = 2.0 * sqrtf(3.0f) * resolution;
#else
        float resolutionDiagonal = 2.0 * ::sqrt((double)3.0f) * resolution;
#endif
        bool updated = false;
        for (size_t i = 0; i < centroids.size(); i++) {
            uint8_t color[4] = {0, 0, 0, 0};
            V3 voxelCenter = centroids[i] + origin;
            V3 voxelCenterInCamera = depthCameraPose.transposeMul(voxelCenter - depthCameraPose.t);
            V3 cameraPos = depthCamera.ProjectPoint(voxelCenterInCamera);
            if (!depthCamera.IsPointOnImage(cameraPos) || voxelCenterInCamera.z < 0) continue;
            float voxelDist = voxelCenterInCamera.z;
            float depth = depthImage.DepthAt((int)cameraPos.y, (int)cameraPos.x);
            if (std::isnan(depth)) continue;
            float truncation = truncator.GetTruncationDistance(depth);
            float surfaceDist = depth - voxelDist;
            if (depth > 100.0f) continue;
            if (std::abs(surfaceDist) < truncation + resolutionDiagonal) {
                V3 voxelCenterInColorCamera = colorCameraPose.transposeMul(voxelCenter - colorCameraPose.t);
                V3 colorCameraPos = colorCamera.ProjectPoint(voxelCenterInColorCamera);
                if (colorCamera.IsPointOnImage(colorCameraPos)) {
                    ColorVoxel &colorVoxel = chunk->colors.at(i);
                    if (colorVoxel.weight < 8) {
                        int r = static_cast<int>(colorCameraPos.y);
                        int c = static_cast<int>(colorCameraPos.x);
                        colorImage.At(r, c, color);
                        colorVoxel.Integrate(color[0], color[1], color[2], 1);
                        tally->col++;
                    } else {
                        tally->colsat++;
                    }
                }
                DistVoxel &voxel = chunk->voxels.at(i);
                voxel.Integrate(surfaceDist, ConstantWeight(weighterWeight, surfaceDist, truncation));
                updated = true;
                tally->sdf++;
            } else if (enableVoxelCarving && surfaceDist > truncation + carvingDist) {
                DistVoxel &voxel = chunk->voxels.at(i);
                if (!chunkIsNew) tally->probe++;
                if (voxel.weight > 0 && voxel.sdf < 1e-5) {
                    if (voxel.weight < 5) {
                        voxel.Reset();  // Carve()
                    } else {
                        voxel.weight = voxel.weight - 1;
                    }
                    updated = true;
                    tally->carved++;
                }
            }
        }
        return updated;
    }

    // -----------------------------------------------------------------------------------------
    // point-cloud fusion mode.  ref: Chisel.cpp:107-157, ProjectionIntegrator.cpp:52-173,
    // ChunkManager.cpp:214-257, geometry/Raycast.cpp:4-128.  fp32 operation order of Eigen 3.3:
    //   Transform * Vec3 (Transform.h, rhs vector of size Dim): the point is extended to (x, y, z, 1) and multiplied
    //   by the 4x4 matrix column by column: ((m_i0 x + m_i1 y) + m_i2 z) + m_i3   (etor_product_packet_impl)
    //   Transform::inverse() (Affine): linear().inverse() by cofactors (InverseImpl.h compute_inverse<3>),
    //   translation = -(inverse_linear * t) with rows summed a0 + (a1 + a2).
    // -----------------------------------------------------------------------------------------
    struct Affine {  // row-major 3x4
        float m[3][4];
        V3 mul(const V3 &p) const {
            return V3(((m[0][0] * p.x + m[0][1] * p.y) + m[0][2] * p.z) + m[0][3],
                      ((m[1][0] * p.x + m[1][1] * p.y) + m[1][2] * p.z) + m[1][3],
                      ((m[2][0] * p.x + m[2][1] * p.y) + m[2][2] * p.z) + m[2][3]);
        }
        V3 translation() const { return V3(m[0][3], m[1][3], m[2][3]); }
        static Affine fromPose(const Pose &q) {
            Affine a;
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) a.m[r][c] = q.R[r][c];
                a.m[r][3] = q.t(r);
            }
            return a;
        }
        float cof(int i, int j) const {  // cofactor_3x3<i, j>
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
        }
        Affine inverse() const {
            Affine r;
            const float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
            const float det = sum3(c0 * m[0][0], c1 * m[1][0], c2 * m[2][0]);
            const float invdet = 1.0f / det;
            r.m[0][0] = c0 * invdet;
            r.m[0][1] = c1 * invdet;
            r.m[0][2] = c2 * invdet;
            r.m[1][0] = cof(0, 1) * invdet;
            r.m[1][1] = cof(1, 1) * invdet;
            r.m[1][2] = cof(2, 1) * invdet;
            r.m[2][0] = cof(0, 2) * invdet;
            r.m[2][1] = cof(1, 2) * invdet;
            r.m[2][2] = cof(2, 2) * invdet;
            for (int i = 0; i < 3; i++) r.m[i][3] = -sum3(r.m[i][0] * m[0][3], r.m[i][1] * m[1][3], r.m[i][2] * m[2][3]);
            return r;
        }
    };

    // Raycast.cpp:4-33.  The unqualified fmod() binds to ::fmod(double, double) (only <cmath> is in scope), so the
    // inner sum is taken in double and the result narrowed on return.
    static float rc_signum(int x) { return x == 0 ? 0 : x < 0 ? -1 : 1; }
    static float rc_mod(float value, float modulus) { return ::fmod(::fmod(value, modulus) + modulus, modulus); }
    static float rc_intbound(float s, int ds) {
        if (ds == 0) return std::numeric_limits<float>::infinity();  // (float)DBL_MAX on x86-64
        if (ds < 0) return rc_intbound(-s, -ds);
        s = rc_mod(s, 1.0f);
        return (1 - s) / ds;
    }
    // floor() of a coordinate to int as x86-64 does it (cvttss2si: INT_MIN for NaN and out-of-range values)
    static int rc_floor_to_int(float v) {
        const float f = std::floor(v);
        if (!(f >= -2147483648.0f && f < 2147483648.0f)) return std::numeric_limits<int>::min();
        return (int)f;
    }
    // Raycast.cpp:35-128.  Deviation: the reference loops until the end cell is reached, and when rounding lets an axis step
    // past its end coordinate it never returns.  Here the walk stops at the moment an axis would step past its end cell
    // (a walk that terminates in the reference never does that, so terminating walks are unchanged).
    // Rays with a coordinate that is not finite or beyond the int range are dropped (on x86-64 such coordinates all become
    // INT_MIN; with every coordinate affected -- the only way they arise from a point and a truncation -- the reference returns
    // at "stepX == 0 && ..." too).
    static void Raycast(const V3 &start, const V3 &end, const I3 &mn, const I3 &mx, std::vector<I3> *output) {
        const float c6[6] = {start.x, start.y, start.z, end.x, end.y, end.z};
        for (float c : c6)
            if (!(std::floor(c) >= -2147483648.0f && std::floor(c) < 2147483648.0f)) return;
        int x = rc_floor_to_int(start.x), y = rc_floor_to_int(start.y), z = rc_floor_to_int(start.z);
        const int endX = rc_floor_to_int(end.x), endY = rc_floor_to_int(end.y), endZ = rc_floor_to_int(end.z);
        const int dx = (int)((unsigned)endX - (unsigned)x), dy = (int)((unsigned)endY - (unsigned)y), dz = (int)((unsigned)endZ - (unsigned)z);
        const int stepX = (int)rc_signum(dx), stepY = (int)rc_signum(dy), stepZ = (int)rc_signum(dz);
        float tMaxX = rc_intbound(start.x, dx), tMaxY = rc_intbound(start.y, dy), tMaxZ = rc_intbound(start.z, dz);
        const float tDeltaX = ((float)stepX) / dx, tDeltaY = ((float)stepY) / dy, tDeltaZ = ((float)stepZ) / dz;
        if (stepX == 0 && stepY == 0 && stepZ == 0) return;
        while (true) {
            if (x >= mn.x && x < mx.x && y >= mn.y && y < mx.y && z >= mn.z && z < mx.z) output->push_back(I3(x, y, z));
            if (x == endX && y == endY && z == endZ) break;
            if (tMaxX < tMaxY) {
                if (tMaxX < tMaxZ) { if (x == endX) break; x += stepX; tMaxX += tDeltaX; }
                else { if (z == endZ) break; z += stepZ; tMaxZ += tDeltaZ; }
            } else {
                if (tMaxY < tMaxZ) { if (y == endY) break; y += stepY; tMaxY += tDeltaY; }
                else { if (z == endZ) break; z += stepZ; tMaxZ += tDeltaZ; }
            }
        }
    }

    // ChunkManager::GetChunkIDsIntersecting(cloud, ...).  ref: ChunkManager.cpp:214-257
    void GetChunkIDsIntersectingCloud(const V3 *points, size_t n, const Affine &cameraTransform, float truncation, float maxDist,
                                      std::vector<I3> *chunkList) const {
        chunkList->clear();
        const float roundX = 1.0f / (chunkSize.x * voxelResolutionMeters);
        const float roundY = 1.0f / (chunkSize.y * voxelResolutionMeters);
        const float roundZ = 1.0f / (chunkSize.z * voxelResolutionMeters);
        ChunkSet map;
        const int imax = std::numeric_limits<int>::max();
        const I3 minVal(-imax, -imax, -imax), maxVal(imax, imax, imax);
        std::vector<I3> intersectingChunks;
        for (size_t p = 0; p < n; p++) {
            V3 end = cameraTransform.mul(points[p]);
            V3 start = cameraTransform.translation();
            float len = norm(end - start);
            if (len > maxDist) continue;
            V3 dir = normalized(end - start);
            V3 truncStart = end - dir * truncation;
            V3 truncEnd = end + dir * truncation;
            V3 startInt(truncStart.x * roundX, truncStart.y * roundY, truncStart.z * roundZ);
            V3 endInt(truncEnd.x * roundX, truncEnd.y * roundY, truncEnd.z * roundZ);
            intersectingChunks.clear();
            Raycast(startInt, endInt, minVal, maxVal, &intersectingChunks);
            for (const I3 &id : intersectingChunks) map[id] = true;
        }
        for (const auto &it : map) chunkList->push_back(it.first);
    }

    // ProjectionIntegrator::IntegratePointCloud / IntegrateColorPointCloud.  ref: ProjectionIntegrator.cpp:52-112 / :114-173
    // (depth limit 2 m without colours, 5 m with; the colour index only advances on points that pass the limit, :68-70)
    bool IntegrateCloudChunk(const V3 *points, size_t n, const V3 *colors, const Affine &cameraPose, const Affine &inversePose,
                             Chunk *chunk, CloudTally *tally) const {
        const bool withColor = colors != nullptr && !chunk->colors.empty();  // ProjectionIntegrator.cpp:42
        const float depthLimit = withColor ? 5.0f : 2.f;
        const float roundX = 1.0f / chunk->voxelResolutionMeters;
        const float roundY = 1.0f / chunk->voxelResolutionMeters;
        const float roundZ = 1.0f / chunk->voxelResolutionMeters;
        std::vector<I3> raycastVoxels;
        const I3 chunkMin(0, 0, 0);
        const I3 chunkMax = chunk->numVoxels;
        bool updated = false;
        size_t i = 0;
        const V3 startCamera = cameraPose.translation();
        for (size_t p = 0; p < n; p++) {
            const V3 &point = points[p];
            V3 worldPoint = cameraPose.mul(point);
            float depth = point.z;
            if (depth > depthLimit) continue;
            V3 dir = normalized(worldPoint - startCamera);
            float truncation = truncator.GetTruncationDistance(depth);
            V3 start = worldPoint - dir * truncation - chunk->origin;
            V3 end = worldPoint + dir * truncation - chunk->origin;
            start.x *= roundX; start.y *= roundY; start.z *= roundZ;
            end.x *= roundX; end.y *= roundY; end.z *= roundZ;
            raycastVoxels.clear();
            Raycast(start, end, chunkMin, chunkMax, &raycastVoxels);
            for (const I3 &voxelCoords : raycastVoxels) {
                int id = chunk->GetVoxelID(voxelCoords.x, voxelCoords.y, voxelCoords.z);
                DistVoxel &distVoxel = chunk->voxels.at(id);
                V3 centroid = centroids[id] + chunk->origin;
                float u = depth - (inversePose.mul(centroid) - startCamera).z;
                float weight = ConstantWeight(weighterWeight, u, truncation);
                tally->hits++;
                if (fabs(u) < truncation) {
                    distVoxel.Integrate(u, weight);
                    if (withColor) {
                        const V3 &color = colors[i];
                        chunk->colors.at(id).Integrate((uint8_t)(int)(color.x * 255.0f), (uint8_t)(int)(color.y * 255.0f),
                                                       (uint8_t)(int)(color.z * 255.0f), 1);
                    }
                    updated = true;
                    tally->sdf++;
                } else if (enableVoxelCarving && u > truncation + carvingDist) {
                    if (distVoxel.weight > 0) {
                        distVoxel.Integrate(1.0e-5, 5.0f);
                        updated = true;
                        tally->carved++;
                    }
                }
            }
            i++;
        }
        return updated;
    }

    // Chisel::IntegratePointCloud.  ref: Chisel.cpp:107-157 (serial here; chunks are independent)
    void IntegratePointCloudScan(const V3 *points, size_t n, const V3 *colors, const Pose &extrinsic, float truncation, float maxDist) {
        memset(counters, 0, sizeof(counters));
        const Affine pose = Affine::fromPose(extrinsic);
        const Affine inversePose = pose.inverse();
        std::vector<I3> chunksIntersecting;
        GetChunkIDsIntersectingCloud(points, n, pose, truncation, maxDist, &chunksIntersecting);
        counters[OC_CNT_CANDIDATES] = chunksIntersecting.size();
        if (chunksIntersecting.size() == 0) return;
        std::vector<I3> garbageChunks;
        CloudTally tally;
        for (const I3 &chunkID : chunksIntersecting) {
            bool chunkNew = false;
            if (!HasChunk(chunkID)) {
                chunkNew = true;
                CreateChunk(chunkID);
                counters[OC_CNT_CREATED]++;
            }
            ChunkPtr chunk = GetChunk(chunkID);
            bool needsUpdate = IntegrateCloudChunk(points, n, colors, pose, inversePose, chunk.get(), &tally);
            if (needsUpdate) {
                MarkNeighbours(chunkID);
                counters[OC_CNT_UPDATED_CHUNKS]++;
            } else if (chunkNew) {
                garbageChunks.push_back(chunkID);
            }
        }
        for (const I3 &id : garbageChunks) {
            chunks.erase(id);
            counters[OC_CNT_COLLECTED]++;
        }
        counters[OC_CNT_SDF] = tally.sdf;
        counters[OC_CNT_CARVED] = tally.carved;
        counters[OC_CNT_VISITED] = tally.hits;
    }


    void MarkNeighbours(const I3 &chunkID) {  // Chisel.h:87-98 / :175-189
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dz = -1; dz <= 1; dz++) meshesToUpdate[chunkID + I3(dx, dy, dz)] = true;
    }

    // Chisel::IntegrateDepthScan<float> (serial).  ref: Chisel.h:59-112
    void IntegrateDepthScan(const DepthView &depthImage, const Pose &extrinsic, const Camera &camera) {
        memset(counters, 0, sizeof(counters));
        auto t0 = std::chrono::steady_clock::now();
        Frustum frustum;
        SetupFrustum(camera, extrinsic, &frustum);
        std::vector<I3> chunksIntersecting;
        GetChunkIDsIntersecting(frustum, &chunksIntersecting);
        auto t1 = std::chrono::steady_clock::now();
        std::vector<I3> garbageChunks;
        Tally tally;
        for (const I3 &chunkID : chunksIntersecting) {
            bool chunkNew = false;
            if (!HasChunk(chunkID)) {
                chunkNew = true;
                CreateChunk(chunkID);
                counters[OC_CNT_CREATED]++;
            }
            ChunkPtr chunk = GetChunk(chunkID);
            bool needsUpdate = Integrate(depthImage, camera, extrinsic, chunk.get(), chunkNew, &tally);
            if (needsUpdate) {
                MarkNeighbours(chunkID);
                counters[OC_CNT_UPDATED_CHUNKS]++;
            } else if (chunkNew) {
                garbageChunks.push_back(chunkID);
            }
        }
        auto t2 = std::chrono::steady_clock::now();
        for (const I3 &id : garbageChunks) {  // Chisel::GarbageCollect Chisel.cpp:61-67
            chunks.erase(id);
            counters[OC_CNT_COLLECTED]++;
        }
        PrintMemoryStatisticsSweep();  // Chisel.h:111 (full pass over every voxel; output discarded)
        auto t3 = std::chrono::steady_clock::now();
        counters[OC_CNT_SDF] = tally.sdf;
        counters[OC_CNT_PROBE] = tally.probe;
        counters[OC_CNT_CARVED] = tally.carved;
        counters[OC_CNT_CANDIDATES] = chunksIntersecting.size();
        counters[OC_CNT_VISITED] = (uint64_t)chunksIntersecting.size() * centroids.size();
        phaseMs[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
        phaseMs[1] = 0.0;
        phaseMs[2] = std::chrono::duration<double, std::milli>(t2 - t1).count();
        phaseMs[3] = std::chrono::duration<double, std::milli>(t3 - t2).count();
    }

    volatile double statsSink = 0;
    void PrintMemoryStatisticsSweep() {  // ChunkManager.cpp:641-678, Chunk.cpp:89-116 (printf dropped)
        size_t inside = 0, outside = 0, unknown = 0;
        float totalWeight = 0;
        for (const auto &c : chunks)
            for (const DistVoxel &vox : c.second->voxels) {
                float weight = vox.weight;
                if (weight > 0) {
                    if (vox.sdf < 0) inside++; else outside++;
                } else {
                    unknown++;
                }
                totalWeight += weight;
            }
        statsSink = (double)inside + outside + unknown + totalWeight;
    }

    // Chisel::IntegrateDepthScanColor<float,uint8_t> (16 threads, static blocks).  ref: Chisel.h:114-213
    void IntegrateDepthScanColor(const DepthView &depthImage, const Pose &depthExtrinsic, const Camera &depthCamera,
                                 const ColorView &colorImage, const Pose &colorExtrinsic, const Camera &colorCamera) {
        memset(counters, 0, sizeof(counters));
        auto t0 = std::chrono::steady_clock::now();
        Frustum frustum;
        SetupFrustum(depthCamera, depthExtrinsic, &frustum);
        std::vector<I3> chunksIntersecting;
        GetChunkIDsIntersecting(frustum, &chunksIntersecting);
        auto t1 = std::chrono::steady_clock::now();
        int n = chunksIntersecting.size();
        std::vector<char> isNew(n), isGarbage(n), isUpdated(n);
        std::vector<ChunkMap::iterator> newChunks(n);
        for (int i = 0; i < n; i++) {
            isNew[i] = 0;
            isGarbage[i] = 0;
            isUpdated[i] = 0;
            const I3 &chunkID = chunksIntersecting[i];
            if (!HasChunk(chunkID)) {
                isNew[i] = 1;
                newChunks[i] = CreateChunk(chunkID);
                counters[OC_CNT_CREATED]++;
            }
        }
        auto t2 = std::chrono::steady_clock::now();
        int nThread = nThreads;
        std::vector<std::thread> threads;
        std::vector<Tally> tallies(nThread);
        std::mutex m;
        int blockSize = (n + nThread - 1) / nThread;
        for (int i = 0; i < nThread; i++) {
            int s = i * blockSize;
            threads.push_back(std::thread([&, s, i]() {
                for (int j = 0, k = s; j < blockSize && k < n; j++, k++) {
                    const I3 &chunkID = chunksIntersecting[k];
                    ChunkPtr chunk = this->GetChunk(chunkID);
                    bool needsUpdate = IntegrateColor(depthImage, depthCamera, depthExtrinsic, colorImage, colorCamera,
                                                      colorExtrinsic, chunk.get(), isNew[k] != 0, &tallies[i]);
                    if (!needsUpdate && isNew[k]) isGarbage[k] = 1;
                    if (needsUpdate) {
                        isUpdated[k] = 1;
                        m.lock();
                        MarkNeighbours(chunkID);
                        m.unlock();
                    }
                }
            }));
        }
        for (int i = 0; i < nThread; i++) threads[i].join();
        auto t3 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++)
            if (isGarbage[i]) {
                chunks.erase(newChunks[i]);
                counters[OC_CNT_COLLECTED]++;
            }
        auto t4 = std::chrono::steady_clock::now();
        for (const Tally &t : tallies) {
            counters[OC_CNT_SDF] += t.sdf;
            counters[OC_CNT_COL] += t.col;
            counters[OC_CNT_COL_SAT] += t.colsat;
            counters[OC_CNT_PROBE] += t.probe;
            counters[OC_CNT_CARVED] += t.carved;
        }
        for (int i = 0; i < n; i++) counters[OC_CNT_UPDATED_CHUNKS] += isUpdated[i];
        counters[OC_CNT_CANDIDATES] = n;
        counters[OC_CNT_VISITED] = (uint64_t)n * centroids.size();
        phaseMs[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
        phaseMs[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
        phaseMs[2] = std::chrono::duration<double, std::milli>(t3 - t2).count();
        phaseMs[3] = std::chrono::duration<double, std::milli>(t4 - t3).count();
    }

    // ---- meshing ----------------------------------------------------------------------------
    void ExtractInsideVoxelMesh(const ChunkPtr &chunk, const I3 &index, const V3 &coords, size_t *nextMeshIndex,
                                Mesh *mesh) const {  // ChunkManager.cpp:259-294
        V3 cornerCoords[8];
        float cornerSDF[8];
        bool allNeighborsObserved = true;
        for (int i = 0; i < 8; ++i) {
            I3 ci(index.x + kCubeOff[i][0], index.y + kCubeOff[i][1], index.z + kCubeOff[i][2]);
            const DistVoxel &thisVoxel = chunk->voxels.at(chunk->GetVoxelID(ci.x, ci.y, ci.z));
            if (thisVoxel.weight <= 0.5) {
                allNeighborsObserved = false;
                break;
            }
            cornerCoords[i] = coords + V3((float)kCubeOff[i][0] * voxelResolutionMeters, (float)kCubeOff[i][1] * voxelResolutionMeters,
                                          (float)kCubeOff[i][2] * voxelResolutionMeters);
            cornerSDF[i] = thisVoxel.sdf;
        }
        if (allNeighborsObserved) {
            MeshCube(cornerCoords, cornerSDF, nextMeshIndex, mesh);
            if (IsOccupied(cornerSDF)) mesh->grids.push_back(coords);
        }
    }
    void ExtractBorderVoxelMesh(const ChunkPtr &chunk, const I3 &index, const V3 &coordinates, size_t *nextMeshIndex,
                                Mesh *mesh) const {  // ChunkManager.cpp:296-379
        V3 cornerCoords[8];
        float cornerSDF[8];
        bool allNeighborsObserved = true;
        for (int i = 0; i < 8; ++i) {
            int c[3] = {index.x + kCubeOff[i][0], index.y + kCubeOff[i][1], index.z + kCubeOff[i][2]};
            const int cs[3] = {chunkSize.x, chunkSize.y, chunkSize.z};
            const DistVoxel *thisVoxel = nullptr;
            if (chunk->IsCoordValid(c[0], c[1], c[2])) {
                thisVoxel = &chunk->voxels.at(chunk->GetVoxelID(c[0], c[1], c[2]));
            } else {
                int off[3] = {0, 0, 0};
                for (int j = 0; j < 3; j++) {
                    if (c[j] < 0) {
                        off[j] = -1;
                        c[j] = cs[j] - 1;
                    } else if (c[j] >= cs[j]) {
                        off[j] = 1;
                        c[j] = 0;
                    }
                }
                I3 neighborID = I3(off[0], off[1], off[2]) + chunk->ID;
                auto it = chunks.find(neighborID);
                if (it == chunks.end()) {
                    allNeighborsObserved = false;
                    break;
                }
                const ChunkPtr &neighborChunk = it->second;
                if (!neighborChunk->IsCoordValid(c[0], c[1], c[2])) {
                    allNeighborsObserved = false;
                    break;
                }
                thisVoxel = &neighborChunk->voxels.at(neighborChunk->GetVoxelID(c[0], c[1], c[2]));
            }
            if (thisVoxel->weight <= 0.5) {
                allNeighborsObserved = false;
                break;
            }
            cornerCoords[i] = coordinates + V3((float)kCubeOff[i][0] * voxelResolutionMeters, (float)kCubeOff[i][1] * voxelResolutionMeters,
                                               (float)kCubeOff[i][2] * voxelResolutionMeters);
            cornerSDF[i] = thisVoxel->sdf;
        }
        if (allNeighborsObserved) {
            MeshCube(cornerCoords, cornerSDF, nextMeshIndex, mesh);
            if (IsOccupied(cornerSDF)) mesh->grids.push_back(coordinates);
        }
    }
    void GenerateMesh(const ChunkPtr &chunk, Mesh *mesh) const {  // ChunkManager.cpp:381-447
        mesh->Clear();
        const int maxX = chunkSize.x, maxY = chunkSize.y, maxZ = chunkSize.z;
        I3 index;
        int i = 0;
        size_t nextIndex = 0;
        for (index.z = 0; index.z < maxZ - 1; index.z++)
            for (index.y = 0; index.y < maxY - 1; index.y++)
                for (index.x = 0; index.x < maxX - 1; index.x++) {
                    i = chunk->GetVoxelID(index.x, index.y, index.z);
                    ExtractInsideVoxelMesh(chunk, index, centroids.at(i) + chunk->origin, &nextIndex, mesh);
                }
        index.x = maxX - 1;  // max X plane
        for (index.z = 0; index.z < maxZ - 1; index.z++)
            for (index.y = 0; index.y < maxY; index.y++) {
                i = chunk->GetVoxelID(index.x, index.y, index.z);
                ExtractBorderVoxelMesh(chunk, index, centroids.at(i) + chunk->origin, &nextIndex, mesh);
            }
        index.y = maxY - 1;  // max Y plane
        for (index.z = 0; index.z < maxZ - 1; index.z++)
            for (index.x = 0; index.x < maxX - 1; index.x++) {
                i = chunk->GetVoxelID(index.x, index.y, index.z);
                ExtractBorderVoxelMesh(chunk, index, centroids.at(i) + chunk->origin, &nextIndex, mesh);
            }
        index.z = maxZ - 1;  // max Z plane
        for (index.y = 0; index.y < maxY; index.y++)
            for (index.x = 0; index.x < maxX; index.x++) {
                i = chunk->GetVoxelID(index.x, index.y, index.z);
                ExtractBorderVoxelMesh(chunk, index, centroids.at(i) + chunk->origin, &nextIndex, mesh);
            }
    }

    bool GetSDF(const V3 &posf, double *dist) const {  // ChunkManager.cpp:476-499
        ChunkPtr chunk = GetChunkAt(posf);
        if (chunk) {
            V3 relativePos = posf - chunk->origin;
            I3 coords = chunk->GetVoxelCoords(relativePos);
            int id = chunk->GetVoxelID(coords.x, coords.y, coords.z);
            if (id >= 0 && id < chunk->GetTotalNumVoxels()) {
                const DistVoxel &voxel = chunk->voxels.at(id);
                if (voxel.weight > 1e-12) {
                    *dist = voxel.sdf;
                    return true;
                }
            }
            return false;
        }
        return false;
    }
    bool GetSDFAndGradient(const V3 &pos, double *dist, V3 *grad) const {  // ChunkManager.cpp:449-474
        const float r = voxelResolutionMeters;
        V3 posf = V3(std::floor(pos.x / r) * r + r / 2.0f, std::floor(pos.y / r) * r + r / 2.0f,
                     std::floor(pos.z / r) * r + r / 2.0f);
        if (!GetSDF(posf, dist)) return false;
        double ddxplus, ddyplus, ddzplus = 0.0;
        double ddxminus, ddyminus, ddzminus = 0.0;
        if (!GetSDF(posf + V3(r, 0, 0), &ddxplus)) return false;
        if (!GetSDF(posf + V3(0, r, 0), &ddyplus)) return false;
        if (!GetSDF(posf + V3(0, 0, r), &ddzplus)) return false;
        if (!GetSDF(posf - V3(r, 0, 0), &ddxminus)) return false;
        if (!GetSDF(posf - V3(0, r, 0), &ddyminus)) return false;
        if (!GetSDF(posf - V3(0, 0, r), &ddzminus)) return false;
        *grad = V3((float)(ddxplus - ddxminus), (float)(ddyplus - ddyminus), (float)(ddzplus - ddzminus));
        float z = squaredNorm(*grad);  // grad->normalize() (Eigen 3.3: only when z > 0)
        if (z > 0.0f) *grad = *grad / std::sqrt(z);
        return true;
    }
    void ComputeNormalsFromGradients(Mesh *mesh) const {  // ChunkManager.cpp:609-626
        double dist;
        V3 grad;
        for (size_t i = 0; i < mesh->vertices.size(); i++) {
            const V3 &vertex = mesh->vertices.at(i);
            if (GetSDFAndGradient(vertex, &dist, &grad)) {
                float mag = norm(grad);
                if (mag > 1e-12) mesh->normals[i] = grad * (1.0f / mag);
            }
        }
    }
    const ColorVoxel *GetColorVoxel(const V3 &pos) const {  // ChunkManager.cpp:588-607
        ChunkPtr chunk = GetChunkAt(pos);
        if (chunk.get()) {
            V3 rel = (pos - chunk->origin);
            I3 c = chunk->GetVoxelCoords(rel);
            int id = chunk->GetVoxelID(c.x, c.y, c.z);
            if (id >= 0 && id < chunk->GetTotalNumVoxels()) return &(chunk->colors.at(id));
            return nullptr;
        }
        return nullptr;
    }
    V3 InterpolateColor(const V3 &colorPos) const {  // ChunkManager.cpp:501-573
        const float &x = colorPos.x;
        const float &y = colorPos.y;
        const float &z = colorPos.z;
        const float r = voxelResolutionMeters;
        const int x_0 = static_cast<int>(std::floor(x / r));
        const int y_0 = static_cast<int>(std::floor(y / r));
        const int z_0 = static_cast<int>(std::floor(z / r));
        const int x_1 = x_0 + 1, y_1 = y_0 + 1, z_1 = z_0 + 1;
        // sic: integer voxel indices passed as metric positions (:506-520)
        const ColorVoxel *v_000 = GetColorVoxel(V3(x_0, y_0, z_0));
        const ColorVoxel *v_001 = GetColorVoxel(V3(x_0, y_0, z_1));
        const ColorVoxel *v_011 = GetColorVoxel(V3(x_0, y_1, z_1));
        const ColorVoxel *v_111 = GetColorVoxel(V3(x_1, y_1, z_1));
        const ColorVoxel *v_110 = GetColorVoxel(V3(x_1, y_1, z_0));
        const ColorVoxel *v_100 = GetColorVoxel(V3(x_1, y_0, z_0));
        const ColorVoxel *v_010 = GetColorVoxel(V3(x_0, y_1, z_0));
        const ColorVoxel *v_101 = GetColorVoxel(V3(x_1, y_0, z_1));
        if (!v_000 || !v_001 || !v_011 || !v_111 || !v_110 || !v_100 || !v_010 || !v_101) {
            I3 chunkID = GetIDAt(colorPos);
            auto it = chunks.find(chunkID);
            if (it == chunks.end()) return V3(0, 0, 0);
            return it->second->GetColorAt(colorPos);
        }
        float xd = (x - x_0) / (x_1 - x_0);
        float yd = (y - y_0) / (y_1 - y_0);
        float zd = (z - z_0) / (z_1 - z_0);
        float out[3];
        for (int ch = 0; ch < 3; ch++) {
            auto g = [ch](const ColorVoxel *v) -> uint8_t { return ch == 0 ? v->red : (ch == 1 ? v->green : v->blue); };
            float c_00 = g(v_000) * (1 - xd) + g(v_100) * xd;
            float c_10 = g(v_010) * (1 - xd) + g(v_110) * xd;
            float c_01 = g(v_001) * (1 - xd) + g(v_101) * xd;
            float c_11 = g(v_011) * (1 - xd) + g(v_111) * xd;
            float c_0 = c_00 * (1 - yd) + c_10 * yd;
            float c_1 = c_01 * (1 - yd) + c_11 * yd;
            float c = c_0 * (1 - zd) + c_1 * zd;
            out[ch] = c / 255.0f;
        }
        return V3(out[0], out[1], out[2]);
    }
    void ColorizeMesh(Mesh *mesh) const {  // ChunkManager.cpp:628-639
        mesh->colors.clear();
        mesh->colors.resize(mesh->vertices.size());
        for (size_t i = 0; i < mesh->vertices.size(); i++) mesh->colors[i] = InterpolateColor(mesh->vertices.at(i));
    }
    void RecomputeMesh(const I3 &chunkID, std::mutex &mutex) {  // ChunkManager.cpp:91-128
        if (!HasChunk(chunkID)) return;
        MeshPtr mesh;
        mutex.lock();  // the reference reads allMeshes unlocked here (latent race); same result
        auto it = allMeshes.find(chunkID);
        if (it == allMeshes.end()) mesh = std::make_shared<Mesh>(); else mesh = it->second;
        mutex.unlock();
        ChunkPtr chunk = GetChunk(chunkID);
        GenerateMesh(chunk, mesh.get());
        if (useColor) ColorizeMesh(mesh.get());
        ComputeNormalsFromGradients(mesh.get());
        mutex.lock();
        if (!mesh->grids.empty()) allMeshes[chunkID] = mesh;
        mutex.unlock();
    }
    void RecomputeMeshes(const ChunkSet &chunkMeshes) {  // ChunkManager.cpp:130-169
        if (chunkMeshes.empty()) return;
        std::vector<I3> chunkIDList;
        for (const auto &c : chunkMeshes)
            if (c.second) chunkIDList.push_back(c.first);
        int nThread = nThreads;
        std::vector<std::thread> threads;
        std::mutex mutex;
        int n = chunkIDList.size();
        int blockSize = (n + nThread - 1) / nThread;
        for (int i = 0; i < nThread; i++) {
            int s = i * blockSize;
            threads.push_back(std::thread([this, &mutex, &chunkIDList, n, s, blockSize]() {
                for (int j = 0, k = s + j; j < blockSize && k < n; j++, k++) RecomputeMesh(chunkIDList[k], mutex);
            }));
        }
        for (int i = 0; i < nThread; i++) threads[i].join();
    }
    void UpdateMeshes(bool force) {  // Chisel.cpp:50-59
        if (force || (updateMeshesCalls++ % 10 == 0)) {
            RecomputeMeshes(meshesToUpdate);
            meshesToUpdate.clear();
        }
    }
    bool SaveAllMeshesToPLY(const char *filename) const {  // Chisel.cpp:69-105 + io/PLY.cpp:29-88
        Mesh full;
        size_t v = 0;
        for (const auto &it : allMeshes) {
            for (const V3 &vert : it.second->vertices) {
                full.vertices.push_back(vert);
                full.indices.push_back(v);
                v++;
            }
            for (const V3 &c : it.second->colors) full.colors.push_back(c);
            for (const V3 &nrm : it.second->normals) full.normals.push_back(nrm);
        }
        std::ofstream stream(filename);
        if (!stream) return false;
        size_t numPoints = full.vertices.size();
        stream << "ply" << std::endl;
        stream << "format ascii 1.0" << std::endl;
        stream << "element vertex " << numPoints << std::endl;
        stream << "property float x" << std::endl;
        stream << "property float y" << std::endl;
        stream << "property float z" << std::endl;
        if (!full.colors.empty()) {
            stream << "property uchar red" << std::endl;
            stream << "property uchar green" << std::endl;
            stream << "property uchar blue" << std::endl;
        }
        stream << "element face " << numPoints / 3 << std::endl;
        stream << "property list uchar int vertex_index" << std::endl;
        stream << "end_header" << std::endl;
        size_t vert_idx = 0;
        for (const V3 &vert : full.vertices) {
            stream << vert.x << " " << vert.y << " " << vert.z;
            if (!full.colors.empty()) {
                const V3 &color = full.colors[vert_idx];
                int r = static_cast<int>(color.x * 255.0f);
                int g = static_cast<int>(color.y * 255.0f);
                int b = static_cast<int>(color.z * 255.0f);
                stream << " " << r << " " << g << " " << b;
            }
            stream << std::endl;
            vert_idx++;
        }
        for (size_t i = 0; i < full.indices.size(); i += 3) {
            stream << "3 ";
            for (int j = 0; j < 3; j++) stream << full.indices.at(i + j) << " ";
            stream << std::endl;
        }
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// C interface
// ---------------------------------------------------------------------------------------------
extern "C" {

oc_map *oc_create(int csx, int csy, int csz, float resolution, int use_color) {
    return new oc_map(I3(csx, csy, csz), resolution, use_color != 0);
}
void oc_destroy(oc_map *m) { delete m; }
void oc_reset(oc_map *m) {  // Chisel::Reset Chisel.cpp:44-48 + ChunkManager::Reset ChunkManager.cpp:176-180
    m->allMeshes.clear();
    m->chunks.clear();
    m->meshesToUpdate.clear();
}
void oc_set_integrator(oc_map *m, int trunc_kind, float trunc_param, float weight, int carving_enabled,
                       float carving_dist) {
    m->truncator.kind = trunc_kind;
    m->truncator.param = trunc_param;
    m->weighterWeight = weight;
    m->enableVoxelCarving = carving_enabled != 0;
    m->carvingDist = carving_dist;
}
void oc_set_threads(oc_map *m, int n) { m->nThreads = n < 1 ? 1 : n; }

static Camera makeCamera(float fx, float fy, float cx, float cy, int W, int H, float nearP, float farP) {
    Camera c;
    c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy; c.width = W; c.height = H; c.nearPlane = nearP; c.farPlane = farP;
    return c;
}

void oc_integrate_depth(oc_map *m, const float *depth, int W, int H, const float *pose, float fx, float fy, float cx,
                        float cy, float near_plane, float far_plane) {
    DepthView d{depth, W, H};
    m->IntegrateDepthScan(d, Pose::fromRowMajor3x4(pose), makeCamera(fx, fy, cx, cy, W, H, near_plane, far_plane));
}
void oc_integrate_depth_color(oc_map *m, const float *depth, int W, int H, const float *pose, float fx, float fy,
                              float cx, float cy, float near_plane, float far_plane, const uint8_t *color, int CW,
                              int CH, int channels, const float *color_pose, float cfx, float cfy, float ccx,
                              float ccy) {
    DepthView d{depth, W, H};
    ColorView c{color, CW, CH, channels};
    m->IntegrateDepthScanColor(d, Pose::fromRowMajor3x4(pose), makeCamera(fx, fy, cx, cy, W, H, near_plane, far_plane), c,
                               Pose::fromRowMajor3x4(color_pose),
                               makeCamera(cfx, cfy, ccx, ccy, CW, CH, near_plane, far_plane));
}
void oc_integrate_pointcloud(oc_map *m, const float *points_xyz, int n_points, const float *colors_rgb, const float *pose,
                             float truncation, float max_dist) {
    std::vector<V3> pts((size_t)n_points), cols;
    for (int i = 0; i < n_points; i++) pts[i] = V3(points_xyz[3 * i], points_xyz[3 * i + 1], points_xyz[3 * i + 2]);
    if (colors_rgb) {
        cols.resize((size_t)n_points);
        for (int i = 0; i < n_points; i++) cols[i] = V3(colors_rgb[3 * i], colors_rgb[3 * i + 1], colors_rgb[3 * i + 2]);
    }
    m->IntegratePointCloudScan(pts.data(), pts.size(), colors_rgb ? cols.data() : nullptr, Pose::fromRowMajor3x4(pose), truncation,
                               max_dist);
}
// ChunkManager::GetChunkIDsIntersecting(cloud, cameraTransform, truncation, maxDist, chunkList) on its own (ChunkManager.cpp:214-257);
// returns the number of ids (at most `capacity` are written, in the listing order of the oracle's own hash map)
int oc_cloud_chunk_ids(oc_map *m, const float *points_xyz, int n_points, const float *pose, float truncation, float max_dist, int *ids_xyz,
                       int capacity) {
    std::vector<V3> pts((size_t)n_points);
    for (int i = 0; i < n_points; i++) pts[i] = V3(points_xyz[3 * i], points_xyz[3 * i + 1], points_xyz[3 * i + 2]);
    std::vector<I3> out;
    m->GetChunkIDsIntersectingCloud(pts.data(), pts.size(), oc_map::Affine::fromPose(Pose::fromRowMajor3x4(pose)), truncation, max_dist, &out);
    for (size_t i = 0; i < out.size() && (int)i < capacity; i++) {
        ids_xyz[3 * i] = out[i].x; ids_xyz[3 * i + 1] = out[i].y; ids_xyz[3 * i + 2] = out[i].z;
    }
    return (int)out.size();
}
int oc_raycast(const float *start3, const float *end3, const int *min3, const int *max3, int *cells_xyz, int capacity) {
    std::vector<I3> out;
    oc_map::Raycast(V3(start3[0], start3[1], start3[2]), V3(end3[0], end3[1], end3[2]), I3(min3[0], min3[1], min3[2]),
                    I3(max3[0], max3[1], max3[2]), &out);
    for (size_t i = 0; i < out.size() && (int)i < capacity; i++) {
        cells_xyz[3 * i] = out[i].x; cells_xyz[3 * i + 1] = out[i].y; cells_xyz[3 * i + 2] = out[i].z;
    }
    return (int)out.size();
}
void oc_invert_pose(const float *pose, float *inverse) {
    oc_map::Affine inv = oc_map::Affine::fromPose(Pose::fromRowMajor3x4(pose)).inverse();
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) inverse[4 * r + c] = inv.m[r][c];
}
void oc_get_counters(const oc_map *m, uint64_t *out) { memcpy(out, m->counters, sizeof(m->counters)); }
void oc_get_phase_ms(const oc_map *m, double *out4) { memcpy(out4, m->phaseMs, sizeof(m->phaseMs)); }

int oc_num_chunks(const oc_map *m) { return (int)m->chunks.size(); }
void oc_list_chunks(const oc_map *m, int *ids) {
    int i = 0;
    for (const auto &c : m->chunks) {
        ids[i++] = c.first.x; ids[i++] = c.first.y; ids[i++] = c.first.z;
    }
}
int oc_has_chunk(const oc_map *m, int x, int y, int z) { return m->HasChunk(I3(x, y, z)) ? 1 : 0; }
int oc_get_chunk(const oc_map *m, int x, int y, int z, float *sdf, float *weight, uint8_t *rgbw) {
    auto it = m->chunks.find(I3(x, y, z));
    if (it == m->chunks.end()) return 0;
    const Chunk &c = *it->second;
    size_t n = c.voxels.size();
    for (size_t i = 0; i < n; i++) {
        sdf[i] = c.voxels[i].sdf;
        weight[i] = c.voxels[i].weight;
    }
    if (rgbw && !c.colors.empty())
        for (size_t i = 0; i < n; i++) {
            rgbw[4 * i + 0] = c.colors[i].red; rgbw[4 * i + 1] = c.colors[i].green;
            rgbw[4 * i + 2] = c.colors[i].blue; rgbw[4 * i + 3] = c.colors[i].weight;
        }
    return 1;
}
int oc_remove_chunk(oc_map *m, int x, int y, int z) { return m->chunks.erase(I3(x, y, z)) ? 1 : 0; }

int oc_num_meshes_to_update(const oc_map *m) { return (int)m->meshesToUpdate.size(); }
void oc_list_meshes_to_update(const oc_map *m, int *ids) {
    int i = 0;
    for (const auto &c : m->meshesToUpdate) {
        ids[i++] = c.first.x; ids[i++] = c.first.y; ids[i++] = c.first.z;
    }
}
void oc_update_meshes(oc_map *m, int force) { m->UpdateMeshes(force != 0); }
int oc_num_meshes(const oc_map *m) { return (int)m->allMeshes.size(); }
void oc_list_meshes(const oc_map *m, int *ids) {
    int i = 0;
    for (const auto &c : m->allMeshes) {
        ids[i++] = c.first.x; ids[i++] = c.first.y; ids[i++] = c.first.z;
    }
}
int oc_mesh_size(const oc_map *m, int x, int y, int z, int *nv, int *ng) {
    auto it = m->allMeshes.find(I3(x, y, z));
    if (it == m->allMeshes.end()) return 0;
    *nv = (int)it->second->vertices.size();
    *ng = (int)it->second->grids.size();
    return 1;
}
int oc_get_mesh(const oc_map *m, int x, int y, int z, float *vertices, float *normals, float *colors, float *grids) {
    auto it = m->allMeshes.find(I3(x, y, z));
    if (it == m->allMeshes.end()) return 0;
    const Mesh &ms = *it->second;
    for (size_t i = 0; i < ms.vertices.size(); i++) {
        if (vertices) { vertices[3 * i] = ms.vertices[i].x; vertices[3 * i + 1] = ms.vertices[i].y; vertices[3 * i + 2] = ms.vertices[i].z; }
        if (normals) { normals[3 * i] = ms.normals[i].x; normals[3 * i + 1] = ms.normals[i].y; normals[3 * i + 2] = ms.normals[i].z; }
        if (colors && !ms.colors.empty()) { colors[3 * i] = ms.colors[i].x; colors[3 * i + 1] = ms.colors[i].y; colors[3 * i + 2] = ms.colors[i].z; }
    }
    if (grids)
        for (size_t i = 0; i < ms.grids.size(); i++) { grids[3 * i] = ms.grids[i].x; grids[3 * i + 1] = ms.grids[i].y; grids[3 * i + 2] = ms.grids[i].z; }
    return 1;
}
int oc_save_ply(const oc_map *m, const char *path) { return m->SaveAllMeshesToPLY(path) ? 1 : 0; }
int oc_get_sdf(const oc_map *m, float x, float y, float z, double *dist) { return m->GetSDF(V3(x, y, z), dist) ? 1 : 0; }
int oc_get_sdf_and_gradient(const oc_map *m, float x, float y, float z, double *dist, float *g) {
    V3 grad;
    bool ok = m->GetSDFAndGradient(V3(x, y, z), dist, &grad);
    if (ok) { g[0] = grad.x; g[1] = grad.y; g[2] = grad.z; }
    return ok ? 1 : 0;
}

float oc_truncation(int kind, float param, float depth) { return Truncator(kind, param).GetTruncationDistance(depth); }
float oc_weight(float weight, float sd, float trunc) { return ConstantWeight(weight, sd, trunc); }
void oc_dist_integrate(float *sdf, float *weight, float d, float wu) {
    DistVoxel v;
    v.sdf = *sdf; v.weight = *weight;
    v.Integrate(d, wu);
    *sdf = v.sdf; *weight = v.weight;
}
void oc_color_integrate(uint8_t *rgbw, uint8_t r, uint8_t g, uint8_t b, uint8_t wu) {
    ColorVoxel v;
    v.red = rgbw[0]; v.green = rgbw[1]; v.blue = rgbw[2]; v.weight = rgbw[3];
    v.Integrate(r, g, b, wu);
    rgbw[0] = v.red; rgbw[1] = v.green; rgbw[2] = v.blue; rgbw[3] = v.weight;
}
void oc_color_at(const uint8_t *data, int width, int channels, int row, int col, uint8_t *rgba) {
    ColorView c{data, width, 0, channels};
    c.At(row, col, rgba);
}
uint64_t oc_chunk_hash(int x, int y, int z) { return (uint64_t)ChunkHasher()(I3(x, y, z)); }
void oc_project_point(float fx, float fy, float cx, float cy, const float *p, float *out) {
    Camera c = makeCamera(fx, fy, cx, cy, 0, 0, 0, 0);
    V3 r = c.ProjectPoint(V3(p[0], p[1], p[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oc_frustum(const float *pose, float nearP, float farP, float fy, float cy, int W, int H, float *corners,
                float *planes, float *aabb) {
    Frustum f;
    SetupFrustum(makeCamera(fy, fy, 0, cy, W, H, nearP, farP), Pose::fromRowMajor3x4(pose), &f);
    for (int i = 0; i < 8; i++) { corners[3 * i] = f.corners[i].x; corners[3 * i + 1] = f.corners[i].y; corners[3 * i + 2] = f.corners[i].z; }
    const Plane *pl[6] = {&f.far_, &f.near_, &f.top, &f.bottom, &f.left, &f.right};
    for (int i = 0; i < 6; i++) { planes[4 * i] = pl[i]->normal.x; planes[4 * i + 1] = pl[i]->normal.y; planes[4 * i + 2] = pl[i]->normal.z; planes[4 * i + 3] = pl[i]->distance; }
    AABB b;
    f.ComputeBoundingBox(&b);
    aabb[0] = b.min.x; aabb[1] = b.min.y; aabb[2] = b.min.z; aabb[3] = b.max.x; aabb[4] = b.max.y; aabb[5] = b.max.z;
}
int oc_candidates(const oc_map *m, const float *pose, float nearP, float farP, float fy, float cy, int W, int H,
                  int *ids, int max_ids) {
    Frustum f;
    SetupFrustum(makeCamera(fy, fy, 0, cy, W, H, nearP, farP), Pose::fromRowMajor3x4(pose), &f);
    std::vector<I3> list;
    m->GetChunkIDsIntersecting(f, &list);
    int n = (int)list.size();
    for (int i = 0; i < n && i < max_ids; i++) { ids[3 * i] = list[i].x; ids[3 * i + 1] = list[i].y; ids[3 * i + 2] = list[i].z; }
    return n;
}
int oc_mesh_cube(const float *s, const float *o, float res, float *verts, float *normals) {
    V3 coords[8];
    for (int i = 0; i < 8; i++)
        coords[i] = V3(o[0], o[1], o[2]) + V3((float)kCubeOff[i][0] * res, (float)kCubeOff[i][1] * res, (float)kCubeOff[i][2] * res);
    Mesh mesh;
    size_t next = 0;
    MeshCube(coords, s, &next, &mesh);
    for (size_t i = 0; i < mesh.vertices.size(); i++) {
        verts[3 * i] = mesh.vertices[i].x; verts[3 * i + 1] = mesh.vertices[i].y; verts[3 * i + 2] = mesh.vertices[i].z;
        normals[3 * i] = mesh.normals[i].x; normals[3 * i + 1] = mesh.normals[i].y; normals[3 * i + 2] = mesh.normals[i].z;
    }
    return (int)mesh.vertices.size();
}
// geometry/Interpolate.h:28-36 (off the live path: only DepthImage::BilinearInterpolateDepth calls it and
// its call sites, ProjectionIntegrator.h:72,131, are commented out).  Restated for the known-answer test.
float oc_bilinear_interpolate(float c00, float c10, float c01, float c11, float tx, float ty) {
    float a = c00 + (c10 - c00) * tx;
    float b = c01 + (c11 - c01) * tx;
    return a + (b - a) * ty;
}
void oc_triangle_table_row(int index, int *row16) { memcpy(row16, triTable().rows[index & 255], 16 * sizeof(int)); }

}  // extern "C"
