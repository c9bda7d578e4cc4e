// marching_cubes/MarchingCubes.h:41-146 of the reference for the chisel_hip facade: the public statics of chisel::MarchingCubes over the
// C ABI (chisel_hip_mc_tables, chisel_hip_mesh_cube_values, chisel_hip_interpolate_vertex: the arithmetic runs on the device, as every
// other marching cube of this library does; ChunkManager::GenerateMesh does not come through here but through the per-chunk kernels).
#ifndef CHISEL_HIP_FACADE_MARCHINGCUBES_H_
#define CHISEL_HIP_FACADE_MARCHINGCUBES_H_
#include <chisel_hip.h>

#include <cassert>
#include <stdexcept>
#include <vector>

#include "../geometry/Geometry.h"
#include "../mesh/Mesh.h"

namespace chisel {

typedef std::vector<Mat3x3, Eigen::aligned_allocator<Mat3x3>> TriangleVector;

class MarchingCubes {
  public:
    typedef Eigen::Matrix<float, 3, 8> CornerCoords;
    typedef Eigen::Matrix<float, 8, 1> CornerSDF;
    typedef Eigen::Matrix<float, 3, 12> EdgeCoords;

    // the case table and the edge -> corner pairs (MarchingCubes.cpp:29-302), filled from the library on first use
    struct Tables {
        int triangleTable[256][16];
        int edgeIndexPairs[12][2];
        Tables() {
            if (chisel_hip_mc_tables(&triangleTable[0][0], &edgeIndexPairs[0][0]) != CHISEL_HIP_OK) throw std::runtime_error(chisel_hip_last_error());
        }
    };
    static const Tables &tables() {
        static const Tables t;
        return t;
    }
    // (the reference's `static int triangleTable[256][16]` / `edgeIndexPairs[12][2]`, as references to the same data)
    static const int (&triangleTable())[256][16] { return tables().triangleTable; }
    static const int (&edgeIndexPairs())[12][2] { return tables().edgeIndexPairs; }

    MarchingCubes() {}
    virtual ~MarchingCubes() {}

    static bool IsOccupied(const CornerSDF &vertexSDF) { return tables().triangleTable[CalculateVertexConfiguration(vertexSDF)][0] != -1; }

    static void MeshCube(const CornerCoords &vertex_coordinates, const CornerSDF &vertexSDF, TriangleVector *triangles) {
        assert(triangles != nullptr);
        float v[45], n[45];
        const int nv = cube(vertex_coordinates, vertexSDF, nullptr, v, n);
        for (int t = 0; t < nv; t += 3) {  // the device emits (t + 2, t + 1, t) as MeshCube(.., Mesh*) pushes them; a Mat3x3 holds (t, t + 1, t + 2) in its columns
            Mat3x3 tri;
            for (int c = 0; c < 3; c++)
                for (int r = 0; r < 3; r++) tri(r, c) = v[3 * (t + 2 - c) + r];
            triangles->push_back(tri);
        }
    }

    static void MeshCube(const CornerCoords &vertexCoords, const CornerSDF &vertexSDF, VertIndex *nextIDX, Mesh *mesh) {
        assert(nextIDX != nullptr);
        assert(mesh != nullptr);
        float v[45], n[45];
        const int nv = cube(vertexCoords, vertexSDF, nullptr, v, n);
        for (int i = 0; i < nv; i++) {
            mesh->vertices.emplace_back(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
            mesh->normals.emplace_back(n[3 * i], n[3 * i + 1], n[3 * i + 2]);
            mesh->indices.push_back(*nextIDX + (VertIndex)(i % 3));
            if (i % 3 == 2) *nextIDX += 3;
        }
    }

    static int CalculateVertexConfiguration(const CornerSDF &vertexSDF) {
        int index = 0;
        const CornerCoords none;
        cube(none, vertexSDF, &index, nullptr, nullptr);
        return index;
    }

    static void InterpolateEdgeVertices(const CornerCoords &vertexCoords, const CornerSDF &vertSDF, EdgeCoords *edgeCoords) {
        assert(edgeCoords != nullptr);
        float e[36];
        int nv = 0, index = 0;
        if (chisel_hip_mesh_cube_values(vertexCoords.data(), vertSDF.data(), e, &index, nullptr, nullptr, &nv) != CHISEL_HIP_OK)
            throw std::runtime_error(chisel_hip_last_error());
        const Tables &T = tables();
        for (int i = 0; i < 12; i++) {  // only the edges with a sign change are written (the reference leaves the other columns as they are)
            const float s0 = vertSDF(T.edgeIndexPairs[i][0]), s1 = vertSDF(T.edgeIndexPairs[i][1]);
            if ((s0 < 0 && s1 >= 0) || (s0 >= 0 && s1 < 0)) edgeCoords->col(i) = Vec3(e[3 * i], e[3 * i + 1], e[3 * i + 2]);
        }
    }

    static inline Vec3 InterpolateVertex(const Vec3 &vertex1, const Vec3 &vertex2, const float &sdf1, const float &sdf2) {
        const float a[3] = {vertex1(0), vertex1(1), vertex1(2)}, b[3] = {vertex2(0), vertex2(1), vertex2(2)};
        float out[3];
        if (chisel_hip_interpolate_vertex(a, b, sdf1, sdf2, out) != CHISEL_HIP_OK) throw std::runtime_error(chisel_hip_last_error());
        return Vec3(out[0], out[1], out[2]);
    }

  private:
    static int cube(const CornerCoords &coords, const CornerSDF &sdf, int *index, float *v, float *n) {
        int nv = 0, idx = 0;
        if (chisel_hip_mesh_cube_values(coords.data(), sdf.data(), nullptr, &idx, v, n, &nv) != CHISEL_HIP_OK) throw std::runtime_error(chisel_hip_last_error());
        if (index) *index = idx;
        return nv;
    }
};

}  // namespace chisel
#endif
