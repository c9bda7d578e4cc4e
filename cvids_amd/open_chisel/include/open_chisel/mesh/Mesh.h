// mesh/Mesh.h:33-60 of the reference
#ifndef CHISEL_HIP_FACADE_MESH_H_
#define CHISEL_HIP_FACADE_MESH_H_
#include <memory>
#include <vector>
#include "../geometry/Geometry.h"
namespace chisel {
typedef size_t VertIndex;
typedef std::vector<VertIndex> VertIndexList;
class Mesh {
  public:
    bool HasVertices() const { return !vertices.empty(); }
    bool HasNormals() const { return !normals.empty(); }
    bool HasColors() const { return !colors.empty(); }
    bool HasIndices() const { return !indices.empty(); }
    void Clear() {
        vertices.clear(); normals.clear(); colors.clear(); indices.clear(); grids.clear();
    }
    Vec3List vertices;
    VertIndexList indices;
    Vec3List normals;
    Vec3List colors;
    Vec3List grids;
};
typedef std::shared_ptr<Mesh> MeshPtr;
typedef std::shared_ptr<const Mesh> MeshConstPtr;
}  // namespace chisel
#endif
