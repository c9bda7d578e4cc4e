// open_chisel/io/PLY.h -- SaveMeshPLYASCII of the reference (io/PLY.h:31, src/io/PLY.cpp:29-88) for a mesh the caller holds: the text is
// written by the library (chisel_hip_write_mesh_ply, the writer behind Chisel::SaveAllMeshesToPLY), here the mesh is only flattened.
#ifndef CHISEL_HIP_FACADE_PLY_H_
#define CHISEL_HIP_FACADE_PLY_H_
#include <string>
#include <vector>
#include <chisel_hip.h>
#include "../mesh/Mesh.h"
namespace chisel {
inline bool SaveMeshPLYASCII(const std::string &fileName, const MeshConstPtr &mesh) {
    const size_t n = mesh->vertices.size();
    std::vector<float> v(3 * n), c(mesh->HasColors() ? 3 * n : 0);
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            v[3 * i + k] = mesh->vertices[i](k);
            if (!c.empty()) c[3 * i + k] = mesh->colors[i](k);
        }
    std::vector<int64_t> idx(mesh->indices.begin(), mesh->indices.end());
    return chisel_hip_write_mesh_ply(fileName.c_str(), v.data(), c.empty() ? nullptr : c.data(), (int64_t)n, idx.data(), (int64_t)idx.size()) == CHISEL_HIP_OK;
}
}  // namespace chisel
#endif
