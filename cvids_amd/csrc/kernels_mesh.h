// kernels_mesh.h -- marching-cubes mesh extraction (placeholder until the mesh milestone lands).
#pragma once
#include "chisel_device.h"
namespace chisel_hip {
struct MeshBuffers {
    void *p = nullptr;
};
inline void free_mesh_buffers(MeshBuffers &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
}
}  // namespace chisel_hip
